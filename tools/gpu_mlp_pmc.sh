#!/bin/bash
# PMC passes over the policy layers (tools/gpu_mlp_layers.py; MLP_TOOL=tools/gpu_mlp_bwd_layers.py: the backward): matrix-core busy cycles and LDS bank
# conflicts per kernel.
# Counters in their own runs with --kernel-trace only (gpurun refuses --pmc together with the API trace domains).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_mlp; rm -rf $out; mkdir -p $out
pass() { name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python ${MLP_TOOL:-tools/gpu_mlp_layers.py} ${MLP_M:-16384} ${MLP_K:-80} > $out/$name.log 2>&1 || { tail -5 $out/$name.log; return 1; }
}
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES || exit 1
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS || exit 1
pass grbm GRBM_GUI_ACTIVE || exit 1
python - <<'PY'
import collections, csv, glob, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mlp/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"(mlp_[a-z0-9_]+(<[^>]*>)?|prepare_input_kernel|reduce_[a-z]+_kernel)", row["Kernel_Name"])
        if m:
            acc[m.group(1) + " grid " + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/pmc_mlp/summary.csv", "w") as out:
    out.write("kernel,dispatches,counter,mean_per_dispatch\n")
    for k in sorted(acc):
        m = {c: sum(v) / len(v) for c, v in acc[k].items()}
        for c in sorted(m):
            out.write("\"%s\",%d,%s,%.1f\n" % (k, len(acc[k][c]), c, m[c]))
        if m.get("SQ_INSTS_MFMA", 0) > 0 and "GRBM_GUI_ACTIVE" in m:
            # SQ_* are chip-wide sums (SQ_INSTS_MFMA equals workgroups x waves x MFMAs per wave exactly); GRBM_GUI_ACTIVE is summed over the
            # 8 XCDs, so the launch lasted GRBM_GUI_ACTIVE / 8 shader cycles on each of 1024 SIMDs
            util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
            conf = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)
            line = "%-52s MFMA pipes busy %.1f %% of the launch's SIMD-cycles (%.0f MFMAs x %.0f cycles); LDS bank-conflict cycles %.1f %% of LDS-active" % (
                k, 100 * util, m["SQ_INSTS_MFMA"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"], 100 * conf)
            print(line)
            out.write("# " + line + "\n")
PY
