#!/bin/bash
# Evidence for the policy backward at the learner's minibatch: per-layer times next to PyTorch / hipBLASLt, rocprofv3 kernel stats, PMC (matrix-core
# busy cycles, LDS bank conflicts of the transposed reads).  Run on the GPU box: bash tools/gpu_mlp_bwd_prof.sh [tag]
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_mlp_bwd_layers.py 32768 313 > gpurun_out/${tag}_mlp_bwd_layers_32768.txt 2>&1 || { tail -5 gpurun_out/${tag}_mlp_bwd_layers_32768.txt; exit 1; }
cat gpurun_out/${tag}_mlp_bwd_layers_32768.txt
rm -rf gpurun_out/prof_bwd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -- python tools/gpu_mlp_bwd_layers.py 32768 313 > gpurun_out/prof_bwd.log 2>&1 || { tail -5 gpurun_out/prof_bwd.log; exit 1; }
f=$(find gpurun_out/prof_bwd -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/${tag}_mlp_bwd_kernel_stats.csv; head -12 $f
MLP_TOOL=tools/gpu_mlp_bwd_layers.py MLP_M=32768 MLP_K=313 bash tools/gpu_mlp_pmc.sh && cp gpurun_out/pmc_mlp/summary.csv gpurun_out/${tag}_mlp_bwd_pmc_summary.csv
