#!/bin/bash
# Round 4: the pipelined rollout — K env groups on K streams, each group's layer launches sized for a share of the CUs;
# then the kernel trace of the two-group configuration and how much of each kernel's time another kernel ran beside it.
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
CFG=${PIPE_CFG:-"1,2,2:128,2:160,4:64"}
PIPE_ATTACH=0 timeout -k 10 200 python tools/gpu_rollout_pipeline.py 4096 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rollout_pipeline2.txt || exit 1
timeout -k 10 400 python tools/gpu_rollout_pipeline.py 4096 "$CFG" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/rollout_pipeline2.txt || exit 1
for c in ${PIPE_TRACE:-1 2:128}; do
  d=gpurun_out/prof_pipe_${c/:/_}; rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python tools/gpu_rollout_pipeline.py 4096 $c > $d.log 2>&1 || { tail -20 $d.log; exit 1; }
  f=$(ls $d/*/*kernel_trace.csv | head -1)
  echo "== kernel trace, configuration $c" | tee -a gpurun_out/pipe_overlap.txt
  python tools/trace_overlap.py $f 0.5 | head -14 | tee -a gpurun_out/pipe_overlap.txt
  rm -f $f   # the raw trace is large; the summary above is what is kept
done
