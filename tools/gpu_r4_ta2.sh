#!/bin/bash
set -o pipefail
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_ta_physics.py tests/test_ta_golden.py tests/test_policy_mlp.py tests/test_urdf.py tests/test_isaacgymenvs_shim.py tests/test_collector.py -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_ta.log 2>&1 || { tail -30 gpurun_out/pytest_ta.log; exit 1; }
tail -2 gpurun_out/pytest_ta.log
rm -f gpurun_out/ta_ab.txt
bash tools/gpu_r4_ta_ab.sh || exit 1
timeout -k 10 400 python tools/gpu_ta_chain_stamps.py 4096 > gpurun_out/ta_chain_stamps.txt 2>&1 || { tail -20 gpurun_out/ta_chain_stamps.txt; exit 1; }
grep "inputs staged\|s1 begin\|B1 leave\|B2\|B3\|stored\|^end\|span" gpurun_out/ta_chain_stamps.txt
