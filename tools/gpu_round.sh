#!/bin/bash
# The full evidence set of a build: GPU tests, smoke, bench line (with cpu_baseline), rocprofv3 kernel trace, PMC passes.
set -o pipefail
bash tools/gpu_check.sh || exit 1
rm -rf gpurun_out/pmc; bash tools/gpu_pmc.sh || exit 1
