#!/bin/bash
# step time against the number of envs on one GPU (the headline variant and the 27-dof task), final build
for v in TT:4096 TT:8192 TT:16384 TT:32768 TT:65536 TT:131072 TA:1024 TA:2048 TA:4096 TA:8192 TA:16384 TA:65536 T4:8192 T4:32768; do
  timeout -k 10 300 python bench.py --variant ${v%%:*} --num-envs ${v##*:} --steps 1024 --warmup 128 --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v%%:*} N=%-7d kernel %7.2f us   %8.1f M env-steps/s   roofline.frac %.4f' % (d['config']['num_envs_per_gpu'], d['roofline']['avg_kernel_us'], d['value']/1e6, d['roofline']['frac']))" || exit 1
done
