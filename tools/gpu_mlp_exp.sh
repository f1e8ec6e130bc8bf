for e in 0 1 2 4 3 7; do PP_EXP=$e timeout -k 10 200 python tools/gpu_mlp_stamps.py 16384 2>&1 | tail -5 || exit 1; done
