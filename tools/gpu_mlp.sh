#!/bin/bash
# policy MLP: parity tests, then the forward alone (eager, per tile choice) at the two rollout sizes
timeout -k 10 300 python -m pytest tests/test_policy_mlp.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
for tile in 128 256 0; do
PPENV_MLP_TILE=$tile timeout -k 10 300 python - <<'PY'
import os, time, torch
from isaacgym_amd.policy import NativeMLP, UNITS
dev = torch.device("cuda", 0)
for m, k, a in ((4096, 313, 27), (16384, 80, 7)):
    torch.manual_seed(0)
    def mlp(n_out):
        d, out = k, []
        for u in UNITS + [n_out]:
            lin = torch.nn.Linear(d, u); out.append((lin.weight, lin.bias)); d = u
        return out
    net = NativeMLP(mlp(a), mlp(1), k, dev, mean=torch.zeros(k), var=torch.ones(k), max_rows=m)
    obs = torch.randn(m, k, device=dev)
    for _ in range(10): net.forward(obs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): net.forward(obs)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = NativeMLP.flops(m, k, UNITS, a)
    print("tile=%s M=%d K0=%d: forward %.1f us  %.0f TFLOP/s  %.1f %% of 2.5 PF" % (os.environ["PPENV_MLP_TILE"], m, k, us, fl / us / 1e6, 100 * fl / us / 1e6 / 2500))
PY
done
