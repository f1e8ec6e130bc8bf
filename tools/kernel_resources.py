#!/usr/bin/env python3
"""Register / scratch / LDS use of the step kernels from a `hipcc --save-temps` listing (the code object's own metadata):

    hipcc --offload-arch=gfx950 <flags of isaacgym_amd/_lib.py> -c --save-temps=obj -o /tmp/x.o isaacgym_amd/csrc/ppenv.hip
    python tools/kernel_resources.py /tmp/ppenv-hip-amdgcn-amd-amdhsa-gfx950.s [substring ...]
"""
import re
import subprocess
import sys

path, keys = sys.argv[1], sys.argv[2:] or ["step_kernel", "ta_chain", "ta_sim"]
txt = open(path).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for body in re.split(r"\n  - ", meta)[1:]:
    mm = re.search(r"\.name:\s+(\S+)", body)
    if not mm:
        continue
    name = mm.group(1)
    if not any(k in name for k in keys):
        continue
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    name = re.sub(r"\(pp::.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))

    def g(k):
        mm = re.search(re.escape(k) + r":\s+(\d+)", body)
        return mm.group(1) if mm else "-"
    print(f"{name:60s} vgpr {g('.vgpr_count'):>4s} agpr {g('.agpr_count'):>4s} vgpr_spill {g('.vgpr_spill_count'):>4s} sgpr_spill {g('.sgpr_spill_count'):>4s} "
          f"scratch {g('.private_segment_fixed_size'):>5s} B  lds {g('.group_segment_fixed_size'):>6s} B")
