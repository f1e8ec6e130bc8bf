"""Load the reference's task files in THIS container only (never on the GPU box).

The reference (`/root/reference/tasks/*.py`) needs `isaacgym`, `isaacgymenvs`,
`tensorboardX` and `cv2`, none of which exist offline.  Only the module-level
pure-torch functions (reward / observation) are wanted, so the missing
packages are registered as inert stub modules before each file is loaded with
importlib (recipe: SURVEY.md Appendix B).  The quaternion helpers the files
star-import from `isaacgymenvs.utils.torch_jit_utils` are restated here from
their published definitions (quaternions are xyzw).

Used by tools/gen_golden.py to write tests/golden/*.npz.  Nothing in the
product, tests, smoke() or bench imports this file.
"""
import importlib.util
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("PPENV_REFERENCE_ROOT", "/root/reference")


class _Dummy:
    """Permissive placeholder: any attribute / call yields another dummy."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __call__(self, *a, **k):
        return _Dummy()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    if "__getattr__" not in attrs:
        def _ga(attr, _n=name):
            if attr.startswith("__"):
                raise AttributeError(attr)
            return _Dummy()
        m.__getattr__ = _ga
    sys.modules[name] = m
    return m


# ---- published torch_jit_utils helpers (xyzw quaternions) -------------------
def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


def quat_from_angle_axis(angle, axis):
    theta = (angle / 2).unsqueeze(-1)
    xyz = normalize(axis) * theta.sin()
    w = theta.cos()
    return normalize(torch.cat([xyz, w], dim=-1))


def my_quat_rotate(q, v):
    shape = q.shape
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(shape[0], 1, 3), v.view(shape[0], 3, 1)).squeeze(-1) * 2.0
    return a + b + c


def calc_heading(q):
    ref_dir = torch.zeros_like(q[..., 0:3])
    ref_dir[..., 0] = 1
    rot_dir = my_quat_rotate(q, ref_dir)
    return torch.atan2(rot_dir[..., 1], rot_dir[..., 0])


def calc_heading_quat_inv(q):
    heading = calc_heading(q)
    axis = torch.zeros_like(q[..., 0:3])
    axis[..., 2] = 1
    return quat_from_angle_axis(-heading, axis)


def calc_heading_quat(q):
    heading = calc_heading(q)
    axis = torch.zeros_like(q[..., 0:3])
    axis[..., 2] = 1
    return quat_from_angle_axis(heading, axis)


def to_torch(x, dtype=torch.float, device="cpu", requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def install_stubs():
    _stub("isaacgym")
    _stub("isaacgym.gymtorch")
    _stub("isaacgym.gymapi")
    _stub("isaacgym.gymutil")
    tu = types.ModuleType("isaacgym.terrain_utils")
    sys.modules["isaacgym.terrain_utils"] = tu
    _stub("isaacgymenvs")
    _stub("isaacgymenvs.utils")
    tj = types.ModuleType("isaacgymenvs.utils.torch_jit_utils")
    tj.__dict__.update(dict(
        normalize=normalize, quat_from_angle_axis=quat_from_angle_axis,
        my_quat_rotate=my_quat_rotate, calc_heading=calc_heading,
        calc_heading_quat_inv=calc_heading_quat_inv, calc_heading_quat=calc_heading_quat,
        to_torch=to_torch, torch=torch))
    sys.modules["isaacgymenvs.utils.torch_jit_utils"] = tj
    _stub("isaacgymenvs.tasks")
    _stub("isaacgymenvs.tasks.base")
    _stub("isaacgymenvs.tasks.base.vec_task", VecTask=object)
    _stub("isaacgymenvs.tasks.interos")
    _stub("isaacgymenvs.tasks.interos.motion_lib", MotionLib=object)
    _stub("isaacgymenvs.tasks.interos.poselib")
    _stub("isaacgymenvs.tasks.interos.poselib.skeleton")
    _stub("isaacgymenvs.tasks.interos.poselib.skeleton.skeleton3d", SkeletonTree=object)
    _stub("tensorboardX", SummaryWriter=object)
    _stub("cv2")


def load_task(filename):
    """Return the module object for /root/reference/tasks/<filename>."""
    install_stubs()
    path = os.path.join(REF_ROOT, "tasks", filename)
    name = "ref_" + os.path.splitext(filename)[0]
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
