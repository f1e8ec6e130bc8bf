"""Loader of the HIP library `isaacgym_amd/lib/libppenv.so` (C ABI: include/ppenv.h).

There is no CPU fallback: if the library is missing or does not load, importing the
environment fails loudly.  `build()` compiles it with hipcc for gfx950 (cross-compiles
without a GPU); the built .so lives in-tree so it travels with the source snapshot.
"""
import ctypes as C
import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

from . import scene

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("PPENV_LIB", os.path.join(_PKG, "lib", "libppenv.so"))   # PPENV_LIB: profiling builds only
SOURCES = [os.path.join(_PKG, "csrc", "ppenv.hip"), os.path.join(_PKG, "csrc", "ppenv_ta.hip"), os.path.join(_PKG, "csrc", "ppenv_ta_sim.hip"),
           os.path.join(_PKG, "csrc", "ppenv_ta_chain.hip"), os.path.join(_PKG, "csrc", "ppenv_policy.hip"), os.path.join(_PKG, "csrc", "ppenv_policy_bwd.hip")]
HEADERS = [os.path.join(_PKG, "csrc", "ppenv_device.h"), os.path.join(_PKG, "csrc", "ppenv_model_g1.h"), os.path.join(_PKG, "csrc", "ppenv_ta_device.h"), os.path.join(_PKG, "csrc", "ppenv_ta_task.h"), os.path.join(_PKG, "csrc", "ppenv_ta_chain.h"), os.path.join(_PKG, "csrc", "ppenv_model_g1_ta.h"),
           os.path.join(ROOT, "include", "ppenv.h"), os.path.join(ROOT, "include", "ppenv_policy.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-fno-signed-zeros", "-ffinite-math-only", "-fPIC", "-shared"]

_lib = None


class PPEnvError(RuntimeError):
    pass


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force=False, verbose=False, out=None, extra_flags=(), extra_deps=()):
    """hipcc --offload-arch=gfx950 -> isaacgym_amd/lib/libppenv.so.  out / extra_flags: a build of the same sources for another
    compiled-in arm model (-DPPENV_MODEL_HEADER=..., see build_for_arm_model) — loaded with load(), never the default library."""
    out = LIB_PATH if out is None else out
    if not force and os.path.exists(out):
        t = os.path.getmtime(out)
        if not any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + list(extra_deps)):
            return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    tmp = f"{out}.{os.getpid()}.tmp"          # built aside and renamed: another process never sees half a library
    objdir = tempfile.mkdtemp(prefix="ppenv_build_")
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + compile_flags + ["-c", "-o", obj, src]
        return obj, cmd, subprocess.run(cmd, capture_output=True, text=True)
    try:
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:      # one translation unit per core
            done = list(pool.map(compile_one, SOURCES))
        for obj, cmd, res in done:
            if res.returncode != 0:
                raise PPEnvError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stderr[-4000:])
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + [obj for obj, _, _ in done]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise PPEnvError("hipcc (link) failed:\n" + " ".join(cmd) + "\n" + res.stderr[-4000:])
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
        shutil.rmtree(objdir, ignore_errors=True)
    if verbose:
        print(" ".join(cmd))
    return out


def build_for_arm_model(config, out_dir, force=False):
    """The library with ANOTHER 7-dof arm compiled in: `config` carries the model (scene.build_config after
    scene.use_arm_tables(urdf.arm_specs(...))); modelgen writes its header into out_dir and the same sources are built against it.
    -> path of out_dir/libppenv.so (load it with load(); ppenv_create of THAT library accepts the config, the default one refuses it)."""
    from . import modelgen
    os.makedirs(out_dir, exist_ok=True)
    header = os.path.join(out_dir, "ppenv_model.h")
    text = modelgen.generate(config)
    if not os.path.exists(header) or open(header).read() != text:
        with open(header, "w") as fh:
            fh.write(text)
    return build(force=force, out=os.path.join(out_dir, "libppenv.so"), extra_flags=[f'-DPPENV_MODEL_HEADER="{header}"'], extra_deps=[header])


def build_for_ta_model(model, out_dir, scene_cfg=None, force=False):
    """The library with ANOTHER 27-dof tree compiled into the chain-wave kernel (TA:470 `g1_27dof.urdf`): `model` is a
    scene.TAModel (isaacgym_amd.urdf.ta_model of the asset), `scene_cfg` the scene whose collision shapes ride on it (default
    scene.build_ta_scene).  modelgen_ta writes the tree's header into out_dir and the same sources are built against it
    (-DPPENV_TA_MODEL_HEADER=...): ppenv_ta_sim_create of THAT library then selects ta_chain_kernel for the model — the default
    library keeps its table-driven kernels for it (and refuses PPENV_TA_KERNEL=chain).  The tree must have the G1's topology: limbs of
    6 + 6 + 3 + 7 + 5 links on the pelvis / torso; anything else fails the kernel's static_asserts at compile time, not at run time.
    -> path of out_dir/libppenv.so (load()).  The 7-dof kernels of that library carry the stock arm."""
    from . import modelgen_ta
    os.makedirs(out_dir, exist_ok=True)
    header = os.path.join(out_dir, "ppenv_model_ta.h")
    text = modelgen_ta.generate(scene_cfg if scene_cfg is not None else scene.build_ta_scene(1), model)
    if not os.path.exists(header) or open(header).read() != text:
        with open(header, "w") as fh:
            fh.write(text)
    return build(force=force, out=os.path.join(out_dir, "libppenv.so"), extra_flags=[f'-DPPENV_TA_MODEL_HEADER="{header}"'], extra_deps=[header])


def lib():
    """The loaded library with argtypes set.  Raises PPEnvError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PPEnvError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback.")
    _lib = load(LIB_PATH)
    return _lib


def load(path):
    """A libppenv build with its argtypes set (lib() for the default one; build_for_arm_model's output for another arm)."""
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise PPEnvError(f"could not load {path}: {e}") from e
    if L.ppenv_abi_version() != scene.ABI_VERSION:
        raise PPEnvError("libppenv.so ABI version does not match isaacgym_amd.scene; rebuild the library")
    vp, sz = C.c_void_p, C.c_size_t
    cfgp = C.POINTER(scene.Config)
    L.ppenv_last_error.restype = C.c_char_p
    L.ppenv_arena_bytes.restype = sz
    L.ppenv_arena_bytes.argtypes = [cfgp]
    L.ppenv_create.argtypes = [cfgp, vp, sz, vp, C.POINTER(vp)]
    L.ppenv_destroy.restype = None
    L.ppenv_destroy.argtypes = [vp]
    L.ppenv_buffers_of.argtypes = [vp, C.POINTER(scene.Buffers)]
    L.ppenv_config_of.argtypes = [vp, cfgp]
    L.ppenv_step.argtypes = [vp, vp, vp]
    L.ppenv_step_into.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ppenv_step_sequence.argtypes = [vp, vp, C.c_int32, vp]
    L.ppenv_reset_all.argtypes = [vp, vp]
    L.ppenv_reduce_stats.argtypes = [vp, vp, vp]
    L.ppenv_reset_idx.argtypes = [vp, vp, C.c_int32, C.c_int, vp]
    L.ppenv_pd_targets.argtypes = [vp, vp, vp, vp]
    L.ppenv_serve_from_draws.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.ppenv_set_randomization.argtypes = [vp, C.POINTER(scene.Randomization)]
    L.ppenv_set_gravity.argtypes = [vp, C.c_float]
    L.ppenv_status.restype = C.c_uint32
    L.ppenv_status.argtypes = [vp]
    L.ppenv_step_kernel_name.restype = C.c_char_p
    L.ppenv_step_kernel_name.argtypes = [vp]
    L.ppenv_ta_sim_set_gravity.argtypes = [vp, C.c_float, vp]
    L.ppenv_ta_sim_kernel_name.restype = C.c_char_p
    L.ppenv_ta_sim_kernel_name.argtypes = [vp]
    L.ppenv_ta_pd_targets.argtypes = [vp, C.c_int32, vp, vp, vp]
    L.ppenv_ta_serve_from_draws.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.ppenv_ta_sim_device.argtypes = [vp]
    L.ppenv_ta_sim_status.restype = C.c_uint32
    L.ppenv_ta_sim_status.argtypes = [vp]
    L.ppenv_ta_sim_kernel.argtypes = [vp]
    L.ppenv_ta_sim_set_policy_input.argtypes = [vp, vp, vp, C.c_float, vp, C.c_int32]
    L.ppenv_ta_sim_set_randomization.argtypes = [vp, C.POINTER(scene.Randomization)]
    L.ppenv_ta_model_is_compiled.argtypes = [cfgp, C.POINTER(scene.TAModel)]
    L.ppenv_post_physics_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    for name in ("ppenv_refresh_root_states", "ppenv_refresh_dof_states", "ppenv_refresh_dof_force",
                 "ppenv_refresh_rigid_body_states"):
        getattr(L, name).argtypes = [vp, vp, vp]
    L.ppenv_set_serve_override.argtypes = [vp, vp, C.c_int, vp]
    L.ppenv_ta_post_physics_step.argtypes = [C.POINTER(scene.TAParams)] + [vp] * 15
    L.ppenv_t4_rewards.argtypes = [C.POINTER(scene.T4Params)] + [vp] * 15
    L.ppenv_ta_sim_create.argtypes = [cfgp, C.POINTER(scene.TAModel), vp, C.POINTER(vp)]
    L.ppenv_ta_sim_destroy.restype = None
    L.ppenv_ta_sim_destroy.argtypes = [vp]
    L.ppenv_ta_simulate.argtypes = [vp, C.c_int32] + [vp] * 7
    L.ppenv_ta_forward_kinematics.argtypes = [vp, C.c_int32] + [vp] * 4
    L.ppenv_ta_step.argtypes = [vp, C.POINTER(scene.TAParams)] + [vp] * 16
    L.ppenv_state_bytes.restype = sz
    L.ppenv_state_bytes.argtypes = [vp]
    L.ppenv_get_state.argtypes = [vp, vp, sz]
    L.ppenv_set_state.argtypes = [vp, vp, sz]
    return L


def check(rc, L=None):
    if rc != 0:
        raise PPEnvError(f"ppenv error {rc}: {(L if L is not None else lib()).ppenv_last_error().decode()}")
