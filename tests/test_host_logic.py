"""Host-side logic and the C-ABI surface; no GPU needed."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from isaacgym_amd import _lib, modelgen, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    """include/ppenv.h is the contract: every function it declares must be exported by the built .so."""
    _lib.build()
    L = C.CDLL(_lib.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "ppenv.h")).read() + open(os.path.join(ROOT, "include", "ppenv_policy.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(ppenv_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 17, declared
    for name in declared:
        assert hasattr(L, name), f"{name} declared in ppenv.h but not exported"
    assert L.ppenv_abi_version() == scene.ABI_VERSION
    # no torch / C++ types leak through the ABI: the exported ppenv_* names are unmangled C symbols
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(declared) <= exported


def test_config_struct_layout_matches_the_c_header():
    """ctypes mirror vs the C compiler's view of ppenv_config (size and a few offsets)."""
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "ppenv.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ppenv_config), sizeof(ppenv_joint), offsetof(ppenv_config, joint),
               offsetof(ppenv_config, obs_body), offsetof(ppenv_config, table), offsetof(ppenv_config, shape),
               offsetof(ppenv_config, max_episode_length));
        return 0;
    }'''
    exe = os.path.join(ROOT, "tests", "csrc", "layout_probe")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    got = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    os.remove(exe)
    want = [C.sizeof(scene.Config), C.sizeof(scene.Joint), scene.Config.joint.offset, scene.Config.obs_body.offset,
            scene.Config.table.offset, scene.Config.shape.offset, scene.Config.max_episode_length.offset]
    assert got == want


def test_generated_model_header_is_current():
    assert modelgen.is_current(), "isaacgym_amd/csrc/ppenv_model_g1.h is stale: run `python -m isaacgym_amd.modelgen`"


def test_generated_27dof_model_header_is_current_and_equals_the_runtime_tables():
    """csrc/ppenv_model_g1_ta.h (compiled into the chain-wave kernel) is what modelgen_ta writes today, and its values equal, bit for
    bit, what the library derives from the run-time model (make_ta_consts) — otherwise ppenv_ta_step would silently fall back."""
    from isaacgym_amd import modelgen_ta
    assert modelgen_ta.is_current(), "isaacgym_amd/csrc/ppenv_model_g1_ta.h is stale: run `python -m isaacgym_amd.modelgen_ta` and rebuild"
    L = _lib.lib()
    cfg, m = scene.build_ta_scene(4), scene.build_ta_model()
    assert L.ppenv_ta_model_is_compiled(C.byref(cfg), C.byref(m)) == 1
    m.link[5].mass *= 1.5                       # any other model: not the compiled one (the table-driven kernels serve it)
    assert L.ppenv_ta_model_is_compiled(C.byref(cfg), C.byref(m)) == 0


def test_reference_constants_in_config():
    """Constants the reference states verbatim."""
    tt = scene.build_config("TT", num_envs=4)
    assert tt.max_episode_length == 140 and tt.substeps == 2                      # HumanoidPingpongTiltG1.yaml:10,80
    assert abs(tt.dt - 0.0083) < 1e-9 and abs(tt.gravity_z + 9.8) < 1e-6          # yaml:78, TT:331
    assert list(tt.ball_init_pos) == pytest.approx([3.15, -0.28, 1.1])            # TT:622
    assert [tt.joint[j].kp for j in range(7)] == [20, 20, 20, 20, 20, 5, 5]       # TT:694-709
    assert [tt.joint[j].kd for j in range(7)] == pytest.approx([0.5] * 5 + [0.125] * 2)   # TT:711
    assert (tt.alpha_velocity_reward, tt.penalty) == (50.0, -200.0)               # yaml:15,19
    assert tt.power_coefficient == pytest.approx(0.0005)
    t3 = scene.build_config("T3", num_envs=4)
    assert list(t3.humanoid_root_quat) == pytest.approx([0, 0, -0.2588, 0.9659], abs=1e-4)   # T3:504-506
    assert t3.max_episode_length == 64 and abs(t3.dt - 0.0166) < 1e-9             # HumanoidPingpongG1.yaml:10,66
    assert (t3.serve_speed_lo, t3.serve_speed_hi) == (6.5, 7.5)
    tn = scene.build_config("TN", num_envs=4)
    assert tn.max_episode_length == 170 and tn.alpha_velocity_reward == 1000.0    # NoEarlyStop yaml:10,15
    assert list(tn.ball_init_pos) == pytest.approx([2.9, -0.28, 1.02])            # TN:628
    off, scale = scene.pd_action_offset_scale(tt)                                 # TT:664-665
    for j in range(7):
        assert off[j] == pytest.approx(0.5 * (tt.joint[j].upper + tt.joint[j].lower))
        assert scale[j] == pytest.approx(0.5 * (tt.joint[j].upper - tt.joint[j].lower))
    # restitution above 1 is clamped before combining (documented design parameter restitution_max)
    assert tt.table.restitution == pytest.approx(1.0) and tt.paddle_restitution == pytest.approx(0.8)


def test_bad_cfg_raises_like_the_reference():
    cfg = scene.default_task_cfg("TT")
    del cfg["env"]["hitTableReward"]        # the TT yaml lacks this key; the reference raises KeyError (TT:106)
    with pytest.raises(KeyError):
        scene.build_config("TT", cfg=cfg)
    cfg = scene.default_task_cfg("TT")
    cfg["env"]["bodyStatesId"] = list(range(30))
    with pytest.raises(ValueError):
        scene.build_config("TT", cfg=cfg)


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ or the tests' host shim."""
    roots = [os.path.join(ROOT, d) for d in ("isaacgym_amd", "isaacgymenvs", "isaacgym", "include")]
    for dirpath, _, files in (x for r in roots for x in os.walk(r)):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libppenv_oracle" not in text and "host_shim" not in text.replace("tests/csrc/host_shim.cpp", ""), f


def test_env_creation_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import isaacgym_amd
    with pytest.raises(_lib.PPEnvError):
        isaacgym_amd.make(task="HumanoidPingpongTiltG1", num_envs=8)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libppenv.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.PPEnvError, match="not been built"):
        _lib.lib()


def test_oracle_physics_known_answers(oracle_lib):
    """Analytic checks of the physics specification the oracle restates."""
    cfg = scene.build_config("TT", num_envs=1, seed=0)
    # (1) inverse -> forward dynamics round trip, and RK4 energy conservation of the free arm
    rng = np.random.default_rng(0)
    q, qd, qdd = rng.uniform(-1, 1, 7), rng.uniform(-2, 2, 7), rng.uniform(-5, 5, 7)
    tau = oracle_lib.arm_inverse_dynamics(cfg, q, qd, qdd)
    assert np.abs(oracle_lib.arm_qdd(cfg, q, qd, tau, np.zeros(7)) - qdd).max() < 1e-9
    f = lambda q_, qd_: oracle_lib.arm_qdd(cfg, q_, qd_, np.zeros(7), np.zeros(7))
    e0, h = oracle_lib.arm_energy(cfg, q, qd), 2e-4
    for _ in range(500):
        k1q, k1v = qd, f(q, qd)
        k2q, k2v = qd + 0.5 * h * k1v, f(q + 0.5 * h * k1q, qd + 0.5 * h * k1v)
        k3q, k3v = qd + 0.5 * h * k2v, f(q + 0.5 * h * k2q, qd + 0.5 * h * k2v)
        k4q, k4v = qd + h * k3v, f(q + h * k3q, qd + h * k3v)
        q, qd = q + h / 6 * (k1q + 2 * k2q + 2 * k3q + k4q), qd + h / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
    assert abs(oracle_lib.arm_energy(cfg, q, qd) - e0) < 1e-5 * abs(e0)
    # (2) ball free flight: z(t) under g = -9.8 with semi-implicit Euler micro-steps matches the closed form
    env = oracle_lib.OracleEnv(scene.build_config("TT", num_envs=1, seed=0))
    env.ball[:, 0] = [2.0, 0.0, 1.5, 0, 0, 0, 1, 0.3, 0.1, 0.2, 0, 0, 0]
    steps = 10
    for _ in range(steps):
        env.step(np.zeros((1, 7), np.float32))
    hb = cfg.dt / cfg.substeps / cfg.ball_substeps
    k = steps * cfg.substeps * cfg.ball_substeps
    z = 1.5 + 0.2 * k * hb - 9.8 * hb * hb * k * (k + 1) / 2
    assert abs(float(env.ball[2, 0]) - z) < 1e-5 and abs(float(env.ball[0, 0]) - (2.0 + 0.3 * k * hb)) < 1e-5
    # (3) drop on the table: restitution e = 1.0 (clamped) returns the ball to (almost) its drop height
    env.ball[:, 0] = [2.6, 0.0, 0.98, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    zs = []
    for _ in range(60):
        env.step(np.zeros((1, 7), np.float32))
        zs.append(float(env.ball[2, 0]))
    low = int(np.argmin(zs))
    assert 0.775 < zs[low] < 0.80                      # bounced at table top (0.76) + radius (0.02)
    assert max(zs[low:]) > 0.95                        # e = 1: rises back near 0.98
    # (4) PD drive: zero action holds the arm near the mid-range targets at rest (steady state, gravity droop small)
    env2 = oracle_lib.OracleEnv(scene.build_config("TN", num_envs=1, seed=0))
    for _ in range(2000):   # lightly damped (Kd = Kp/40): give the ringing time to decay
        env2.step(np.zeros((1, 7), np.float32))
    off, _ = scene.pd_action_offset_scale(cfg)
    assert np.abs(env2.dof_vel[:, 0]).max() < 0.02
    assert np.abs(env2.dof_pos[:, 0] - off).max() < 0.35   # Kp = 20: gravity torque / Kp of droop at most


def _ball_env(oracle_lib, variant="TN", **kw):
    cfg = scene.build_config(variant, num_envs=1, seed=0, **kw)
    env = oracle_lib.OracleEnv(cfg)
    return cfg, env


def test_contact_model_known_answers(oracle_lib):
    """Closed-form checks of the ball contact specification through the oracle."""
    zero = np.zeros((1, 7), np.float32)
    # (a) oblique bounce on the table: e = 1 (clamped) reverses vz; Coulomb friction mu = 0.3 (avg of 0.2, 0.4 in TT)
    #     slows vx by mu * jn while sliding and spins the ball up about +y by (jt / (k r)).
    cfg, env = _ball_env(oracle_lib, "TT")
    vx0, vz0 = 3.0, -2.0
    env.ball[:, 0] = [2.6, 0.0, 0.76 + 0.02 + 0.004, 0, 0, 0, 1, vx0, 0.0, vz0, 0, 0, 0]
    env.step(zero)
    vx, vz, wy = float(env.ball[7, 0]), float(env.ball[9, 0]), float(env.ball[11, 0])
    assert vz > 1.9                                          # bounced with e = 1 (minus a little gravity)
    hb = cfg.dt / cfg.substeps / cfg.ball_substeps
    damp = (1 - cfg.ball_angular_damping * hb)
    jn_lo, jn_hi = 2 * 2.0, 2 * (2.0 + 9.8 * 0.0083)          # normal impulse per unit mass, bracketing the gravity kick
    mu = cfg.table.friction
    assert vx0 - mu * jn_hi - 1e-3 <= vx <= vx0 - mu * jn_lo + 1e-3      # slipping: tangential impulse = mu * jn
    jt = vx0 - vx
    k, r = cfg.ball_inertia_factor, cfg.ball_radius
    # spin about +y of magnitude jt / (k r), then a few micro-steps of angular damping
    assert abs(wy) == pytest.approx(jt / (k * r), rel=0.02) and wy > 0
    # (b) slow oblique contact sticks: the stick impulse s k/(1+k) leaves v' = v/(1+k) (3/5 for a thin shell, the classic
    #     5/7 for a solid sphere) and the ball rolls without slipping afterwards
    cfg, env = _ball_env(oracle_lib, "TT")
    env.ball[:, 0] = [2.6, 0.0, 0.76 + 0.02 + 0.0005, 0, 0, 0, 1, 0.05, 0.0, -3.0, 0, 0, 0]
    env.step(zero)
    vx, wy = float(env.ball[7, 0]), float(env.ball[11, 0])
    assert vx == pytest.approx(0.05 / (1 + k), rel=0.03)
    assert vx - wy * r == pytest.approx(0.0, abs=2e-3)              # rolling without slipping: v = w r
    # (c) below the bounce threshold (0.2 m/s) restitution is off: the ball stays on the table
    cfg, env = _ball_env(oracle_lib, "TN")
    env.ball[:, 0] = [2.6, 0.0, 0.76 + 0.02 + 0.0001, 0, 0, 0, 1, 0.0, 0.0, -0.1, 0, 0, 0]
    for _ in range(20):
        env.step(zero)
    assert abs(float(env.ball[2, 0]) - 0.78) < 2e-3 and abs(float(env.ball[9, 0])) < 0.06
    # (d) the net stops a low ball: e = 0.5 against the default-material net, vx changes sign
    cfg, env = _ball_env(oracle_lib, "TT")
    env.ball[:, 0] = [1.80, 0.0, 0.85, 0, 0, 0, 1, -6.0, 0.0, 0.0, 0, 0, 0]
    for _ in range(3):
        env.step(zero)
    assert float(env.ball[7, 0]) == pytest.approx(0.5 * 6.0, rel=0.05) and float(env.ball[0, 0]) > 1.75
    # (e) a ball above the net height flies over it
    cfg, env = _ball_env(oracle_lib, "TT")
    env.ball[:, 0] = [1.80, 0.0, 1.0, 0, 0, 0, 1, -6.0, 0.0, 0.0, 0, 0, 0]
    for _ in range(3):
        env.step(zero)
    assert float(env.ball[7, 0]) == pytest.approx(-6.0) and float(env.ball[0, 0]) < 1.7


def test_paddle_contact_known_answer(oracle_lib):
    """A ball thrown at the resting paddle face comes back with the combined restitution 0.8 (ball 1.0 clamped, humanoid 0.6)."""
    cfg = scene.build_config("TN", num_envs=1, seed=0)
    env = oracle_lib.OracleEnv(cfg)
    zero = np.zeros((1, 7), np.float32)
    for _ in range(2500):            # let the arm settle at its mid-range pose
        env.step(zero)
    rb = env.refresh_rigid_body_states()[0]
    paddle_pos, paddle_q = rb[39, 0:3].astype(np.float64), rb[39, 3:7].astype(np.float64)
    n_local = np.array(list(cfg.paddle_normal), np.float64)
    normal = scene.quat_to_rot(paddle_q) @ n_local
    speed = 4.0
    start = paddle_pos + normal * 0.08
    env.ball[:, 0] = list(start) + [0, 0, 0, 1] + list(-normal * speed) + [0, 0, 0]
    blob = env.get_state()
    vn = []
    for _ in range(6):
        env.step(zero)
        vn.append(float(np.dot(env.ball[7:10, 0].astype(np.float64), normal)))
    assert min(vn) < -3.5                                    # approached ...
    assert max(vn) == pytest.approx(cfg.paddle_restitution * speed, rel=0.12)   # ... and left with e = 0.8 (gravity perturbs slightly)
    assert cfg.paddle_restitution == pytest.approx(0.8)


def test_policy_entry_points_reject_bad_arguments_without_a_gpu():
    """include/ppenv_policy.h: every entry validates its arguments before any device call — NULL pointers, strides shorter than the
    rows, unaligned or odd output strides — and reports PPENV_EINVAL with a message (no compute happens here: there is no GPU)."""
    import ctypes as C
    from isaacgym_amd import _lib
    from isaacgym_amd.policy import MLPLayer, _lib_policy
    L = _lib_policy()
    EINVAL = -1
    d = MLPLayer()
    assert L.ppenv_mlp_layer_forward(C.byref(d), None) == EINVAL                       # all NULL
    d.m, d.n, d.k, d.batch = 8, 8, 16, 1
    d.in_, d.w, d.out = 0x1000, 0x2000, 0x3000                                          # never dereferenced: the sizes fail first
    d.lda, d.ldw, d.ldo = 8, 16, 8                                                      # lda < k
    assert L.ppenv_mlp_layer_forward(C.byref(d), None) == EINVAL
    assert b"lda >= k" in L.ppenv_last_error()
    f = C.c_void_p(0x1000)
    assert L.ppenv_mlp_prepare_input(None, 4, 8, 8, None, None, 5.0, f, 8, None) == EINVAL
    assert L.ppenv_mlp_prepare_input(f, 4, 8, 8, None, None, 5.0, f, 12, None) == EINVAL      # ld_out not a multiple of 8
    assert L.ppenv_mlp_prepare_input(f, 4, 8, 8, f, None, 5.0, f, 8, None) == EINVAL          # mean without inv_std
    assert L.ppenv_mlp_sample_actions(f, 4, 300, 300, f, 0, 0, -1.0, 1.0, f, None, None) == EINVAL   # more than 256 actions
    assert L.ppenv_mlp_sample_actions(f, 4, 8, 4, f, 0, 0, -1.0, 1.0, f, None, None) == EINVAL       # ld_mu < a
    d.lda, d.ldw, d.ldo, d.out_f32, d.n = 16, 16, 40, 1, 40                              # a heads layer wider than the skinny kernel takes
    assert L.ppenv_mlp_heads_sample(C.byref(d), 7, f, 0, 0, -1.0, 1.0, f, None, None) == EINVAL
    assert L.ppenv_gae(f, f, 1, 8, f, 0, 8, 0.99, 0.95, 1.0, f, f, None) == EINVAL                 # horizon 0
    L.ppenv_ta_sim_set_policy_input.restype = C.c_int
    assert L.ppenv_ta_sim_set_policy_input(None, None, None, 5.0, None, 0) == EINVAL                 # NULL handle
