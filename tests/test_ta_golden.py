"""27-DoF variant (tasks/humanoid_pingpong_3_actor_all_dof.py, "TA"), tensor-API mode: the oracle and the HIP
kernel against the reference's own post_physics_step (TA:1145-1192) outputs in tests/golden/post_physics_TA.npz."""
import numpy as np
import pytest

from helpers import GOLDEN_DIR, assert_close
from isaacgym_amd import scene


def load():
    return np.load(f"{GOLDEN_DIR}/post_physics_TA.npz")


def params_for(g):
    n = g["out_rew"].shape[1]
    return scene.build_ta_params(n, env=dict(episodeLength=int(g["episode_length"])))


def check_step(g, t, obs, rew, reset, progress, flags, root, dof):
    np.testing.assert_array_equal(reset, g["out_reset"][t], err_msg=f"reset, step {t}")
    np.testing.assert_array_equal(progress, g["out_progress"][t], err_msg=f"progress, step {t}")
    np.testing.assert_array_equal(flags, g["out_flags"][t], err_msg=f"flags, step {t}")
    assert_close(rew, g["out_rew"][t], f"rew, step {t}", atol=2e-3)      # rewards reach 2e4: atol = 1e-4 x 20
    assert_close(obs, g["out_obs"][t], f"obs, step {t}", atol=2e-5)
    assert_close(root, g["out_root"][t], f"root after reset, step {t}", rtol=0, atol=1e-7)
    assert_close(dof, g["out_dof"][t], f"dof after reset, step {t}", rtol=0, atol=0)


def test_oracle_ta_matches_reference(oracle_lib):
    g = load()
    p = params_for(g)
    T, n = g["out_rew"].shape
    irb = np.ascontiguousarray(np.broadcast_to(g["initial_bodies42"], (n, 42, 13)), np.float32)
    flags, episode, progress = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int64)
    for t in range(T):
        root, dof = g["in_root"][t].copy(), g["in_dof"][t].copy()
        obs, rew, reset = oracle_lib.ta_post_physics_step(p, np.ascontiguousarray(g["in_bodies42"][t]), irb, root, dof,
                                                          g["in_dof_force"][t].copy(), g["in_pre_vx"][t].copy(),
                                                          np.nan_to_num(g["reset_override"][t]), flags, episode, progress)
        check_step(g, t, obs, rew, reset, progress, flags, root, dof)
    assert g["out_reset"].sum() > 40 and (g["out_flags"] & 0x1F0).any()


@pytest.mark.gpu
def test_hip_ta_matches_reference():
    import torch
    from isaacgym_amd import tensor_api
    g = load()
    p = params_for(g)
    T, n = g["out_rew"].shape
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    irb = dev(np.broadcast_to(g["initial_bodies42"], (n, 42, 13)).astype(np.float32))
    st = tensor_api.TAState(p, device="cuda:0")
    for t in range(T):
        root, dof = dev(g["in_root"][t]), dev(g["in_dof"][t])
        st.post_physics_step(dev(g["in_bodies42"][t]), irb, root, dof, dev(g["in_dof_force"][t]), dev(g["in_pre_vx"][t]),
                             reset_override=dev(np.nan_to_num(g["reset_override"][t])))
        check_step(g, t, st.obs_buf.cpu().numpy(), st.rew_buf.cpu().numpy(), st.reset_buf.cpu().numpy(), st.progress_buf.cpu().numpy(),
                   st.flags.cpu().numpy().view(np.uint32), root.cpu().numpy(), dof.cpu().numpy())
