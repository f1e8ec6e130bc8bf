"""The CPU oracle against the reference's own outputs (tests/golden, made by tools/gen_golden.py
from the reference's post_physics_step: TT:1022-1052 and the T3 / TN counterparts)."""
import numpy as np
import pytest

from helpers import assert_close, expand_bodies, golden_config, load_golden


@pytest.mark.parametrize("variant", ["TT", "TN", "T3"])
def test_oracle_post_physics_matches_reference(oracle_lib, variant):
    g = load_golden(variant)
    env = oracle_lib.OracleEnv(golden_config(variant, g))
    T = g["out_rew"].shape[0]
    n_resets = 0
    for t in range(T):
        serve = np.nan_to_num(g["serve"][t], nan=0.0)
        env.set_serve_override(serve, on=True)
        root = g["in_root"][t].copy()
        dof = g["in_dof"][t].copy()
        env.post_physics_step(expand_bodies(g["in_bodies"][t]), root, dof, g["in_dof_force"][t].copy(),
                              g["in_pre_vx"][t].copy())
        np.testing.assert_array_equal(env.reset_buf, g["out_reset"][t], err_msg=f"reset_buf, step {t}")
        np.testing.assert_array_equal(env.progress_buf, g["out_progress"][t], err_msg=f"progress_buf, step {t}")
        np.testing.assert_array_equal(env.flags, g["out_flags"][t], err_msg=f"flags, step {t}")
        assert_close(env.rew_buf, g["out_rew"][t], f"rew_buf, step {t}")
        assert_close(env.obs_buf, g["out_obs"][t], f"obs_buf, step {t}")
        assert_close(root, g["out_root"][t], f"root_states after reset, step {t}", rtol=0, atol=1e-7)
        assert_close(dof, g["out_dof"][t], f"dof_states after reset, step {t}", rtol=0, atol=0)
        n_resets += int(env.reset_buf.sum())
    assert n_resets > 50  # the fixture exercises the reset path


def test_golden_covers_every_reward_branch():
    """The TT fixture must hit each branch of TT:1105-1270 (otherwise the pin proves little)."""
    g = load_golden("TT")
    ball, pre_vx = g["in_root"][:, :, 2, :], g["in_pre_vx"]
    bx, by, bz, vx = ball[..., 0], ball[..., 1], ball[..., 2], ball[..., 7]
    bounce = (bz < 0.83) & (vx > 0) & (np.abs(by) < 0.6)
    assert ((pre_vx < 0) & (vx > 0)).sum() > 10          # paddle return
    assert (bx < -0.05).sum() > 10                       # missed ball
    assert ((bx < 2.44) & bounce).sum() > 10             # early bounce
    assert ((bx > 2.44) & (bx < 3.1) & bounce).sum() > 5  # good bounce
    assert ((bx >= 3.1) & (vx > 0)).sum() > 10           # overshoot
    net = (bx > 1.7) & (bx < 1.8) & (vx > 0) & (np.abs(by) < 0.4) & (bz > 0.98) & (bz < 1.14)
    assert net.sum() > 3                                 # net-crossing bonus
    assert (bz < 0.1).sum() > 5                          # ball on the floor
    assert (g["out_rew"] > 1500).sum() > 3               # hitTableReward actually paid
    assert (g["out_reset"][(g["in_root"][:, :, 2, 2] >= 0.1)] == 1).sum() > 20   # time-outs
