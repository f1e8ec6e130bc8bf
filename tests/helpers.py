"""Shared helpers for the parity tests."""
import os

import numpy as np

from isaacgym_amd import scene

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BODY_IDS = [0, 31, 32, 33, 34, 35, 36, 37, 38, 39]

# fp32 tolerances of the parity bar (BASELINE.json north_star: "fp32, rtol 1e-4")
RTOL = 1e-4
ATOL = 1e-5


def load_golden(variant):
    return np.load(os.path.join(GOLDEN_DIR, f"post_physics_{variant}.npz"))


def golden_config(variant, g, **kw):
    """The task config the golden fixture was generated with (tools/gen_golden.py)."""
    cfg = scene.default_task_cfg(variant)
    cfg["env"]["episodeLength"] = int(g["episode_length"])
    n = g["out_rew"].shape[1]
    return scene.build_config(variant, cfg=cfg, num_envs=n, **kw)


def expand_bodies(compact):
    """[N,10,13] observed rows -> the reference's [N,42,13] rigid-body tensor (other rows zero)."""
    n = compact.shape[0]
    full = np.zeros((n, scene.NUM_BODIES, 13), np.float32)
    full[:, BODY_IDS, :] = compact
    return full


def assert_close(actual, expected, what, rtol=RTOL, atol=ATOL):
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    err = np.abs(actual - expected)
    tol = atol + rtol * np.abs(expected)
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.size} beyond rtol={rtol} atol={atol}; worst at {idx}: "
                             f"got {actual[idx]!r} want {expected[idx]!r}")
