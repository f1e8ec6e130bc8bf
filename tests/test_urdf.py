"""URDF -> model tables (isaacgym_amd/urdf.py).  The real assets (TT:415, TA:470) are not available; the fixture
tests/golden/g1_27dof_placeholder.urdf is the placeholder model written out by urdf.write_g1_urdf (data, not reference source),
so these tests pin the importer — tree walking, merging of welded bodies, frames, limits — against the hand-built tables."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR
from isaacgym_amd import scene, urdf

FIXTURE = os.path.join(GOLDEN_DIR, "g1_27dof_placeholder.urdf")


def _fields_equal(a, b, struct):
    for name, _ in struct._fields_:
        x, y = getattr(a, name), getattr(b, name)
        if hasattr(x, "__len__"):
            x, y = np.array(x[:] if not hasattr(x[0], "__len__") else [list(r) for r in x]), np.array(y[:] if not hasattr(y[0], "__len__") else [list(r) for r in y])
            if not np.array_equal(x, y):
                return name
        elif x != y:
            return name
    return None


def test_fixture_is_the_written_placeholder_model():
    assert open(FIXTURE).read() == urdf.write_g1_urdf()


def test_27dof_tree_from_urdf_equals_the_hand_built_model():
    robot = urdf.load(FIXTURE)
    assert robot.root() == "pelvis" and len(robot.links) == 40 and len(robot.joints) == 39
    assert sum(j.type == "revolute" for j in robot.joints.values()) == 27
    got = urdf.ta_model(robot, urdf.ta_dof_joint_names(), urdf.G1_BODY_NAMES)
    want = scene.build_ta_model()
    for i in range(scene.TA_NUM_LINKS):
        assert _fields_equal(got.link[i], want.link[i], scene.TALink) is None, (i, _fields_equal(got.link[i], want.link[i], scene.TALink))
    for k in range(scene.TA_NUM_FIXED):
        assert _fields_equal(got.fixed[k], want.fixed[k], scene.TAFixed) is None, k
    assert got.num_contacts == want.num_contacts and list(got.contact_link) == list(want.contact_link)
    assert got.bound_link == want.bound_link and got.ground_z == want.ground_z


def test_7dof_chain_from_urdf_equals_the_arm_tables():
    robot = urdf.parse(urdf.write_g1_urdf(weld_right_elbow=False))
    names = [f"right_{n}_joint" for n in ("shoulder_pitch", "shoulder_roll", "shoulder_yaw", "elbow", "wrist_roll", "wrist_pitch", "wrist_yaw")]
    specs = urdf.arm_specs(robot, names, {n: i for i, n in enumerate(urdf.G1_BODY_NAMES)})
    for got, want in zip(specs, scene.G1_RIGHT_ARM):
        for k in ("xyz", "rpy", "limits", "com", "inertia"):
            np.testing.assert_array_equal(np.asarray(got[k], float), np.asarray(want[k], float), err_msg=f"{want['name']} {k}")
        assert (got["name"], got["body"], got["axis"], got["mass"], got["effort"], got["vel"]) == \
               (want["name"], want["body"], want["axis"], want["mass"], want["effort"], want["vel"])
    # feeding them back leaves the C config bit-identical (what ppenv_create checks against the compiled-in model)
    before = bytes(scene.build_config("TT", num_envs=4))
    saved = scene.G1_RIGHT_ARM
    try:
        scene.use_arm_tables(specs)
        assert bytes(scene.build_config("TT", num_envs=4)) == before
    finally:
        scene.use_arm_tables(saved)


def test_importer_rejects_what_the_kernels_cannot_represent():
    text = urdf.write_g1_urdf()
    with pytest.raises(ValueError, match="axis"):
        urdf.ta_model(urdf.parse(text.replace('<axis xyz="0 0 1"/>', '<axis xyz="0 0.6 0.8"/>', 1)), urdf.ta_dof_joint_names(), urdf.G1_BODY_NAMES)
    with pytest.raises(ValueError, match="does not exist"):
        urdf.parse(text.replace('<parent link="torso_link"/>', '<parent link="nowhere"/>', 1))
    with pytest.raises(ValueError, match="not supported"):
        urdf.parse(text.replace('type="revolute"', 'type="prismatic"', 1))
    names = urdf.ta_dof_joint_names()
    with pytest.raises(ValueError, match="not one of the 27 dofs"):
        urdf.ta_model(urdf.parse(urdf.write_g1_urdf(weld_right_elbow=False)), names, urdf.G1_BODY_NAMES)   # a 29-dof arm under a 27-dof list
