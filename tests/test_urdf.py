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


# A second, hand-written asset: every expected number below is typed here, not generated from scene.py.
HAND_URDF = """<?xml version="1.0"?>
<robot name="toy">
  <link name="base">
    <inertial><origin xyz="0 0 0.1"/><mass value="2.0"/><inertia ixx="0.02" iyy="0.03" izz="0.04" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0 0 0.05"/><geometry><sphere radius="0.12"/></geometry></collision>
  </link>
  <link name="upper">
    <inertial><origin xyz="0.1 0 0"/><mass value="1.0"/><inertia ixx="0.001" iyy="0.01" izz="0.01" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0.15 0 0" rpy="0 1.5707963267948966 0"/><geometry><cylinder radius="0.03" length="0.2"/></geometry></collision>
  </link>
  <link name="foot">
    <inertial><origin xyz="0.03 0 -0.02"/><mass value="0.5"/><inertia ixx="0.0004" iyy="0.0009" izz="0.001" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0.04 0 -0.03"/><geometry><box size="0.2 0.08 0.02"/></geometry></collision>
  </link>
  <link name="blade">
    <inertial><origin xyz="0 0 0"/><mass value="0.1"/><inertia ixx="0.0001" iyy="0.0001" izz="0.0002" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0 0.01 0" rpy="1.5707963267948966 0 0"/><geometry><cylinder radius="0.075" length="0.012"/></geometry></collision>
  </link>
  <joint name="j1" type="revolute"><origin xyz="0 0.1 0.2"/><parent link="base"/><child link="upper"/><axis xyz="0 1 0"/><limit lower="-1" upper="2" effort="30" velocity="10"/></joint>
  <joint name="j2" type="revolute"><origin xyz="0.3 0 0"/><parent link="upper"/><child link="foot"/><axis xyz="1 0 0"/><limit lower="-0.5" upper="0.5" effort="20" velocity="8"/></joint>
  <joint name="weld" type="fixed"><origin xyz="0.05 0 0.02"/><parent link="foot"/><child link="blade"/></joint>
</robot>
"""


def test_collision_geometry_of_a_hand_written_urdf():
    """<collision> primitives -> ball shapes, paddle blade and ground-contact points; the expected tables are typed in, the asset is
    not derived from scene.py."""
    robot = urdf.parse(HAND_URDF)
    movable = ["base", "upper", "foot"]
    assert [c.kind for c in robot.links["foot"].collisions] == ["box"] and robot.links["upper"].collisions[0].size == (0.03, 0.2)
    shapes = urdf.ball_shapes(robot, movable, ["base", "upper"])
    assert [s["link"] for s in shapes] == [0, 1]
    np.testing.assert_allclose(shapes[0]["a"], (0, 0, 0.05), atol=1e-12)
    np.testing.assert_allclose(shapes[0]["b"], (0, 0, 0.05), atol=1e-12)                 # a sphere: a == b
    assert shapes[0]["radius"] == 0.12
    # the cylinder's axis is its local z, pitched by 90 degrees onto the link's x: ends at x = 0.15 -+ 0.1
    np.testing.assert_allclose(shapes[1]["a"], (0.05, 0, 0), atol=1e-12)
    np.testing.assert_allclose(shapes[1]["b"], (0.25, 0, 0), atol=1e-12)
    assert shapes[1]["radius"] == 0.03
    # the blade hangs on the foot through a fixed joint: centre = weld offset + collision origin, normal = the cylinder's axis (local z
    # rolled by +90 degrees about x: z -> -y), half thickness = length / 2
    blade = urdf.paddle_blade(robot, movable, "blade")
    assert blade["link"] == 2 and blade["radius"] == 0.075 and abs(blade["half_thickness"] - 0.006) < 1e-15
    np.testing.assert_allclose(blade["center"], (0.05, 0.01, 0.02), atol=1e-12)
    np.testing.assert_allclose(blade["normal"], (0, -1, 0), atol=1e-12)
    # ground contacts: the bottom face of the foot box (centre (0.04, 0, -0.03), size 0.2 x 0.08 x 0.02), then the sphere's low point
    pts = urdf.ground_contacts(robot, movable, ["foot", "base"])
    assert [li for li, _ in pts] == [2, 2, 2, 2, 0]
    np.testing.assert_allclose([p for _, p in pts[:4]], [(-0.06, -0.04, -0.04), (-0.06, 0.04, -0.04), (0.14, -0.04, -0.04), (0.14, 0.04, -0.04)], atol=1e-12)
    np.testing.assert_allclose(pts[4][1], (0, 0, 0.05 - 0.12), atol=1e-12)
    with pytest.raises(ValueError, match="not supported"):
        urdf.parse(HAND_URDF.replace('<sphere radius="0.12"/>', '<cone radius="0.1"/>'))
    with pytest.raises(ValueError, match="one cylinder"):
        urdf.paddle_blade(robot, movable, "foot")


def test_foot_contact_points_of_the_27dof_model_from_collision_boxes():
    """The placeholder G1's sole corners (scene.TA_FOOT['points'], typed in scene.py) come out of a foot <collision> box of the same
    extent, through ground_contacts -> ta_model(contacts=...)."""
    text = urdf.write_g1_urdf()
    box = '<collision><origin xyz="0.035 0 -0.03"/><geometry><box size="0.17 0.06 0.01"/></geometry></collision>'
    for side in ("left", "right"):
        text = text.replace(f'<link name="{side}_ankle_roll_link"><inertial>', f'<link name="{side}_ankle_roll_link">{box}<inertial>')
    robot = urdf.parse(text)
    names = urdf.ta_dof_joint_names()
    movable = [robot.root()] + [robot.joints[n].child for n in names]
    feet = urdf.ground_contacts(robot, movable, ["left_ankle_roll_link", "right_ankle_roll_link"])
    assert [li for li, _ in feet] == [6] * 4 + [12] * 4
    np.testing.assert_allclose([p for _, p in feet[:4]], [(-0.05, -0.03, -0.035), (-0.05, 0.03, -0.035), (0.12, -0.03, -0.035), (0.12, 0.03, -0.035)], atol=1e-12)
    m = urdf.ta_model(robot, names, urdf.G1_BODY_NAMES, contacts=feet + list(scene.TA_BODY_CONTACTS))
    want = scene.build_ta_model()
    assert m.num_contacts == want.num_contacts
    np.testing.assert_allclose(np.array([list(m.contact_point[k]) for k in range(m.num_contacts)]),
                               np.array([list(want.contact_point[k]) for k in range(want.num_contacts)]), atol=1e-7)
