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


def test_features_of_a_real_file_rotated_inertials_off_diagonals_nested_welds():
    """What a manufacturer's URDF carries that the placeholder does not: <inertial> frames with an rpy, inertia tensors
    with off-diagonal terms, welded bodies hanging on welded bodies through rotated fixed joints.  The 27-dof file is edited that way — torso:
    rotated inertial frame + products of inertia; mid360_link re-hung under head_link, both welds rotated — and the importer's merged torso
    and weld frames are compared with a composition written out here (homogeneous frames + parallel axes), not with scene.composite_inertial."""
    import xml.etree.ElementTree as ET
    root = ET.fromstring(open(FIXTURE).read())
    link = {e.get("name"): e for e in root.findall("link")}
    joint = {e.find("child").get("link"): e for e in root.findall("joint")}
    rot = scene.rpy_to_rot
    # the torso's own inertial: frame rotated, tensor with products of inertia (given in that rotated frame, as the URDF convention says)
    t_in = link["torso_link"].find("inertial")
    t_in.find("origin").set("rpy", "0.3 -0.2 0.5")
    for k, v in (("ixy", "1.1e-3"), ("ixz", "-0.7e-3"), ("iyz", "0.4e-3")):
        t_in.find("inertia").set(k, v)
    # head welded with a rotation; mid360 re-hung under the head, rotated again; the head gets a rotated inertial too
    joint["head_link"].find("origin").set("rpy", "0.0 0.25 0.1")
    joint["mid360_link"].find("parent").set("link", "head_link")
    joint["mid360_link"].find("origin").set("xyz", "0.01 -0.02 0.12")
    joint["mid360_link"].find("origin").set("rpy", "-0.4 0.0 0.2")
    link["head_link"].find("inertial").find("origin").set("rpy", "0.1 0.2 0.3")
    link["head_link"].find("inertial").find("inertia").set("ixy", "2.0e-4")
    text = ET.tostring(root, encoding="unicode")
    robot = urdf.parse(text)
    m = urdf.ta_model(robot, urdf.ta_dof_joint_names(), urdf.G1_BODY_NAMES)

    def f(e, key, n=3):
        return np.array([float(x) for x in e.get(key, " ".join(["0"] * n)).split()])

    def frame_in_torso(name):          # (offset, rotation) of a body's link frame in the torso's, by walking the file's own joints
        off, r = np.zeros(3), np.eye(3)
        while name != "torso_link":
            o = joint[name].find("origin")
            jr = rot(*f(o, "rpy"))
            off, r = f(o, "xyz") + jr @ off, jr @ r
            name = joint[name].find("parent").get("link")
        return off, r

    ti = m.link[15]                    # link 15 = the torso (12 leg dofs + waist yaw, roll, torso joint)
    assert ti.body == urdf.G1_BODY_NAMES.index("torso_link")
    bodies = ["torso_link"] + [n for n in urdf.G1_BODY_NAMES if n != "torso_link" and n in joint and joint[n].get("type") == "fixed"
                               and _welded_to(joint, n) == "torso_link"]
    mass, first, second = 0.0, np.zeros(3), np.zeros((3, 3))
    for n in bodies:
        inert = link[n].find("inertial")
        if inert is None:
            continue
        off, r = frame_in_torso(n)
        mb = float(inert.find("mass").get("value"))
        c = off + r @ f(inert.find("origin"), "xyz")
        ri = r @ rot(*f(inert.find("origin"), "rpy"))
        i = inert.find("inertia")
        g = lambda k: float(i.get(k, 0.0))
        ib = ri @ np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]]) @ ri.T
        mass += mb
        first += mb * c
        second += ib + mb * (c.dot(c) * np.eye(3) - np.outer(c, c))          # about the torso frame's origin
    com = first / mass
    about_com = second - mass * (com.dot(com) * np.eye(3) - np.outer(com, com))
    assert ti.mass == pytest.approx(mass, rel=1e-6)
    np.testing.assert_allclose(list(ti.com), com, rtol=0, atol=2e-7)
    want6 = [about_com[0, 0], about_com[1, 1], about_com[2, 2], about_com[0, 1], about_com[0, 2], about_com[1, 2]]
    np.testing.assert_allclose(list(ti.inertia), want6, rtol=2e-6, atol=2e-9)
    assert abs(want6[3]) > 5e-4 and abs(want6[4]) > 3e-4                        # the products of inertia did arrive
    # the nested weld: mid360's frame in the torso = head weld o mid360 weld
    fx = {m.fixed[k].body: m.fixed[k] for k in range(scene.TA_NUM_FIXED)}
    off, r = frame_in_torso("mid360_link")
    mid = fx[urdf.G1_BODY_NAMES.index("mid360_link")]
    assert mid.link == 15
    np.testing.assert_allclose(list(mid.xyz), off, atol=2e-7)
    np.testing.assert_allclose(np.array(list(mid.rot)).reshape(3, 3), r, atol=2e-7)
    assert np.linalg.norm(off - f(joint["mid360_link"].find("origin"), "xyz")) > 0.1     # not the one-level value
    # and what the tables cannot carry is refused by name
    ET.SubElement(joint["left_elbow_link"], "mimic", joint="right_elbow_joint", multiplier="1")
    with pytest.raises(ValueError, match="mimic"):
        urdf.parse(ET.tostring(root, encoding="unicode"))


def _welded_to(joint, name):
    """The first ancestor reached through fixed joints only that itself hangs on a movable joint (or is the root)."""
    while name in joint and joint[name].get("type") == "fixed":
        name = joint[name].find("parent").get("link")
    return name


def test_parser_ignores_what_a_real_file_carries_beside_the_dynamics():
    """A manufacturer's URDF is full of elements the tables have no use for — <visual> meshes and materials, <dynamics>, <safety_controller>,
    <gazebo> / <transmission> blocks, comments, a <limit> without lower / upper on a continuous joint: none of them may change the model or trip
    the parser, and the elements that DO matter are read from the same joint."""
    text = """<?xml version="1.0"?>
    <!-- exported by some CAD plug-in -->
    <robot name="excerpt" xmlns:xacro="http://www.ros.org/wiki/xacro">
      <material name="dark"><color rgba="0.2 0.2 0.2 1"/></material>
      <link name="base">
        <inertial><origin xyz="0 0 0.1" rpy="0 0 0"/><mass value="2.0"/><inertia ixx="0.01" ixy="0" ixz="0" iyy="0.02" iyz="0" izz="0.03"/></inertial>
        <visual><origin xyz="0 0 0"/><geometry><mesh filename="meshes/base.STL" scale="1 1 1"/></geometry><material name="dark"/></visual>
        <collision><origin xyz="0 0 0.1"/><geometry><box size="0.2 0.2 0.2"/></geometry></collision>
      </link>
      <link name="arm">
        <inertial><origin xyz="0.1 0 0"/><mass value="0.5"/><inertia ixx="0.001" iyy="0.002" izz="0.002"/></inertial>
        <visual><geometry><mesh filename="meshes/arm.STL"/></geometry></visual>
        <collision><origin xyz="0.1 0 0" rpy="0 1.5708 0"/><geometry><cylinder radius="0.02" length="0.2"/></geometry></collision>
      </link>
      <link name="wheel"><inertial><mass value="0.1"/><inertia ixx="1e-4" iyy="1e-4" izz="1e-4"/></inertial></link>
      <joint name="shoulder" type="revolute">
        <origin xyz="0 0 0.2" rpy="0 0 0"/><parent link="base"/><child link="arm"/><axis xyz="0 1 0"/>
        <limit lower="-1.0" upper="2.0" effort="25" velocity="37"/>
        <dynamics damping="0.001" friction="0.1"/>
        <safety_controller soft_lower_limit="-0.9" soft_upper_limit="1.9" k_position="100" k_velocity="10"/>
      </joint>
      <joint name="spin" type="continuous"><origin xyz="0.2 0 0"/><parent link="arm"/><child link="wheel"/><axis xyz="1 0 0"/><limit effort="5" velocity="22"/></joint>
      <transmission name="t1"><type>transmission_interface/SimpleTransmission</type><joint name="shoulder"/></transmission>
      <gazebo reference="arm"><material>Gazebo/Grey</material></gazebo>
    </robot>"""
    r = urdf.parse(text)
    assert set(r.links) == {"base", "arm", "wheel"} and set(r.joints) == {"shoulder", "spin"}
    j = r.joints["shoulder"]
    assert (j.lower, j.upper, j.effort, j.velocity) == (-1.0, 2.0, 25.0, 37.0) and tuple(j.axis) == (0.0, 1.0, 0.0)
    c = r.joints["spin"]
    assert c.type == "continuous" and c.effort == 5.0 and c.lower == pytest.approx(-np.pi) and c.upper == pytest.approx(np.pi)
    assert [col.kind for col in r.links["base"].collisions] == ["box"] and [col.kind for col in r.links["arm"].collisions] == ["cylinder"]   # visuals are not collisions
    assert r.links["arm"].mass == 0.5 and np.allclose(r.links["arm"].inertia, np.diag([0.001, 0.002, 0.002]))                                   # missing ixy / ixz / iyz are zero
    assert not r.dropped_mesh_collisions
    specs = urdf.arm_specs(r, ["shoulder", "spin"])
    assert [s["axis"] for s in specs] == [1, 0] and specs[0]["limits"] == (-1.0, 2.0)
    with pytest.raises(ValueError, match="not \\+x, \\+y or \\+z"):          # a flipped axis changes the sign convention of q: refused by name, not silently mirrored
        urdf.arm_specs(urdf.parse(text.replace('<axis xyz="0 1 0"/>', '<axis xyz="0 -1 0"/>')), ["shoulder"])
    with pytest.raises(ValueError, match="prismatic"):
        urdf.parse(text.replace('type="continuous"', 'type="prismatic"'))


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


# ------------------------------------------------------------------------------------------------------------------
# <mesh> collisions (the real G1 asset's collision geometry is mostly meshes): a stand-in per link or a loud failure
MESH_URDF = HAND_URDF.replace('<collision><origin xyz="0.15 0 0" rpy="0 1.5707963267948966 0"/><geometry><cylinder radius="0.03" length="0.2"/></geometry></collision>',
                              '<collision><origin xyz="0.15 0 0" rpy="0 1.5707963267948966 0"/><geometry><mesh filename="meshes/upper.STL"/></geometry></collision>') \
                     .replace('<collision><origin xyz="0 0 0.05"/><geometry><sphere radius="0.12"/></geometry></collision>',
                              '<collision><origin xyz="0 0 0.05"/><geometry><mesh filename="meshes/base.STL"/></geometry></collision>')


def test_mesh_collisions_are_never_dropped_in_silence():
    assert MESH_URDF.count("<mesh") == 2
    with pytest.raises(ValueError) as e:
        urdf.parse(MESH_URDF)
    assert "base (meshes/base.STL)" in str(e.value) and "upper (meshes/upper.STL)" in str(e.value) and "2 <mesh>" in str(e.value)
    with pytest.raises(ValueError, match="upper"):                      # one stand-in given, the other still missing: still an error
        urdf.parse(MESH_URDF, mesh_bounds={"base": dict(kind="sphere", radius=0.12)})
    with pytest.warns(UserWarning, match="base .meshes/base.STL., upper"):
        robot = urdf.parse(MESH_URDF, on_mesh="warn")
    assert robot.dropped_mesh_collisions == [("base", "meshes/base.STL"), ("upper", "meshes/upper.STL")]
    assert robot.links["upper"].collisions == [] and len(robot.links["foot"].collisions) == 1
    with pytest.raises(ValueError, match="does not have"):
        urdf.parse(MESH_URDF, mesh_bounds={"nowhere": dict(kind="sphere", radius=0.1)}, on_mesh="warn")
    with pytest.raises(ValueError, match="not sphere"):
        urdf.parse(MESH_URDF, mesh_bounds={"base": dict(kind="cone", radius=0.1)}, on_mesh="warn")


def test_mesh_stand_ins_sit_at_the_mesh_origin():
    """A capsule named for the `upper` mesh is placed at the mesh's own <origin> (its axis = the origin's local z, pitched onto the
    link's x): the same two end points the analytic cylinder of HAND_URDF gives; the sphere likewise."""
    bounds = {"base": dict(kind="sphere", radius=0.12), "upper": [dict(kind="capsule", radius=0.03, length=0.2)]}
    robot = urdf.parse(MESH_URDF, mesh_bounds=bounds)
    assert robot.dropped_mesh_collisions == []
    shapes = urdf.ball_shapes(robot, ["base", "upper", "foot"], ["base", "upper"])
    want = urdf.ball_shapes(urdf.parse(HAND_URDF), ["base", "upper", "foot"], ["base", "upper"])
    for g, w in zip(shapes, want):
        assert g["link"] == w["link"] and g["radius"] == w["radius"]
        np.testing.assert_allclose(g["a"], w["a"], atol=1e-12)
        np.testing.assert_allclose(g["b"], w["b"], atol=1e-12)
    # a stand-in with its own offset is composed with the mesh origin: 0.05 along the mesh frame's z = the link's x
    robot = urdf.parse(MESH_URDF, mesh_bounds={"base": bounds["base"], "upper": dict(kind="sphere", radius=0.04, xyz=(0, 0, 0.05))})
    s = urdf.ball_shapes(robot, ["base", "upper", "foot"], ["upper"])[0]
    np.testing.assert_allclose(s["a"], (0.20, 0, 0), atol=1e-12)


# ------------------------------------------------------------------------------------------------------------------
# A URDF-derived model that DIFFERS from the placeholder tables, all the way into the kernels' arithmetic (tests/urdf_assets.py)
def test_second_27dof_asset_differs_and_runs_on_the_kernel_arithmetic(oracle_lib):
    """urdf.perturbed -> urdf.ta_model: not the compiled-in model; the table-driven kernel arithmetic (ppenv_ta_device.h on the host)
    follows it and agrees with the oracle given the same tables — and both differ from the placeholder model's step."""
    import ctypes as C

    import shim_binding as sb
    import urdf_assets
    from test_ta_physics import ball_switch_probe, check_step, initial_tensors
    m, base = urdf_assets.second_27dof_model(), scene.build_ta_model()
    assert abs(sum(m.link[i].mass for i in range(28)) / sum(base.link[i].mass for i in range(28)) - 1.3) < 1e-6
    assert abs(m.link[4].origin_xyz[2] - (base.link[4].origin_xyz[2] - 0.03)) < 1e-7          # left knee 3 cm lower
    assert (m.link[16].lower, m.link[16].upper) == (-1.0, np.float32(1.2))
    n = 48
    cfg = scene.build_ta_scene(n)
    root, dof = initial_tensors(n, seed=5)
    rng = np.random.default_rng(6)
    moved = 0.0
    for t in range(60):
        if t % 4 == 0:
            act = rng.uniform(-1.1, 1.1, (n, 27)).astype(np.float32)
        r2, d2, r3, d3 = root.copy(), dof.copy(), root.copy(), dof.copy()
        root0, dof0 = root.copy(), dof.copy()
        rb, frc, _ = oracle_lib.ta_simulate(cfg, m, act, root, dof, threads=8)
        rb2, frc2, _ = sb.ta_simulate(cfg, m, act, r2, d2)
        oracle_lib.ta_simulate(cfg, base, act, r3, d3, threads=8)
        moved = max(moved, float(np.abs(d3[..., 1] - dof[..., 1]).max()))
        hit = ball_switch_probe(oracle_lib, cfg, m, act, root0, dof0, root, seed=t)
        check_step((r2[~hit], d2[~hit], rb2[~hit], frc2[~hit]), (root[~hit], dof[~hit], rb[~hit], frc[~hit]), f"second asset, step {t}")
    assert moved > 0.5          # rad/s: the placeholder model would have stepped elsewhere — the tables did reach the arithmetic


def test_second_arm_asset_through_modelgen_into_the_kernel_arithmetic(oracle_lib, tmp_path):
    """7-dof path: urdf.arm_specs -> scene.use_arm_tables -> modelgen.generate -> the kernel arithmetic rebuilt against that header
    agrees with the oracle on the changed arm; the stock build refuses the config (pp::model_matches)."""
    import ctypes as C

    import shim_binding as sb
    import urdf_assets
    from helpers import ExclusionLog, SensitivityProbe, assert_close, assert_state_close, mask_envs, obs_atol, reward_atol
    from isaacgym_amd import modelgen
    n = 128
    stock_cfg = scene.build_config("TT", num_envs=n, seed=3)
    with urdf_assets.second_arm_tables():
        cfg = scene.build_config("TT", num_envs=n, seed=3)
    assert bytes(cfg) != bytes(stock_cfg)
    assert abs(cfg.joint[3].mass / stock_cfg.joint[3].mass - 1.3) < 1e-6 and cfg.joint[1].upper == np.float32(0.9)
    assert sb.lib().shim_model_matches(C.byref(cfg)) == 0                          # the stock compiled-in model is stale for this asset
    header = tmp_path / "ppenv_model.h"
    header.write_text(modelgen.generate(cfg))
    L = sb.lib_for_model(str(header), str(tmp_path / "libshim_second_arm.so"))
    assert L.shim_model_matches(C.byref(cfg)) == 1 and L.shim_model_matches(C.byref(stock_cfg)) == 0
    o, s, o_stock = oracle_lib.OracleEnv(cfg), sb.ShimEnv(cfg, L=L), oracle_lib.OracleEnv(stock_cfg)
    probe = SensitivityProbe(oracle_lib, cfg)
    log = ExclusionLog("host shim rebuilt for the second arm asset vs oracle [TT]", bound=0.005)
    rng = np.random.default_rng(1)
    oa, ra = obs_atol(), reward_atol(cfg)
    moved = 0.0
    for t in range(80):
        actions = rng.uniform(-1.2, 1.2, (n, 7)).astype(np.float32)
        s.copy_state_from(o)
        st = o.get_state()
        o_stock.set_state(st)
        o.step(actions)
        s.step(actions)
        o_stock.step(actions)
        moved = max(moved, float(np.abs(o_stock.dof_vel - o.dof_vel).max()))
        keep = ~probe.sensitive(st, actions, o)
        log.add(keep)
        sm, om = mask_envs(s, keep), mask_envs(o, keep)
        np.testing.assert_array_equal(sm.reset_buf, om.reset_buf)
        np.testing.assert_array_equal(sm.flags, om.flags)
        assert_state_close(sm, om, f"second arm, step {t}")
        assert_close(sm.obs_buf, om.obs_buf, f"second arm, obs step {t}", atol=oa)
        assert_close(sm.rew_buf, om.rew_buf, f"second arm, rew step {t}", atol=ra)
    log.close()
    assert moved > 0.5          # rad/s: the stock arm steps elsewhere from the same state


# ------------------------------------------------------------------------------------------------------------------
# The scene's other two assets: pingpong_table.urdf / small_ball.urdf (TT:496,502) -> the run-time slabs and ball constants
def test_table_and_ball_urdf_importers_against_typed_in_numbers():
    import urdf_assets
    t = urdf.table_scene(urdf.parse(urdf_assets.TABLE_URDF))
    # slab: the first <collision> (shape 0, the one the reference gives the table material, TT:580-582): 2.70 x 1.50 x 0.025 centred at
    # (0.01, 0, 0.7375) -> top face at 0.75; net: 0.006 x 1.80 x 0.14 on a fixed joint at (0.01, 0, 0.75), collision origin +0.07 in z
    expect = dict(length=2.70, width=1.50, top_z=0.75, slab=0.025, net_height=0.14, net_overhang=0.15, net_half_thickness=0.003, net_bottom_z=0.75)
    for k, v in expect.items():
        assert abs(t[k] - v) < 1e-12, (k, t[k], v)
    assert t["offset_xy"] == (0.01, 0.0) and t["net_offset_xy"] == (0.01, 0.0)
    assert t["ignored"] == [("leg_near", 0), ("leg_far", 0)]                      # legs below the surface, inside the footprint: no ball shape
    b = urdf.ball_params(urdf.parse(urdf_assets.BALL_URDF))
    assert b["radius"] == 0.0205 and b["mass"] == 0.0027
    assert abs(b["inertia_factor"] - 6.8079e-07 / (0.0027 * 0.0205 ** 2)) < 1e-15 and abs(b["inertia_factor"] - 0.6) < 1e-4
    # ... and into the config the kernels and the oracle read
    cfg, stock = scene.build_config("TT", num_envs=4, table=t, ball=b), scene.build_config("TT", num_envs=4)
    np.testing.assert_allclose(list(cfg.table.center), (1.75 + 0.01, 0.0, 0.7375), atol=1e-6)           # table actor at (1.75, 0, 0), TT:575
    np.testing.assert_allclose(list(cfg.table.half), (1.35, 0.75, 0.0125), atol=1e-7)
    np.testing.assert_allclose(list(cfg.net.center), (1.76, 0.0, 0.82), atol=1e-6)
    np.testing.assert_allclose(list(cfg.net.half), (0.003, 0.90, 0.07), atol=1e-7)
    assert cfg.ball_radius == np.float32(0.0205) and abs(cfg.ball_inertia_factor - 0.6) < 1e-4
    assert (cfg.table.restitution, cfg.table.friction) == (stock.table.restitution, stock.table.friction)   # materials are the task's, not the file's
    assert abs(stock.table.center[2] - 0.745) < 1e-6 and stock.ball_radius == np.float32(0.02)             # the placeholders differ
    ta = scene.build_ta_scene(4, table=t, ball=b)
    assert list(ta.table.half) == list(cfg.table.half) and ta.ball_radius == cfg.ball_radius


def test_table_and_ball_importers_reject_what_the_kernels_cannot_represent():
    import urdf_assets
    T, B = urdf_assets.TABLE_URDF, urdf_assets.BALL_URDF
    with pytest.raises(ValueError, match="rotated off"):                             # a slab yawed by 30 degrees is not an axis-aligned box
        urdf.table_scene(urdf.parse(T.replace('<origin xyz="0.01 0 0.7375" rpy="0 0 0"/>', '<origin xyz="0.01 0 0.7375" rpy="0 0 0.5236"/>')))
    with pytest.raises(ValueError, match="must be boxes"):
        urdf.table_scene(urdf.parse(T.replace('<box size="0.006 1.80 0.14"/>', '<sphere radius="0.07"/>')))
    with pytest.raises(ValueError, match="fixed joints only"):
        urdf.table_scene(urdf.parse(T.replace('<joint name="net_joint" type="fixed">', '<joint name="net_joint" type="revolute">')))
    with pytest.raises(ValueError, match="exactly one box standing"):                # no net: the net joint dropped 5 cm into the slab
        urdf.table_scene(urdf.parse(T.replace('<origin xyz="0.01 0 0.75" rpy="0 0 0"/><parent link="table_top"/><child link="net"/>',
                                              '<origin xyz="0.01 0 0.70" rpy="0 0 0"/><parent link="table_top"/><child link="net"/>')))
    with pytest.raises(ValueError, match="sticks out"):                              # a leg outside the slab's footprint
        urdf.table_scene(urdf.parse(T.replace('<origin xyz="1.0 0 0.3625" rpy="0 0 0"/>', '<origin xyz="1.5 0 0.3625" rpy="0 0 0"/>')))
    with pytest.raises(ValueError, match="not its largest"):                         # shape 0 must be the playing surface
        urdf.table_scene(urdf.parse(T.replace('<box size="2.70 1.50 0.025"/>', '<box size="0.1 0.1 0.025"/>')))
    with pytest.raises(ValueError, match="not isotropic"):
        urdf.ball_params(urdf.parse(B.replace('iyy="6.8079e-07"', 'iyy="9e-07"')))
    with pytest.raises(ValueError, match="one sphere"):
        urdf.ball_params(urdf.parse(B.replace('<sphere radius="0.0205"/>', '<box size="0.04 0.04 0.04"/>')))
    with pytest.raises(ValueError, match="centre of mass"):
        urdf.ball_params(urdf.parse(B.replace('<inertial><origin xyz="0 0 0"/>', '<inertial><origin xyz="0.001 0 0"/>')))
    with pytest.raises(ValueError, match="unknown keys"):
        scene.build_config("TT", num_envs=1, table=dict(lenght=2.7))


def test_table_and_ball_from_urdf_reach_the_kernel_arithmetic(oracle_lib, tmp_path):
    """The kernels' per-env arithmetic (host shim) follows the oracle on a scene whose table and ball come from the two URDFs — through
    the cfg hook a task uses (scene.table_urdf / ball_urdf paths) — and the placeholder scene steps elsewhere from the same state."""
    import shim_binding as sb
    import urdf_assets
    from helpers import ExclusionLog, SensitivityProbe, assert_close, assert_state_close, mask_envs, obs_atol, reward_atol
    (tmp_path / "pingpong_table.urdf").write_text(urdf_assets.TABLE_URDF)
    (tmp_path / "small_ball.urdf").write_text(urdf_assets.BALL_URDF)
    task_cfg = scene.default_task_cfg("TT")
    task_cfg["scene"] = dict(task_cfg["scene"], table_urdf=str(tmp_path / "pingpong_table.urdf"), ball_urdf=str(tmp_path / "small_ball.urdf"))
    table, ball = scene.asset_geometry(task_cfg["scene"])
    assert (table, ball) == urdf_assets.second_scene_geometry()
    n = 128
    cfg, stock_cfg = scene.build_config("TT", cfg=task_cfg, num_envs=n, seed=4, table=table, ball=ball), scene.build_config("TT", num_envs=n, seed=4)
    o, s, o_stock = oracle_lib.OracleEnv(cfg), sb.ShimEnv(cfg), oracle_lib.OracleEnv(stock_cfg)
    probe = SensitivityProbe(oracle_lib, cfg)
    log = ExclusionLog("host shim on the URDF table + ball vs oracle [TT]", bound=0.005)
    rng = np.random.default_rng(3)
    oa, ra = obs_atol(), reward_atol(cfg)
    moved = 0.0
    for t in range(120):
        actions = rng.uniform(-1.0, 1.0, (n, 7)).astype(np.float32)
        s.copy_state_from(o)
        st = o.get_state()
        o_stock.set_state(st)
        o.step(actions)
        s.step(actions)
        o_stock.step(actions)
        same_reset = o_stock.reset_buf == o.reset_buf
        moved = max(moved, float(np.abs(o_stock.ball[:3, same_reset] - o.ball[:3, same_reset]).max()) if same_reset.any() else 0.0)     # ball is SoA [13][N]
        keep = ~probe.sensitive(st, actions, o)
        log.add(keep)
        sm, om = mask_envs(s, keep), mask_envs(o, keep)
        np.testing.assert_array_equal(sm.reset_buf, om.reset_buf)
        np.testing.assert_array_equal(sm.flags, om.flags)
        assert_state_close(sm, om, f"urdf table + ball, step {t}")
        assert_close(sm.obs_buf, om.obs_buf, f"urdf table + ball, obs step {t}", atol=oa)
        assert_close(sm.rew_buf, om.rew_buf, f"urdf table + ball, rew step {t}", atol=ra)
    log.close()
    assert moved > 5e-3          # m: a bounce off a table 1 cm lower with a 0.5 mm larger ball lands elsewhere than the placeholder scene's


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mapping", ["default", "lane"])
def test_gpu_27dof_kernels_follow_a_urdf_model_that_differs_from_the_placeholder(oracle_lib, monkeypatch, mapping):
    """The changed asset -> urdf.ta_model -> TASim(model=...): the library sees it is not the compiled-in model, takes a table-driven
    kernel (the quad mapping by default — same tree — or one lane per env) and steps it like the oracle does with the same tables."""
    import ctypes as C

    import torch
    import urdf_assets
    from helpers import ExclusionLog
    from isaacgym_amd import _lib
    from isaacgym_amd.tensor_api import TASim
    from test_ta_physics import ball_switch_probe, check_step, initial_tensors
    if mapping == "lane":
        monkeypatch.setenv("PPENV_TA_KERNEL", "lane")
    else:
        monkeypatch.delenv("PPENV_TA_KERNEL", raising=False)
    n = 200
    cfg, m = scene.build_ta_scene(n), urdf_assets.second_27dof_model()
    assert _lib.lib().ppenv_ta_model_is_compiled(C.byref(cfg), C.byref(m)) == 0
    sim = TASim(n, device="cuda:0", model=m)
    assert sim.kernel == ("lane" if mapping == "lane" else "quad")
    base = scene.build_ta_model()
    root, dof = initial_tensors(n, seed=11)
    rng = np.random.default_rng(12)
    dev = lambda a: torch.from_numpy(a).cuda()
    rb_d, frc_d, pvx_d = torch.zeros(n, 42, 13, device="cuda"), torch.zeros(n, 27, device="cuda"), torch.zeros(n, device="cuda")
    log = ExclusionLog(f"gpu 27-dof step from the second URDF asset vs oracle [{sim.kernel}, n={n}]", bound=0.005)
    moved = 0.0
    for t in range(80):
        if t % 4 == 0:
            act = rng.uniform(-1.2, 1.2, (n, 27)).astype(np.float32)
            act[: n // 3] *= 0.1
        root_d, dof_d = dev(root), dev(dof)
        sim.simulate(dev(act), root_d, dof_d, rb_d, frc_d, pvx_d)
        root0, dof0 = root.copy(), dof.copy()
        r3, d3 = root.copy(), dof.copy()
        rb, frc, pvx = oracle_lib.ta_simulate(cfg, m, act, root, dof, threads=8)
        oracle_lib.ta_simulate(cfg, base, act, r3, d3, threads=8)
        moved = max(moved, float(np.abs(d3[..., 1] - dof[..., 1]).max()))
        keep = ~ball_switch_probe(oracle_lib, cfg, m, act, root0, dof0, root, seed=300 + t)
        log.add(keep)
        got = (root_d.cpu().numpy()[keep], dof_d.cpu().numpy()[keep], rb_d.cpu().numpy()[keep], frc_d.cpu().numpy()[keep])
        check_step(got, (root[keep], dof[keep], rb[keep], frc[keep]), f"second asset on the GPU, step {t}")
    log.close()
    assert moved > 0.5
    sim.close()
    monkeypatch.setenv("PPENV_TA_KERNEL", "chain")                  # the compile-time-tree kernel is for the compiled model only
    with pytest.raises(_lib.PPEnvError, match="differs from the one compiled"):
        TASim(n, device="cuda:0", model=m)


@pytest.mark.gpu
def test_gpu_7dof_step_on_a_library_built_for_the_second_arm_asset(oracle_lib):
    """urdf.arm_specs -> use_arm_tables -> modelgen -> _lib.build_for_arm_model: the fused 7-dof step of THAT library follows the
    changed arm (vs the oracle on the same config); the default library refuses the config instead of stepping the stale model."""
    import torch
    import urdf_assets
    from helpers import ExclusionLog, SensitivityProbe, assert_close, assert_state_close, mask_envs, obs_atol, reward_atol
    from isaacgym_amd import _lib
    from isaacgym_amd.env import PPEnv
    from test_gpu_parity import DevView
    n = 512
    with urdf_assets.second_arm_tables():
        cfg, cfg2 = scene.build_config("TT", num_envs=n, seed=3), scene.build_config("TT", num_envs=n, seed=3)
    with pytest.raises(_lib.PPEnvError, match="differs from the one compiled"):
        PPEnv(cfg2, device="cuda:0")
    L = _lib.load(urdf_assets.build_second_arm_library())
    env = PPEnv(cfg2, device="cuda:0", library=L)
    o = oracle_lib.OracleEnv(cfg)
    probe = SensitivityProbe(oracle_lib, cfg)
    log = ExclusionLog(f"gpu fused step of the library built for the second arm asset vs oracle [TT, n={n}]", bound=0.005)
    rng = np.random.default_rng(2)
    oa, ra = obs_atol(), reward_atol(cfg)

    for t in range(80):
        a = rng.uniform(-1.2, 1.2, (n, 7)).astype(np.float32)
        st = o.get_state()
        env.set_state(st)
        o.step(a)
        env.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        keep = ~probe.sensitive(st, a, o)
        log.add(keep)
        g = DevView(env)
        gm, om = mask_envs(g, keep), mask_envs(o, keep)
        np.testing.assert_array_equal(gm.reset_buf, om.reset_buf)
        np.testing.assert_array_equal(gm.flags, om.flags)
        assert_state_close(gm, om, f"second arm on the GPU, step {t}")
        assert_close(gm.obs_buf, om.obs_buf, f"second arm on the GPU, obs step {t}", atol=oa)
        assert_close(gm.rew_buf, om.rew_buf, f"second arm on the GPU, rew step {t}", atol=ra)
    log.close()
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["TT", "T4"])
def test_gpu_fused_step_on_a_table_and_ball_from_urdf(oracle_lib, variant):
    """pingpong_table.urdf / small_ball.urdf stand-ins -> urdf.table_scene / ball_params -> scene.build_config(table=, ball=): the fused HIP
    step (two-wave kernel; the 4-actor three-wave kernel) follows the oracle on that scene, and the placeholder scene steps elsewhere."""
    import torch
    import urdf_assets
    from helpers import ExclusionLog, SensitivityProbe, assert_close, assert_state_close, mask_envs, obs_atol, reward_atol
    from isaacgym_amd.env import PPEnv
    from test_gpu_parity import DevView
    n = 512
    table, ball = urdf_assets.second_scene_geometry()
    cfg, cfg2 = (scene.build_config(variant, num_envs=n, seed=6, table=table, ball=ball) for _ in range(2))
    stock_cfg = scene.build_config(variant, num_envs=n, seed=6)
    assert bytes(cfg) != bytes(stock_cfg)
    env = PPEnv(cfg2, device="cuda:0")
    o, o_stock = oracle_lib.OracleEnv(cfg), oracle_lib.OracleEnv(stock_cfg)
    probe = SensitivityProbe(oracle_lib, cfg)
    log = ExclusionLog(f"gpu fused step on the URDF table + ball vs oracle [{variant}, n={n}]", bound=0.005)
    rng = np.random.default_rng(8)
    A = 2 if variant == "T4" else 1
    oa, ra = obs_atol(), A * reward_atol(cfg)
    moved = 0.0
    for t in range(120):
        a = rng.uniform(-1.0, 1.0, (A * n, 7)).astype(np.float32)
        st = o.get_state()
        env.set_state(st)
        o_stock.set_state(st)
        o.step(a)
        o_stock.step(a)
        env.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        same_reset = o_stock.reset_buf[::A] == o.reset_buf[::A]
        if same_reset.any():
            moved = max(moved, float(np.abs(o_stock.ball[:3, same_reset] - o.ball[:3, same_reset]).max()))
        keep = ~probe.sensitive(st, a, o)
        log.add(keep)
        gm, om = mask_envs(DevView(env), keep, A), mask_envs(o, keep, A)
        np.testing.assert_array_equal(gm.reset_buf, om.reset_buf)
        np.testing.assert_array_equal(gm.flags, om.flags)
        assert_state_close(gm, om, f"urdf table + ball on the GPU, step {t}")
        assert_close(gm.obs_buf, om.obs_buf, f"urdf table + ball on the GPU, obs step {t}", atol=oa)
        assert_close(gm.rew_buf, om.rew_buf, f"urdf table + ball on the GPU, rew step {t}", atol=ra)
    log.close()
    assert moved > 5e-3
    env.close()


@pytest.mark.gpu
def test_gpu_chain_kernel_of_a_library_built_for_the_second_27dof_asset(oracle_lib, monkeypatch):
    """urdf.ta_model -> _lib.build_for_ta_model: THAT library runs the changed tree on ta_chain_kernel (the 30-us-class kernel the default
    library reserves for its own compiled tree) and follows the oracle given the same tables, on a scene whose table and ball come from the
    URDF stand-ins too; the stock tree is in turn NOT this library's compiled model."""
    import ctypes as C

    import torch
    import urdf_assets
    from helpers import ExclusionLog
    from isaacgym_amd import _lib
    from isaacgym_amd.tensor_api import TAEnv, TASim
    from test_ta_physics import initial_tensors, run_chain_step_parity
    monkeypatch.delenv("PPENV_TA_KERNEL", raising=False)
    n = 640
    table, ball = urdf_assets.second_scene_geometry()
    cfg, m, base = scene.build_ta_scene(n, table=table, ball=ball), urdf_assets.second_27dof_model(), scene.build_ta_model()
    L = _lib.load(urdf_assets.build_second_ta_library())
    assert L.ppenv_ta_model_is_compiled(C.byref(cfg), C.byref(m)) == 1 and _lib.lib().ppenv_ta_model_is_compiled(C.byref(cfg), C.byref(m)) == 0
    assert L.ppenv_ta_model_is_compiled(C.byref(cfg), C.byref(base)) == 0
    stock = TASim(n, device="cuda:0", scene_cfg=cfg, model=base, library=L)            # the other way round: table-driven for the stock tree
    assert stock.kernel == "quad"
    stock.close()
    env = TAEnv(n, device="cuda:0", seed=13, env={"episodeLength": 40}, materialize_rb=True, scene_cfg=cfg, model=m, library=L)
    assert env.sim.kernel == "chain"
    # the placeholder tree steps elsewhere from the same state: the compiled tables are the asset's
    root, dof = initial_tensors(64, seed=21)
    act = np.random.default_rng(22).uniform(-1.2, 1.2, (64, 27)).astype(np.float32)
    r3, d3 = root.copy(), dof.copy()
    cfg64 = scene.build_ta_scene(64, table=table, ball=ball)
    for _ in range(8):
        oracle_lib.ta_simulate(cfg64, m, act, root, dof, threads=8)
        oracle_lib.ta_simulate(cfg64, base, act, r3, d3, threads=8)
    assert float(np.abs(d3[..., 1] - dof[..., 1]).max()) > 0.5
    run_chain_step_parity(oracle_lib, env, cfg, m, f"gpu chain-wave kernel compiled for the second URDF asset vs oracle [n={n}]", joint_probe=True)
    env.close()
