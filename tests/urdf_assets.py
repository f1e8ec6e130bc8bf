"""A SECOND asset for the URDF path: the placeholder G1 written out as URDF and then changed where a real
asset would differ from the placeholder tables — every mass and inertia x 1.3, two link lengths, one joint range — so that
tests drive the kernels from a URDF-derived model that is NOT what is compiled in / typed into scene.py.
TEST INFRASTRUCTURE (also used by __graft_entry__.build() to prebuild the library variant the GPU test loads)."""
import contextlib
import os

from isaacgym_amd import scene, urdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT_DIR = os.path.join(ROOT, "build_variants", "second_arm")          # git-ignored; travels to the GPU box with the snapshot
TA_VARIANT_DIR = os.path.join(ROOT, "build_variants", "second_ta")        # the chain-wave kernel compiled for second_27dof_model()
ARM_JOINTS = [f"right_{n}_joint" for n in ("shoulder_pitch", "shoulder_roll", "shoulder_yaw", "elbow", "wrist_roll", "wrist_pitch", "wrist_yaw")]

MASS_SCALE = 1.3
TA_SHIFTS = {"left_knee_joint": (0.0, 0.0, -0.03), "left_elbow_joint": (0.02, 0.0, 0.0)}    # a longer left thigh and left upper arm
TA_LIMITS = {"left_shoulder_pitch_joint": (-1.0, 1.2)}
ARM_SHIFTS = {"right_elbow_joint": (0.025, 0.0, -0.01), "right_wrist_roll_joint": (0.015, 0.0, 0.0)}
ARM_LIMITS = {"right_shoulder_roll_joint": (-1.4, 0.9)}


def second_27dof_model():
    """scene.TAModel of the changed 27-dof asset, through the importer."""
    text = urdf.perturbed(urdf.write_g1_urdf(), MASS_SCALE, TA_SHIFTS, TA_LIMITS)
    return urdf.ta_model(urdf.parse(text), urdf.ta_dof_joint_names(), urdf.G1_BODY_NAMES)


def second_arm_specs():
    text = urdf.perturbed(urdf.write_g1_urdf(weld_right_elbow=False), MASS_SCALE, ARM_SHIFTS, ARM_LIMITS)
    return urdf.arm_specs(urdf.parse(text), ARM_JOINTS, {n: i for i, n in enumerate(urdf.G1_BODY_NAMES)})


@contextlib.contextmanager
def second_arm_tables():
    """scene's 7-dof chain tables replaced by the changed asset's for the duration (scene.build_config then describes that arm)."""
    saved = scene.G1_RIGHT_ARM
    scene.use_arm_tables(second_arm_specs())
    try:
        yield
    finally:
        scene.use_arm_tables(saved)


def build_second_arm_library(force=False):
    """libppenv with the changed arm compiled in -> its path (isaacgym_amd._lib.build_for_arm_model)."""
    from isaacgym_amd import _lib
    with second_arm_tables():
        cfg = scene.build_config("TT", num_envs=1)
    return _lib.build_for_arm_model(cfg, VARIANT_DIR, force=force)


def build_second_ta_library(force=False):
    """libppenv with the changed 27-dof tree compiled into the chain-wave kernel -> its path (isaacgym_amd._lib.build_for_ta_model)."""
    from isaacgym_amd import _lib
    return _lib.build_for_ta_model(second_27dof_model(), TA_VARIANT_DIR, force=force)


# ------------------------------------------------------------------------------------------------------------------
# The scene's other two assets (the reference's pingpong_table.urdf / small_ball.urdf, TT:496,502 — neither is in the reference).
# Hand-written stand-ins that DIFFER from scene.TABLE_GEOM / BALL_GEOM; tests/test_urdf.py types the expected numbers in.
TABLE_URDF = """<?xml version="1.0"?>
<robot name="pingpong_table">
  <link name="table_top">
    <inertial><mass value="40"/><inertia ixx="8" iyy="25" izz="32" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0.01 0 0.7375" rpy="0 0 0"/><geometry><box size="2.70 1.50 0.025"/></geometry></collision>
  </link>
  <link name="net">
    <collision><origin xyz="0 0 0.07" rpy="0 0 0"/><geometry><box size="0.006 1.80 0.14"/></geometry></collision>
  </link>
  <joint name="net_joint" type="fixed"><origin xyz="0.01 0 0.75" rpy="0 0 0"/><parent link="table_top"/><child link="net"/></joint>
  <link name="leg_near">
    <collision><origin xyz="0 0 0" rpy="0 0 1.5707963267948966"/><geometry><box size="1.2 0.05 0.725"/></geometry></collision>
  </link>
  <joint name="leg_near_joint" type="fixed"><origin xyz="-1.0 0 0.3625" rpy="0 0 0"/><parent link="table_top"/><child link="leg_near"/></joint>
  <link name="leg_far">
    <collision><origin xyz="0 0 0" rpy="0 0 1.5707963267948966"/><geometry><box size="1.2 0.05 0.725"/></geometry></collision>
  </link>
  <joint name="leg_far_joint" type="fixed"><origin xyz="1.0 0 0.3625" rpy="0 0 0"/><parent link="table_top"/><child link="leg_far"/></joint>
</robot>
"""

BALL_URDF = """<?xml version="1.0"?>
<robot name="small_ball">
  <link name="ball">
    <inertial><origin xyz="0 0 0"/><mass value="0.0027"/>
      <inertia ixx="6.8079e-07" iyy="6.8079e-07" izz="6.8079e-07" ixy="0" ixz="0" iyz="0"/></inertial>
    <collision><origin xyz="0 0 0"/><geometry><sphere radius="0.0205"/></geometry></collision>
  </link>
</robot>
"""


def second_scene_geometry():
    """(table, ball) dicts for scene.build_config / build_ta_scene, through the importers."""
    return urdf.table_scene(urdf.parse(TABLE_URDF)), urdf.ball_params(urdf.parse(BALL_URDF))
