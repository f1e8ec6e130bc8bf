"""A SECOND asset for the URDF path: the placeholder G1 written out as URDF and then changed where a real
asset would differ from the placeholder tables — every mass and inertia x 1.3, two link lengths, one joint range — so that
tests drive the kernels from a URDF-derived model that is NOT what is compiled in / typed into scene.py.
TEST INFRASTRUCTURE (also used by __graft_entry__.build() to prebuild the library variant the GPU test loads)."""
import contextlib
import os

from isaacgym_amd import scene, urdf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT_DIR = os.path.join(ROOT, "build_variants", "second_arm")          # git-ignored; travels to the GPU box with the snapshot
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
