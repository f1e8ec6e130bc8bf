"""`import isaacgym` of the reference's train.py (train.py:80) on an MI355X: NVIDIA's Isaac Gym binary does not exist here.

This package only marks the simulator as provided natively — the rigid-body step, the tensor API and the VecTask surface of the
HumanoidPingpong tasks live in `isaacgym_amd` (HIP kernels behind include/ppenv.h), reached through the `isaacgymenvs` shim next to
this directory.  The reference's own task classes (tasks/*.py) call gymapi / gymtorch directly and are NOT what runs here: the task
names of `isaacgymenvs.tasks.isaacgym_task_map` resolve to the native classes of isaacgym_amd.vec_task.
"""
NATIVE_BACKEND = "isaacgym_amd"


def __getattr__(name):   # gymapi, gymtorch, gymutil ...: there is no PhysX to talk to
    raise AttributeError(f"isaacgym.{name} does not exist on this platform: the HumanoidPingpong tasks run on the native MI355X "
                         f"environment (isaacgym_amd); use isaacgymenvs.make / isaacgymenvs.tasks.isaacgym_task_map")
