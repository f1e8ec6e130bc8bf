"""isaacgymenvs.tasks (train.py:85): task name -> class.  The four registered HumanoidPingpong names (reference tasks/__init__.py:49-53,
118-122) plus the unregistered 27-dof class under its yaml's name, all MI355X-native (isaacgym_amd.vec_task)."""
from isaacgym_amd.tasks import isaacgym_task_map  # noqa: F401
