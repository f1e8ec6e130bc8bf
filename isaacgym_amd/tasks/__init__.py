"""Task registry: the names the reference registers for this path (tasks/__init__.py:49-53,118-120)."""
from ..vec_task import HumanoidPingpong, HumanoidPingpongTilt, HumanoidPingpongTiltNoEarlyStop

isaacgym_task_map = {
    "HumanoidPingpongG1": HumanoidPingpong,
    "HumanoidPingpongTiltG1": HumanoidPingpongTilt,
    "HumanoidPingpongTiltNoEarlyStopG1": HumanoidPingpongTiltNoEarlyStop,
}
