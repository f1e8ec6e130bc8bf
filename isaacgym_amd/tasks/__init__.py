"""Task registry: the names the reference registers for this path (tasks/__init__.py:49-53,118-122)."""
from ..vec_task import (Humanoid12PingpongTilt, HumanoidPingpong, HumanoidPingpongTilt, HumanoidPingpongTiltNESSparse27DOF,
                        HumanoidPingpongTiltNoEarlyStop)

isaacgym_task_map = {
    "HumanoidPingpongG1": HumanoidPingpong,
    "HumanoidPingpongTiltG1": HumanoidPingpongTilt,
    "HumanoidPingpongTiltNoEarlyStopG1": HumanoidPingpongTiltNoEarlyStop,
    "Humanoid12PingpongTiltG1": Humanoid12PingpongTilt,
    # not registered by the reference; named after its yaml, cfg/task/HumanoidPingpongTiltNESSparse27DOFG1.yaml
    "HumanoidPingpongTiltNESSparse27DOFG1": HumanoidPingpongTiltNESSparse27DOF,
}
