"""CPU oracle for the HumanoidPingpong step — test infrastructure only (see ppenv_oracle.c)."""
