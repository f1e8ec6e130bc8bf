import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """What the probe-using parity tests excluded, and the worst error they retained (tests/helpers.py ExclusionLog)."""
    import helpers
    if helpers.PARITY_REPORT:
        terminalreporter.write_sep("-", "parity: probe exclusions and worst retained errors")
        for line in helpers.PARITY_REPORT:
            terminalreporter.write_line(line)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    binding.build()
    return binding
