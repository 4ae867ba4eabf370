import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mppi_playground_amd  # noqa: E402,F401  (puts pi_mpc/ and envs/ on sys.path like the reference's src/)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _name_the_running_test(request):
    """tests/parity_report.py attributes what the checks measure to the test that is running."""
    import parity_report

    parity_report.current_test = request.node.name
    yield
