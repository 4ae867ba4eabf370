import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import mppi_playground_amd  # noqa: E402,F401  (puts pi_mpc/ and envs/ on sys.path like the reference's src/)


def pytest_addoption(parser):
    parser.addoption("--strict-parity", action="store_true", default=False,
                     help="hold the full-size configurations (C2 / C3 / C5 against the reference itself) to the north star's plain "
                          "1e-5 instead of max(1e-5, the reference's own measured spread)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import parity_report

    parity_report.strict = bool(config.getoption("--strict-parity"))


@pytest.fixture(autouse=True)
def _name_the_running_test(request):
    """tests/parity_report.py attributes what the checks measure to the test that is running."""
    import parity_report

    parity_report.current_test = request.node.name
    yield
