import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# glibc writes its fatal messages (heap consistency checks, fortify, stack protector) to the controlling terminal when there is one, and
# only otherwise to stderr: a session that aborts must say why in its log ("open at the end of round 6")
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
