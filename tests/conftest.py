import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure), built on demand from oracle/nkml_oracle.c."""
    so = os.path.join(ROOT, "oracle", "nkml_oracle.so")
    src = os.path.join(ROOT, "oracle", "nkml_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "nkml_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    from tests.helpers import Oracle
    return Oracle
