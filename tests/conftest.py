import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# The tests select the forms of single stages through PSACX_* variables (monkeypatch.setenv between constructions of one process).
# The library never reads the environment; this switch makes the Python wrappers call its debug shims (psacx_configure_from_env,
# psacx_multi_configure_from_env) before every call -- also in the Python subprocesses the tests start, which inherit the variable.
os.environ["PSACX_ENV_KNOBS"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
