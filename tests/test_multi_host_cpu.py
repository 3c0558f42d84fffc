"""Host logic of the shipped multi-GPU engine (psac_amd/csrc/multi.hpp) that runs without a GPU: the shared-memory
link of the process-per-rank deployment, with real processes (world size 3 and 5)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("P", [2, 3, 5])
def test_shm_link_between_processes(tmp_path, P):
    # P forked processes: all-gathers through the slots, a stream through the boxes in rounds, barrier time-out on a
    # missing peer (tests/cpp/test_shm_link.cpp over psac_amd/csrc/shm_link.hpp)
    exe = str(tmp_path / "test_shm_link")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(HERE, "cpp", "test_shm_link.cpp"), "-lrt", "-pthread"])
    r = subprocess.run([exe, str(P)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
