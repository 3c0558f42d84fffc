"""The planning code of the shipped multi-GPU engine (psac_amd/csrc/multi_plan.hpp: block distribution, dealing of the top-digit
buckets with its messages and in-place re-balance, sample-sort splitters and exact re-balance, slice shapes) compiled with g++ --
no hipcc, no GPU -- and exercised by tests/cpp/test_multi_plan.cpp, which plays every planned exchange on host arrays for
P in {1, 2, 3, 7, 8} including skewed digit counts and a bucket longer than a block.  multi.hpp executes the same functions' plans
with ncclSend / ncclRecv.  The block sizes it prints are compared with the Python harness's own mxx::blk_dist."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def plan_output(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("plan") / "test_multi_plan")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(HERE, "cpp", "test_multi_plan.cpp")])
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:]
    return res.stdout


def test_plans_hold_on_host_arrays(plan_output):
    for part in ("ok blk", "ok deal", "ok sort", "ok slices"):
        assert part in plan_output
    assert "FAIL" not in plan_output


def test_block_distribution_agrees_with_the_harness(plan_output):
    from dist_harness import dist as D
    seen = 0
    for m in re.finditer(r"^blk n=(\d+) P=(\d+) sizes=([\d,]+)$", plan_output, re.M):
        n, P = int(m.group(1)), int(m.group(2))
        assert [int(x) for x in m.group(3).split(",")] == list(D.blk_sizes(n, P)), (n, P)
        seen += 1
    assert seen >= 20


def test_plan_header_has_no_device_code():
    # the header must stay compilable without HIP: no runtime call, no kernel, no device pointer arithmetic
    src = open(os.path.join(ROOT, "psac_amd", "csrc", "multi_plan.hpp")).read()
    for word in ("hipLaunch", "hipMalloc", "hipMemcpy", "__global__", "hipStream", "#include <hip"):
        assert word not in src, word
    # ... and the engine (multi.hpp and the headers that define its members) really calls it: the plans are not a test-only copy
    eng = "".join(open(os.path.join(ROOT, "psac_amd", "csrc", f)).read() for f in ("multi.hpp", "multi_first_round.hpp", "multi_queries.hpp"))
    for call in ("plan::deal_top_digit_buckets", "plan::in_place_messages", "plan::choose_splitters", "plan::sample_positions",
                 "plan::rebalance_bounds", "plan::slice_shape", "plan::follows_blk_dist"):
        assert call in eng, call
