"""CPU-side checks of the drop-in boundary: the library loads and exports every
symbol include/psacx.h declares; no compute call is made (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "psacx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psacx_[a-z0-9_]+)\s*\(", src)))


def declared_op_symbols():
    src = open(os.path.join(ROOT, "include", "psacx_ops.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(psacx_op_[a-z0-9_]+?)(?:_##S)?\s*\(", src))
    out = set()
    for n in names:
        if n == "psacx_op_char_hist":
            out.add(n)
        else:
            out.add(n + "_u32"); out.add(n + "_u64")
    return sorted(out)


def test_ops_header_symbols_are_exported():
    from psac_amd import _lib
    from dist_harness import dist_ops
    lib = _lib.load()
    syms = declared_op_symbols()
    assert len(syms) == 43
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(dist_ops.OP_EXPORTS) == syms


def test_header_symbols_are_exported():
    from psac_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.EXPORTS) == syms


def test_strerror_and_no_gpu_behaviour():
    import ctypes as C
    from psac_amd import _lib
    lib = _lib.load()
    assert lib.psacx_strerror(0) == b"ok"
    assert lib.psacx_strerror(-2) == b"input too long for the index type"
    assert lib.psacx_create(None, 0, None) == -1
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert lib.psacx_create(C.byref(h), 0, None) == -6      # PSACX_ENOGPU: fails loudly, no CPU fallback


def test_cpp_mirror_compiles_as_cxx11_and_fails_loudly_without_a_gpu(tmp_path):
    # include/suffix_array.hpp (the class contract, suffix_array.hpp:170-228) with every instantiation tests/cpp/test_header.cpp makes --
    # char / int symbols, 32- and 64-bit indices, more than 256 distinct symbols -- builds warning-free as C++11 against the library;
    # without a GPU the program ends with the library's error, not with results from somewhere else
    import subprocess
    exe = str(tmp_path / "test_header")
    lib = os.path.join(ROOT, "psac_amd", "lib")
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_header.cpp"),
           "-L" + lib, "-lpsacx", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
        assert r.returncode != 0 and "cpp header tests passed" not in r.stdout
        assert "psacx" in (r.stdout + r.stderr)


def test_product_does_not_touch_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "psac_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if "oracle" in txt.lower() and f not in ("_lib.py",):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
