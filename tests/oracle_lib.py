"""ctypes wrapper around oracle/libpsac_oracle.so (the CPU restatement of psac).

Test infrastructure: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only -- never from psac_amd/.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libpsac_oracle.so")


class Trace(C.Structure):
    _fields_ = [("h", C.c_uint64), ("unfinished_buckets", C.c_uint64),
                ("unfinished_elements", C.c_uint64), ("phase", C.c_uint32), ("pad", C.c_uint32)]


def build():
    src = os.path.join(ROOT, "oracle", "psac_ref.cpp")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libpsac_oracle.so"],
                              stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.psac_ref_fnv64_u64.restype = C.c_uint64
        _lib.psac_ref_fnv64_u32.restype = C.c_uint64
        _lib.psac_ref_range_min_u32.restype = C.c_uint32
        _lib.psac_ref_range_min_u64.restype = C.c_uint64
    return _lib


_SO_MT = os.path.join(ROOT, "oracle", "libpsac_oracle_mt.so")


def construct_all_cores(text, bits=32):
    """The same restatement built with OpenMP loops and the libstdc++ parallel-mode sort
    (oracle/Makefile: libpsac_oracle_mt.so), for bench.py's CPU baseline leg.  Returns (SA, LCP)."""
    src = os.path.join(ROOT, "oracle", "psac_ref.cpp")
    if (not os.path.exists(_SO_MT)) or os.path.getmtime(_SO_MT) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libpsac_oracle_mt.so"], stdout=subprocess.DEVNULL)
    mt = C.CDLL(_SO_MT)
    t = as_text(text)
    n = t.size
    dt = _dt(bits)
    SA = np.zeros(n, dt); ISA = np.zeros(n, dt); LCP = np.zeros(n, dt)
    rc = getattr(mt, "psac_ref_construct_u%d" % bits)(_p(t), C.c_uint64(n), C.c_int(1), C.c_uint(0), _p(SA), _p(ISA), _p(LCP),
                                                     None, C.c_uint32(0), None, None, None)
    if rc != 0:
        raise RuntimeError("oracle construct failed rc=%d" % rc)
    return SA, LCP


def reference_sa_lcp_cached(tag, text, bits=32, isa=None):
    """SA by libdivsufsort and LCP by Kasai (what test/test_psac.cpp:50-98 checks psac with) for a text several test modules check
    against (the 2^27-character tandem twin of configs[4]; the restatement takes 80 s on it, these 20): the first caller of a session
    leaves both under the system's temporary directory, keyed by the tag and a checksum of the text.  isa: an inverse of the SA the
    caller has verified (saves the inversion on the host)."""
    import tempfile
    import zlib
    t = as_text(text)
    key = "%s_%d_%08x_u%d" % (tag, t.size, zlib.crc32(t[:: max(1, t.size // (1 << 20))].tobytes()), bits)
    d = os.path.join(tempfile.gettempdir(), "psacx_oracle_cache")
    fa, fl = os.path.join(d, key + ".sa.npy"), os.path.join(d, key + ".lcp.npy")
    if os.path.exists(fa) and os.path.exists(fl):
        try:
            SA, LCP = np.load(fa), np.load(fl)
            if SA.size == t.size and LCP.size == t.size:
                return SA, LCP
        except Exception as e:
            sys.stderr.write("[oracle cache] unreadable %s: %r\n" % (key, e))
    SA = divsufsort(t, bits)
    if isa is None or not np.array_equal(isa[SA.astype(np.int64)], np.arange(t.size, dtype=SA.dtype)):
        isa = inverse(SA)
    LCP = kasai(t, SA, isa)
    try:
        os.makedirs(d, exist_ok=True)
        np.save(fa, SA); np.save(fl, LCP)
    except Exception as e:
        sys.stderr.write("[oracle cache] not saved %s: %r\n" % (key, e))
    return SA, LCP


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _dt(bits):
    return np.uint32 if bits == 32 else np.uint64


def as_text(x):
    if isinstance(x, (bytes, bytearray)):
        return np.frombuffer(bytes(x), dtype=np.uint8).copy()
    if isinstance(x, str):
        return np.frombuffer(x.encode("latin-1"), dtype=np.uint8).copy()
    return np.ascontiguousarray(x, dtype=np.uint8)


def construct(text, bits=32, fast=True, k=0, lcp=True):
    """psac's suffix_array<>::construct at p=1.  Returns dict(SA, ISA, LCP, trace, k, l)."""
    t = as_text(text)
    n = t.size
    dt = _dt(bits)
    SA = np.zeros(n, dt); ISA = np.zeros(n, dt)
    LCP = np.zeros(n, dt) if lcp else None
    tr = (Trace * 256)()
    trn = C.c_uint32(0); ku = C.c_uint32(0); lu = C.c_uint32(0)
    f = getattr(lib(), "psac_ref_construct_u%d" % bits)
    rc = f(_p(t), C.c_uint64(n), C.c_int(1 if fast else 0), C.c_uint(k), _p(SA), _p(ISA),
           _p(LCP) if lcp else None, tr, C.c_uint32(256), C.byref(trn), C.byref(ku), C.byref(lu))
    if rc != 0:
        raise RuntimeError("oracle construct failed rc=%d" % rc)
    trace = [(tr[i].h, tr[i].unfinished_buckets, tr[i].unfinished_elements, tr[i].phase)
             for i in range(min(trn.value, 256))]
    return dict(SA=SA, ISA=ISA, LCP=LCP, trace=trace, k=ku.value, l=lu.value)


def construct_lc(text, bits=32, fast=True, k=0):
    """suffix_array<char, T, true, true>::construct at p=1: dict(SA, ISA, LCP, Lc)."""
    t = as_text(text)
    n = t.size
    dt = _dt(bits)
    SA = np.zeros(n, dt); ISA = np.zeros(n, dt); LCP = np.zeros(n, dt); Lc = np.zeros(n, np.uint8)
    f = getattr(lib(), "psac_ref_construct_lc_u%d" % bits)
    rc = f(_p(t), C.c_uint64(n), C.c_int(1 if fast else 0), C.c_uint(k), _p(SA), _p(ISA), _p(LCP), _p(Lc))
    if rc != 0:
        raise RuntimeError("oracle construct_lc failed rc=%d" % rc)
    return dict(SA=SA, ISA=ISA, LCP=LCP, Lc=Lc)


def flatten(strings):
    """Strings back to back plus their offsets (what simple_dstringset holds, stringset.hpp:33-81)."""
    parts = [as_text(x) for x in strings]
    off = np.zeros(len(parts) + 1, np.uint64)
    off[1:] = np.cumsum([p.size for p in parts])
    return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), off


def construct_ss(strings, bits=64, k=0, lcp=True):
    """suffix_array<char, T, LCP>::construct_ss at p=1 (suffix_array.hpp:267-363)."""
    t, off = flatten(strings)
    n = t.size
    dt = _dt(bits)
    SA = np.zeros(n, dt); ISA = np.zeros(n, dt)
    LCP = np.zeros(n, dt) if lcp else None
    tr = (Trace * 256)()
    trn = C.c_uint32(0)
    f = getattr(lib(), "psac_ref_construct_ss_u%d" % bits)
    rc = f(_p(t), C.c_uint64(n), _p(off), C.c_uint64(len(off) - 1), C.c_uint(k), _p(SA), _p(ISA),
           _p(LCP) if lcp else None, tr, C.c_uint32(256), C.byref(trn))
    if rc != 0:
        raise RuntimeError("oracle construct_ss failed rc=%d" % rc)
    trace = [(tr[i].h, tr[i].unfinished_buckets, tr[i].unfinished_elements, tr[i].phase)
             for i in range(min(trn.value, 256))]
    return dict(SA=SA, ISA=ISA, LCP=LCP, trace=trace, text=t, off=off)


def gsa_by_definition(strings):
    """Suffixes of every string (ended by a marker below all characters), ties by position; LCP
    bounded by the string ends.  Quadratic; small inputs only."""
    t, off = flatten(strings)
    sufs = []
    for s in range(len(off) - 1):
        b, e = int(off[s]), int(off[s + 1])
        for i in range(b, e):
            sufs.append((bytes(t[i:e]), i))
    sufs.sort()
    SA = np.array([i for _, i in sufs], np.uint64)
    LCP = np.zeros(len(sufs), np.uint64)
    for j in range(1, len(sufs)):
        a, b = sufs[j - 1][0], sufs[j][0]
        c = 0
        while c < len(a) and c < len(b) and a[c] == b[c]:
            c += 1
        LCP[j] = c
    return SA, LCP


def left_chars_by_definition(text, SA, LCP):
    """desa.hpp:262-264: Lc[i] = S[SA[i-1] + LCP[i]] (0 past the end and at i = 0)."""
    t = as_text(text)
    n = t.size
    out = np.zeros(n, np.uint8)
    if n > 1:
        p = SA[:-1].astype(np.int64) + LCP[1:].astype(np.int64)
        ok = p < n
        out[1:][ok] = t[p[ok]]
    return out


def kasai(text, SA, ISA):
    t = as_text(text)
    bits = SA.dtype.itemsize * 8
    out = np.zeros(t.size, SA.dtype)
    getattr(lib(), "psac_ref_kasai_u%d" % bits)(_p(t), C.c_uint64(t.size), _p(np.ascontiguousarray(SA)),
                                                 _p(np.ascontiguousarray(ISA)), _p(out))
    return out


def check_sa(text, SA, ISA):
    t = as_text(text)
    bits = SA.dtype.itemsize * 8
    return getattr(lib(), "psac_ref_check_sa_u%d" % bits)(_p(t), C.c_uint64(t.size),
                                                          _p(np.ascontiguousarray(SA)),
                                                          _p(np.ascontiguousarray(ISA)))


def naive_sa(text, bits=32):
    t = as_text(text)
    out = np.zeros(t.size, _dt(bits))
    getattr(lib(), "psac_ref_naive_sa_u%d" % bits)(_p(t), C.c_uint64(t.size), _p(out))
    return out


def kmers(text, k, bits=32):
    t = as_text(text)
    out = np.zeros(t.size, _dt(bits))
    getattr(lib(), "psac_ref_kmers_u%d" % bits)(_p(t), C.c_uint64(t.size), C.c_uint(k), _p(out))
    return out


def alphabet(text):
    t = as_text(text)
    code = np.zeros(256, np.uint16)
    s = C.c_uint32(0); b = C.c_uint32(0)
    lib().psac_ref_alphabet(_p(t), C.c_uint64(t.size), _p(code), C.byref(s), C.byref(b))
    return code, s.value, b.value


def optimal_k(word_bits, l, n, k=0):
    return lib().psac_ref_optimal_k(C.c_uint(word_bits), C.c_uint(l), C.c_uint64(n), C.c_uint(k))


def rebucket(b1, b2):
    bits = b1.dtype.itemsize * 8
    v1 = np.ascontiguousarray(b1).copy(); v2 = np.ascontiguousarray(b2)
    ub = C.c_uint64(0); ue = C.c_uint64(0)
    getattr(lib(), "psac_ref_rebucket_u%d" % bits)(_p(v1), _p(v2), C.c_uint64(v1.size), C.byref(ub), C.byref(ue))
    return v1, ub.value, ue.value


def lcp_bitwise(x, y, bits, k, l):
    ct = C.c_uint32 if bits == 32 else C.c_uint64
    return getattr(lib(), "psac_ref_lcp_bitwise_u%d" % bits)(ct(x), ct(y), C.c_uint(k), C.c_uint(l))


def leading_zeros(x, bits):
    ct = C.c_uint32 if bits == 32 else C.c_uint64
    return getattr(lib(), "psac_ref_leading_zeros_u%d" % bits)(ct(x))


def trailing_zeros(x, bits):
    ct = C.c_uint32 if bits == 32 else C.c_uint64
    return getattr(lib(), "psac_ref_trailing_zeros_u%d" % bits)(ct(x))


def floorlog2(x):
    return lib().psac_ref_floorlog2(C.c_uint64(x))


def ceillog2(x):
    return lib().psac_ref_ceillog2(C.c_uint64(x))


def rand_dna(n, seed):
    out = np.zeros(n, np.uint8)
    lib().psac_ref_rand_dna(C.c_uint64(n), C.c_int(seed), _p(out))
    return out


def ansv(values, left, kind, nonsv):
    """kind: 0 nearest_sm, 1 nearest_eq, 2 furthest_eq (ansv_common.hpp:20-22)."""
    v = np.ascontiguousarray(values)
    bits = v.dtype.itemsize * 8
    out = np.zeros(v.size, np.uint64)
    getattr(lib(), "psac_ref_ansv_u%d" % bits)(_p(v), C.c_uint64(v.size), C.c_int(1 if left else 0),
                                               C.c_int(kind), C.c_uint64(nonsv), _p(out))
    return out


def ansv_seq(values, left, nonsv):
    v = np.ascontiguousarray(values)
    bits = v.dtype.itemsize * 8
    out = np.zeros(v.size, np.uint64)
    getattr(lib(), "psac_ref_ansv_seq_u%d" % bits)(_p(v), C.c_uint64(v.size), C.c_int(1 if left else 0),
                                                   C.c_uint64(nonsv), _p(out))
    return out


def suffix_tree(text, SA, LCP):
    """construct_suffix_tree at p=1 (suffix_tree.hpp:440-499): (n x (sigma+1)) node table."""
    t = as_text(text)
    bits = SA.dtype.itemsize * 8
    code, sigma, _ = alphabet(t)
    nodes = np.zeros(t.size * (sigma + 1), np.uint64)
    sg = C.c_uint32(0)
    getattr(lib(), "psac_ref_suffix_tree_u%d" % bits)(_p(t), C.c_uint64(t.size), _p(np.ascontiguousarray(SA)),
                                                      _p(np.ascontiguousarray(LCP)), _p(nodes), C.byref(sg))
    return nodes.reshape(t.size, sigma + 1)


def range_min(values, l, r):
    v = np.ascontiguousarray(values)
    bits = v.dtype.itemsize * 8
    return getattr(lib(), "psac_ref_range_min_u%d" % bits)(_p(v), C.c_uint64(v.size), C.c_uint64(l), C.c_uint64(r))


def fnv(arr):
    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint32:
        return lib().psac_ref_fnv64_u32(_p(a), C.c_uint64(a.size))
    a = a.astype(np.uint64)
    return lib().psac_ref_fnv64_u64(_p(a), C.c_uint64(a.size))


# ---------------------------------------------------------------------------------------------------------------
# libdivsufsort, the reference's own checker (test/test_psac.cpp:77-98 compares psac's SA with dss::construct) and
# CPU baseline (src/psac_vs_dss.cpp:87-119), built by oracle/Makefile from the sources under
# /root/reference/ext/libdivsufsort into oracle/_ref/ (git-ignored; the built files travel to the GPU box).
# ---------------------------------------------------------------------------------------------------------------
_DSS = {32: os.path.join(ROOT, "oracle", "_ref", "libdivsufsort.so"), 64: os.path.join(ROOT, "oracle", "_ref", "libdivsufsort64.so")}
_dss_libs = {}


def have_divsufsort():
    if not all(os.path.exists(p) for p in _DSS.values()) and os.path.isdir("/root/reference/ext/libdivsufsort/lib"):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    return all(os.path.exists(p) for p in _DSS.values())


def divsufsort(text, bits=32):
    """dss::construct (divsufsort_wrapper.hpp:54-74): the suffix array by libdivsufsort (divsufsort for 32-bit
    indices, n < 2^31 - 1; divsufsort64 otherwise), returned as unsigned index_t like psac's local_SA."""
    if not have_divsufsort():
        raise RuntimeError("oracle/_ref/libdivsufsort*.so not built")
    t = as_text(text)
    n = t.size
    if bits == 32 and n >= (1 << 31) - 1:
        raise ValueError("Input size is too large for 32bit indexing.")
    if bits not in _dss_libs:
        _dss_libs[bits] = C.CDLL(_DSS[bits])
    L = _dss_libs[bits]
    SA = np.zeros(n, np.int32 if bits == 32 else np.int64)
    fn = L.divsufsort if bits == 32 else L.divsufsort64
    fn.restype = C.c_int32
    rc = fn(_p(t), _p(SA), C.c_int32(n) if bits == 32 else C.c_int64(n))
    if rc != 0:
        raise RuntimeError("divsufsort failed rc=%d" % rc)
    return SA.view(np.uint32 if bits == 32 else np.uint64)


def sufcheck(text, SA):
    """dss::check -> sufcheck (divsufsort_wrapper.hpp:76-100): 0 if SA is the suffix array of text."""
    t = as_text(text)
    bits = SA.dtype.itemsize * 8
    if bits not in _dss_libs:
        _dss_libs[bits] = C.CDLL(_DSS[bits])
    L = _dss_libs[bits]
    fn = L.sufcheck if bits == 32 else L.sufcheck64
    fn.restype = C.c_int32
    a = np.ascontiguousarray(SA).view(np.int32 if bits == 32 else np.int64)
    return fn(_p(t), _p(a), C.c_int32(t.size) if bits == 32 else C.c_int64(t.size), C.c_int32(0))


def inverse(SA):
    isa = np.empty_like(SA)
    isa[SA.astype(np.int64)] = np.arange(SA.size, dtype=SA.dtype)
    return isa


def divsufsort_sa_lcp(text, bits=32):
    """SA by libdivsufsort, ISA by inversion, LCP by Kasai (lcp.hpp:46-77) -- what test/test_psac.cpp compares
    psac's results with (:77-98 SA, :50-74 LCP)."""
    SA = divsufsort(text, bits)
    ISA = inverse(SA)
    return SA, ISA, kasai(text, SA, ISA)
