"""Python mirror of the reference's suffix_array<> interface for one rank / one GPU."""
import ctypes as C
import sys

import numpy as np

from . import _lib
from ._lib import PSACX_LCP, PSACX_NO_FAST, PSACX_PROFILE, PsacxError, Stats

NEAREST_SM, NEAREST_EQ, FURTHEST_EQ = 0, 1, 2      # ansv_common.hpp:20-22


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context(object):
    """One HIP device + stream + reusable HBM workspace (psacx_ctx)."""

    def __init__(self, device=0, stream=None):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.psacx_create(C.byref(h), int(device), C.c_void_p(stream) if stream else None)
        if rc != 0:
            raise PsacxError(rc, self._lib.psacx_strerror(rc).decode())
        self.handle = h
        self.device = device

    def check(self, rc):
        if rc != 0:
            msg = self._lib.psacx_strerror(rc).decode()
            detail = self._lib.psacx_last_hip_error(self.handle).decode()
            raise PsacxError(rc, msg + (" [" + detail + "]" if detail else ""))

    def stats(self):
        s = Stats()
        self.check(self._lib.psacx_get_stats(self.handle, C.byref(s)))
        return s

    def configure(self, **options):
        """psacx_configure: pins the form of single stages (include/psacx.h), e.g. configure(force_diet=1, diet_cap=1 << 20), configure(reset=0)."""
        for name, value in options.items():
            self.check(self._lib.psacx_configure(self.handle, _lib.OPTIONS[name], int(value)))

    def _pre(self):
        """Before every call that runs the engine: with _lib.ENV_KNOBS the options come from PSACX_* variables (debug shim)."""
        if _lib.ENV_KNOBS:
            self.check(self._lib.psacx_configure_from_env(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.psacx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # raw device memory (for callers without their own HIP bindings)
    def alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self._lib.psacx_dev_alloc(self.handle, C.byref(p), nbytes))
        return p.value

    def free(self, p):
        self.check(self._lib.psacx_dev_free(self.handle, C.c_void_p(p)))

    def h2d(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        self.check(self._lib.psacx_copy_h2d(self.handle, C.c_void_p(dptr), _ptr(a), a.nbytes))

    def d2h(self, arr, dptr):
        assert arr.flags["C_CONTIGUOUS"]
        self.check(self._lib.psacx_copy_d2h(self.handle, _ptr(arr), C.c_void_p(dptr), arr.nbytes))


def parse_stringset(strings, sep=None):
    """simple_dstringset::parse (stringset.hpp:43-72) on one rank: returns (characters of all strings
    back to back as uint8, uint64 offsets[m + 1]).  A flat buffer is cut at runs of `sep`; empty
    strings do not exist in a set."""
    if isinstance(strings, (list, tuple)):
        parts = [np.frombuffer(x.encode("latin-1") if isinstance(x, str) else bytes(x), dtype=np.uint8) for x in strings]
        parts = [p for p in parts if p.size]
        off = np.zeros(len(parts) + 1, np.uint64)
        if parts:
            off[1:] = np.cumsum([p.size for p in parts])
        return (np.concatenate(parts) if parts else np.zeros(0, np.uint8)), off
    flat = strings
    if isinstance(flat, str):
        flat = flat.encode("latin-1")
    flat = np.frombuffer(bytes(flat), dtype=np.uint8) if isinstance(flat, (bytes, bytearray)) \
        else np.ascontiguousarray(flat, dtype=np.uint8)
    s = ord('$') if sep is None else (ord(sep) if isinstance(sep, (str, bytes)) else int(sep))
    keep = flat != s
    text = np.ascontiguousarray(flat[keep])
    # a string starts at every kept character whose predecessor is a separator (or the start)
    prev_sep = np.ones(flat.size, bool)
    prev_sep[1:] = ~keep[:-1]
    starts_flat = np.nonzero(keep & prev_sep)[0]
    kept_before = np.cumsum(keep) - keep            # kept characters before each flat position
    off = np.zeros(starts_flat.size + 1, np.uint64)
    off[:-1] = kept_before[starts_flat]
    off[-1] = text.size
    return text, off


class SuffixArray(object):
    """suffix_array<char, index_t, LCP> for a whole text held by one rank.

    Fields follow the reference (suffix_array.hpp:180-212): n, local_size,
    local_SA, local_B (0-based inverse suffix array after construct()),
    local_LCP (empty unless lcp=True), local_Lc (left-branching characters,
    suffix_array.hpp:211-212, empty unless lc=True; lc implies lcp).  `log` receives the reference's stderr
    lines ("Alphabet: ...", "iteration h: unfinished buckets = ...").
    """

    def __init__(self, index_bits=64, lcp=False, ctx=None, log=None, lc=False):
        if index_bits not in (32, 64):
            raise ValueError("index_bits must be 32 or 64")
        self.index_bits = index_bits
        self.lc = bool(lc)
        self.lcp = bool(lcp) or self.lc
        self.ctx = ctx if ctx is not None else Context(0)
        self.dtype = np.uint32 if index_bits == 32 else np.uint64
        self.log = log
        self.n = 0
        self.local_size = 0
        self.p = 1
        self.local_SA = np.zeros(0, self.dtype)
        self.local_B = np.zeros(0, self.dtype)
        self.local_LCP = np.zeros(0, self.dtype)
        self.local_Lc = np.zeros(0, np.uint8)
        self.k = 0
        self.sigma = 0
        self.bits_per_char = 0
        self.rounds = []

    def _flags(self, fast_resolval, profile):
        f = 0
        if self.lcp:
            f |= PSACX_LCP
        if not fast_resolval:
            f |= PSACX_NO_FAST
        if profile:
            f |= PSACX_PROFILE
        return f

    def _after(self):
        s = self.ctx.stats()
        self.k, self.sigma, self.bits_per_char = s.k, s.sigma, s.bits_per_char
        self.rounds = [(r.h, r.unfinished_buckets, r.unfinished_elements, r.active, r.sort_passes,
                        r.sort_passes_skipped) for r in s.rounds[:s.n_rounds]]
        if self.log is not None:
            for r in self.rounds:
                self.log.write("iteration %d: unfinished buckets = %d, unfinished elements = %d\n" % r[:3])
        return s

    def construct(self, text, fast_resolval=True, k=0, profile=False):
        """suffix_array::construct(begin, end, fast_resolval, k) (suffix_array.hpp:469-486)."""
        if isinstance(text, str):
            text = text.encode("latin-1")
        t = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) \
            else np.ascontiguousarray(text, dtype=np.uint8)
        n = int(t.size)
        if n == 0:
            raise ValueError("empty input")
        if self.index_bits == 32 and n > 0xFFFFFFFE:
            raise PsacxError(-2, "input too long for the index type")
        self.n = self.local_size = n
        self.local_SA = np.empty(n, self.dtype)
        self.local_B = np.empty(n, self.dtype)
        self.local_LCP = np.empty(n, self.dtype) if self.lcp else np.zeros(0, self.dtype)
        self.local_Lc = np.empty(n, np.uint8) if self.lc else np.zeros(0, np.uint8)
        args = [self.ctx.handle, _ptr(t), n, int(k), self._flags(fast_resolval, profile), _ptr(self.local_SA),
                _ptr(self.local_B), _ptr(self.local_LCP) if self.lcp else None]
        if self.lc:
            fn = getattr(self.ctx._lib, "psacx_construct_lc_u%d" % self.index_bits)
            args.append(_ptr(self.local_Lc))
        else:
            fn = getattr(self.ctx._lib, "psacx_construct_u%d" % self.index_bits)
        self.ctx._pre()
        self.ctx.check(fn(*args))
        return self._after()

    def construct_into(self, text, SA, B, LCP=None, fast_resolval=True, k=0):
        """construct() writing into result arrays the caller already holds (a second construct() on the same
        object reuses its vectors in the reference, test/test_psac.cpp:148-170).  Returns (SA, B, LCP)."""
        t = np.ascontiguousarray(text, dtype=np.uint8)
        n = int(t.size)
        assert SA.size == n and B.size == n and SA.dtype == self.dtype and B.dtype == self.dtype
        assert (LCP is not None and LCP.size == n) or not self.lcp
        fn = getattr(self.ctx._lib, "psacx_construct_u%d" % self.index_bits)
        self.ctx._pre()
        self.ctx.check(fn(self.ctx.handle, _ptr(t), n, int(k), self._flags(fast_resolval, False), _ptr(SA), _ptr(B),
                          _ptr(LCP) if self.lcp else None))
        self.n = self.local_size = n
        self._after()
        return SA, B, LCP

    def construct_ss(self, strings, sep=None, k=0, profile=False):
        """suffix_array::construct_ss(simple_dstringset&, alphabet) (suffix_array.hpp:267-363): the
        generalized suffix array of a set of strings.  `strings` is a list of byte strings, or one
        flat buffer cut at runs of `sep` (stringset.hpp:43-72, default '$' as stringset.hpp:147; gsac
        passes '\\n', src/gsac.cpp:170).
        Positions in local_SA count the characters of the strings back to back, separators left out."""
        text, off = parse_stringset(strings, sep)
        n = int(text.size)
        if n == 0:
            raise ValueError("empty input")
        if self.index_bits == 32 and n > 0xFFFFFFFE:
            raise PsacxError(-2, "input too long for the index type")
        if self.lc:
            raise ValueError("left-branching characters are not defined for string sets")
        self.n = self.local_size = n
        self.local_SA = np.empty(n, self.dtype)
        self.local_B = np.empty(n, self.dtype)
        self.local_LCP = np.empty(n, self.dtype) if self.lcp else np.zeros(0, self.dtype)
        fn = getattr(self.ctx._lib, "psacx_construct_gsa_u%d" % self.index_bits)
        self.ctx._pre()
        self.ctx.check(fn(self.ctx.handle, _ptr(text), n, _ptr(off), int(off.size - 1), int(k), self._flags(True, profile),
                          _ptr(self.local_SA), _ptr(self.local_B), _ptr(self.local_LCP) if self.lcp else None))
        self.string_offsets = off
        return self._after()

    def construct_device(self, d_text, n, d_sa, d_isa, d_lcp=None, fast_resolval=True, k=0, profile=False, d_lc=None):
        """Same with every buffer already resident in HBM (raw device addresses)."""
        args = [self.ctx.handle, C.c_void_p(d_text), int(n), int(k), self._flags(fast_resolval, profile),
                C.c_void_p(d_sa), C.c_void_p(d_isa), C.c_void_p(d_lcp) if d_lcp else None]
        if d_lc:
            fn = getattr(self.ctx._lib, "psacx_construct_lc_dev_u%d" % self.index_bits)
            args.append(C.c_void_p(d_lc))
        else:
            fn = getattr(self.ctx._lib, "psacx_construct_dev_u%d" % self.index_bits)
        self.ctx._pre()
        self.ctx.check(fn(*args))
        self.n = self.local_size = int(n)
        return self._after()


def suffix_tree(text, SA, LCP, ctx=None):
    """construct_suffix_tree(sa, begin, end, comm) (suffix_tree.hpp:413-499) at one rank: the
    n x (sigma + 1) node table (see psacx_suffix_tree_* in include/psacx.h)."""
    if isinstance(text, str):
        text = text.encode("latin-1")
    t = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
    sa, lcp = np.ascontiguousarray(SA), np.ascontiguousarray(LCP)
    ctx = ctx if ctx is not None else Context(0)
    fn = getattr(ctx._lib, "psacx_suffix_tree_u%d" % (sa.dtype.itemsize * 8))
    sigma = C.c_uint32(0)
    ctx.check(fn(ctx.handle, _ptr(t), t.size, None, None, None, C.byref(sigma)))
    nodes = np.zeros(t.size * (sigma.value + 1), np.uint64)
    ctx.check(fn(ctx.handle, _ptr(t), t.size, _ptr(sa), _ptr(lcp), _ptr(nodes), C.byref(sigma)))
    return nodes.reshape(t.size, sigma.value + 1)


def check_device(ctx, d_text, n, d_sa, d_isa, d_lcp, index_bits):
    """check_SA / check_lcp on buffers resident in HBM (check_suffix_array.hpp:56-126).  Returns the four
    error counters of psacx_check_dev_*; all zero means correct."""
    err = (C.c_uint64 * 4)()
    fn = getattr(ctx._lib, "psacx_check_dev_u%d" % index_bits)
    ctx.check(fn(ctx.handle, C.c_void_p(d_text), int(n), C.c_void_p(d_sa), C.c_void_p(d_isa),
                 C.c_void_p(d_lcp) if d_lcp else None, err))
    return list(err)


def ansv_device(ctx, d_in, n, d_left, d_right, index_bits, left_type=NEAREST_SM, right_type=NEAREST_SM, nonsv=0):
    """ansv<T, left_type, right_type> with the input (n index_t) and both results (n uint64 each) resident in
    HBM, e.g. over the LCP array construct_device left there (suffix_tree.hpp:62)."""
    fn = getattr(ctx._lib, "psacx_ansv_dev_u%d" % index_bits)
    ctx.check(fn(ctx.handle, C.c_void_p(d_in), int(n), int(left_type), int(right_type), int(nonsv),
                 C.c_void_p(d_left), C.c_void_p(d_right)))


def ansv(values, left_type=NEAREST_SM, right_type=NEAREST_SM, nonsv=0, ctx=None):
    """ansv<T,left_type,right_type>(in, left_nsv, right_nsv, comm) (ansv.hpp:2042-2051), one rank."""
    v = np.ascontiguousarray(values)
    if v.dtype not in (np.uint32, np.uint64):
        raise TypeError("ansv needs uint32 or uint64 input")
    ctx = ctx if ctx is not None else Context(0)
    left = np.empty(v.size, np.uint64)
    right = np.empty(v.size, np.uint64)
    fn = getattr(ctx._lib, "psacx_ansv_u%d" % (v.dtype.itemsize * 8))
    ctx.check(fn(ctx.handle, _ptr(v), v.size, int(left_type), int(right_type), int(nonsv), _ptr(left), _ptr(right)))
    return left, right
