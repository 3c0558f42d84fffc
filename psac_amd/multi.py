"""Python mirror of the multi-GPU boundary (psacx_multi_* in include/psacx.h): suffix_array<> on p ranks, one GPU each.

Two deployments, as in the C ABI:
  MultiContext(dev_ids)                 one process drives all GPUs (ranks 0..p-1); a device listed several times carries
                                        several ranks (the tests' way of running p ranks on a one-GPU box)
  MultiContext.for_rank(rank, p, device, unique_id)
                                        one process per GPU (psac's own model; what bench.py runs under torchrun)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PSACX_LCP, PsacxError, Stats


def unique_id():
    """128 bytes identifying a new communicator; make it on one process and broadcast it to the others."""
    buf = (C.c_uint8 * 128)()
    rc = _lib.load().psacx_multi_unique_id(buf)
    if rc != 0:
        raise PsacxError(rc, "psacx_multi_unique_id failed (librccl not available?)")
    return bytes(buf)


def _env_create_flags(lib):
    """(flags, shm_box_bytes) of a new context from the PSACX_MULTI_* transport variables -- only with _lib.ENV_KNOBS (tests, tools)."""
    flags, box = 0, 0
    if _lib.ENV_KNOBS:
        env = lambda name: lib.psacx_debug_env(name.encode())
        if env("PSACX_MULTI_FORCE_WIRE"):
            flags |= _lib.MULTI_FORCE_WIRE
        if env("PSACX_MULTI_NO_RCCL"):
            flags |= _lib.MULTI_NO_RCCL
        if (env("PSACX_MULTI_TRANSPORT") or b"") == b"shm":
            flags |= _lib.MULTI_SHM
        box = int(env("PSACX_SHM_BOX") or 0)
    return flags, box


class MultiContext(object):
    def __init__(self, dev_ids=None, ndev=None, _handle=None, force_wire=False, no_rccl=False):
        self._lib = _lib.load()
        if _handle is not None:
            self.handle = _handle
        else:
            if dev_ids is None:
                dev_ids = list(range(int(ndev)))
            arr = (C.c_int * len(dev_ids))(*[int(d) for d in dev_ids])
            h = C.c_void_p()
            flags = (_lib.MULTI_FORCE_WIRE if force_wire else 0) | (_lib.MULTI_NO_RCCL if no_rccl else 0) | (_env_create_flags(self._lib)[0] & ~_lib.MULTI_SHM)
            rc = self._lib.psacx_multi_create_ex(C.byref(h), len(dev_ids), arr, flags)
            if rc != 0:
                raise PsacxError(rc, self._lib.psacx_strerror(rc).decode())
            self.handle = h
        self.nranks = self._lib.psacx_multi_nranks(self.handle)
        self.nlocal = self._lib.psacx_multi_nlocal(self.handle)
        self.uses_rccl = bool(self._lib.psacx_multi_uses_rccl(self.handle))
        self.transport = ("copy", "rccl", "shm")[self._lib.psacx_multi_transport(self.handle)]

    @classmethod
    def for_rank(cls, rank, nranks, device, uid, force_wire=False, shm=False, shm_box_bytes=0):
        lib = _lib.load()
        h = C.c_void_p()
        buf = (C.c_uint8 * 128)(*bytearray(uid)) if uid is not None else None
        eflags, ebox = _env_create_flags(lib)
        flags = (_lib.MULTI_FORCE_WIRE if force_wire else 0) | (_lib.MULTI_SHM if shm else 0) | (eflags & ~_lib.MULTI_NO_RCCL)
        rc = lib.psacx_multi_create_rank_ex(C.byref(h), int(rank), int(nranks), int(device), buf, flags, int(shm_box_bytes or ebox))
        if rc != 0:
            raise PsacxError(rc, "psacx_multi_create_rank: %s" % lib.psacx_strerror(rc).decode())
        return cls(_handle=h)

    def check(self, rc):
        if rc != 0:
            msg = self._lib.psacx_strerror(rc).decode() if rc > -7 else "RCCL failure"
            det = self._lib.psacx_multi_last_error(self.handle).decode()
            raise PsacxError(rc, msg + (" [" + det + "]" if det else ""))

    def rank_ctx(self, i):
        """psacx_ctx handle of local rank i (device memory helpers psacx_dev_alloc / psacx_copy_* take it)."""
        return C.c_void_p(self._lib.psacx_multi_ctx(self.handle, int(i)))

    def stats(self):
        s = Stats()
        sent, ex, ga = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self.check(self._lib.psacx_multi_get_stats(self.handle, C.byref(s), C.byref(sent), C.byref(ex), C.byref(ga)))
        return s, sent.value, ex.value, ga.value

    def wire(self):
        """After a call: {"sends", "recvs", "allgathers"}: ncclSend / ncclRecv / ncclAllGather calls this process really
        issued, and "exchange_ms": time the exchanges occupied each local rank's second stream."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        ms = (C.c_double * self.nlocal)()
        self.check(self._lib.psacx_multi_get_wire(self.handle, C.byref(a), C.byref(b), C.byref(c), ms))
        return {"sends": a.value, "recvs": b.value, "allgathers": c.value, "exchange_ms": [round(x, 3) for x in ms]}

    def last_form(self):
        """{"two_word", "reduced_memory", "slice_inversion", "one_word", "tie_slabs"}: the forms the last construction took
        (psacx_multi_last_form); tie_slabs = slabs beyond the first in which the tie stage of the first round ran."""
        f = self._lib.psacx_multi_last_form(self.handle)
        return {"two_word": bool(f & 1), "reduced_memory": bool(f & 2), "slice_inversion": bool(f & 4), "one_word": bool(f & 16), "tie_slabs": (f >> 8) & 255}

    def phases(self):
        """Host wall time (ms) of the phases of the last construction, in order of first appearance."""
        buf = C.create_string_buffer(1 << 14)
        self.check(self._lib.psacx_multi_get_phases(self.handle, buf, len(buf)))
        out = []
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.rsplit("=", 1)
                out.append((k, float(v)))
        return out

    LAYOUT_AUTO, LAYOUT_NORMAL, LAYOUT_REDUCED = 0, 1, 2

    def configure(self, **options):
        """psacx_multi_configure: memory layout of the distributed construction (layout, slab, output_slack) and the forms of single
        stages (trace, wire_piece, pieces, two_word, ...; include/psacx.h)."""
        for name, val in options.items():
            if val is not None:
                self.check(self._lib.psacx_multi_configure(self.handle, _lib.MULTI_OPTIONS[name], int(val)))

    def _pre(self):
        """Before every call that runs the engine: with _lib.ENV_KNOBS the options come from PSACX_* variables (debug shim)."""
        if _lib.ENV_KNOBS:
            self.check(self._lib.psacx_multi_configure_from_env(self.handle))

    def memory(self):
        """(peak bytes of every local rank's block cache, reduced-memory layout used?, refinement rounds run in slabs)
        of the last construction."""
        peak = (C.c_uint64 * self.nlocal)()
        red, slabs = C.c_int32(0), C.c_uint32(0)
        self.check(self._lib.psacx_multi_get_memory(self.handle, peak, C.byref(red), C.byref(slabs)))
        return list(peak), bool(red.value), int(slabs.value)

    def construct(self, text, index_bits=64, lcp=True, k=0):
        """suffix_array<char, index_t, LCP>::construct on p ranks, the whole text and results on this host
        (needs every rank in this process).  Returns (SA, ISA, LCP or None, rounds)."""
        self._pre()
        if isinstance(text, str):
            text = text.encode("latin-1")
        t = np.frombuffer(bytes(text), dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        n = int(t.size)
        dt = np.uint32 if index_bits == 32 else np.uint64
        SA = np.empty(n, dt); ISA = np.empty(n, dt); LCP = np.empty(n, dt) if lcp else None
        fn = getattr(self._lib, "psacx_multi_construct_u%d" % index_bits)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.check(fn(self.handle, p(t), n, int(k), PSACX_LCP if lcp else 0, p(SA), p(ISA), p(LCP) if lcp else None))
        s = self.stats()[0]
        rounds = [(r.h, r.unfinished_buckets, r.unfinished_elements) for r in s.rounds[:s.n_rounds]]
        return SA, ISA, LCP, rounds

    def construct_ss(self, strings, sep=None, index_bits=64, lcp=True, k=0):
        """suffix_array::construct_ss (suffix_array.hpp:267-363) on p ranks: the generalized suffix array of a string set
        (a list of byte strings, or one flat buffer cut at runs of `sep`), the strings back to back in the
        block-distributed text.  Returns (SA, ISA, LCP or None, rounds, string offsets)."""
        self._pre()
        from .suffix_array import parse_stringset
        t, off = parse_stringset(strings, sep)
        n = int(t.size)
        dt = np.uint32 if index_bits == 32 else np.uint64
        SA = np.empty(n, dt); ISA = np.empty(n, dt); LCP = np.empty(n, dt) if lcp else None
        fn = getattr(self._lib, "psacx_multi_construct_gsa_u%d" % index_bits)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        self.check(fn(self.handle, p(t), n, p(off), int(off.size - 1), int(k), PSACX_LCP if lcp else 0, p(SA), p(ISA), p(LCP) if lcp else None))
        s = self.stats()[0]
        rounds = [(r.h, r.unfinished_buckets, r.unfinished_elements) for r in s.rounds[:s.n_rounds]]
        return SA, ISA, LCP, rounds, off

    def construct_device(self, d_text, m, d_sa, d_isa, d_lcp, index_bits, k=0):
        """Blocks resident in HBM: lists (one entry per local rank) of raw device addresses and block lengths."""
        self._pre()
        L = self.nlocal
        vp = C.c_void_p * L
        mm = (C.c_uint64 * L)(*[int(x) for x in m])
        fn = getattr(self._lib, "psacx_multi_construct_dev_u%d" % index_bits)
        lcp_arr = vp(*d_lcp) if d_lcp is not None else None
        self.check(fn(self.handle, vp(*d_text), mm, int(k), PSACX_LCP if d_lcp is not None else 0, vp(*d_sa), vp(*d_isa), lcp_arr))
        return self.stats()

    def check_device(self, d_text, m, d_sa, d_isa, d_lcp, index_bits):
        """Distributed d_check_sa (+ LCP recurrence) over blocks resident in HBM; returns the four error counters
        summed over all ranks (all zero = correct)."""
        self._pre()
        L = self.nlocal
        vp = C.c_void_p * L
        mm = (C.c_uint64 * L)(*[int(x) for x in m])
        err = (C.c_uint64 * 4)()
        fn = getattr(self._lib, "psacx_multi_check_dev_u%d" % index_bits)
        self.check(fn(self.handle, vp(*d_text), mm, vp(*d_sa), vp(*d_isa), vp(*d_lcp) if d_lcp is not None else None, err))
        return list(err)

    def left_chars_device(self, d_text, m, d_sa, d_lcp, d_lc, index_bits):
        """Left-branching characters Lc[i] = S[SA[i-1] + LCP[i]] of block-distributed results resident in HBM
        (suffix_array.hpp:211-212); d_lc[i] receives m[i] bytes."""
        self._pre()
        L = self.nlocal
        vp = C.c_void_p * L
        mm = (C.c_uint64 * L)(*[int(x) for x in m])
        fn = getattr(self._lib, "psacx_multi_left_chars_dev_u%d" % index_bits)
        self.check(fn(self.handle, vp(*d_text), mm, vp(*d_sa), vp(*d_lcp), vp(*d_lc)))

    def suffix_tree_device(self, d_text, m, d_sa, d_lcp, d_nodes, index_bits):
        """construct_suffix_tree on p ranks (suffix_tree.hpp:413-499) over blocks resident in HBM: d_nodes[i] receives the
        m[i] x (sigma + 1) rows of local rank i's LCP indices (uint64 cells); d_nodes=None only returns sigma."""
        self._pre()
        L = self.nlocal
        vp = C.c_void_p * L
        mm = (C.c_uint64 * L)(*[int(x) for x in m])
        sg = C.c_uint32(0)
        fn = getattr(self._lib, "psacx_multi_suffix_tree_dev_u%d" % index_bits)
        if d_nodes is None:
            self.check(fn(self.handle, vp(*d_text), mm, None, None, None, C.byref(sg)))
        else:
            self.check(fn(self.handle, vp(*d_text), mm, vp(*d_sa), vp(*d_lcp), vp(*d_nodes), C.byref(sg)))
        return int(sg.value)

    def ansv_device(self, d_in, m, d_left, d_right, index_bits, left_type=0, right_type=0, nonsv=0):
        """ansv<T, left_type, right_type, global_indexing> over a block-distributed array resident in HBM (lists of raw
        device addresses, one per local rank; results are uint64 global indices)."""
        self._pre()
        L = self.nlocal
        vp = C.c_void_p * L
        mm = (C.c_uint64 * L)(*[int(x) for x in m])
        fn = getattr(self._lib, "psacx_multi_ansv_dev_u%d" % index_bits)
        self.check(fn(self.handle, vp(*d_in), mm, int(left_type), int(right_type), int(nonsv), vp(*d_left), vp(*d_right)))

    def close(self):
        if getattr(self, "handle", None):
            self._lib.psacx_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
