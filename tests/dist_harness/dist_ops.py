"""HIP implementation of the LocalOps interface of tests/dist_harness/dist.py.

Every method is one call into the step-level C ABI (include/psacx_ops.h) on device
tensors that PyTorch merely owns (memory + stream + RCCL); there is no CPU or torch-op
fallback for the arithmetic.  Index arrays are torch.int32 / torch.int64 tensors holding
the unsigned bit patterns the kernels work on.
"""
import ctypes as C

import torch

from psac_amd import _lib


class Boundary(C.Structure):
    _fields_ = [("off", C.c_uint64), ("base", C.c_uint64), ("has_prev", C.c_int32), ("has_next", C.c_int32),
                ("prev", C.c_uint64 * 3), ("next", C.c_uint64 * 3)]


OPS = ["make_keys", "iota", "pair_sort", "split_by", "pair_bounds", "owners", "take", "put", "put_perm", "add_scalar", "finish_b2",
       "last_head", "rebucket_first", "rebucket_refine", "compact", "block_min", "range_min", "rmq_split",
       "rmq_combine", "lcp_apply", "nsv_from"]
OP_EXPORTS = ["psacx_op_char_hist"] + ["psacx_op_%s_%s" % (o, s) for o in OPS for s in ("u32", "u64")]


class HipOps(object):
    def __init__(self, index_bits=32, device=0):
        self.index_bits = index_bits
        self.tdt = torch.int32 if index_bits == 32 else torch.int64
        self.mask = (1 << index_bits) - 1
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.lib = _lib.load()
        self.suf = "u%d" % index_bits
        h = C.c_void_p()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # share PyTorch's stream so that kernels, copies and RCCL collectives stay ordered
        rc = self.lib.psacx_create(C.byref(h), int(device), C.c_void_p(stream if stream else -1))
        if rc != 0:
            raise _lib.PsacxError(rc, self.lib.psacx_strerror(rc).decode())
        self.ctx = h

    # -- plumbing ------------------------------------------------------------------------
    def _f(self, name):
        return getattr(self.lib, "psacx_op_%s_%s" % (name, self.suf))

    def _chk(self, rc):
        if rc != 0:
            msg = self.lib.psacx_strerror(rc).decode()
            det = self.lib.psacx_last_hip_error(self.ctx).decode()
            raise _lib.PsacxError(rc, msg + (" [" + det + "]" if det else ""))

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr() if t is not None and t.numel() else (t.data_ptr() if t is not None else 0))

    def empty_idx(self, m):
        return torch.empty(int(m), dtype=self.tdt, device=self.device)

    def empty_like(self, t):
        return torch.empty_like(t)

    def _u(self, x):
        return int(x) & self.mask

    def value_at(self, t, j):
        return self._u(t[j].item())

    def record_at(self, a, b, c, j):
        return (self._u(a[j].item()), self._u(b[j].item()), self._u(c[j].item()))

    def sample(self, S1, S2, positions):
        if not positions:
            return []
        idx = torch.tensor(positions, dtype=torch.int64, device=self.device)
        a = S1[idx].tolist(); b = S2[idx].tolist()          # device-side row selection, then a small copy
        return [(self._u(x), self._u(y)) for x, y in zip(a, b)]

    def _bd(self, prev, nxt, off, base):
        b = Boundary()
        b.off, b.base = int(off), int(base)
        b.has_prev, b.has_next = int(prev is not None), int(nxt is not None)
        for i in range(3):
            b.prev[i] = int(prev[i]) if prev is not None and i < len(prev) else 0
            b.next[i] = int(nxt[i]) if nxt is not None and i < len(nxt) else 0
        return b

    # -- round 1 -------------------------------------------------------------------------
    def char_hist(self, text):
        h = torch.zeros(256, dtype=torch.int64, device=self.device)
        self._chk(self.lib.psacx_op_char_hist(self.ctx, self._p(text), int(text.numel()), self._p(h)))
        return h

    def make_keys(self, text, halo, m, two_k, codes, l, c1, c2, front=0):
        """front: the arrays get `front` extra entries before the keys (filled in by the caller), so that
        records can be put in front of this rank's own without copying the block."""
        pad = torch.zeros(max(0, two_k - int(halo.numel())), dtype=torch.uint8, device=self.device)
        buf = torch.cat([text, halo[:two_k], pad])            # the block followed by its halo (zeros past the end)
        k1, k2 = self.empty_idx(front + m), self.empty_idx(front + m)
        tab = (C.c_uint16 * 256)(*codes)
        self._chk(self._f("make_keys")(self.ctx, self._p(buf), int(m), int(buf.numel()), tab, l, c1, c2,
                                       self._p(k1[front:]), self._p(k2[front:])))
        return k1, k2

    def iota(self, m, start, front=0):
        t = self.empty_idx(front + m)
        self._chk(self._f("iota")(self.ctx, self._p(t[front:]), int(m), int(start)))
        return t

    # -- sorting -------------------------------------------------------------------------
    def pair_sort(self, K1, K2, V, bits1, bits2, destroy=False):
        """Stable sort by (K1, K2).  The inputs survive unless destroy=True (several radix passes then
        ping-pong through them instead of through copies)."""
        planned = (bits1 + 7) // 8 + (bits2 + 7) // 8
        if not destroy and planned > 1:
            K1, K2, V = K1.clone(), K2.clone(), V.clone()
        a, b, v = self.empty_like(K1), self.empty_like(K2), self.empty_like(V)
        where = C.c_int32(0)
        self._chk(self._f("pair_sort")(self.ctx, self._p(K1), self._p(K2), self._p(V), self._p(a), self._p(b), self._p(v),
                                       int(K1.numel()), int(bits1), int(bits2), C.byref(where)))
        return (a, b, v) if where.value else (K1, K2, V)

    def split_by(self, K1, K2, V, splitters, my_rank):
        """Groups the records by destination rank (see psacx_op_split_by); returns the three grouped arrays
        and the list of group boundaries (len(splitters) + 2 entries)."""
        ns = len(splitters)
        cols = [(C.c_uint64 * max(ns, 1))(*[int(s[i]) for s in splitters]) for i in range(4)]
        o1, o2, ov = self.empty_like(K1), self.empty_like(K2), self.empty_like(V)
        cs = (C.c_uint64 * (ns + 2))()
        self._chk(self._f("split_by")(self.ctx, self._p(K1), self._p(K2), self._p(V), int(K1.numel()), cols[0], cols[1], cols[2],
                                      cols[3], ns, int(my_rank), self._p(o1), self._p(o2), self._p(ov), cs))
        return o1, o2, ov, list(cs)

    def _bounds(self, S1, S2, q1, q2, use_second):
        nq = len(q1)
        if nq == 0:
            return [], []
        A = (C.c_uint64 * nq)(*q1); B = (C.c_uint64 * nq)(*q2)
        lb = (C.c_uint64 * nq)(); ub = (C.c_uint64 * nq)()
        self._chk(self._f("pair_bounds")(self.ctx, self._p(S1), self._p(S2), int(S1.numel()), A, B, nq, int(use_second), lb, ub))
        return list(lb), list(ub)

    def pair_bounds(self, S1, S2, q1, q2):
        return self._bounds(S1, S2, q1, q2, 1)

    def key_bounds(self, S1, qs):
        return self._bounds(S1, S1, qs, [0] * len(qs), 0)[0]

    # -- global indexing -----------------------------------------------------------------
    def owners(self, gidx, n, P):
        out = self.empty_like(gidx)
        self._chk(self._f("owners")(self.ctx, self._p(gidx), int(gidx.numel()), int(n), int(P), self._p(out)))
        return out

    def take(self, block, gidx, off, n):
        out = self.empty_like(gidx)
        self._chk(self._f("take")(self.ctx, self._p(block), self._p(gidx), int(gidx.numel()), int(off), int(n), self._p(out)))
        return out

    def put(self, block, gidx, off, vals, delta):
        self._chk(self._f("put")(self.ctx, self._p(block), self._p(gidx), int(gidx.numel()), int(off), self._p(vals), int(delta)))

    def put_perm(self, block, gidx, off, vals):
        """block[gidx - off] = vals - 1 where gidx is a permutation of the block's positions."""
        s = [self.empty_like(gidx) for _ in range(4)]
        self._chk(self._f("put_perm")(self.ctx, self._p(block), self._p(gidx), int(gidx.numel()), int(off), self._p(vals),
                                      *[self._p(x) for x in s]))

    def add_scalar(self, t, s, cap):
        """min(t + s, cap), formed in 64 bits."""
        out = self.empty_like(t)
        self._chk(self._f("add_scalar")(self.ctx, self._p(t), int(t.numel()), int(s), int(cap), self._p(out)))
        return out

    def finish_b2(self, ans, q, n):
        out = self.empty_like(q)
        self._chk(self._f("finish_b2")(self.ctx, self._p(ans), self._p(q), int(q.numel()), int(n), self._p(out)))
        return out

    # -- re-bucketing --------------------------------------------------------------------
    def last_head_first(self, S1, S2, SA, prev, off, n, shape):
        out = C.c_uint64(0)
        bd = self._bd(prev, None, off, 0)
        self._chk(self._f("last_head")(self.ctx, 0, self._p(S1), self._p(S2), self._p(SA), int(S1.numel()), int(n),
                                       shape[0], shape[1], shape[2], C.cast(C.pointer(bd), C.c_void_p), C.byref(out)))
        return out.value

    def last_head_refine(self, T1, T2, pos, prev):
        out = C.c_uint64(0)
        bd = self._bd(prev, None, 0, 0)
        self._chk(self._f("last_head")(self.ctx, 1, self._p(T1), self._p(T2), self._p(pos), int(T1.numel()), 0, 1, 1, 0,
                                       C.cast(C.pointer(bd), C.c_void_p), C.byref(out)))
        return out.value

    def rebucket_first(self, S1, S2, SA, prev, nxt, off, n, shape, base, want_lcp):
        m = int(S1.numel())
        bsa = self.empty_idx(m)
        lcp = self.empty_idx(m) if want_lcp else None
        na, nu = C.c_uint64(0), C.c_uint64(0)
        bd = self._bd(prev, nxt, off, base)
        self._chk(self._f("rebucket_first")(self.ctx, self._p(S1), self._p(S2), self._p(SA), m, int(n), shape[0], shape[1],
                                            shape[2], C.cast(C.pointer(bd), C.c_void_p), self._p(bsa), self._p(lcp) if want_lcp else None,
                                            C.byref(na), C.byref(nu)))
        return bsa, lcp, na.value, nu.value

    def rebucket_refine(self, T1, T2, TV, pos, prev, nxt, base, h, n, SA, Bsa, off, want_lcp, LCP):
        cnt = int(T1.numel())
        ids = self.empty_idx(cnt)
        qa, ql, qh = self.empty_idx(cnt), self.empty_idx(cnt), self.empty_idx(cnt)
        nq, na, nu = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        bd = self._bd(prev, nxt, off, base)
        self._chk(self._f("rebucket_refine")(self.ctx, self._p(T1), self._p(T2), self._p(TV), self._p(pos), cnt, int(n), int(h),
                                             C.cast(C.pointer(bd), C.c_void_p), self._p(SA), self._p(Bsa), self._p(LCP) if want_lcp else None,
                                             self._p(ids), self._p(qa), self._p(ql), self._p(qh), C.byref(nq), C.byref(na),
                                             C.byref(nu)))
        k = nq.value
        return dict(ids=ids, nact=na.value, nunf=nu.value, q_at=qa[:k], q_lo=ql[:k], q_hi=qh[:k])

    def compact(self, ids, pos, off, pid, nid):
        cnt = int(ids.numel())
        out = self.empty_idx(cnt)
        k = C.c_uint64(0)
        self._chk(self._f("compact")(self.ctx, self._p(ids), self._p(pos) if pos is not None else None, cnt, int(off),
                                     int(pid or 0), int(nid or 0), self._p(out), C.byref(k)))
        return out[:k.value].clone()

    # -- range minima --------------------------------------------------------------------
    def block_min(self, LCP):
        out = C.c_uint64(0)
        self._chk(self._f("block_min")(self.ctx, self._p(LCP), int(LCP.numel()), C.byref(out)))
        return out.value

    def range_min(self, LCP, lo, hi, off):
        out = self.empty_like(lo)
        self._chk(self._f("range_min")(self.ctx, self._p(LCP), int(LCP.numel()), self._p(lo), self._p(hi), int(lo.numel()),
                                       int(off), self._p(out)))
        return out

    def rmq_split(self, lo, hi, offs, sizes):
        n, P = sum(sizes), len(sizes)
        outs = [self.empty_like(lo) for _ in range(8)]
        self._chk(self._f("rmq_split")(self.ctx, self._p(lo), self._p(hi), int(lo.numel()), int(n), int(P),
                                       *[self._p(o) for o in outs]))
        return tuple(outs)

    def rmq_combine(self, a1, a2, ra, rb, mins):
        out = self.empty_like(a1)
        M = (C.c_uint64 * len(mins))(*[int(x) for x in mins])
        self._chk(self._f("rmq_combine")(self.ctx, self._p(a1), self._p(a2), self._p(ra), self._p(rb), int(a1.numel()), M,
                                         len(mins), self._p(out)))
        return out

    def nsv_from(self, block, start, thr, strict, left, off):
        """Per query the nearest element of `block` strictly beyond global position start[j] (int64
        tensor; left: towards lower positions) with value < thr[j] (strict) or <= thr[j].
        Returns (idx, val): global position or all ones, and the value found."""
        idx, val = self.empty_like(thr), self.empty_like(thr)
        self._chk(self._f("nsv_from")(self.ctx, self._p(block), int(block.numel()), int(off), self._p(start), self._p(thr),
                                      int(thr.numel()), int(bool(strict)), int(bool(left)), self._p(idx), self._p(val)))
        return idx, val

    def lcp_apply(self, LCP, at, off, mins, h):
        self._chk(self._f("lcp_apply")(self.ctx, self._p(LCP), self._p(at), int(at.numel()), int(off), self._p(mins), int(h)))

    def profile(self, on):
        self._chk(self.lib.psacx_profile(self.ctx, int(on)))

    def stats(self):
        st = _lib.Stats()
        self._chk(self.lib.psacx_get_stats(self.ctx, C.byref(st)))
        return st

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.psacx_destroy(self.ctx)
            self.ctx = None
