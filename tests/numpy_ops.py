"""CPU implementation of the LocalOps interface of tests/dist_harness/dist.py (numpy on CPU torch
tensors).  Test infrastructure: lets the distributed choreography run on the CPU (gloo /
loopback) where the HIP ops cannot.  Each op states the semantics the HIP op must match."""
import numpy as np
import torch


class NumpyOps(object):
    def __init__(self, index_bits=64):
        self.index_bits = index_bits
        self.udt = np.uint32 if index_bits == 32 else np.uint64
        self.tdt = torch.int32 if index_bits == 32 else torch.int64
        self.device = torch.device("cpu")
        self.INF = int(np.iinfo(self.udt).max)

    # -- helpers -------------------------------------------------------------------------
    def u(self, t):                       # unsigned numpy view of an index tensor
        return t.numpy().view(self.udt)

    def t(self, a):                       # index tensor from an unsigned numpy array
        a = np.ascontiguousarray(a, dtype=self.udt)
        return torch.from_numpy(a.view(np.int32 if self.index_bits == 32 else np.int64).copy())

    def empty_idx(self, m):
        return torch.zeros(m, dtype=self.tdt)

    def empty_like(self, t):
        return torch.zeros_like(t)

    def iota(self, m, start, front=0):
        return self.t(np.concatenate([np.zeros(front, np.uint64), np.arange(start, start + m, dtype=np.uint64)]))

    def value_at(self, t, j):
        return int(self.u(t)[j])

    def record_at(self, a, b, c, j):
        return (int(self.u(a)[j]), int(self.u(b)[j]), int(self.u(c)[j]))

    def add_scalar(self, t, s, cap):
        return self.t(np.minimum(self.u(t).astype(np.uint64) + np.uint64(s), np.uint64(cap)))

    # -- round 1 -------------------------------------------------------------------------
    def char_hist(self, text):
        return torch.from_numpy(np.bincount(text.numpy(), minlength=256).astype(np.int64))

    def make_keys(self, text, halo, m, two_k, codes, l, c1, c2, front=0):
        buf = np.zeros(m + two_k, np.uint64)
        lut = np.array(codes, np.uint64)
        buf[:m] = lut[text.numpy()]
        hl = min(int(halo.numel()), two_k)
        buf[m:m + hl] = lut[halo.numpy()[:hl]]
        k1 = np.zeros(m, np.uint64); k2 = np.zeros(m, np.uint64)
        for t in range(c1):
            k1 = (k1 << np.uint64(l)) | buf[t:t + m]
        for t in range(c2):
            k2 = (k2 << np.uint64(l)) | buf[c1 + t:c1 + t + m]
        z = np.zeros(front, np.uint64)
        return self.t(np.concatenate([z, k1])), self.t(np.concatenate([z, k2]))

    # -- sorting -------------------------------------------------------------------------
    def put_perm(self, block, gidx, off, vals):
        self.put(block, gidx, off, vals, -1)

    def pair_sort(self, K1, K2, V, bits1, bits2, destroy=False):
        a, b = self.u(K1).astype(np.uint64), self.u(K2).astype(np.uint64)
        m1 = np.uint64((1 << bits1) - 1) if bits1 < 64 else np.uint64(2**64 - 1)
        m2 = np.uint64((1 << bits2) - 1) if bits2 < 64 else np.uint64(2**64 - 1)
        order = np.lexsort((b & m2, a & m1))            # stable
        return self.t(self.u(K1)[order]), self.t(self.u(K2)[order]), self.t(self.u(V)[order])

    def split_by(self, K1, K2, V, splitters, my_rank):
        import bisect
        a, b = self.u(K1).tolist(), self.u(K2).tolist()
        sp = [tuple(int(x) for x in s) for s in splitters]
        cls = np.array([bisect.bisect_right(sp, (a[i], b[i], my_rank, i)) for i in range(len(a))], np.int64)
        order = np.argsort(cls, kind="stable")
        starts = [int(np.searchsorted(cls[order], d, side="left")) for d in range(len(sp) + 1)] + [len(a)]
        return self.t(self.u(K1)[order]), self.t(self.u(K2)[order]), self.t(self.u(V)[order]), starts

    def sample(self, S1, S2, positions):
        a, b = self.u(S1), self.u(S2)
        return [(int(a[p]), int(b[p])) for p in positions]

    def pair_bounds(self, S1, S2, q1, q2):
        a, b = self.u(S1).astype(object), self.u(S2).astype(object)
        pairs = list(zip(a.tolist(), b.tolist()))
        import bisect
        lb = [bisect.bisect_left(pairs, (x, y)) for x, y in zip(q1, q2)]
        ub = [bisect.bisect_right(pairs, (x, y)) for x, y in zip(q1, q2)]
        return lb, ub

    def key_bounds(self, S1, qs):
        a = self.u(S1)
        return [int(np.searchsorted(a, self.udt(q), side="left")) for q in qs]

    # -- global indexing -----------------------------------------------------------------
    def owners(self, gidx, n, P):
        g = np.minimum(self.u(gidx).astype(np.uint64), np.uint64(n - 1))
        div, mod = n // P, n % P
        big = np.uint64((div + 1) * mod)
        own = np.where(g < big, g // np.uint64(div + 1), np.uint64(mod) + (g - big) // np.uint64(max(div, 1)))
        return self.t(own)

    def take(self, block, gidx, off, n):
        g = np.minimum(self.u(gidx).astype(np.uint64), np.uint64(n - 1)) - np.uint64(off)
        return self.t(self.u(block)[g.astype(np.int64)])

    def put(self, block, gidx, off, vals, delta):
        g = (self.u(gidx).astype(np.uint64) - np.uint64(off)).astype(np.int64)
        v = (self.u(vals).astype(np.int64) + delta).astype(np.uint64)
        self.u(block)[g] = v.astype(self.udt)

    def finish_b2(self, ans, q, n):
        qq = self.u(q).astype(np.uint64)
        return self.t(np.where(qq < np.uint64(n), self.u(ans).astype(np.uint64) + np.uint64(1), np.uint64(0)))

    # -- re-bucketing --------------------------------------------------------------------
    def _window_lcp(self, x1, x2, y1, y2, shape):
        l, c1, c2 = shape
        W = self.index_bits

        def clz(v):
            v = int(v)
            return W - v.bit_length()
        if x1 != y1:
            return (clz(x1 ^ y1) - (W - c1 * l)) // l
        if x2 != y2:
            return c1 + (clz(x2 ^ y2) - (W - c2 * l)) // l
        return c1 + c2

    def _first_heads(self, S1, S2, SA, prev, off, n, shape):
        a, b, sa = self.u(S1), self.u(S2), self.u(SA)
        m = a.size
        two_k = shape[1] + shape[2]
        heads = np.zeros(m, bool); lcps = np.zeros(m, np.uint64)
        for e in range(m):
            if off + e == 0:
                heads[e] = True; lcps[e] = 0; continue
            p = prev if e == 0 else (a[e - 1], b[e - 1], sa[e - 1])
            c = self._window_lcp(int(p[0]), int(p[1]), int(a[e]), int(b[e]), shape)
            c = min(c, n - int(p[2]), n - int(sa[e]))
            heads[e] = c < two_k; lcps[e] = c
        return heads, lcps

    def last_head_first(self, S1, S2, SA, prev, off, n, shape):
        heads, _ = self._first_heads(S1, S2, SA, prev, off, n, shape)
        idx = np.nonzero(heads)[0]
        return int(off + idx[-1] + 1) if idx.size else 0

    def _fill(self, heads, ids_at_heads, base):
        out = np.zeros(heads.size, np.uint64)
        run = base
        for e in range(heads.size):
            if heads[e]:
                run = int(ids_at_heads[e])
            out[e] = run
        return out

    def _activity(self, heads, next_head):
        hn = np.append(heads[1:], next_head) if heads.size else heads
        act = (~heads) | (~hn)
        ub = heads & (~hn)
        return int(act.sum()), int(ub.sum())

    def rebucket_first(self, S1, S2, SA, prev, nxt, off, n, shape, base, want_lcp):
        heads, lcps = self._first_heads(S1, S2, SA, prev, off, n, shape)
        m = heads.size
        ids = self._fill(heads, off + np.arange(m, dtype=np.uint64) + 1, base)
        next_head = True
        if nxt is not None and m:
            a, b, sa = self.u(S1), self.u(S2), self.u(SA)
            c = self._window_lcp(int(a[-1]), int(b[-1]), nxt[0], nxt[1], shape)
            c = min(c, n - int(sa[-1]), n - nxt[2])
            next_head = c < shape[1] + shape[2]
        nact, nunf = self._activity(heads, next_head)
        LCP = None
        if want_lcp:
            LCP = self.t(np.where(heads, lcps, np.uint64(n)))
        return self.t(ids), LCP, nact, nunf

    def _refine_heads(self, T1, T2, prev):
        a, b = self.u(T1), self.u(T2)
        m = a.size
        heads = np.zeros(m, bool)
        for j in range(m):
            if j == 0:
                p = prev
                heads[j] = (p is None) or int(a[0]) != p[0] or int(b[0]) != p[1] or int(b[0]) == 0
            else:
                heads[j] = a[j] != a[j - 1] or b[j] != b[j - 1] or b[j] == 0
        return heads

    def last_head_refine(self, T1, T2, pos, prev):
        heads = self._refine_heads(T1, T2, prev)
        idx = np.nonzero(heads)[0]
        return int(self.u(pos)[idx[-1]]) + 1 if idx.size else 0

    def rebucket_refine(self, T1, T2, TV, pos, prev, nxt, base, h, n, SA, Bsa, off, want_lcp, LCP):
        a, b, v, ps = self.u(T1), self.u(T2), self.u(TV), self.u(pos)
        m = a.size
        heads = self._refine_heads(T1, T2, prev)
        ids = self._fill(heads, ps.astype(np.uint64) + 1, base)
        next_head = True
        if nxt is not None and m:
            next_head = int(a[-1]) != nxt[0] or int(b[-1]) != nxt[1] or nxt[1] == 0
        nact, nunf = self._activity(heads, next_head)
        loc = (ps.astype(np.uint64) - np.uint64(off)).astype(np.int64)
        self.u(SA)[loc] = v
        self.u(Bsa)[loc] = ids.astype(self.udt)
        q_at, q_lo, q_hi = [], [], []
        if want_lcp:
            L = self.u(LCP)
            for j in range(m):
                p = prev if j == 0 else (int(a[j - 1]), int(b[j - 1]))
                if p is None or not heads[j] or int(a[j]) != p[0]:
                    continue            # not a boundary inside an old bucket
                x, y = p[1], int(b[j])
                if x == 0 or y == 0:
                    if int(L[loc[j]]) == n:
                        L[loc[j]] = h
                else:
                    q_at.append(int(ps[j])); q_lo.append(min(x, y)); q_hi.append(max(x, y))
        return dict(ids=self.t(ids), nact=nact, nunf=nunf, q_at=self.t(np.array(q_at, np.uint64)),
                    q_lo=self.t(np.array(q_lo, np.uint64)), q_hi=self.t(np.array(q_hi, np.uint64)))

    def compact(self, ids, pos, off, pid, nid):
        x = self.u(ids).astype(np.uint64)
        m = x.size
        if m == 0:
            return self.t(np.zeros(0, np.uint64))
        prev = np.concatenate(([np.uint64(pid if pid is not None else 0)], x[:-1]))
        nxt = np.concatenate((x[1:], [np.uint64(nid if nid is not None else 0)]))
        act = (x == prev) | (x == nxt)
        p = self.u(pos).astype(np.uint64) if pos is not None else np.uint64(off) + np.arange(m, dtype=np.uint64)
        return self.t(p[act])

    # -- range minima --------------------------------------------------------------------
    def block_min(self, LCP):
        a = self.u(LCP)
        return int(a.min()) if a.size else self.INF

    def range_min(self, LCP, lo, hi, off):
        a = self.u(LCP)
        l = self.u(lo).astype(np.int64) - off
        r = self.u(hi).astype(np.int64) - off
        out = np.full(l.size, self.INF, np.uint64)
        for i in range(l.size):
            if r[i] > l[i]:
                out[i] = a[l[i]:r[i]].min()
        return self.t(out)

    def nsv_from(self, block, start, thr, strict, left, off):
        a = self.u(block).astype(np.uint64)
        st = start.numpy().astype(np.int64) - off
        th = self.u(thr).astype(np.uint64)
        m = a.size
        none = np.uint64(self.INF)
        idx = np.full(th.size, none, np.uint64); val = np.zeros(th.size, np.uint64)
        for j in range(th.size):
            rng = range(min(int(st[j]), m) - 1, -1, -1) if left else range(max(int(st[j]), -1) + 1, m)
            for e in rng:
                if (a[e] < th[j]) if strict else (a[e] <= th[j]):
                    idx[j] = off + e; val[j] = a[e]
                    break
        return self.t(idx), self.t(val)

    def rmq_split(self, lo, hi, offs, sizes):
        l, r = self.u(lo).astype(np.int64), self.u(hi).astype(np.int64)
        ends = np.array(offs) + np.array(sizes)
        pl = np.searchsorted(ends, l, side="right")
        pr = np.searchsorted(ends, r - 1, side="right")
        same = pl == pr
        hi1 = np.where(same, r, ends[pl])
        lo2 = np.where(same, r, np.array(offs)[pr])
        z = lambda x: self.t(np.asarray(x, np.uint64))
        return z(pl), z(l), z(hi1), z(pr), z(lo2), z(r), (pl + 1).tolist(), pr.tolist()

    def rmq_combine(self, a1, a2, ra, rb, mins):
        x = np.minimum(self.u(a1).astype(np.uint64), self.u(a2).astype(np.uint64))
        for i in range(x.size):
            if rb[i] > ra[i]:
                x[i] = min(int(x[i]), min(mins[ra[i]:rb[i]]))
        return self.t(x)

    def lcp_apply(self, LCP, at, off, mins, h):
        loc = (self.u(at).astype(np.uint64) - np.uint64(off)).astype(np.int64)
        self.u(LCP)[loc] = (self.u(mins).astype(np.uint64) + np.uint64(h)).astype(self.udt)
