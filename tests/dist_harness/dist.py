"""Block-distributed SA / ISA / LCP construction: one rank per GPU.

The text is block-partitioned like psac's (suffix_array.hpp:183-194, mxx::blk_dist):
rank r owns text / SA / ISA / LCP positions [off_r, off_r + m_r).  The loop is the
single-GPU one (psac_amd/csrc/construct.hpp) with every global step made explicit:

  psac step (reference)                         here
  -------------------------------------------   ------------------------------------------
  alphabet allreduce     alphabet.hpp:98        all_reduce of the 256-bin histogram
  k-mer halo             kmer.hpp:142           first 2k characters sent to the left rank
  mxx::sort of tuples    idxsort.hpp:60-62      dist_sort: local radix sort, sampled
                                                splitters, all-to-all, local sort, exact
                                                re-balance to the block sizes
  right_shift of the last tuple                 all_gather of first / last records
                         bucketing.hpp:77,100
  exscan(max) of bucket ids bucketing.hpp:39    all_gather of per-rank last head ids
  bulk_permute_inplace   bulk_permute.hpp:14    dist_put: partition by owner, all-to-all,
                                                local scatter
  sparse_get_b2/bulk_rma suffix_array.hpp:972   dist_take: queries to owners, answers back
  bulk_rmq_v2            par_rmq.hpp:199-332    edge sub-queries to owners + all-gathered
                                                per-rank minima for whole ranks in between

All arithmetic on the arrays happens in a LocalOps object (HIP kernels on the GPU:
tests/dist_harness/dist_ops.py); this module only slices, concatenates and exchanges tensors.
Written as a generator so that it runs unchanged under torch.distributed or under the
in-process LoopbackWorld (tests/dist_harness/comm.py).
"""
import torch

SAMPLES_PER_RANK = 256


def blk_sizes(n, P):
    """mxx::blk_dist: the first n mod P ranks hold one element more."""
    return [n // P + (1 if r < n % P else 0) for r in range(P)]


def prefix(xs):
    out, s = [], 0
    for x in xs:
        out.append(s)
        s += x
    return out


def build_alphabet(hist):
    """alphabet.hpp:147-164: codes 1..sigma in byte order, l = ceil(log2(sigma + 1))."""
    codes, nxt = [0] * 256, 1
    for ch in range(256):
        if hist[ch]:
            codes[ch] = nxt
            nxt += 1
    sigma = nxt - 1
    l = 0
    while (1 << l) < sigma + 1:
        l += 1
    return codes, sigma, l


def choose_k(word_bits, l, min_local, P, k):
    """kmer.hpp:26-40."""
    max_k = word_bits // l
    if k == 0 or k > max_k:
        k = max_k
    if k >= min_local:
        k = min_local
        if P == 1 and k > 1:
            k -= 1
    return k


def bits_for(v):
    return max(1, int(v).bit_length())


# ---------------------------------------------------------------------------------------
# distributed primitives
# ---------------------------------------------------------------------------------------
def dist_sort(comm, ops, records, targets, bits1, bits2):
    """records: list [K1, K2, V]; it is emptied, so the inputs can be released as soon as the partition pass
    has copied them (they are three of the thirteen words per record this path holds at its peak).
    Globally sorts records by (K1, K2); rank r ends with exactly targets[r] records, the
    concatenation over ranks being sorted (the contract psac needs from mxx::sort,
    idxsort.hpp:67-79).  Sample sort: sampled splitters -> one partition pass by destination ->
    all-to-all -> one local radix sort -> exact re-balance to the target sizes."""
    P, r = comm.size, comm.rank
    K1, K2, V = records
    del records[:]
    if P == 1:
        return ops.pair_sort(K1, K2, V, bits1, bits2, destroy=True)
    c = int(K1.numel())
    # regular samples of the local records, made unique by (rank, index) so that ties are divided
    pos = sorted(set((c * (2 * i + 1)) // (2 * SAMPLES_PER_RANK) for i in range(SAMPLES_PER_RANK))) if c else []
    keys = ops.sample(K1, K2, pos)
    mine = [(k1, k2, r, p) for (k1, k2), p in zip(keys, pos)]
    allsamp = yield from comm.all_gather_obj(mine)
    flat = sorted(s for lst in allsamp for s in lst)
    splitters = [flat[min(len(flat) - 1, (len(flat) * d) // P)] for d in range(1, P)] if flat else []
    splitters = sorted(set(splitters))
    G1, G2, GV, bounds = ops.split_by(K1, K2, V, splitters, r)
    del K1, K2, V
    bounds = list(bounds[:len(splitters) + 1]) + [c] * (P + 1 - (len(splitters) + 1))   # empty trailing groups
    parts, _ = yield from comm.exchange([G1, G2, GV], bounds)
    del G1, G2, GV
    R1, R2, RV = ops.pair_sort(parts[0], parts[1], parts[2], bits1, bits2, destroy=True)
    # exact re-balance: global index of my j-th record is G[r] + j
    c2 = int(R1.numel())
    counts = yield from comm.all_gather_obj(c2)
    G, TP = prefix(counts), prefix(targets)
    if counts == list(targets):
        return R1, R2, RV                       # already balanced: nothing moves
    # my records [cut[d], cut[d+1]) belong to rank d (contiguous ranges, ascending in d)
    cut = [min(max(TP[d] - G[r], 0), c2) for d in range(P)] + [c2]
    out, _ = yield from comm.exchange([R1, R2, RV], cut)
    return out[0], out[1], out[2]


def _route(comm, ops, gidx, payload, n):
    """Stable partition of `gidx` (global positions) and one payload array by owner rank: one
    radix pass over the owner word carries the two other words along."""
    P = comm.size
    own = ops.owners(gidx, n, P)
    o, g, v = ops.pair_sort(own, gidx, payload, bits_for(P - 1), 0)
    bounds = ops.key_bounds(o, list(range(P))) + [int(o.numel())]
    return g, v, bounds


def dist_put(comm, ops, block, off, gidx, vals, delta, n, permutation=False):
    """block[gidx - off_owner] = vals + delta on the owner of each global position
    (bulk_permute_inplace, bulk_permute.hpp:14-73).  permutation=True: every position of every
    block is written exactly once (the SA -> ISA inversion of the first round, delta = -1)."""
    P = comm.size
    if P == 1:
        gi, vi = gidx, vals
    else:
        g, v, bounds = _route(comm, ops, gidx, vals, n)
        (gi, vi), _ = yield from comm.exchange([g, v], bounds)
    if permutation and delta == -1:
        ops.put_perm(block, gi, off, vi)
    else:
        ops.put(block, gi, off, vi, delta)


def dist_take(comm, ops, block, off, gidx, n):
    """Returns block_owner[gidx - off_owner] for every global position in gidx, in the order of
    gidx (bulk_rma, bulk_rma.hpp:13-135).  Positions >= n are clamped (callers mask them)."""
    P = comm.size
    if P == 1:
        return ops.take(block, gidx, off, n)
    idx = ops.iota(int(gidx.numel()), 0)
    g, back, bounds = _route(comm, ops, gidx, idx, n)
    (q,), lens = yield from comm.exchange([g], bounds)
    ans = ops.take(block, q, off, n)
    (got,), _ = yield from comm.exchange([ans], prefix(lens) + [sum(lens)])
    res = ops.empty_like(gidx)
    ops.put(res, back, 0, got, 0)          # undo the routing permutation
    return res


def _neighbours(comm, first, last, has):
    """Last record of the nearest non-empty rank below and first record of the nearest one above."""
    allrec = yield from comm.all_gather_obj((has, first, last))
    r = comm.rank
    prev = nxt = None
    for s in range(r - 1, -1, -1):
        if allrec[s][0]:
            prev = allrec[s][2]
            break
    for s in range(r + 1, comm.size):
        if allrec[s][0]:
            nxt = allrec[s][1]
            break
    return prev, nxt


def dist_range_min(comm, ops, lcp_block, off, sizes, lo, hi, n):
    """min(LCP[lo .. hi)) over the block-distributed LCP array for every query
    (bulk_rmq_v2, par_rmq.hpp:199-332): the part inside the first and the last rank is
    answered by their owners, whole ranks in between from the all-gathered block minima."""
    P = comm.size
    if P == 1:
        return ops.range_min(lcp_block, lo, hi, off)
    mins = yield from comm.all_gather_obj(ops.block_min(lcp_block))
    offs = prefix(sizes)
    own1, lo1, hi1, own2, lo2, hi2, ra, rb = ops.rmq_split(lo, hi, offs, sizes)
    answers = []
    for own, a, b in ((own1, lo1, hi1), (own2, lo2, hi2)):
        idx = ops.iota(int(a.numel()), 0)
        # route by owner; carry the upper end and the original slot along
        o, ga, gb = ops.pair_sort(own, a, b, bits_for(P - 1), 0)
        _, _, back = ops.pair_sort(own, a, idx, bits_for(P - 1), 0)
        bounds = ops.key_bounds(o, list(range(P))) + [int(o.numel())]
        (qa, qb), lens = yield from comm.exchange([ga, gb], bounds)
        res = ops.range_min(lcp_block, qa, qb, off)
        (got,), _ = yield from comm.exchange([res], prefix(lens) + [sum(lens)])
        ordered = ops.empty_like(a)
        ops.put(ordered, back, 0, got, 0)
        answers.append(ordered)
    return ops.rmq_combine(answers[0], answers[1], ra, rb, mins)


# ---------------------------------------------------------------------------------------
# all nearest smaller values over a block-distributed array
# ---------------------------------------------------------------------------------------
NEAREST_SM, NEAREST_EQ, FURTHEST_EQ = 0, 1, 2       # ansv_common.hpp:20-22


def _as_i64(ops, t):
    """Unsigned value of an index tensor as int64 (values stay below 2^62 on this path)."""
    x = t.to(torch.int64)
    return x & 0xFFFFFFFF if ops.index_bits == 32 else x


def dist_search(comm, ops, block, off, sizes, mins, start, thr, strict, left):
    """For every query the nearest element of the distributed array strictly beyond global position
    start[j] (int64 tensor; -1 and n are allowed) whose value is < thr[j] (strict) or <= thr[j];
    left: towards lower positions.  Returns (idx, val) index tensors, idx = all ones where no such
    element exists.  Queries are answered by the block holding `start`, then by the nearest further
    block whose minimum qualifies (its edge element search cannot fail)."""
    P, r = comm.size, comm.rank
    n = sum(sizes)
    offs = prefix(sizes)
    cnt = int(thr.numel())
    none = -1
    if P == 1:
        return ops.nsv_from(block, start, thr, strict, left, off)
    dev = thr.device
    # block that holds the start position (clamped: -1 -> first block, n -> last block)
    ends = torch.tensor([o + s for o, s in zip(offs, sizes)], dtype=torch.int64, device=dev)
    own = torch.searchsorted(ends, start.clamp(0, max(n - 1, 0)), right=True).clamp(max=P - 1)
    idx = ops.empty_like(thr); idx.fill_(none)
    val = ops.empty_like(thr); val.zero_()
    thr64 = _as_i64(ops, thr)

    def ask(target, st):
        """Sends query j to rank target[j] (>= 0) with start st[j]; returns (idx, val) in query order."""
        sel = torch.nonzero(target >= 0).reshape(-1)
        oi = ops.empty_like(thr); oi.fill_(none)
        ov = ops.empty_like(thr); ov.zero_()
        order = torch.argsort(target[sel], stable=True)
        sel = sel[order]
        tgt = target[sel]
        bounds = torch.searchsorted(tgt, torch.arange(P + 1, dtype=torch.int64, device=dev)).tolist()
        (qs, qt), lens = yield from comm.exchange([st[sel].contiguous(), thr[sel].contiguous()], bounds)
        ri, rv = ops.nsv_from(block, qs, qt, strict, left, off)
        (bi, bv), _ = yield from comm.exchange([ri, rv], prefix(lens) + [sum(lens)])
        oi[sel] = bi; ov[sel] = bv
        return oi, ov

    # phase A: the block of the start position
    i1, v1 = yield from ask(own.clone(), start)
    idx, val = i1, v1
    # phase B: the nearest block beyond it whose minimum qualifies
    open_q = idx == none
    target = torch.full((cnt,), -1, dtype=torch.int64, device=dev)
    order = range(P - 2, -1, -1) if left else range(1, P)
    for b in (order if left else order):
        ok = (thr64 > mins[b]) if strict else (thr64 >= mins[b])
        beyond = (own > b) if left else (own < b)
        take = open_q & (target < 0) & ok & beyond & (sizes[b] > 0)
        target[take] = b
    edge = torch.full((cnt,), n if left else -1, dtype=torch.int64, device=dev)   # beyond the target's far edge
    i2, v2 = yield from ask(target, edge)
    got = target >= 0
    idx[got] = i2[got]; val[got] = v2[got]
    return idx, val


def dist_ansv(comm, ops, block, left_type=NEAREST_SM, right_type=NEAREST_SM):
    """ansv<T, left_type, right_type, global_indexing> (ansv.hpp:2042-2051) over a block-distributed
    array: for every local element the global index of its nearest smaller value on each side
    (type 0: strictly smaller; 1: smaller or equal; 2: the furthest element of the run of equal
    values that nothing smaller interrupts, ansv_common.hpp:20-22); all ones where none exists.
    Returns (left, right) index tensors."""
    P, r = comm.size, comm.rank
    m = int(block.numel())
    sizes = yield from comm.all_gather_obj(m)
    offs = prefix(sizes)
    off = offs[r]
    n = sum(sizes)
    mins = yield from comm.all_gather_obj(ops.block_min(block) if m else (1 << 64) - 1)
    none = -1
    here = torch.arange(off, off + m, dtype=torch.int64, device=block.device)
    out = []
    for left, typ in ((True, left_type), (False, right_type)):
        if typ == NEAREST_SM:
            idx, _ = yield from dist_search(comm, ops, block, off, sizes, mins, here, block, True, left)
        else:
            j, u = yield from dist_search(comm, ops, block, off, sizes, mins, here, block, False, left)
            idx = j
            if typ == FURTHEST_EQ:
                # first strictly smaller value beyond j, then back towards i: the first value <= in[j]
                jj = _as_i64(ops, j)
                found = j != none
                start2 = torch.where(found, jj, torch.full_like(jj, n if left else -1))
                s, _ = yield from dist_search(comm, ops, block, off, sizes, mins, start2, u, True, left)
                ss = _as_i64(ops, s)
                start3 = torch.where(s != none, ss, torch.full_like(ss, -1 if left else n))
                f, _ = yield from dist_search(comm, ops, block, off, sizes, mins, start3, u, False, not left)
                idx = torch.where(found, f, j)
        out.append(idx)
    return out[0], out[1]


# ---------------------------------------------------------------------------------------
# the construction
# ---------------------------------------------------------------------------------------
def construct(comm, ops, text_block, want_lcp=True, k_req=0, log=None):
    """Generator.  text_block: uint8 tensor holding this rank's block of the text.
    Returns dict(SA, ISA, LCP, k, l, sigma, rounds) with this rank's blocks."""
    P, r = comm.size, comm.rank
    m = int(text_block.numel())
    sizes = yield from comm.all_gather_obj(m)
    n = sum(sizes)
    if sizes != blk_sizes(n, P):      # suffix_array.hpp:226-227
        raise RuntimeError("The input string must be equally block decomposed accross all MPI processes.")
    offs = prefix(sizes)
    off = offs[r]
    word_bits = ops.index_bits

    hist = yield from comm.all_reduce_sum(ops.char_hist(text_block))
    codes, sigma, l = build_alphabet([int(x) for x in hist.tolist()])
    k = choose_k(word_bits, l, min(sizes), P, k_req)
    two_k = 2 * k
    if P > 1 and min(sizes) < two_k:
        raise RuntimeError("text blocks shorter than 2k characters are not supported with more than one rank")
    # The 2k-character window is packed without an end-marker code (codes 0..sigma-1, lc bits each;
    # see key_pairs_kernel): 40 instead of 60 key bits for DNA and 32-bit words.  The suffixes
    # shorter than 2k -- the last 2k - 1 positions of the text -- then tie with longer ones on the
    # zero padding; they are moved to the very front of the record order (rank 0, shortest first),
    # where the (rank, index) tie-break of the stable distributed sort leaves them first.
    lc = max(1, (sigma - 1).bit_length())
    pcodes = [c - 1 if c else 0 for c in codes]
    c1 = min(two_k, word_bits // lc)
    c2 = two_k - c1

    # halo: the first 2k characters of the right neighbour (zeros past the end of the text)
    if P > 1:
        send = [text_block[:0]] * P
        if r > 0:
            send[r - 1] = text_block[:two_k]
        got = yield from comm.all_to_all_v(send)
        halo = got[r + 1] if r + 1 < P else text_block[:0]
    else:
        halo = text_block[:0]
    spec = min(two_k - 1, n)
    front = spec if r == 0 else 0                        # room for the moved records in front of rank 0's own
    K1, K2 = ops.make_keys(text_block, halo, m, two_k, pcodes, lc, c1, c2, front)
    V = ops.iota(m, off, front)
    mine = min(m, max(0, off + m - (n - spec)))          # short suffixes in this block (its tail)
    tails = [torch.flip(a[front + m - mine:], [0]) for a in (K1, K2, V)]
    if P > 1:
        moved, rc = yield from comm.exchange(tails, [0] + [mine] * P)      # everything to rank 0
        if r == 0:
            cuts = prefix(rc) + [sum(rc)]
            moved = [torch.cat([a[cuts[s]:cuts[s + 1]] for s in range(P - 1, -1, -1)]) for a in moved]
    else:
        moved = tails
    if r == 0:
        for a, mv in zip((K1, K2, V), moved):
            a[:front] = mv                               # (the exchange hands rank 0 exactly `spec` records)
    K1, K2, V = [a[:front + m - mine] for a in (K1, K2, V)]
    recs = [K1, K2, V]
    del K1, K2, V
    S1, S2, SA = yield from dist_sort(comm, ops, recs, sizes, c1 * lc, c2 * lc)

    rounds = []
    shape = (lc, c1, c2)

    def neighbours(a1, a2, a3):
        has = int(a1.numel()) > 0
        first = ops.record_at(a1, a2, a3, 0) if has else None
        last = ops.record_at(a1, a2, a3, int(a1.numel()) - 1) if has else None
        return (yield from _neighbours(comm, first, last, has))

    prev, nxt = yield from neighbours(S1, S2, SA)
    lh = ops.last_head_first(S1, S2, SA, prev, off, n, shape)
    heads = yield from comm.all_gather_obj(lh)
    base = max([0] + heads[:r])
    Bsa, LCP, nact, nunf = ops.rebucket_first(S1, S2, SA, prev, nxt, off, n, shape, base, want_lcp)
    del S1, S2
    ISA = ops.empty_idx(m)
    yield from dist_put(comm, ops, ISA, off, SA, Bsa, -1, n, permutation=True)

    def boundary_ids(ids):
        has = int(ids.numel()) > 0
        first = ops.value_at(ids, 0) if has else None
        last = ops.value_at(ids, int(ids.numel()) - 1) if has else None
        return (yield from _neighbours(comm, first, last, has))

    pid, nid = yield from boundary_ids(Bsa)
    pos = ops.compact(Bsa, None, off, pid, nid)
    tot = yield from comm.all_gather_obj((nact, nunf))
    unf_e, unf_b = sum(t[0] for t in tot), sum(t[1] for t in tot)
    rounds.append((k, unf_b, unf_e))
    if log is not None and r == 0:
        log.write("iteration %d: unfinished buckets = %d, unfinished elements = %d\n" % rounds[-1])

    id_bits = bits_for(n)
    h = two_k
    while unf_b > 0 and h < n:
        cnt = int(pos.numel())
        counts = yield from comm.all_gather_obj(cnt)
        # B2 = rank of the suffix h further (sparse_get_b2, suffix_array.hpp:972-996)
        sa_act = ops.take(SA, pos, off, n)
        q = ops.add_scalar(sa_act, h, n)        # saturates at n: SA + h must not wrap a 32-bit index
        ans = yield from dist_take(comm, ops, ISA, off, q, n)
        K2 = ops.finish_b2(ans, q, n)
        K1 = ops.take(Bsa, pos, off, n)
        recs = [K1, K2, sa_act]
        del K1, K2, sa_act
        T1, T2, TV = yield from dist_sort(comm, ops, recs, counts, id_bits, id_bits)
        prev, nxt = yield from neighbours(T1, T2, TV)
        lh = ops.last_head_refine(T1, T2, pos, prev)
        heads = yield from comm.all_gather_obj(lh)
        base = max([0] + heads[:r])
        out = ops.rebucket_refine(T1, T2, TV, pos, prev, nxt, base, h, n, SA, Bsa, off, want_lcp, LCP)
        yield from dist_put(comm, ops, ISA, off, TV, out["ids"], -1, n)
        if want_lcp:
            mins = yield from dist_range_min(comm, ops, LCP, off, sizes, out["q_lo"], out["q_hi"], n)
            ops.lcp_apply(LCP, out["q_at"], off, mins, h)
        pid, nid = yield from boundary_ids(out["ids"])
        pos = ops.compact(out["ids"], pos, off, pid, nid)
        tot = yield from comm.all_gather_obj((out["nact"], out["nunf"]))
        unf_e, unf_b = sum(t[0] for t in tot), sum(t[1] for t in tot)
        rounds.append((h, unf_b, unf_e))
        if log is not None and r == 0:
            log.write("iteration %d: unfinished buckets = %d, unfinished elements = %d\n" % rounds[-1])
        h *= 2
    return dict(SA=SA, ISA=ISA, LCP=LCP if want_lcp else None, k=k, l=l, sigma=sigma, rounds=rounds, n=n, off=off)


def run(gen):
    """Drives a construct() generator to completion under a back-end whose collectives complete
    immediately (TorchComm)."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value
