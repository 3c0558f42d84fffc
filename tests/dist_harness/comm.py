"""Communication layer of the distributed construction (one process per GPU).

The distributed algorithm in dist.py is written as a generator per rank: every
collective is `result = yield from comm.<collective>(...)`.  Two back-ends:

* TorchComm     -- torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
                   "gloo" in the CPU tests).  Collectives run immediately.
* LoopbackWorld -- P virtual ranks inside ONE process, stepped in lockstep.  Used by
                   the tests (CPU reference ops, or the HIP ops on a single GPU) to
                   exercise the exact choreography without P devices.

Replaces the mxx collectives psac uses on this path:
all2allv (bulk_permute.hpp:60-61, par_rmq.hpp:273-293, bulk_rma.hpp:20-49),
allgather / allreduce / exscan of scalars (bucketing.hpp:39,70,117), left/right
shift of boundary records (bucketing.hpp:77,100; kmer.hpp:142).
"""
import pickle

import numpy as np
import torch


class TorchComm(object):
    """group: the process group that moves tensors (RCCL on the GPU box).  obj_group: optional
    second group for the small Python objects (counts, boundary records); a gloo group keeps
    them off the GPU: all_gather_object over RCCL pickles to a device tensor and runs two
    collectives plus a host synchronisation per call, and a round issues a dozen of them."""

    OBJ_CAP = 1 << 16          # bytes reserved per rank for one pickled object

    def __init__(self, group=None, obj_group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.obj_group = obj_group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.on_gpu = dist.get_backend(group) == "nccl"

    def _gather_objects(self, obj):
        if self.obj_group is not None or not self.on_gpu:
            out = [None] * self.size
            self.dist.all_gather_object(out, obj, group=self.obj_group if self.obj_group is not None else self.group)
            return out
        # RCCL: one fixed-size all-gather of the pickled bytes and one copy back to the host
        # (all_gather_object would run a size and a payload collective with a sync after each)
        data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        if len(data) + 8 > self.OBJ_CAP:
            raise ValueError("object of %d bytes is too large for the small-object channel" % len(data))
        host = np.zeros(self.OBJ_CAP, np.uint8)
        host[:8] = np.frombuffer(np.uint64(len(data)).tobytes(), np.uint8)
        host[8:8 + len(data)] = np.frombuffer(data, np.uint8)
        mine = torch.from_numpy(host).cuda()
        gathered = torch.empty(self.size * self.OBJ_CAP, dtype=torch.uint8, device=mine.device)
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        raw = gathered.cpu().numpy()
        out = []
        for s in range(self.size):
            blk = raw[s * self.OBJ_CAP:(s + 1) * self.OBJ_CAP]
            ln = int(np.frombuffer(blk[:8].tobytes(), np.uint64)[0])
            out.append(pickle.loads(blk[8:8 + ln].tobytes()))
        return out

    # every method is a generator so that call sites are identical for both back-ends
    def all_gather_obj(self, obj):
        """Small Python objects (ints, tuples): list indexed by rank."""
        return self._gather_objects(obj)
        yield  # pragma: no cover  (makes this a generator)

    def exchange(self, arrays, bounds):
        """All-to-all of several 1-D arrays that share one partition: records bounds[d] .. bounds[d+1]
        of every array go to rank d.  One exchange of the counts serves all arrays, the arrays are
        sent in place (no staging copy) and each result comes back as one contiguous tensor (records
        ordered by source rank).  Returns (list of received arrays, counts received per source)."""
        P = self.size
        counts = [int(bounds[d + 1]) - int(bounds[d]) for d in range(P)]
        allc = self._gather_objects(counts)
        recv_counts = [allc[s][self.rank] for s in range(P)]
        lo, hi = int(bounds[0]), int(bounds[P])
        out = []
        for a in arrays:
            send = a[lo:hi]
            if not send.is_contiguous():
                send = send.contiguous()
            recv = a.new_empty(sum(recv_counts))
            self.dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=counts,
                                        group=self.group)
            out.append(recv)
        return out, recv_counts
        yield  # pragma: no cover

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t
        yield  # pragma: no cover

    def all_to_all_v(self, chunks):
        """chunks[d] = 1-D tensor for rank d (any lengths).  Returns the list of tensors
        received from every rank (index = source)."""
        counts = [int(c.numel()) for c in chunks]
        rcounts = self._gather_objects(counts)
        recv_counts = [rcounts[s][self.rank] for s in range(self.size)]
        ref = chunks[0]
        send = torch.cat([c.reshape(-1) for c in chunks]) if sum(counts) else ref.new_empty(0)
        recv = ref.new_empty(sum(recv_counts))
        self.dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=counts,
                                    group=self.group)
        return list(torch.split(recv, recv_counts))
        yield  # pragma: no cover


class _Request(object):
    __slots__ = ("kind", "payload")

    def __init__(self, kind, payload):
        self.kind = kind
        self.payload = payload


class LoopbackComm(object):
    """Rank handle of a LoopbackWorld: collectives yield a request to the scheduler."""

    def __init__(self, rank, size):
        self.rank = rank
        self.size = size

    def all_gather_obj(self, obj):
        res = yield _Request("gather", obj)
        return res

    def all_reduce_sum(self, t):
        res = yield _Request("reduce", t)
        return res

    def all_to_all_v(self, chunks):
        res = yield _Request("a2a", chunks)
        return res

    def exchange(self, arrays, bounds):
        import torch
        out, recv_counts = [], None
        for a in arrays:
            got = yield _Request("a2a", [a[int(bounds[d]):int(bounds[d + 1])] for d in range(self.size)])
            recv_counts = [int(t.numel()) for t in got]
            out.append(torch.cat(got) if got else a[:0])
        return out, recv_counts


class LoopbackWorld(object):
    """Runs `fn(comm, *args_of_rank)` for P virtual ranks in lockstep.

    fn must be a generator function following the `yield from comm.x(...)` convention;
    its return value is collected per rank."""

    def __init__(self, size):
        self.size = size

    def run(self, fn, per_rank_args):
        P = self.size
        gens = [fn(LoopbackComm(r, P), *per_rank_args[r]) for r in range(P)]
        results = [None] * P
        pending = [None] * P
        alive = [True] * P
        # prime
        for r in range(P):
            try:
                pending[r] = next(gens[r])
            except StopIteration as e:
                results[r] = e.value
                alive[r] = False
        while any(alive):
            if not all(alive):
                raise RuntimeError("loopback: ranks left the collective sequence at different points")
            kinds = set(p.kind for p in pending)
            if len(kinds) != 1:
                raise RuntimeError("loopback: mismatched collectives %s" % kinds)
            kind = kinds.pop()
            if kind == "gather":
                objs = [p.payload for p in pending]
                answers = [list(objs) for _ in range(P)]
            elif kind == "reduce":
                total = pending[0].payload.clone()
                for p in pending[1:]:
                    total += p.payload.to(total.device)
                answers = [total.to(pending[r].payload.device).clone() for r in range(P)]
            elif kind == "a2a":
                answers = []
                for r in range(P):
                    dev = pending[r].payload[0].device
                    answers.append([pending[s].payload[r].to(dev).clone() for s in range(P)])
            else:
                raise RuntimeError(kind)
            for r in range(P):
                try:
                    pending[r] = gens[r].send(answers[r])
                except StopIteration as e:
                    results[r] = e.value
                    alive[r] = False
        return results
