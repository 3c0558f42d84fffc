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
import torch


class TorchComm(object):
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    # every method is a generator so that call sites are identical for both back-ends
    def all_gather_obj(self, obj):
        """Small Python objects (ints, tuples): list indexed by rank."""
        out = [None] * self.size
        self.dist.all_gather_object(out, obj, group=self.group)
        return out
        yield  # pragma: no cover  (makes this a generator)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t
        yield  # pragma: no cover

    def all_to_all_v(self, chunks):
        """chunks[d] = 1-D tensor for rank d (any lengths).  Returns the list of tensors
        received from every rank (index = source)."""
        counts = [int(c.numel()) for c in chunks]
        rcounts = [None] * self.size
        self.dist.all_gather_object(rcounts, counts, group=self.group)
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
