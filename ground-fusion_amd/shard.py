"""Multi-GPU sharding of independent sequences: sequences are dealt to ranks in contiguous shards; there is no
data-path collective.  The only exchange is the gather of the newest pose of every sequence (7 doubles each), over RCCL on GPUs
(torch.distributed backend "nccl") or gloo in the CPU tests.  Shards may be uneven (n_sequences % world != 0): the gather pads
every rank's block to the largest shard and strips the padding again."""
import torch


def partition(n_sequences, world):
    """contiguous, balanced shards: returns list of (first, count) per rank"""
    base, rem = divmod(n_sequences, world)
    out, first = [], 0
    for r in range(world):
        c = base + (1 if r < rem else 0)
        out.append((first, c))
        first += c
    return out


class Plan:
    """who owns which sequence; every helper derives from the one partition table"""

    def __init__(self, n_sequences, world):
        self.n, self.world = n_sequences, world
        self.parts = partition(n_sequences, world)

    def first(self, rank):
        return self.parts[rank][0]

    def count(self, rank):
        return self.parts[rank][1]

    def counts(self):
        return [c for _, c in self.parts]

    def owner_of(self, seq):
        for r, (f, c) in enumerate(self.parts):
            if f <= seq < f + c:
                return r
        raise IndexError("sequence %d outside 0..%d" % (seq, self.n - 1))


def first_sequence(rank, per_gpu):
    """uniform shards of per_gpu sequences (weak scaling: the bench's layout)"""
    return rank * per_gpu


def owner_of(seq, per_gpu):
    return seq // per_gpu


class PoseGather:
    """The per-step exchange on persistent buffers: one [world, cmax, 7] receive buffer and one [cmax, 7] send buffer per rank, allocated once;
    `all_gather_into_tensor` (one flat collective, no per-step tensor lists) where the backend has it -- RCCL does --, else `all_gather` into views of
    the same persistent buffer (gloo in the CPU tests)."""

    def __init__(self, dist, world, counts, device, dtype=torch.float64):
        self.dist, self.world, self.counts = dist, world, list(counts)
        self.cmax = max(self.counts)
        self.uniform = all(c == self.cmax for c in self.counts)
        self.send = torch.zeros((self.cmax, 7), dtype=dtype, device=device)
        self.recv = torch.zeros((world, self.cmax, 7), dtype=dtype, device=device)
        self.views = list(self.recv.unbind(0))
        self.flat = None      # None: not probed yet; True / False: all_gather_into_tensor works on this backend

    def __call__(self, newest):
        """newest: [count(rank), 7] on this rank's device -> [sum(counts), 7], sequences in global order (a view of the persistent buffer when the shards are even)"""
        if self.dist is None or self.world == 1:
            return newest
        n = newest.shape[0]
        self.send[:n].copy_(newest)
        if self.flat is not False:
            try:
                self.dist.all_gather_into_tensor(self.recv, self.send)
                self.flat = True
            except (RuntimeError, NotImplementedError, AttributeError):
                if self.flat:          # it worked before: a real failure, not a missing feature
                    raise
                self.flat = False
        if self.flat is False:
            self.dist.all_gather(self.views, self.send)
        if self.uniform:
            return self.recv.view(-1, 7)
        return torch.cat([self.recv[r, :c] for r, c in enumerate(self.counts)], 0)


def gather_poses(newest, dist, world, counts=None):
    """newest: [count(rank), 7] float64 tensor on this rank's device.  Returns [sum(counts), 7] on every rank, sequences in global
    order (all_gather keeps ranks symmetric; 56 B per sequence, latency-bound -- one collective per step).  counts: sequences per
    rank when the shards are uneven (Plan.counts()); None = every rank holds newest.shape[0]."""
    if dist is None or world == 1:
        return newest.clone()
    if counts is None:
        counts = [newest.shape[0]] * world
    cmax = max(counts)
    mine = newest.contiguous()
    if mine.shape[0] < cmax:   # pad to the largest shard: all_gather needs equal shapes
        mine = torch.cat([mine, torch.zeros((cmax - mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=mine.device)], 0)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return torch.cat([o[:c] for o, c in zip(out, counts)], 0)
