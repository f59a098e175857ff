"""Multi-GPU sharding of independent sequences: sequence k lives on rank k // per_gpu; there is no
data-path collective.  The only exchange is the gather of the newest pose of every sequence (7 doubles each) onto rank 0,
over RCCL on GPUs (torch.distributed backend "nccl") or gloo in the CPU tests."""
import torch


def first_sequence(rank, per_gpu):
    return rank * per_gpu


def owner_of(seq, per_gpu):
    return seq // per_gpu


def partition(n_sequences, world):
    """contiguous, balanced shards: returns list of (first, count) per rank"""
    base, rem = divmod(n_sequences, world)
    out, first = [], 0
    for r in range(world):
        c = base + (1 if r < rem else 0)
        out.append((first, c))
        first += c
    return out


def gather_poses(newest, dist, world):
    """newest: [per_gpu, 7] float64 tensor on this rank's device. Returns [world*per_gpu, 7] on every rank (all_gather keeps
    ranks symmetric; 56 B per sequence, latency-bound — one collective per step at most)."""
    if dist is None or world == 1:
        return newest.clone()
    out = [torch.empty_like(newest) for _ in range(world)]
    dist.all_gather(out, newest.contiguous())
    return torch.cat(out, 0)
