"""Multi-GPU: shard the TENSOR LIST, nothing else.

Every tensor -- indeed every block -- dequantizes independently: no reduction, no halo, no
exchange step, hence no RCCL collective and no xGMI traffic on the data path (SURVEY.md
section 8e).  One process per GPU takes the tensors ``partition()`` assigns to its rank
(greedy longest-processing-time bin packing on read+write bytes, deterministic, so every rank
computes the same assignment without communicating) and runs the single-GPU path on them.
torch.distributed is used by bench.py only to fence the timed region and take the max time.
"""
from .qtypes import algorithmic_bytes


def tensor_cost(entry):
    """entry = (name, qtype, shape): bytes the dequant moves."""
    _, qtype, shape = entry[:3]
    n = 1
    for s in shape:
        n *= int(s)
    return algorithmic_bytes(qtype, n)


def partition(manifest, world_size):
    """-> list (one per rank) of lists of manifest indices.  Deterministic LPT bin packing."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(range(len(manifest)), key=lambda i: (-tensor_cost(manifest[i]), i))
    loads = [0] * world_size
    bins = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        bins[r].append(i)
        loads[r] += tensor_cost(manifest[i])
    for b in bins:
        b.sort()
    return bins


def shard(manifest, rank, world_size):
    """The manifest entries owned by ``rank``."""
    return [manifest[i] for i in partition(manifest, world_size)[rank]]


def imbalance(manifest, world_size):
    """max rank load / mean rank load (1.0 = perfect)."""
    bins = partition(manifest, world_size)
    loads = [sum(tensor_cost(manifest[i]) for i in b) for b in bins]
    mean = sum(loads) / world_size
    return max(loads) / mean if mean else 1.0
