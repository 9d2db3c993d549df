"""Range-sharding of independent sites over the GPUs of one node.

Every site (feature row) is independent (the forward has no cross-row term, models.py:178-240), so the N
ranks take contiguous ranges with NO data-path collective.  The only exchanges are (1) an all_gather of one
integer per rank (rows in my byte range) so that each rank knows the GLOBAL index of its first row -- the
in-kernel Philox initial states are keyed by it, which makes results independent of N -- and (2) the
optional final gather of per-site probabilities.  Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests."""
from __future__ import annotations

import os


def env_world():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def split_range(n, world, rank):
    """Contiguous split of n items: rank r gets [r*ceil(n/world), (r+1)*ceil(n/world)) clipped to n
    (SURVEY.md 8(e) 'Partitioning')."""
    per = (n + world - 1) // world
    a = min(n, rank * per)
    return a, min(n, a + per)


def align_to_line_start(mm, pos, size):
    """Smallest offset >= pos that begins a row (pos itself when it is 0 or follows a newline)."""
    if pos <= 0:
        return 0
    if pos >= size:
        return size
    if mm[pos - 1:pos] == b"\n":
        return pos
    nl = mm.find(b"\n", pos)
    return size if nl < 0 else nl + 1


def byte_range_for_rank(mm, size, world, rank):
    """Byte range [a, b) of a plain-text feature file owned by `rank`: the raw split points advanced to the
    next row start, so every row belongs to exactly one rank."""
    a, b = split_range(size, world, rank)
    return align_to_line_start(mm, a, size), align_to_line_start(mm, b, size)


def exclusive_prefix(counts, rank):
    return int(sum(counts[:rank]))


def all_gather_ints(value, world, device=None):
    """One integer per rank -> list of all ranks' integers (torch.distributed must be initialised)."""
    if world == 1:
        return [int(value)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def gather_probs(probs, world, dst=0):
    """Optional final gather of per-site probabilities [n_r, C] to rank `dst` (ragged: sizes exchanged
    first).  One collective on RCCL over xGMI; ~8 B/site."""
    if world == 1:
        return [probs]
    import torch
    import torch.distributed as dist
    sizes = all_gather_ints(probs.shape[0], world, probs.device if probs.is_cuda else None)
    mx = max(sizes)
    pad = torch.zeros((mx, probs.shape[1]), dtype=probs.dtype, device=probs.device)
    pad[:probs.shape[0]] = probs
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[:s] for o, s in zip(out, sizes)] if dist.get_rank() == dst else None


def gather_columns(cols, world, device=None):
    """Ragged gather of equally long 1-D int64 columns to rank 0: sizes exchanged first, then one padded all_gather
    per column (RCCL when `device` is a GPU).  Returns the concatenated columns on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    n = int(cols[0].numel())
    sizes = all_gather_ints(n, world, device)
    mx = max(max(sizes), 1)
    out = []
    for c in cols:
        src = c if device is not None else c.cpu()
        pad = torch.zeros(mx, dtype=src.dtype, device=src.device)
        pad[:n] = src
        got = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(got, pad)
        if dist.get_rank() == 0:
            out.append(torch.cat([g[:k] for g, k in zip(got, sizes)]))
    return out if dist.get_rank() == 0 else None
