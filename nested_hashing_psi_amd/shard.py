"""Bin-layer sharding of BatchedFHEHIPPIE::run() across GPUs (one process per GPU).

The outer loop over bin layers (reference BatchedFHEHIPPIE.cpp:91) has independent iterations: each
reads its own K*E plaintexts and mask, shares the read-only index / minus ciphertexts and the
relinearisation key, and writes its own resultList[bin] (.cpp:127).  Rank r therefore owns a contiguous
slice of bin layers, holds only that slice of the packed database, and the single collective of the
path is the final gather of result ciphertexts (RCCL all-gather over xGMI; `gloo` in the CPU tests).
"""
import torch
import torch.distributed as dist


def bin_slice(b, rank, world):
    """[lo, hi): the bin layers rank `rank` of `world` evaluates (sizes differ by at most one)"""
    return (b * rank) // world, (b * (rank + 1)) // world


def max_bins(b, world):
    return -(-b // world)


def gather_bins(local, b, world, out=None, group=None):
    """local: [b_local, W] result ciphertexts of this rank's slice (any integer dtype, device tensor for
    RCCL / CPU tensor for gloo).  Returns [b, W] in bin order on every rank."""
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = bin_slice(b, rank, world)
    assert local.shape[0] == hi - lo, "local result count does not match this rank's bin slice"
    if world == 1:
        return local
    bmax = max_bins(b, world)
    W = local.shape[1]
    padded = torch.zeros((bmax, W), dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    if out is None:
        out = torch.empty((world * bmax, W), dtype=local.dtype, device=local.device)
    parts = list(out.view(world, bmax, W).unbind(0))
    dist.all_gather(parts, padded, group=group)
    if b == world * bmax:
        return out
    keep = []
    for r in range(world):
        rlo, rhi = bin_slice(b, r, world)
        keep.append(out[r * bmax: r * bmax + (rhi - rlo)])
    return torch.cat(keep, dim=0)
