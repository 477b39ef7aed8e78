"""Bin-layer sharding of BatchedFHEHIPPIE::run() across GPUs (one process per GPU).

The outer loop over bin layers (reference BatchedFHEHIPPIE.cpp:91) has independent iterations: each
reads its own K*E plaintexts and mask, shares the read-only index / minus ciphertexts and the
relinearisation key, and writes its own resultList[bin] (.cpp:127).  Rank r therefore owns a contiguous
slice of bin layers, holds only that slice of the packed database, and the single collective of the
path is the final gather of result ciphertexts (RCCL over xGMI; `gloo` in the CPU tests): to the rank that
answers the client (gather_bins_to), or to every rank (gather_bins).  With point-to-point xGMI links the
gather to one rank moves each slice over its own link once; the all-gather moves 8x the bytes.
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


def gather_bins_to(local, b, world, dst=0, out=None, group=None, async_op=False):
    """As gather_bins, but only rank `dst` receives: the server process that sends the result list to the client
    (reference BatchedFHEPSIServer.cpp:143-152).  Slices are padded to max_bins(b, world) rows; `out` ([world * bmax, W],
    rank dst only) receives them in rank order.  Returns (out or None, work handle or None)."""
    if world == 1:
        return local, None
    rank = dist.get_rank(group)
    bmax = max_bins(b, world)
    W = local.shape[1]
    lo, hi = bin_slice(b, rank, world)
    if local.shape[0] == bmax:
        padded = local
    else:
        padded = torch.zeros((bmax, W), dtype=local.dtype, device=local.device)
        padded[: hi - lo] = local[: hi - lo]
    parts = None
    if rank == dst:
        if out is None:
            out = torch.empty((world * bmax, W), dtype=local.dtype, device=local.device)
        parts = list(out.view(world, bmax, W).unbind(0))
    work = dist.gather(padded, gather_list=parts, dst=dst, group=group, async_op=async_op)
    return (out if rank == dst else None), work
