"""Multi-GPU execution: chains are independent, so they are the only sharded axis.

One process per GPU (``torchrun``; backend ``nccl`` = RCCL over xGMI on ROCm, ``gloo`` in the CPU
tests).  Rank r owns the contiguous block of chains ``shard_chains(C, r, world)``; the device RNG is
keyed by GLOBAL chain id (``chain_offset``), so the union of all ranks' samples is identical --
bit for bit -- to a single-GPU run over all C chains (``Sampler.HMC_NUTS``: the shared step size is adapted on
the acceptance statistic of all ranks' chains -- one tiny all-reduce per burn-in trajectory -- so it is the
single-GPU step size up to the summation order of that mean).  No collective runs during sampling; the
only communication is one optional gather of the sample tensor at the end (41 MB per rank for
BASELINE config 5, well under a millisecond on the 7-link xGMI mesh).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_chains(total_chains: int, rank: int, world: int):
    """(offset, count) of the contiguous chain block of `rank`; blocks differ by at most one chain."""
    base, rem = divmod(int(total_chains), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def gather_samples(local: torch.Tensor, total_chains: int, group=None, dst=None):
    """All ranks' ``samples[S, C_local, D]`` -> ``[S, C, D]`` in global chain order.

    dst=None: every rank gets the result (all_gather); dst=r: only rank r (others get None).
    Uneven blocks are padded to the largest block for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    S, _, D = local.shape
    counts = [shard_chains(total_chains, r, world)[1] for r in range(world)]
    cmax = max(counts)
    pad = local
    if local.shape[1] < cmax:
        pad = torch.zeros(S, cmax, D, dtype=local.dtype, device=local.device)
        pad[:, :local.shape[1]] = local
    pad = pad.contiguous()
    home = pad.device
    if pad.is_cuda and dist.get_backend(group) == "gloo":      # gloo moves host memory only (CPU tests, several ranks on one GPU)
        pad = pad.cpu()
    if dst is None:
        out = torch.empty(world * S, cmax, D, dtype=local.dtype, device=pad.device)
        dist.all_gather_into_tensor(out, pad, group=group)      # rank r's block lands in rows [r*S, (r+1)*S)
        out = out.view(world, S, cmax, D)
        parts = [out[r, :, :counts[r]] for r in range(world)]
        return torch.cat(parts, dim=1).to(home)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][:, :counts[r]] for r in range(world)], dim=1).to(home)


def sample_sharded(sample_fn, params_init_all: torch.Tensor, *args, gather=True, group=None, **kwargs):
    """Run ``sample_fn`` (``hamiltorch_amd.sample`` / ``sample_model`` ...) on this rank's block of the
    ``[C, D]`` initial states and (optionally) gather the list of ``[C, D]`` samples on every rank.

    ``params_init_all`` must be the same tensor on every rank (values, not necessarily device);
    ``seed`` must be passed explicitly so that all ranks share the Philox key."""
    if "seed" not in kwargs or kwargs["seed"] is None:
        raise ValueError("sample_sharded needs an explicit seed= shared by all ranks")
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    C = params_init_all.shape[0]
    off, cnt = shard_chains(C, rank, world)
    local_init = params_init_all[off:off + cnt].contiguous()
    from . import samplers
    token = None
    if world > 1:
        def reduce_(a_sum, a_cnt, bad):          # Sampler.HMC_NUTS: one step size adapted on the chains of ALL ranks
            t = torch.tensor([a_sum, a_cnt, float(bad)], dtype=torch.float64,
                             device=local_init.device if dist.get_backend(group) != "gloo" else "cpu")
            dist.all_reduce(t, group=group)
            return float(t[0]), float(t[1]), bool(t[2] > 0)
        token = samplers._nuts_reduce.set(reduce_)          # context-local: other threads' sample() calls do not see it
    try:
        out = sample_fn(*args, params_init=local_init, chain_offset=off, **kwargs)
    finally:
        if token is not None:
            samplers._nuts_reduce.reset(token)
    extra = None
    if isinstance(out, tuple):
        out, extra = out
    local = torch.stack(out)
    if gather and world > 1:
        local = gather_samples(local, C, group=group)
    rows = list(local.unbind(0))
    return rows if extra is None else (rows, extra)
