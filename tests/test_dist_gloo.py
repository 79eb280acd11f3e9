"""CPU, world_size 2 over gloo: the N > 1 path -- chain sharding by global id and the one-off gather.
(The kernels cannot run here; every rank fabricates its block of samples from the global-chain-id-keyed
Philox stream, exactly as the device RNG is keyed.)"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_local_sampler(params_init, chain_offset, seed, num_samples, **kw):
    """Stand-in for hamiltorch_amd.sample with the same sharding contract: row n of chain c depends only on
    (seed, global chain id, n)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hmc_oracle as O
    C, D = params_init.shape
    rows = [params_init]
    for n in range(1, num_samples):
        z = O.philox_normals(seed, chain_offset + np.arange(C), n, D, dtype=np.float32)
        rows.append(torch.from_numpy(z))
    return rows


def _worker(rank, world, port, C, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from hamiltorch_amd import dist as hd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        init = torch.arange(C * 3, dtype=torch.float32).reshape(C, 3)
        rows = hd.sample_sharded(_fake_local_sampler, init, seed=99, num_samples=5)
        full = torch.stack(rows)
        off, cnt = hd.shard_chains(C, rank, world)
        only0 = hd.gather_samples(full[:, off:off + cnt].contiguous(), C, dst=0)
        # Sampler.HMC_NUTS under sharding: the acceptance statistic is reduced over ALL ranks' chains (samplers._nuts_reduce)
        seen = {}

        def probe_sampler(params_init, chain_offset, seed, **kw):
            from hamiltorch_amd import samplers
            seen["hook"] = samplers._nuts_reduce.get()
            seen["red"] = samplers._nuts_reduce.get()(float(params_init.shape[0]) * 0.5, float(params_init.shape[0]), rank == 1)
            import threading
            other = []
            t = threading.Thread(target=lambda: other.append(samplers._nuts_reduce.get()))     # multi_chain(parallel=True)'s threads
            t.start(); t.join()
            seen["other_thread"] = other[0]
            return [params_init]
        hd.sample_sharded(probe_sampler, init, seed=1, gather=False)
        from hamiltorch_amd import samplers
        assert seen["hook"] is not None and samplers._nuts_reduce.get() is None        # installed for the call only
        assert seen["red"] == (0.5 * C, float(C), True), seen["red"]
        assert seen["other_thread"] is None                                              # context-local, not a module global
        q.put((rank, full.numpy(), None if only0 is None else only0.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("C", [8, 7])
def test_sharded_sampling_equals_single_process(C):
    from hamiltorch_amd import dist as hd
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + C
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    init = torch.arange(C * 3, dtype=torch.float32).reshape(C, 3)
    single = torch.stack(_fake_local_sampler(init, 0, 99, 5)).numpy()
    for rank, full, only0 in res:
        assert full.shape == (5, C, 3)
        assert np.array_equal(full, single), "rank %d: gathered samples differ from the 1-process run" % rank
        if rank == 0:
            assert np.array_equal(only0, single)
        else:
            assert only0 is None


def test_shard_chains_partitions_exactly():
    from hamiltorch_amd import dist as hd
    for C in (1, 7, 8, 1024, 8192, 8191):
        for world in (1, 2, 3, 4, 8):
            blocks = [hd.shard_chains(C, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == C
            for (o1, c1), (o2, _) in zip(blocks, blocks[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
