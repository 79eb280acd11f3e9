#!/usr/bin/env python3
"""RCCL on the one-GPU box: a world-size-1 process group on backend "nccl" (= RCCL on ROCm) running exactly the collectives
hamiltorch_amd/dist.py and bench.py issue on the 8-GPU node - the float64 all-reduce of the NUTS statistic, the all-gather /
gather of a cfg5-sized sample block (samples[101, 1024, 100] fp32 = 41 MB), the all-reduce of ones behind `ranks_seen`, the
barrier.  It cannot show a link rate (one rank: the collectives are device-local copies); it shows that the library loads,
initialises on gfx950 with the box's environment (HSA_ENABLE_IPC_MODE_LEGACY=0, 127.0.0.1 rendezvous) and accepts the dtypes,
shapes and call forms of the path.  Prints one JSON line; never raises (a failure is a finding, with the message).

    python tools/rccl_world1.py > gpurun_out/rccl_world1.json
"""
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = {"backend": "nccl", "world_size": 1}
    try:
        import torch
        import torch.distributed as dist
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        out["device"] = torch.cuda.get_device_name(0)
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            out["rccl_version"] = "unknown (%s)" % type(e).__name__
        t0 = time.perf_counter()
        dist.init_process_group("nccl", device_id=dev)
        out["init_s"] = round(time.perf_counter() - t0, 3)

        def timed(name, fn, reps=3):
            fn(); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out[name + "_ms"] = round((time.perf_counter() - t) / reps * 1e3, 4)

        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        out["ranks_seen"] = int(one.item())
        stat = torch.tensor([0.75, 1024.0, 0.0], dtype=torch.float64, device=dev)           # dist.sample_sharded's NUTS reducer
        timed("all_reduce_f64x3", lambda: dist.all_reduce(stat))
        out["all_reduce_f64x3_value_ok"] = bool(abs(float(stat[1]) - 1024.0) < 1e-9)
        block = torch.randn(101, 1024, 100, device=dev)                                      # BASELINE config 5: one rank's samples
        gathered = torch.empty_like(block)
        timed("all_gather_41MB", lambda: dist.all_gather_into_tensor(gathered, block))
        out["all_gather_equal"] = bool(torch.equal(gathered, block))
        bufs = [torch.empty_like(block)]
        timed("gather_41MB", lambda: dist.gather(block, bufs, dst=0))
        out["gather_equal"] = bool(torch.equal(bufs[0], block))
        timed("barrier", lambda: dist.barrier())
        # the package's own entry points on this group (world 1: gather_samples / sample_sharded take their single-rank path)
        import hamiltorch_amd as ht
        from hamiltorch_amd import dist as hdist
        tgt = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=torch.eye(3, device=dev))
        th0 = 0.1 * torch.randn(64, 3, device=dev)
        rows = hdist.sample_sharded(ht.sample, th0, tgt, num_samples=20, num_steps_per_sample=5, step_size=0.3, seed=5, verbose=False)
        ref = ht.sample(tgt, th0, num_samples=20, num_steps_per_sample=5, step_size=0.3, seed=5, verbose=False)
        out["sample_sharded_equals_sample"] = bool(torch.equal(torch.stack(rows), torch.stack(list(ref))))
        dist.destroy_process_group()
        out["ok"] = True
    except Exception as e:  # noqa: BLE001 - a failure here is a finding, not a crash
        out["ok"] = False
        out["error"] = "%s: %s" % (type(e).__name__, str(e)[:500])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
