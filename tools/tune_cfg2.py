"""Measurement helper: cfg2 kernel time vs chains-per-block (launch shape knob)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
for cpb in (64, 32, 16, 128, 256):
    _abi.set_tuning("small_chains_per_block", cpb)
    w = bench.Cfg2(dev, 1024, 1000, 0)
    w.step(0); torch.cuda.synchronize()
    ts = []
    for k in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); w.step(k + 1); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ms = sorted(ts)[len(ts) // 2]
    print("chains/block %4d: %.3f ms per launch, %.3e chain-steps/s, %.1f ns per leapfrog step per wave"
          % (cpb, ms, 1024 * 1000 * 25 / (ms * 1e-3), ms * 1e6 / (1000 * 25)))
