"""Measurement helper: cfg2 kernel time vs chains-per-block (launch shape knob)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hamiltorch_amd import _abi
dev = torch.device("cuda:0")
for cpb, use_ws, quad in ((64, True, 0), (64, False, 0), (128, True, 0), (256, True, 0)):
    _abi.set_tuning("small_chains_per_block", cpb)
    w = bench.Cfg2(dev, 1024, 1000, 0)
    if not use_ws:
        w.ws = None
    w.step(0); torch.cuda.synchronize()
    ts = []
    for k in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); w.step(k + 1); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ms = sorted(ts)[len(ts) // 2]
    print("quad=%d ws=%d chains/block %4d: %.3f ms per launch, %.3e chain-steps/s, %.1f ns per leapfrog step per wave"
          % (quad, use_ws, cpb, ms, 1024 * 1000 * 25 / (ms * 1e-3), ms * 1e6 / (1000 * 25)))
