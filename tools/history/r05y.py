"""r05y: cfg2's trajectory launch timed without bench.py's sample check (ablation builds store nothing)."""
import sys, time, torch
sys.path.insert(0, ".")
from benchlib.workloads import Cfg2
dev = torch.device("cuda", 0)
w = Cfg2(dev, None, None, chain_offset=0)
for k in range(5): w.step(k)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(50): w.step(10 + k)
    torch.cuda.synchronize()
    print(sys.argv[1], "ms per step %.5f" % ((time.perf_counter() - t0) * 1e3 / 50), getattr(w, "route", ""))
