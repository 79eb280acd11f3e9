"""Developer check: the bench's cfg3-eig extra at 1024 chains, in the bench's own order (256 chains x 20 trajectories first), wall against kernel time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchlib.measure import measure, result_of
from benchlib.workloads import Cfg3
dev = torch.device("cuda:0")
for rep in range(2):
    w = Cfg3(dev, None, 20, 0, jacobi=True)
    m = measure(w, 5, 1, 1, None, dev, 1)
    r = result_of(w, Cfg3, *m, 5, 1, 1)
    print("256:", r["value"], r["ms_per_step"], r["roofline"].get("kernel_ms_per_step"), flush=True)
    del w
    torch.cuda.empty_cache()
    w = Cfg3(dev, 1024, 5, 0, jacobi=True)
    for k in range(3):
        m1k = measure(w, 3, 1, 1, None, dev, 1)
        r1k = result_of(w, Cfg3, *m1k, 3, 1, 1)
        print("1024:", r1k["value"], r1k["ms_per_step"], r1k["roofline"].get("kernel_ms_per_step"), flush=True)
    del w
    torch.cuda.empty_cache()
