"""r05v: the host-buffer boundary of sample(): D2H of a cfg2 call's samples (pageable .cpu() against a pinned staging copy) and the
PCIe-inclusive time of one sample() call with host-resident params_init and store_on_GPU=False."""
import statistics
import time

import torch

import hamiltorch_amd as ht
from benchlib.workloads import Cfg2

dev = torch.device("cuda", 0)
s = torch.randn(1001, 1024, 3, device=dev)
torch.cuda.synchronize()


def med(f, n=9):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts)


def pinned():
    out = torch.empty(s.shape, dtype=s.dtype, pin_memory=True)
    out.copy_(s, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return out


keep = []
print("bytes", s.numel() * 4)
print("pageable .cpu()            ms (median, min)", med(lambda: s.cpu()))
print("pinned, buffer released    ms", med(lambda: pinned()))
print("pinned, buffer kept        ms", med(lambda: keep.append(pinned())))
w = Cfg2(dev, None, None, chain_offset=0)
th_host = w.theta0.cpu()
for store in (True, False):
    for host_in in (False, True):
        f = lambda: ht.sample(w.tgt, th_host if host_in else w.theta0, num_samples=w.T, num_steps_per_sample=w.L, step_size=w.eps, burn=-1,
                              verbose=False, seed=7, store_on_GPU=store)
        f()
        print("sample(): params_init on %s, store_on_GPU=%s: ms per call (median, min)" % ("host" if host_in else "device", store), med(f))
