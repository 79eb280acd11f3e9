"""r05ac: the callback path's launches per step split into the library's own (hta::) and torch's."""
import sys, torch
sys.path.insert(0, ".")
from benchlib.workloads import FunnelHMC, FunnelRMHMC
dev = torch.device("cuda", 0)
for W in (FunnelHMC,):
    w = W(dev, None, None, chain_offset=0)
    w.step(0); torch.cuda.synchronize()
    n = w._launches()
    print(W.key, "launches per step", n, "of which hta::", w._hta_launches, "leapfrog steps per step", w.T * w.L,
          "=> per leapfrog step: %.1f in all, %.2f the library's" % (n / (w.T * w.L), w._hta_launches / (w.T * w.L)))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        w._sample(w.fn, 200, w.T); torch.cuda.synchronize()
    evs = sorted((e for e in prof.key_averages() if "cuda" in str(getattr(e, "device_type", "")).lower()), key=lambda e: -e.count)
    for e in evs[:40]: print("  %6d  %s" % (e.count, str(e.key)[:110]))
