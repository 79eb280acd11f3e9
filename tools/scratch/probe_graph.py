import sys, os, warnings
sys.path.insert(0, "/root/repo")
import torch
import hamiltorch_amd as ht
from hamiltorch_amd import util
warnings.simplefilter("always")
dev = torch.device("cuda:0")
f = lambda w: -(w * w).sum() * 0.5 - 0.1 * (w ** 4).sum()
gv = torch.func.vmap(torch.func.grad_and_value(f))
th = torch.randn(64, 5, device=dev)
static = th.clone()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): gv(static)
torch.cuda.current_stream().wait_stream(s)
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = gv(static)
    g.replay(); torch.cuda.synchronize()
    print("capture ok", out[0].shape)
except Exception as e:
    import traceback; traceback.print_exc()
