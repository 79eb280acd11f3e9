"""Rates of the small-network kernel (csrc/netn_hmc.hip) on the reference notebooks' shapes against the callback path:
   python tools/netn_speed.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hamiltorch_amd as ht
dev = torch.device("cuda:0")


class Net(torch.nn.Module):
    def __init__(self, ls):
        super().__init__()
        self.n = len(ls) - 1
        self.l1 = torch.nn.Linear(ls[0], ls[1])
        if self.n > 1: self.l2 = torch.nn.Linear(ls[1], ls[2])
        if self.n > 2: self.l3 = torch.nn.Linear(ls[2], ls[3])

    def forward(self, x):
        x = self.l1(x)
        if self.n > 1: x = self.l2(torch.relu(x))
        if self.n > 2: x = self.l3(torch.relu(x))
        return x


def run(name, dims, loss, N, C, NS, L, eps, split):
    torch.manual_seed(0)
    net = Net(dims).to(dev)
    X = torch.randn(N, dims[0])
    Y = torch.randint(0, dims[-1], (N, 1)).float() if loss == "multi_class_linear_output" else torch.sin(3 * X.sum(1, keepdim=True)) + 0.1 * torch.randn(N, 1)
    D = sum(p.numel() for p in net.parameters())
    th0 = 0.1 * torch.randn(C, D, device=dev)
    for native in (True, False):
        ns = NS if native else max(2, NS // 20)
        def go():
            if split:
                loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=N // split, shuffle=False)
                return ht.sample_split_model(net, loader, th0, split, model_loss=loss, num_samples=ns, num_steps_per_sample=L, step_size=eps,
                                             tau_out=10.0, verbose=False, seed=1, native=native)
            return ht.sample_model(net, X, Y, th0, model_loss=loss, num_samples=ns, num_steps_per_sample=L, step_size=eps, tau_out=1.0,
                                   verbose=False, seed=1, native=native)
        go(); torch.cuda.synchronize()
        t0 = time.perf_counter(); go(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        steps = C * ns * L * (1 if not split else 1)
        print("%-44s %-8s %9.3e chain-steps/s  (%d chains x %d trajectories x L=%d in %.3f s)" % (name, "native" if native else "callback", steps / dt, C, ns, L, dt))


run("Net([1,10,10,1]) regression, N=400, split M=4", [1, 10, 10, 1], "regression", 400, 1024, 100, 10, 5e-4, 4)
run("Net([1,10,10,1]) regression, N=400, full data", [1, 10, 10, 1], "regression", 400, 1024, 100, 10, 5e-4, 0)
run("Linear(4,3) softmax (Iris shape), N=150", [4, 3], "multi_class_linear_output", 150, 1024, 200, 10, 1e-2, 0)
