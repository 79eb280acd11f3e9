"""The reference's largest example: the convolutional network of notebooks/hamiltorch_Bayesian_NN_example.ipynb (cells 24-27:
Conv(1,20,5)-pool-Conv(20,50,5)-pool-Linear(800,500)-Linear(500,10), D = 431 080 parameters, 100 training digits, softmax
likelihood, HMC with L = 20, step 0.001, tau = 10, tau_out = 1; the notebook reports 11.22 samples/s for its one chain on the
author's GPU).  No native kernel covers a convolutional model: it runs on the generic path - torch evaluates the functional
model and its gradient for ALL chains at once (torch.func.vmap, replayed as a HIP graph), the HIP kernels do the momentum draw,
kick / drift, the Hamiltonian and the Metropolis bookkeeping.  MNIST is not available offline: synthetic 28 x 28 "digits"
(class-dependent blobs) stand in for it.

    python examples/bnn_cnn.py [chains]     (needs a GPU)
"""
import os
import sys
import time
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import hamiltorch_amd as hamiltorch  # noqa: E402


class Net(nn.Module):
    """ConvNet -> Max_Pool -> RELU -> ConvNet -> Max_Pool -> RELU -> FC -> RELU -> FC (the notebook's class)"""
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 20, 5, 1)
        self.conv2 = nn.Conv2d(20, 50, 5, 1)
        self.fc1 = nn.Linear(4 * 4 * 50, 500)
        self.fc2 = nn.Linear(500, 10)

    def forward(self, x):
        x = F.relu(self.conv1(x))
        x = F.max_pool2d(x, 2, 2)
        x = F.relu(self.conv2(x))
        x = F.max_pool2d(x, 2, 2)
        x = x.view(-1, 4 * 4 * 50)
        x = F.relu(self.fc1(x))
        return self.fc2(x)


def digits(n, g):
    """n synthetic 28 x 28 images in 10 classes: a bright blob whose position encodes the class, plus noise."""
    y = torch.arange(n) % 10
    x = 0.1 * torch.rand(n, 1, 28, 28, generator=g)
    for i in range(n):
        r, c = 4 + 2 * int(y[i]) // 2 * 2 % 20, 4 + (int(y[i]) * 7) % 20
        x[i, 0, r:r + 6, c:c + 6] += 0.8
    return x.clamp(0, 1), y.reshape(-1, 1).float()


def main():
    dev = torch.device("cuda:0")
    chains = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    hamiltorch.set_random_seed(123)
    g = torch.Generator().manual_seed(0)
    x_train, y_train = digits(100, g)
    x_val, y_val = digits(300, g)
    net = Net().to(dev)
    theta0 = hamiltorch.util.flatten(net).detach().repeat(chains, 1).contiguous()
    print("Parameter size:", theta0.shape[1], "| chains:", chains)
    tau_list = torch.full((len(list(net.parameters())),), 10.0, device=dev)
    kw = dict(model_loss="multi_class_linear_output", num_steps_per_sample=20, step_size=0.001, tau_out=1.0, tau_list=tau_list,
              verbose=False)
    hamiltorch.sample_model(net, x_train.to(dev), y_train.to(dev), theta0, num_samples=3, **kw)     # warm-up: tracing, graph capture
    torch.cuda.synchronize()
    N = 30
    t0 = time.time()
    samples = hamiltorch.sample_model(net, x_train.to(dev), y_train.to(dev), theta0, num_samples=N, **kw)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("sample_model (HMC, L = 20): %d chains x %d samples in %.2f s = %.1f samples/s over all chains, %.2f per chain "
          "(the notebook: 11.22 samples/s, one chain)" % (chains, N, dt, chains * N / dt, N / dt))
    one_chain = [s[0] for s in samples]
    pred, _ = hamiltorch.predict_model(net, one_chain, x=x_val.to(dev), y=y_val.to(dev), model_loss="multi_class_linear_output",
                                       tau_out=1.0, tau_list=tau_list)
    acc = float((pred.mean(0).argmax(-1) == y_val.to(dev).flatten()).float().mean())
    print("posterior-mean accuracy of chain 0 on 300 held-out synthetic digits: %.2f" % acc)


if __name__ == "__main__":
    main()
