"""The models of the reference's BNN notebooks, unmodified user code: a softmax regression on a 4-feature / 3-class data set
(hamiltorch_Bayesian_NN_example: ``Net([4, 3])``, ``model_loss='multi_class_linear_output'`` - sample_model's default) and the
two-hidden-layer regression net of hamiltorch_split_HMC_BNN_example (``Net([1, 10, 10, 1])``), both written as plain
``nn.Module`` classes with ``torch.relu`` calls in ``forward``.  hamiltorch_amd recognises such a forward by tracing it and runs
the whole (split-)HMC loop in the small-network kernel (csrc/netn_hmc.hip), hundreds of chains per call.

    python examples/bnn_notebook_models.py     (needs a GPU)
"""
import os
import sys
import time
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import hamiltorch_amd as hamiltorch  # noqa: E402


class Net(nn.Module):                                   # as in the notebooks
    def __init__(self, layer_sizes, bias=True):
        super().__init__()
        self.layer_sizes = layer_sizes
        self.l1 = nn.Linear(layer_sizes[0], layer_sizes[1], bias=bias)
        if len(layer_sizes) > 2:
            self.l2 = nn.Linear(layer_sizes[1], layer_sizes[2], bias=bias)
            self.l3 = nn.Linear(layer_sizes[2], layer_sizes[3], bias=bias)

    def forward(self, x):
        x = self.l1(x)
        if len(self.layer_sizes) > 2:
            x = torch.relu(x)
            x = self.l2(x)
            x = torch.relu(x)
            x = self.l3(x)
        return x


def main():
    dev = torch.device("cuda:0")
    hamiltorch.set_random_seed(0)
    g = torch.Generator().manual_seed(0)
    chains = 512

    # ---- softmax regression, three classes
    centres = torch.tensor([[2.0, 0.0, 0.0, 1.0], [-1.0, 2.0, 0.0, 0.0], [0.0, -1.0, 2.0, -1.0]])
    y = torch.randint(0, 3, (150,), generator=g)
    x = centres[y] + 0.7 * torch.randn(150, 4, generator=g)
    net = Net([4, 3]).to(dev)
    theta0 = hamiltorch.util.flatten(net).detach().repeat(chains, 1).contiguous()
    t0 = time.time()
    samples = hamiltorch.sample_model(net, x.to(dev), y.float().reshape(-1, 1).to(dev), theta0, model_loss="multi_class_linear_output",
                                      num_samples=300, num_steps_per_sample=10, step_size=0.02, burn=100, tau_out=1.0, verbose=False)
    torch.cuda.synchronize()
    th = torch.stack(samples)                           # [samples, chains, 15]
    W = th[..., :12].reshape(-1, 3, 4); b = th[..., 12:].reshape(-1, 3)
    pred = (torch.einsum("soi,ni->sno", W, x.to(dev)) + b[:, None, :]).softmax(-1).mean(0).argmax(-1)
    print("softmax regression:  %d chains x %d samples (L=10) in %.2f s; posterior-mean accuracy on the training set %.2f"
          % (chains, len(samples), time.time() - t0, (pred.cpu() == y).float().mean()))

    # ---- two hidden layers, regression with data splitting
    X = torch.linspace(-2, 2, 400).reshape(-1, 1)
    Y = torch.sin(3 * X) + 0.1 * torch.randn(400, 1, generator=g)
    net = Net([1, 10, 10, 1]).to(dev)
    theta0 = hamiltorch.util.flatten(net).detach().repeat(chains, 1).contiguous()
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=100, shuffle=False)
    t0 = time.time()
    samples = hamiltorch.sample_split_model(net, loader, theta0, 4, model_loss="regression", num_samples=200, num_steps_per_sample=10,
                                            step_size=5e-4, burn=50, inv_mass=torch.ones(theta0.shape[1], device=dev), tau_out=100.0,
                                            verbose=False)
    torch.cuda.synchronize()
    pred, _ = hamiltorch.predict_model(net, [s[0] for s in samples[-50:]], x=X.to(dev), y=Y.to(dev), model_loss="regression", tau_out=100.0)
    print("Net([1,10,10,1]):    %d chains x %d samples (L=10, 4 splits) in %.2f s; rmse of chain 0's posterior mean %.3f"
          % (chains, len(samples), time.time() - t0, float(((pred.mean(0) - Y.to(dev)) ** 2).mean().sqrt())))


if __name__ == "__main__":
    main()
