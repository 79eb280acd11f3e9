"""Bayesian neural-network regression as in the reference's BNN notebooks: sample_model (full data, HMC) and
sample_split_model (symmetric data splitting) for hundreds of chains at once, then the posterior predictive.

    python examples/bnn_regression.py     (needs a GPU)
"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import hamiltorch_amd as hamiltorch  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hamiltorch.set_random_seed(0)
    g = torch.Generator().manual_seed(0)
    X = torch.randn(400, 8, generator=g)
    w = torch.randn(8, 1, generator=g)
    Y = torch.sin(X @ w) + 0.1 * torch.randn(400, 1, generator=g)
    net = torch.nn.Sequential(torch.nn.Linear(8, 100), torch.nn.ReLU(), torch.nn.Linear(100, 1)).to(dev)
    chains = 512
    theta0 = hamiltorch.util.flatten(net).detach().repeat(chains, 1).contiguous()
    tau_list = torch.ones(4, device=dev)

    t0 = time.time()
    samples = hamiltorch.sample_model(net, X.to(dev), Y.to(dev), theta0, model_loss="regression", num_samples=60,
                                      num_steps_per_sample=20, step_size=5e-4, burn=20, tau_out=100.0, tau_list=tau_list,
                                      verbose=False)
    torch.cuda.synchronize()
    print("sample_model:        %d chains x %d samples (L=20) in %.2f s" % (chains, len(samples), time.time() - t0))

    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(X, Y), batch_size=100, shuffle=False)
    t0 = time.time()
    samples = hamiltorch.sample_split_model(net, loader, theta0, 4, model_loss="regression", num_samples=60,
                                            num_steps_per_sample=10, step_size=5e-4, burn=20,
                                            inv_mass=torch.ones(theta0.shape[1], device=dev), tau_out=100.0, tau_list=tau_list,
                                            verbose=False)
    torch.cuda.synchronize()
    print("sample_split_model:  %d chains x %d samples (L=10, M=4) in %.2f s" % (chains, len(samples), time.time() - t0))

    one_chain = [s[0] for s in samples]                                    # posterior predictive of chain 0
    pred, _ = hamiltorch.predict_model(net, one_chain, x=X.to(dev), y=Y.to(dev), model_loss="regression", tau_out=100.0,
                                       tau_list=tau_list)
    rmse = float(((pred.mean(0) - Y.to(dev)) ** 2).mean().sqrt())
    print("posterior-mean RMSE on the training points (chain 0): %.3f (noise level 0.1)" % rmse)


if __name__ == "__main__":
    main()
