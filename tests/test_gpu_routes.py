"""GPU: routing decisions of the API mirror -- which callables reach the fused kernels, which fall back to the
generic-callback path, and that HIP-graph replays of user callbacks can never serve a stale value."""
import json
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

import hmc_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIGMA3 = [[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]]


@pytest.fixture(scope="module")
def ht():
    import hamiltorch_amd
    assert torch.cuda.is_available()
    return hamiltorch_amd


def dev():
    return torch.device("cuda:0")


def _launches(fn):
    """Number of profiled trajectory-kernel launches (the library's own HIP-event hook) while fn() runs."""
    from hamiltorch_amd import _abi
    _abi.set_tuning("profile", 1)
    try:
        out = fn()
        torch.cuda.synchronize()
        _, n = _abi.profile_collect()
    finally:
        _abi.set_tuning("profile", 0)
    return out, n


def test_reference_test_closure_runs_on_the_fused_kernel(ht):
    """tests/test_util.py:98-101 of the reference: an opaque closure around MultivariateNormal(...).log_prob(w).sum().
    sample() recognises it by its curvature and runs hmc_gauss_quad_kernel: one trajectory-kernel launch for the whole run,
    no per-step torch callbacks, samples bit-identical to the explicit GaussianTarget."""
    def log_prob(omega):
        mean = torch.zeros(2, device=dev())
        var = torch.tensor([.10, .10], device=dev())
        return torch.distributions.MultivariateNormal(mean, torch.diag(var)).log_prob(omega).sum()

    C = 256
    init = torch.ones(C, 2, device=dev())
    kw = dict(num_samples=50, num_steps_per_sample=10, step_size=0.1, verbose=False, seed=3)
    calls = [0]

    def counted(w):
        calls[0] += 1
        return log_prob(w)
    out, n = _launches(lambda: ht.sample(counted, init, **kw))
    assert n == 1, "expected ONE fused launch, saw %d profiled launches" % n
    assert calls[0] < 40, "the closure was evaluated %d times: the run went through the callback path" % calls[0]
    t = ht.GaussianTarget(torch.zeros(2, device=dev()), covariance=torch.diag(torch.tensor([.10, .10], device=dev())))
    want = ht.sample(t, init, **kw)
    np.testing.assert_allclose(torch.stack(out).cpu().numpy(), torch.stack(want).cpu().numpy(), rtol=0, atol=2e-5)
    # opt-out: native=False keeps the callback path (and agrees)
    out_g = ht.sample(log_prob, init[:32], native=False, **kw)
    err = (torch.stack(out_g) - torch.stack(want)[:, :32]).abs().amax(dim=(0, 2))
    assert float((err > 2e-4).float().mean()) <= 0.05
    # one chain, the reference's call shape
    one = ht.sample(log_prob, torch.tensor([1., 1.], device=dev()), num_samples=20, num_steps_per_sample=10, step_size=0.1,
                    verbose=False, seed=3)
    assert len(one) == 20 and one[0].shape == (2,)


def test_notebook_closure_matches_oracle_on_fused_route(ht):
    """The notebooks' `MVN(mean, cov).log_prob(w).sum()` closure at cfg2's shape against the oracle, chain by chain."""
    mean = torch.zeros(3, device=dev()); cov = torch.tensor(SIGMA3, device=dev())

    def lp(w):
        return torch.distributions.MultivariateNormal(mean, cov).log_prob(w).sum()
    C, N, L, eps, seed = 512, 60, 25, 0.3, 17
    th0 = (0.1 * O.philox_normals(seed, np.arange(C), 0, 3, O.PURPOSE_INIT)).astype(np.float32)
    (out, acc), n = _launches(lambda: ht.sample(lp, torch.tensor(th0, device=dev()), num_samples=N, num_steps_per_sample=L,
                                                step_size=eps, debug=2, verbose=False, seed=seed))
    assert n == 1
    P = np.linalg.inv(np.array(SIGMA3))
    o = O.GaussianTarget(np.zeros(3, np.float32), P.astype(np.float32), float(-1.5 * np.log(2 * np.pi) + 0.5 * np.linalg.slogdet(P)[1]))
    ref, info = O.sample_hmc(o, th0, N, L, eps, 0, None, O.PhiloxDraws(seed, np.arange(C)))
    err = np.abs(torch.stack(out).cpu().numpy() - np.stack(ref)).max(axis=(0, 2))
    assert (err > 2e-4).mean() <= 0.01, err.max()


def test_probe_mismatch_falls_back_to_generic_path(ht):
    """A piecewise-quadratic callable whose probe points all sit in one piece: the fused run is discarded when the closed
    form disagrees with the callable on the sampled states, and the generic path's result is returned."""
    def lp(w):      # Gaussian inside |w| < 1.5, much narrower outside
        return -0.5 * (w * w).sum() - 200.0 * torch.relu(w.abs() - 1.5).pow(2).sum() * (w.abs().max() < 900.0)
    th0 = torch.zeros(64, 2, device=dev())
    kw = dict(num_samples=40, num_steps_per_sample=8, step_size=0.4, verbose=False, seed=5)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = ht.sample(lp, th0, **kw)
    want = ht.sample(lp, th0, native=False, **kw)
    if any("looked like a Gaussian" in str(w.message) for w in caught):
        assert torch.equal(torch.stack(out), torch.stack(want))
    s = torch.stack(out)
    assert float(s.abs().max()) < 2.5          # the walls were felt: this is not N(0, I)


def test_gaussian_beyond_kernel_range_uses_generic_path(ht):
    """D > 1024 is outside hta_hmc_gaussian_sample: sample() must take the callback path instead of raising."""
    D, C = 1100, 4
    t = ht.GaussianTarget(torch.zeros(D, device=dev()), precision=torch.diag(torch.linspace(0.5, 2.0, D, device=dev())), normalized=False)
    th0 = 0.1 * torch.randn(C, D, generator=torch.Generator().manual_seed(0)).to(dev())
    out, acc = ht.sample(t, th0, num_samples=4, num_steps_per_sample=3, step_size=0.05, debug=2, verbose=False, seed=1)
    s = torch.stack(out)
    assert s.shape == (4, C, D) and torch.isfinite(s).all() and float(acc.mean()) > 0.5
    # round 5: on that path the target brings its batched closed form - ONE [C, D] x [D, D] product per gradient (rocBLAS) instead
    # of autograd under vmap; same values as the opaque closure over the same matrix
    from hamiltorch_amd import samplers
    cb = samplers._BatchedCallback(t)
    assert getattr(cb, "closed_form", False)
    g, lp = cb.grad(th0)
    opaque = samplers._BatchedCallback(lambda w: t(w))
    g2, lp2 = opaque.grad(th0)
    assert torch.allclose(g, g2, rtol=1e-5, atol=1e-6) and torch.allclose(lp, lp2, rtol=1e-5, atol=1e-5)
    out2, acc2 = ht.sample(lambda w: t(w), th0, num_samples=4, num_steps_per_sample=3, step_size=0.05, debug=2, verbose=False, seed=1, native=False)
    assert torch.allclose(torch.stack(out2), s, rtol=1e-4, atol=1e-5)


def test_mlp_with_large_data_set_falls_back_instead_of_raising(ht):
    """The native MLP kernels stage the data set in LDS; the reference works for any N.  N = 60000 points exceed the staging:
    sample_model must run (generic path) rather than fail."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(2, 4), torch.nn.Tanh(), torch.nn.Linear(4, 1)).to(dev())
    X = torch.randn(60000, 2); Y = torch.sin(X.sum(1, keepdim=True))
    th0 = ht.util.flatten(net).detach().repeat(8, 1).contiguous()
    out = ht.sample_model(net, X, Y, th0, model_loss="regression", num_samples=3, num_steps_per_sample=2, step_size=1e-4,
                          tau_out=1.0, verbose=False, seed=2)
    s = torch.stack(out)
    assert s.shape == (3, 8, 17) and torch.isfinite(s).all()


def test_graph_replay_cannot_serve_stale_values(ht):
    """util.GraphedCallable only keeps a graph that was seen to follow its input: mutate the input, replay, compare with
    eager.  A function that bakes a value in at capture time, one that records no device work, and one that raises
    under capture all end up evaluated eagerly -- silently (no warnings), with the reason in util.graph_log."""
    from hamiltorch_amd import util
    x = torch.randn(8, 5, device=dev())
    good = util.GraphedCallable(torch.func.vmap(torch.func.grad(lambda w: -(w ** 4).sum())))
    for k in range(4):
        xx = x * (1.0 + k)
        np.testing.assert_allclose(good(xx).cpu().numpy(), (-4 * xx ** 3).cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert good.capturable() and good.cache
    baked = {"v": None}

    def bakes(w):                                   # host round trip outside the graph: the classic stale-value shape
        if baked["v"] is None or not torch.cuda.is_current_stream_capturing():
            baked["v"] = w.detach().clone()
        return baked["v"] * 2.0
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        gb = util.GraphedCallable(bakes)
        for k in range(3):
            xx = x + k
            np.testing.assert_allclose(gb(xx).cpu().numpy(), (2.0 * xx).cpu().numpy(), rtol=1e-6)
        empty = util.GraphedCallable(lambda w: w)                               # no device work at all
        assert empty(x) is x or torch.equal(empty(x), x)
        raises = util.GraphedCallable(lambda w: w * float(w.sum()))             # .item() under capture
        np.testing.assert_allclose(raises(x).cpu().numpy(), (x * float(x.sum())).cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(raises(2 * x).cpu().numpy(), (2 * x * float(2 * x.sum())).cpu().numpy(), rtol=1e-5)
    assert not raises.capturable() and not empty.capturable()
    assert util.graph_log


def test_generic_paths_emit_no_capture_warnings(ht):
    """A green run prints no graph-capture warnings: BNN closures are capturable now (host-scalar precisions), and whatever
    is not capturable is evaluated eagerly without comment."""
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(2, 3), torch.nn.Tanh(), torch.nn.Linear(3, 3), torch.nn.Tanh(), torch.nn.Linear(3, 1)).to(dev())
    X = torch.randn(20, 2); Y = torch.sin(X.sum(1, keepdim=True))
    th0 = ht.util.flatten(net).detach().repeat(8, 1).contiguous()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = ht.sample_model(net, X, Y, th0, model_loss="regression", num_samples=8, num_steps_per_sample=3, step_size=1e-3,
                              tau_out=1.0, verbose=False, seed=2)

        def host_scalars(w):
            return torch.distributions.Normal(0, 3, validate_args=False).log_prob(w).sum() - 0.1 * (w ** 4).sum()
        out2 = ht.sample(host_scalars, torch.zeros(8, 3, device=dev()), num_samples=8, num_steps_per_sample=3, step_size=0.1,
                         verbose=False, seed=1)
    assert torch.isfinite(torch.stack(out)).all() and torch.isfinite(torch.stack(out2)).all()


def test_bench_gpus_flag_launches_the_ranks_itself(ht):
    """`python bench.py --gpus 2` with no launcher around it re-executes itself as two ranks (ADVICE r1 / VERDICT r1 item 3).
    On this one-GPU box the ranks share the device and talk over gloo (HTA_BENCH_BACKEND); the driver's 8-GPU run uses RCCL."""
    env = dict(os.environ, HTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "cfg5", "--traj", "6", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and len(j["rank_devices"]) == 2
    assert j["config"]["chains_total"] == 2048 and j["gather_ms"] > 0 and j["value"] > 0
    # a mismatch between --gpus and the launcher's world size is an error, not a silent 1-GPU number
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                         "--no-cpu-baseline"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def _free_port():
    import socket
    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


@pytest.mark.parametrize("extra", [[], ["--workload", "cfg5", "--traj", "4"]])
def test_drivers_eight_rank_command_line_prints_the_compact_line(ht, extra):
    """The exact launcher line of the driver's scaling run - `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W` - for the default workload (cfg2 weak
    scaling) and for `--workload cfg5` (cfg3 sharded, with the gather): eight ranks share this box's one GPU over gloo
    (HTA_BENCH_BACKEND; RCCL needs eight devices), so that the first real 8-GPU run cannot fail on plumbing.  The LAST stdout
    line must be the < 4 KB record with n_gpus / ranks_seen / rank_devices / collective_backend / gather_ms."""
    env = dict(os.environ, HTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 4096 and last.startswith("{")
    j = json.loads(last)
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and len(j["rank_devices"]) == 8
    assert sorted(rd[0] for rd in j["rank_devices"]) == list(range(8))
    assert j["collective_backend"] == "gloo" and j["launcher"] == "torch.distributed.run"        # "rccl" on the driver's node
    assert j["scaling"] == "weak" and j["config"]["chains_total"] == 8 * j["config"]["chains_per_gpu"] == 8192
    assert j["value"] > 0 and "roofline" in j and j["roofline"]["kernel"]
    if extra:
        assert j["gather_ms"] > 0 and j["config"]["workload"] == "cfg5"
    else:
        assert j["config"]["workload"] == "cfg2" and "gather_ms" not in j
        # round 4: the default multi-GPU line also carries the SECOND north-star target - D = 100 explicit RMHMC
        # at 1024 chains per GPU on every rank - with the gather, so one scaling run yields both curves
        s5 = j["secondary"][0]
        assert s5["key"] == "cfg5" and s5["n_gpus"] == 8 and s5["ranks_seen"] == 8 and s5["gather_ms"] > 0 and s5["value"] > 0
        assert s5["chains"] == 1024 and s5["kernel"].startswith("rmhmc_uvc2")
        # round 6: the STRONG-scaling form beside it - 8192 chains in all, 8192 / N per GPU
        ss = j["secondary"][1]
        assert ss["key"] == "cfg5-strong" and ss["scaling"] == "strong" and ss["chains_total"] == 8192 and ss["chains"] == 1024
        assert ss["n_gpus"] == 8 and ss["value"] > 0


_SHARD_WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import hamiltorch_amd as ht
from hamiltorch_amd import dist as hd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
cov = torch.tensor([[1.0, 0.6, 0.2], [0.6, 2.0, 0.5], [0.2, 0.5, 0.5]], device=dev)
t = ht.GaussianTarget(torch.zeros(3, device=dev), covariance=cov)
C = 37                                                  # uneven blocks: 19 + 18
init = (0.1 * torch.randn(C, 3, generator=torch.Generator().manual_seed(5))).to(dev)
kw = dict(num_samples=30, num_steps_per_sample=7, step_size=0.3, burn=4, verbose=False, seed=99)
rows = hd.sample_sharded(ht.sample, init, t, **kw)
nuts, eps = hd.sample_sharded(ht.sample, init, t, sampler=ht.Sampler.HMC_NUTS, debug=2, desired_accept_rate=0.7,
                              **dict(kw, step_size=0.02, burn=10))
if rank == 0:
    full = torch.stack(ht.sample(t, init, **kw))
    nfull, neps = ht.sample(t, init, sampler=ht.Sampler.HMC_NUTS, debug=2, desired_accept_rate=0.7, **dict(kw, step_size=0.02, burn=10))
    print(json.dumps({"equal": bool(torch.equal(torch.stack(rows), full)), "shape": list(torch.stack(rows).shape),
                      "eps": eps, "eps_single": neps,
                      "nuts_err": float((torch.stack(nuts) - torch.stack(nfull)).abs().max())}))
dist.barrier()
dist.destroy_process_group()
'''


def test_sample_sharded_on_the_device_equals_single_process(ht, tmp_path):
    """dist.sample_sharded with the real kernels: two ranks (gloo, sharing this box's one GPU) over 37 chains - uneven blocks -
    reproduce the single-process sample() bit for bit (global chain ids key the RNG), and Sampler.HMC_NUTS adapts the same
    step size on the all-reduced acceptance statistic (sum order aside)."""
    script = tmp_path / "shard_worker.py"
    script.write_text(_SHARD_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29600 + os.getpid() % 300), str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["equal"] and j["shape"] == [26, 37, 3], j
    assert abs(j["eps"] - j["eps_single"]) <= 1e-6 * j["eps_single"], j
    assert j["nuts_err"] < 1e-3, j


def test_rccl_selftest(ht):
    """tools/rccl_selftest.py with the devices this box shows.  >= 2 GPUs: one rank per GPU over RCCL -
    gather_samples (even / uneven, dst=None / dst=0), sample_sharded (HMC bit for bit, RMHMC to rounding, NUTS step size) and the timed
    cfg5-sized gather.  One GPU (the builder's box): the same checks with two ranks over gloo sharing it, and RCCL as a world-size-1
    group.  Never skipped: the record says which form ran."""
    import torch
    tool = os.path.join(ROOT, "tools", "rccl_selftest.py")
    runs = [[]] if torch.cuda.device_count() >= 2 else [["--world", "2", "--backend", "gloo"], ["--world", "1", "--backend", "nccl"]]
    for extra in runs:
        r = subprocess.run([sys.executable, tool] + extra, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert lines, (r.returncode, r.stderr[-2000:])
        j = json.loads(lines[-1])
        assert j["ok"], j
        assert j["ranks_seen"] == j["world"]
        for k in ("gather_even_all_gather", "gather_even_dst0", "gather_uneven_all_gather", "gather_uneven_dst0",
                  "sample_sharded_hmc_bit_identical", "nuts_step_size_equal", "sample_sharded_rmhmc_equal_to_rounding"):
            assert j["checks"][k] is True, (k, j)
        assert j["gather_cfg5_ms"] > 0
        if torch.cuda.device_count() >= 2:
            assert j["backend"] == "rccl" and j["form"] == "one rank per GPU over RCCL"
        out_dir = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out_dir):
            with open(os.path.join(out_dir, "rccl_selftest_%s_world%d.json" % (j["backend"], j["world"])), "w") as f:
                json.dump(j, f)


def test_reference_cnn_example_runs_on_the_callback_path(ht):
    """The reference's largest model (notebooks/hamiltorch_Bayesian_NN_example.ipynb cells 24-27: two convolutions, two
    linear layers, D = 431 080, softmax likelihood) has no native kernel: sample_model evaluates the functional model for all
    chains with torch.func and the HIP kernels integrate.  Chain c of a 3-chain run equals a 1-chain run with chain_offset = c
    (the Philox streams are keyed by the global chain id, so batching changes nothing), the log-probability the engine
    reports agrees with a direct torch evaluation of S:1145-1199, and the samples move."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bnn_cnn", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                         "examples", "bnn_cnn.py"))
    ex = importlib.util.module_from_spec(spec); spec.loader.exec_module(ex)
    torch.manual_seed(0)
    net = ex.Net().to(dev())
    g = torch.Generator().manual_seed(1)
    x, y = ex.digits(20, g)
    x, y = x.to(dev()), y.to(dev())
    D = sum(p.numel() for p in net.parameters())
    assert D == 431080
    th0 = ht.util.flatten(net).detach()
    th3 = (th0[None] + 0.01 * torch.randn(3, D, generator=g).to(dev())).contiguous()
    tau_list = torch.full((8,), 10.0, device=dev())
    kw = dict(model_loss="multi_class_linear_output", num_samples=3, num_steps_per_sample=4, step_size=0.001, tau_out=1.0,
              tau_list=tau_list, verbose=False, seed=9)
    out3 = torch.stack(list(ht.sample_model(net, x, y, th3, **kw)))
    assert out3.shape == (3, 3, D) and torch.isfinite(out3).all()
    for c in (0, 2):
        out1 = torch.stack(list(ht.sample_model(net, x, y, th3[c:c + 1].clone(), chain_offset=c, **kw)))
        err = (out1[:, 0] - out3[:, c]).abs().max()
        assert float(err) <= 2e-5, float(err)
    assert float((out3[-1] - out3[0]).abs().max()) > 1e-4
    # the closure's value at the initial point against plain torch
    sizes = [p.numel() for p in net.parameters()]; shapes = [p.shape for p in net.parameters()]
    f = ht.samplers.define_model_log_prob(net, "multi_class_linear_output", x, y, sizes, shapes, tau_list, 1.0, device=dev())
    lp = float(f(th3[1]))
    with torch.no_grad():
        params = ht.util.unflatten(net, th3[1])
        names = [n for n, _ in net.named_parameters()]
        out = torch.func.functional_call(net, dict(zip(names, params)), (x,))
        want = -torch.nn.functional.cross_entropy(out, y.long().flatten(), reduction="sum")
        want = want + sum(-0.5 * 10.0 * (p ** 2).sum() + 0.5 * p.numel() * (np.log(10.0) - np.log(2 * np.pi)) for p in params)
    assert abs(lp - float(want)) <= 2e-3 * abs(float(want)), (lp, float(want))
