"""Developer check: aggregate rate of the cfg2 CPU baseline vs number of single-threaded processes on this host."""
import os, sys, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="")
for procs in (1, 4, 8, 16, 32, 64):
    t0 = time.time()
    ps = [subprocess.Popen([sys.executable, ROOT + "/oracle/torch_port.py", "cfg2", str(1000 + i), "25", "0.3", "4.0"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(procs)]
    res = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in ps]
    rate = sum(r["n"] * r["L"] for r in res) / max(r["dt"] for r in res)
    print("procs %3d: %.0f steps/s aggregate, %.0f per process, wall %.1f s" % (procs, rate, rate / procs, time.time() - t0))
