"""Developer check (CPU, numpy): what the bfloat16 formation of the metric kernel's fast solve ("metric_bx3" = 2) costs in accuracy.  The solve
x = G^-1 m through first-order + closed-form second-order refinement of the shared basis, with F = V0^T diag(e) V0 taken (a) in fp32, (b) as three
products of W = diag(sqrt e) V0 split into two bfloat16 terms, (c) exactly; everything around F in float64, so that only F's precision and the
method's own truncation show: max |x - x64| / max |x64| over 20 systems as (max, mean).   python tools/scratch/bf16_formation_err.py"""
import sys, os
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import hmc_oracle as O
def trunc_bf16(x):
    b = x.astype(np.float32).view(np.uint32) & np.uint32(0xffff0000)
    return b.view(np.float32)
def split2(x):
    hi = trunc_bf16(x); lo = trunc_bf16((x - hi).astype(np.float32)); return hi, lo
def run(D, alpha, jitter, seed=3):
    rng = np.random.default_rng(seed)
    # cfg3-like target: spectrum 0.5..2 random orthogonal
    Q,_ = np.linalg.qr(rng.standard_normal((D,D)))
    lam_true = np.linspace(0.5, 2.0, D)
    P = (Q*lam_true)@Q.T; P = 0.5*(P+P.T)
    lam0, V0 = np.linalg.eigh(P)
    V0f = V0.astype(np.float32); lam0f = lam0.astype(np.float32)
    B = 20
    errs = {k: [] for k in ("fp32","bx3","f64")}
    for b in range(B):
        e = (jitter*rng.uniform(size=D))
        m = rng.standard_normal(D)
        G64 = None
        A64 = P + np.diag(e)
        l64, Q64 = np.linalg.eigh(A64)
        lt64 = l64/np.tanh(alpha*l64)
        x64 = Q64 @ ((Q64.T@m)/lt64)
        ef = e.astype(np.float32)
        F_exact = (V0f.astype(np.float64).T * ef.astype(np.float64)) @ V0f.astype(np.float64)
        F32 = ((V0f.T * ef) @ V0f).astype(np.float32)
        W = (V0f * np.sqrt(ef)[:,None]).astype(np.float32)          # W[k][i] = sqrt(e_k) V0[k][i]
        Wh, Wl = split2(W)
        Fbx = (Wh.T.astype(np.float64)@Wh.astype(np.float64) + Wh.T.astype(np.float64)@Wl.astype(np.float64) + Wl.T.astype(np.float64)@Wh.astype(np.float64)).astype(np.float32)
        for name, F in (("fp32",F32),("bx3",Fbx),("f64",F_exact)):
            F = F.astype(np.float64)
            lam = lam0f.astype(np.float64) + np.diag(F)
            Foff = F - np.diag(np.diag(F))
            gap = lam[None,:]-lam[:,None]; np.fill_diagonal(gap, 1.0)
            E1 = Foff/gap; np.fill_diagonal(E1, 0.0)
            M = Foff@E1
            lam2 = lam + np.diag(M)
            gap2 = lam2[None,:]-lam2[:,None]; np.fill_diagonal(gap2, 1.0)
            E2 = M/gap2; np.fill_diagonal(E2, -0.5*(E1**2).sum(1))
            X = (np.eye(D)+E1)@(np.eye(D)+E2)
            Qa = V0f.astype(np.float64)@X
            lt = lam2/np.tanh(alpha*lam2)
            x = Qa@((Qa.T@m)/lt)
            errs[name].append(np.abs(x-x64).max()/np.abs(x64).max())
    return {k: (np.max(v), np.mean(v)) for k,v in errs.items()}
for D, alpha, jit in [(100,1e6,1e-3),(100,1.3,1e-3),(100,1e6,3e-3),(64,2.0,5e-4)]:
    print(D, alpha, jit, run(D, alpha, jit))
