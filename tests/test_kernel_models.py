"""CPU: numpy models of the data movement of two kernels, checking the invariants their layouts rely on (the kernels
themselves are compared with the oracle in the -m gpu tests)."""
import numpy as np
import pytest


def wave_cholesky_model(P, ev, z, NB):
    """rmhmc_momentum_wave_kernel (csrc/rmhmc_fused.hip) in numpy, lane for lane: lane (ty, tx) owns the elements
    (ty + 8a, tx + 8b), b <= a; rows / columns >= D are CLAMPED copies of P; panels of 4 columns pass through `pan` / `pan2`;
    p = L z is accumulated while the panels finish."""
    D = P.shape[0]
    NR = 8 * NB
    W = np.zeros((8, 8, NB, NB), np.float64)                # [ty][tx][a][b]
    for ty in range(8):
        for tx in range(8):
            for a in range(NB):
                for b in range(a + 1):
                    W[ty, tx, a, b] = P[min(ty + 8 * a, D - 1), min(tx + 8 * b, D - 1)]
            if ty == tx:
                for a in range(NB):
                    i = ty + 8 * a
                    W[ty, tx, a, a] += ev[i] if i < D else 0.0
    pan = np.zeros((NR, 4)); pan2 = np.zeros((NR, 4)); pacc = np.zeros(NR)
    zz = np.zeros(NR); zz[:D] = z
    for bp in range(NB):
        if 8 * bp >= D:
            break
        for half in range(2):
            kb = 8 * bp + 4 * half
            if kb >= D:
                continue
            for ty in range(8):
                for tx in range(8):
                    if (tx >> 2) == half:
                        for a in range(bp, NB):
                            pan[ty + 8 * a, tx & 3] = W[ty, tx, a, bp]
            Ld = np.eye(4)
            for c in range(4):
                if kb + c < D:
                    Ld[c, :c + 1] = pan[kb + c, :c + 1]
            rinv = np.zeros(4)
            for c in range(4):
                for c2 in range(c):
                    Ld[c, c2] = (Ld[c, c2] - Ld[c, :c2] @ Ld[c2, :c2]) * rinv[c2]
                v = Ld[c, c] - Ld[c, :c] @ Ld[c, :c]
                rinv[c] = 1.0 / np.sqrt(v)
                Ld[c, c] = v * rinv[c]
            for r in range(NR):
                out = np.zeros(4)
                if r >= kb + 4:
                    for c in range(4):
                        out[c] = (pan[r, c] - out[:c] @ Ld[c, :c]) * rinv[c]
                        pacc[r] += out[c] * zz[kb + c]
                elif r >= kb:
                    c = r - kb
                    pacc[r] += Ld[c, :c + 1] @ zz[kb:kb + c + 1]
                if r >= kb:
                    pan2[r] = out
            for ty in range(8):
                for tx in range(8):
                    for a in range(bp, NB):
                        for b in range(min(a, NB - 1) + 1):
                            W[ty, tx, a, b] -= pan2[ty + 8 * a] @ pan2[tx + 8 * b]
    return pacc[:D]


@pytest.mark.parametrize("D,NB", [(5, 4), (8, 4), (13, 4), (30, 4), (33, 7), (50, 7)])
def test_wave_cholesky_layout_gives_chol_times_z(D, NB):
    """The clamped padding never reaches the leading D x D block (a Cholesky factor's leading block depends on nothing
    beyond it), zeros in `pan2` make the updates of finished columns no-ops, and the panel-order accumulation is L z."""
    rng = np.random.default_rng(D)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    P = (Q * rng.uniform(0.5, 2.0, D)) @ Q.T
    P = 0.5 * (P + P.T)
    ev = 1e-3 * rng.uniform(size=D)
    z = rng.standard_normal(D)
    got = wave_cholesky_model(P, ev, z, NB)
    want = np.linalg.cholesky(P + np.diag(ev)) @ z
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("D", [7, 64, 65, 100])
def test_mfma4_operand_layout(D):
    """rmhmc_mfma4_kernel: v_mfma_f32_4x4x1_16b (16 blocks of 4 x 4, K = 1) with lane l = 4 * block + i supplying A[64 w + l][k]
    and lane 4 * block + n supplying X[n][k] for every block; D[block][i][n] lands in lane 4 * block + n, register i.  Two waves
    then give S X for 128 padded rows x 4 chains, and lane (block, n) owns rows 64 w + 4 block .. + 3 of chain n."""
    rng = np.random.default_rng(D)
    S = rng.standard_normal((D, D)); S = S + S.T
    X = rng.standard_normal((4, D))                         # [chain][k]
    out = np.zeros((2, 64, 4))                              # [wave][lane][register]
    for w in range(2):
        A = np.zeros((64, D))                               # lane l: column arow of the symmetric S = row arow
        for l in range(64):
            arow = 64 * w + l
            if arow < D:
                A[l] = S[:, arow]
        for k in range(D):
            for blk in range(16):
                for n in range(4):
                    for i in range(4):
                        out[w, 4 * blk + n, i] += A[4 * blk + i, k] * X[n, k]
    want = S @ X.T                                          # [row][chain]
    for w in range(2):
        for l in range(64):
            blk, n = l >> 2, l & 3
            for e in range(4):
                row = 64 * w + 4 * blk + e
                assert abs(out[w, l, e] - (want[row, n] if row < D else 0.0)) < 1e-9
