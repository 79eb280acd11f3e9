"""The closed-form second refinement pass of csrc/rmhmc_metric_mfma.hip (ph_refine_E2, tuning key "metric_second") restated in
float64 numpy (CPU): after the first pass X1 = I + E1, E1_ij = F_ij / (lam_j - lam_i) (F = A - diag A), second-order perturbation
theory gives

    M = F E1,   lam_i' = lam_i + M_ii,   E2_ij = M_ij / (lam_j' - lam_i'),   E2_ii = -1/2 sum_k E1_ki^2,   X2 = X1 (I + E2)

- ONE product instead of the three (A X1, X1^T A X1, X1^T X1) of the Ogita-Aishima pass it replaces.  These tests pin the MATH at
BASELINE config 3's spectrum (the kernel itself is checked on the GPU: test_second_pass_in_closed_form_equals_the_three_product_pass):
the truncation leaves ||A X2 - X2 Lam'|| ~ |F| d^2 (d = max |E1_ij|), far below fp32 rounding wherever the kernel takes the pass
(d <= 8e-3), and the eigenvalues agree with LAPACK's to third order."""
import numpy as np
import pytest


def cfg3_A(rng, D=100, jitter=1e-3):
    Q = np.linalg.qr(rng.standard_normal((D, D)))[0]
    lam = np.linspace(0.5, 2.0, D)
    P = (Q * lam) @ Q.T
    P = 0.5 * (P + P.T)
    lam0, V0 = np.linalg.eigh(P)
    e = jitter * rng.random(D)
    A = np.diag(lam0) + V0.T @ (e[:, None] * V0)                # what the kernel forms: diag(lam0) + V0^T diag(e) V0 ...
    return 0.5 * (A + A.T)                                      # ... exactly symmetric there (the product mirrors its upper tiles)


def first_pass(A):
    D = A.shape[0]
    lam = np.diag(A).copy()
    F = A - np.diag(lam)
    den = lam[None, :] - lam[:, None]
    np.fill_diagonal(den, 1.0)
    E1 = F / den
    np.fill_diagonal(E1, 0.0)
    return lam, F, E1


def closed_form_pass(lam, F, E1):
    D = len(lam)
    M = F @ E1
    lam2 = lam + np.diag(M)
    den = lam2[None, :] - lam2[:, None]
    np.fill_diagonal(den, 1.0)
    E2 = M / den
    np.fill_diagonal(E2, -0.5 * (E1 ** 2).sum(0))
    return lam2, (np.eye(D) + E1) @ (np.eye(D) + E2)


def three_product_pass(A, E1):
    D = A.shape[0]
    X1 = np.eye(D) + E1
    S, G = X1.T @ A @ X1, X1.T @ X1
    lam3 = np.diag(S) / np.diag(G)
    den = lam3[None, :] - lam3[:, None]
    np.fill_diagonal(den, 1.0)
    E3 = (S - lam3[None, :] * G) / den
    np.fill_diagonal(E3, 0.5 * (1.0 - np.diag(G)))
    return lam3, X1 @ (np.eye(D) + E3)


@pytest.mark.parametrize("jitter", [1e-3, 5e-4])
def test_closed_form_pass_at_cfg3(jitter):
    rng = np.random.default_rng(0)
    worst_d = worst_res = worst_orth = worst_lam = 0.0
    for _ in range(60):
        A = cfg3_A(rng, 100, jitter)
        lam, F, E1 = first_pass(A)
        d = np.abs(E1).max()
        assert np.allclose(E1, -E1.T, rtol=0, atol=0)                    # antisymmetric exactly: X1^T v = 2 v - X1 v in the kernel's solve
        lam2, X2 = closed_form_pass(lam, F, E1)
        res = np.abs(A @ X2 - X2 * lam2[None, :]).max()
        orth = np.abs(X2.T @ X2 - np.eye(100)).max()
        w = np.linalg.eigvalsh(A)
        worst_d, worst_res, worst_orth = max(worst_d, d), max(worst_res, res), max(worst_orth, orth)
        worst_lam = max(worst_lam, np.abs(np.sort(lam2) - w).max())
        assert res <= 40.0 * np.abs(F).max() * d * d + 1e-15            # third order: |F| d^2 (x a modest constant)
    assert worst_d <= 8e-3                                                # every evaluation of cfg3 takes the closed form (kSecondE)
    assert worst_res <= 2e-8 and worst_orth <= 2e-8 and worst_lam <= 2e-8   # all below fp32 rounding (6e-8 x |lam| ~ 1e-7)


def test_closed_form_pass_equals_the_three_product_pass_to_third_order():
    rng = np.random.default_rng(1)
    A = cfg3_A(rng, 100, 1e-3)
    lam, F, E1 = first_pass(A)
    d = np.abs(E1).max()
    lam2, X2 = closed_form_pass(lam, F, E1)
    lam3, X3 = three_product_pass(A, E1)
    assert np.abs(lam2 - lam3).max() <= 10.0 * np.abs(F).max() * d * d
    # eigenvector entries differ by the third-order terms, which mix only nearly degenerate pairs: |dX| <~ d^3 x D^(1/2)
    assert np.abs(X2 - X3).max() <= 20.0 * d ** 3
    # ... and a matrix function - what the kernel computes: G^-1 m, log|G| - does not see that mixing
    m = rng.standard_normal(100)
    f2 = X2 @ ((X2.T @ m) / lam2)
    f3 = X3 @ ((X3.T @ m) / lam3)
    ref = np.linalg.solve(A, m)
    assert np.abs(f2 - ref).max() <= 2e-8 * np.abs(ref).max() and np.abs(f3 - ref).max() <= 2e-8 * np.abs(ref).max()


def test_first_pass_update_beyond_the_bound_needs_the_full_pass():
    """A jitter 30 x cfg3's: first-pass updates of ~0.1 - the truncation error of the closed form is no longer negligible (the
    kernel takes the three-product pass above 8e-3, and the Jacobi fallback above 0.03)."""
    rng = np.random.default_rng(2)
    A = cfg3_A(rng, 100, 3e-2)
    lam, F, E1 = first_pass(A)
    assert np.abs(E1).max() > 8e-3
    lam2, X2 = closed_form_pass(lam, F, E1)
    assert np.abs(A @ X2 - X2 * lam2[None, :]).max() > 1e-6
