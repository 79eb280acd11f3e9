"""Effective sample size of batched chains (FFT autocorrelation + Geyer initial-positive-sequence
truncation, multi-chain variance as in Vehtari et al. 2021 without rank-normalisation).

The reference has no ESS routine; BASELINE.json's metric asks for
ESS/sec, so this is the measurement-side definition, applied identically to device samples and
to the CPU baseline's samples.
"""
from __future__ import annotations

import math

import torch


def ess_bulk(x: torch.Tensor) -> float:
    """x[S, C]: S draws of C chains of one scalar quantity -> effective sample size."""
    x = x.to(torch.float64)
    S, C = x.shape
    xc = x - x.mean(dim=0, keepdim=True)
    n = 1 << int(math.ceil(math.log2(2 * S)))
    f = torch.fft.rfft(xc, n=n, dim=0)
    acov = torch.fft.irfft(f * f.conj(), n=n, dim=0)[:S] / S
    chain_var = acov[0] * S / (S - 1.0)
    mean_var = chain_var.mean()
    var_plus = mean_var * (S - 1.0) / S
    if C > 1:
        var_plus = var_plus + x.mean(dim=0).var(unbiased=True)
    rho = (1.0 - (mean_var - acov.mean(dim=1)) / var_plus).cpu().tolist()
    rho[0] = 1.0
    tau, t = -1.0, 0
    while t + 1 < S:
        pair = rho[t] + rho[t + 1]
        if pair < 0:
            break
        tau += 2.0 * pair
        t += 2
    tau = max(tau, 1.0 / math.log10(S * C + 10))
    return S * C / tau


def ess_min(samples: torch.Tensor) -> float:
    """samples[S, C, D] -> min over dimensions of ess_bulk."""
    return min(ess_bulk(samples[:, :, d]) for d in range(samples.shape[2]))
