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


def rhat_split(x: torch.Tensor) -> float:
    """x[S, C] -> split-R-hat (Gelman et al. 2013: each chain cut in halves, between- over within-half variance); 1 = the halves of all
    chains agree.  The measurement's guard on its own ESS numbers: chains that have not left their starting points give R-hat >> 1,
    and an ESS computed from them measures the spread of the starts, not mixing."""
    x = x.to(torch.float64)
    S, C = x.shape
    h = S // 2
    if h < 2:
        return float("nan")
    y = torch.cat([x[:h], x[S - h:]], dim=1)                      # [h, 2 C]
    w = y.var(dim=0, unbiased=True).mean()
    b = h * y.mean(dim=0).var(unbiased=True)
    if float(w) <= 0.0:
        return float("inf") if float(b) > 0.0 else float("nan")
    return float(torch.sqrt(((h - 1.0) / h * w + b / h) / w))


def rhat_max(samples: torch.Tensor) -> float:
    """samples[S, C, D] -> max over dimensions of rhat_split (NaN dimensions ignored)."""
    vals = [rhat_split(samples[:, :, d]) for d in range(samples.shape[2])]
    vals = [v for v in vals if v == v]
    return max(vals) if vals else float("nan")
