"""Dispatch keys of the reference API (hamiltorch/samplers.py:11-31), same names and values."""
from enum import Enum


class Sampler(Enum):
    HMC = 1
    RMHMC = 2
    HMC_NUTS = 3


class Integrator(Enum):
    EXPLICIT = 1
    IMPLICIT = 2
    S3 = 3
    SPLITTING = 4
    SPLITTING_RAND = 5
    SPLITTING_KMID = 6


class Metric(Enum):
    HESSIAN = 1
    SOFTABS = 2
    JACOBIAN_DIAG = 3
