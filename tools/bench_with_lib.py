#!/usr/bin/env python
"""python tools/bench_with_lib.py <library.so> <bench.py arguments ...>: bench.py on a developer build of the library (A/B runs)."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hamiltorch_amd import _abi  # noqa: E402

_abi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
