#!/usr/bin/env python3
"""bench.py on another build of the library (A/B measurements; single GPU):
    python tools/bench_with_lib.py tools/libgdpso_<name>.so [bench.py arguments]"""
import importlib
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("gan-discovery-pso_amd._lib")
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
