"""Builds libgdpso.so (HIP, gfx950) in-tree with hipcc.  Run: python -m build or call build()."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgdpso.so")
SOURCES = ["pack.cpp", "conv_gemm.hip", "nets.hip", "stage_resident.hip", "pso.hip", "vq.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _flags_changed(stamp: str, key: str) -> bool:
    """True when the objects on disk were built with other flags (or another compiler) than `key`."""
    try:
        with open(stamp) as f:
            return f.read() != key
    except OSError:
        return True


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "gdpso.h")]
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    stamp = os.path.join(CSRC, "build", "flags.txt")
    key = " ".join([hipcc] + FLAGS)
    if _flags_changed(stamp, key):
        force = True
    for src in SOURCES:
        obj = os.path.join(CSRC, "build", os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        spath = os.path.join(CSRC, src)
        if force or _stale(obj, [spath] + headers):
            cmd = [hipcc, "-x", "hip", "-c", spath, "-o", obj] + FLAGS
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(key)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
