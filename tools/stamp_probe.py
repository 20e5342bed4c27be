#!/usr/bin/env python3
"""Where a convolution launch spends its time, from in-kernel time stamps (diagnostic build).

    python tools/stamp_probe.py build                      # here (needs hipcc): builds tools/libgdpso_stamps.so
                                                           # (GDPSO_SKIP_BITS=n in the environment compiles parts of the K loop out)
    python tools/stamp_probe.py n hin cin cout k stride pad tile splits res    # on the GPU

Every workgroup stamps s_memtime at entry, after its prologue (first K-tile in LDS), after its
K loop and after its stores have left the CU, plus s_memrealtime (100 MHz) at entry and exit.
Prints the spread of workgroup start times and the median / max of each phase in microseconds.
"""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.environ.get("GDPSO_STAMPS_LIB", os.path.join(HERE, "libgdpso_stamps.so"))  # (another diagnostic build of the library)
CSRC = os.path.join(ROOT, "gan-discovery-pso_amd", "csrc")


def build():
    srcs = ["pack.cpp", "conv_gemm.hip", "nets.hip", "stage_resident.hip", "pso.hip", "vq.hip"]
    cmd = (["/opt/rocm/bin/hipcc", "-shared", "-o", LIB, "-x", "hip"] + [os.path.join(CSRC, s) for s in srcs] +
           ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DGDPSO_STAMPS"])
    if os.environ.get("GDPSO_SKIP_BITS"):  # 1: no MFMAs, 2: no global loads, 4: no LDS stores in the K loop (sum them)
        cmd.append("-DGDPSO_SKIP_BITS=" + os.environ["GDPSO_SKIP_BITS"])
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def main():
    if sys.argv[1] == "build":
        return build()
    import numpy as np
    import torch
    a = [int(x) for x in sys.argv[1:11]]
    lib = C.CDLL(LIB, mode=C.RTLD_GLOBAL)
    lib.gdpso_last_error.restype = C.c_char_p
    lib.gdpso_conv_probe.argtypes = [C.c_int] * 11 + [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.gdpso_conv_stamps_set.argtypes = [C.c_void_p]
    buf = torch.zeros(1 << 16, 16, dtype=torch.int64, device="cuda")
    assert lib.gdpso_conv_stamps_set(buf.data_ptr()) == 0, lib.gdpso_last_error()
    us, tile, splits = C.c_double(), C.c_int(), C.c_int()
    rc = lib.gdpso_conv_probe(*a, 5, C.byref(us), C.byref(tile), C.byref(splits))
    assert rc == 0, lib.gdpso_last_error()
    torch.cuda.synchronize()
    s = buf.cpu().numpy().astype(np.int64)
    live = s[:, 0] != 0
    s = s[live]
    n = s.shape[0]
    ghz = float(np.median((s[:, 3] - s[:, 0]) / np.maximum(1, (s[:, 6] - s[:, 7]) * 10.0)))
    start = (s[:, 7] - s[:, 7].min()) * 0.01
    end = (s[:, 6] - s[:, 7].min()) * 0.01
    ph = [(s[:, i + 1] - s[:, i]) / (ghz * 1e3) for i in range(3)]
    print(f"launch {us.value:.1f} us (tile {tile.value}, splits {splits.value}); {n} workgroups stamped; in-kernel clock ~{ghz:.2f} GHz")
    print(f"workgroup start: median {np.median(start):.2f} us, last {start.max():.2f} us; last exit {end.max():.2f} us after the first entry")
    for name, v in zip(("prologue (to first K-tile in LDS)", "K loop", "epilogue (to stores drained)"), ph):
        print(f"  {name:36s} median {np.median(v):6.2f} us   p90 {np.percentile(v, 90):6.2f}   max {v.max():6.2f}")
    if (s[:, 8] != 0).all():
        for name, a, b in (("  entry -> phase header in SGPRs", 0, 8), ("  -> first K-tile loads issued", 8, 9),
                           ("  -> loads returned", 9, 10), ("  -> LDS stores + barrier", 10, 1)):
            v = (s[:, b] - s[:, a]) / (ghz * 1e3)
            print(f"  {name:36s} median {np.median(v):6.2f} us   p90 {np.percentile(v, 90):6.2f}   max {v.max():6.2f}")
    life = (s[:, 3] - s[:, 0]) / (ghz * 1e3)
    print(f"  workgroup lifetime                   median {np.median(life):6.2f} us   max {life.max():6.2f}")
    # placement: HW_REG_HW_ID (cu_id bits 11:8, sh_id 12, se_id 15:13) and HW_REG_XCC_ID per workgroup
    full = buf.cpu().numpy().astype(np.int64)
    ids = np.nonzero(full[:, 0] != 0)[0]
    hw = full[ids, 5] & 0xffffffff
    xcc = (full[ids, 5] >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
    print("  XCC of workgroups 0..15:", xcc[:16].tolist())
    for x in sorted(set(xcc.tolist()))[:1]:
        sel = ids[xcc == x]
        cus = cu[xcc == x]
        print(f"  XCC {x}: {len(sel)} workgroups on {len(set(cus.tolist()))} CUs; workgroup ids -> CU (first 48):")
        print("   ", [(int(i), int(c)) for i, c in list(zip(sel, cus))[:48]])
        per = {}
        for i, c in zip(sel, cus):
            per.setdefault(int(c), []).append(int(i) >> 3)
        print("    slots per CU:", sorted(per.items())[:12])
    second = start > np.median(life) * 0.5
    print(f"  workgroups that started after half a median lifetime (a later round): {int(second.sum())}")


if __name__ == "__main__":
    main()
