#!/usr/bin/env python3
"""gdpso_conv_probe on one shape: python tools/probe_one.py n hin cin cout k stride pad tile splits res [reps]"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("gan-discovery-pso_amd._lib")
a = [int(x) for x in sys.argv[1:]]
n, hin, cin, cout, k, stride, pad, tile, splits, res = a[:10]
reps = a[10] if len(a) > 10 else 20
us, tu, su = C.c_double(), C.c_int(), C.c_int()
lib.check(lib.lib().gdpso_conv_probe(n, hin, cin, cout, k, stride, pad, tile, splits, res, reps, C.byref(us), C.byref(tu), C.byref(su)))
hout = (hin + 2 * pad - k) // stride + 1
fl = 2.0 * n * hout * hout * cout * cin * k * k
print(f"M={n*hout*hout} K={cin*k*k} N={cout} tile {tu.value} splits {su.value}: {us.value:.1f} us  {fl/us.value/1e6:.1f} TF")
