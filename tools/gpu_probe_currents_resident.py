"""perf probe (not a test): fleet current estimation, 4096 vessels x 600 samples -- the uploaded-history entry point
(138 MB of pageable host memory per call) against the device-resident rings (one 56-byte sample per vessel and tick)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import mpc_amd as M
from mpc_amd.binding import HistorySample

n, ticks = 4096, 620
s = M.Solver(0)
rng = np.random.RandomState(1)
x = rng.uniform(-50, 50, n); y = rng.uniform(-50, 50, n); hd = rng.uniform(0, 6.28, n); v = rng.uniform(1.0, 2.5, n)
flat = (HistorySample * (n * 600))()
s.history_reset(n)
tick_s = 0.0
for k in range(ticks):
    arr = (HistorySample * n)()
    for p in range(n):
        arr[p] = HistorySample(x[p] + 0.2 * k * np.sin(hd[p]), y[p] + 0.2 * k * np.cos(hd[p]), hd[p], v[p], 100.0 + 0.1 * k, 0.1, 0.7)
        if k >= 20:
            flat[p * 600 + (k - 20)] = arr[p]
    t0 = time.perf_counter()
    s.history_update(arr)
    s.synchronize()
    tick_s += time.perf_counter() - t0
counts = (np.ones(n, dtype=np.uint32) * 600)
actual = [(x[p] + 0.2 * ticks * np.sin(hd[p]) + 3.0, y[p] + 0.2 * ticks * np.cos(hd[p]) - 2.0, 100.0 + 0.1 * ticks) for p in range(n)]
import ctypes as C
act = (C.c_double * (3 * n))(*[float(q) for a in actual for q in a])
out = (C.c_double * (2 * n))()
L = M.load_library()
for name, call in (("uploaded histories (mpc_hip_estimate_currents)", lambda: L.mpc_hip_estimate_currents(s._h, flat, counts.ctypes.data_as(C.POINTER(C.c_uint32)), 600, act, n, 0, out)),
                   ("device-resident rings (mpc_hip_estimate_currents_resident)", lambda: L.mpc_hip_estimate_currents_resident(s._h, act, n, 0, out))):
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        assert call() == 0
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%s: %.2f ms per call (best of 5: %.2f), first estimate (%.6f, %.6f)" % (name, np.median(ts), min(ts), out[0], out[1]), flush=True)
print("history_update: %.3f ms per tick (4096 samples up + one kernel, synchronised)" % (1e3 * tick_s / ticks))
