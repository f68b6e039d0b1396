"""Fleet current estimation (mpc_hip_estimate_currents) timed against the CPU oracle's replay, config-5 sized:
4096 vessels x 600-sample histories (60 s at 10 Hz).  Usage: python tools/gpu_probe_currents.py [n_vessels]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: before the library)

import mpc_amd as M
import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
stride = 600
rng = np.random.RandomState(7)
# one simulated track, shifted per vessel: the replay's cost does not depend on where the vessel is
v = O.vehicle(0.0, 0.0, 1.0, 2.0, 100.0)
track = []
r, t = 0.0, 0.5
for i in range(stride):
    if i % 7 == 0:
        r, t = float(rng.randint(-16, 17)) * 0.0625, float(rng.randint(0, 9)) * 0.125
    track.append((v.s.x, v.s.y, v.s.heading, v.s.speed, v.s.time, r, t))
    v = O.simulate(v, r, t, 0.1, 0.3, -0.2)
L = M.load_library()
solver = M.Solver(0)
flat = (M.HistorySample * (n * stride))()
base = np.array(track)
arr = np.frombuffer(flat, dtype=np.float64).reshape(n, stride, 7)
off = rng.uniform(-500, 500, size=(n, 2))
arr[:] = base[None]
arr[:, :, 0] += off[:, :1]; arr[:, :, 1] += off[:, 1:]
counts = (C.c_uint32 * n)(*([stride] * n))
act = (C.c_double * (3 * n))()
a = np.frombuffer(act, dtype=np.float64).reshape(n, 3)
a[:, 0] = v.s.x + off[:, 0]; a[:, 1] = v.s.y + off[:, 1]; a[:, 2] = v.s.time
out = (C.c_double * (2 * n))()
for rep in range(3):
    t0 = time.perf_counter()
    st = L.mpc_hip_estimate_currents(solver._h, flat, counts, stride, act, n, 0, out)
    dt = time.perf_counter() - t0
    assert st == 0
g = np.frombuffer(out, dtype=np.float64).reshape(n, 2)
# CPU oracle on a sample of the vessels
m = min(n, 64)
worst, cpu = 0.0, 0.0
for p in range(m):
    hist = (O.HistoryEntry * stride)(*[O.HistoryEntry(O.State(*arr[p, i, :5]), arr[p, i, 5], arr[p, i, 6]) for i in range(stride)])
    cx, cy = C.c_double(), C.c_double()
    actual = O.State(a[p, 0], a[p, 1], 0.0, 0.0, a[p, 2])
    t0 = time.perf_counter()
    O.lib().orc_current_estimate(hist, stride, 1, actual, C.byref(cx), C.byref(cy))       # the C restatement alone
    cpu += time.perf_counter() - t0
    worst = max(worst, abs(cx.value - g[p, 0]), abs(cy.value - g[p, 1]))
cpu /= m
print("vessels=%d samples=%d  gpu call (H2D of %.0f MB + kernel + D2H) %.2f ms  | cpu oracle %.3f ms per vessel = %.0f ms for the fleet"
      "  | max |gpu - oracle| over %d vessels %.2e" % (n, stride, n * stride * 56 / 1e6, dt * 1e3, cpu * 1e3, cpu * n * 1e3, m, worst))
