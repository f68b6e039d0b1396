"""Writes tests/golden/*.json.  Two kinds of vectors:

survey_probe.json  -- outputs of the REFERENCE ITSELF: the survey session compiled src/controller.cpp,
    src/CurrentEstimator.cpp and src/VehicleState.h unmodified (g++ 11.4 -O2 -funsafe-math-optimizations, the
    flags of CMakeLists.txt:4-6) and recorded these numbers in SURVEY.md sections 4, 7 and 8c.  They are typed in
    here from that document, not recomputed; the oracle is checked AGAINST them (tests/test_oracle_golden.py).
candidates.json    -- known answers of the seeded-candidate generator include/mpc_candidates.h, produced by
    compiling that header with gcc (through oracle/libmpc_oracle.so); pins the numpy and HIP restatements.

Run from the repository root:  python tests/golden/make_goldens.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

survey = {
    "source": "SURVEY.md sections 4, 7 (hard part 1) and 8c: probe of the reference sources compiled unmodified",
    "compiler": "g++ 11.4.0 / ROCm clang 22, -O0 / -O2 / -O2 -funsafe-math-optimizations, x86-64 glibc 2.35 (bit-identical)",
    "simulate": {
        "start": [10.0, -5.0, 1.25, 2.0, 100.0], "rudder": 0.5, "throttle": 0.75, "t": 1.0, "current": [0.3, -0.2],
        "x": 12.361913597579116, "y": -4.7866410095042875, "heading": 1.471070251202069,
        "speed": 2.1681194741262995, "rpm": 2400.0, "cmg": 1.5643423697798238, "sog": 2.4573982973073774,
        "time": 100.99999999999994
    },
    "schedule": {
        "1.0": {"count": 11, "last": 1.3877787807814457e-16},
        "2.0": {"count": 20, "last": 0.09999999999999937},
        "5.0": {"count": 51}
    },
    "tree": {
        "plan": {"type": "LRL", "qi": [0.0, 0.0, 1.5707963267948966], "param": [0.0, 3.141592653589793, 0.0],
                 "rho": 8.0, "speed": 2.5, "start_time": 4.0},
        "start": [0.0, 0.0, 0.0, 2.5, 4.0], "current": [0.2, -0.1], "config": "cfg/mpc.cfg defaults",
        "note": "Score values were printed by the -funsafe-math-optimizations build, whose compareStates sum is "
                "reassociated as w_d*d + (w_h*hd + w_s*sd); the oracle keeps source order, hence score_ulps.",
        "cases": [
            {"plan_seconds": 3.5, "levels": 2, "rudder": 0.0625, "throttle": 1.0, "score": 5.8203995935608095},
            {"plan_seconds": 4.5, "levels": 3, "rudder": -0.0625, "throttle": 1.0, "score": 7.1873595784095548,
             "x": 0.1531378173801084, "y": 2.5300638769535997},
            {"plan_seconds": 5.5, "levels": 4, "rudder": -0.0625, "throttle": 1.0, "score": 7.1873595784095548,
             "x": 0.1531378173801084, "y": 2.5300638769535997}
        ],
        "score_ulps": 2
    },
    "current_estimator_test1": {"expected_by_stale_test": [-0.4, 0.0], "reference_returns": [0.0, 0.0]},
    "reference_test_properties": {
        "estimateTest1_max_ulps": 4, "estimateTest2_tolerance": 1e-3,
        "semicircle_end_time": "8*pi/2.5 + 4 (ASSERT_DOUBLE_EQ, test_controller.cpp:467)"
    }
}
json.dump(survey, open(os.path.join(HERE, "survey_probe.json"), "w"), indent=1)

import oracle as O
import ctypes as C
L = O.lib()
cases = []
for seed, ordinal, level in [(0xC0FFEE, 0, 0), (0xC0FFEE, 1, 3), (0xC0FFEE, 2, 0), (0xC0FFEE, 65535, 19), (0xC0FFEE, 16777215, 29),
                             (1, 2, 3), (0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFF, 63), (12345678901234567, 987654321, 7)]:
    r, t = C.c_double(), C.c_double()
    L.orc_seeded_control(seed, ordinal, level, 0.25, 0.5, C.byref(r), C.byref(t))
    cases.append({"seed": seed, "ordinal": ordinal, "level": level, "last": [0.25, 0.5],
                  "hash": L.orc_seeded_hash(seed, ordinal, level), "rudder": r.value, "throttle": t.value})
json.dump({"source": "include/mpc_candidates.h compiled with gcc (oracle/libmpc_oracle.so)", "cases": cases},
          open(os.path.join(HERE, "candidates.json"), "w"), indent=1)
print("wrote", len(cases), "candidate cases")
