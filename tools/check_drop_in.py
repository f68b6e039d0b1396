#!/usr/bin/env python3
"""Compile check of the drop-in against the REFERENCE'S OWN headers (CPU only, no device, nothing is run).

INTEGRATION.md section 1 says: copy mpc_amd/host/controller_mpc_hip.cpp into the reference's src/, next to its
controller.h, and put include/ on the include path.  This script does exactly that in a scratch directory --
controller_mpc_hip.cpp beside COPIES of the reference's src/controller.h, control_receiver.h, VehicleState.h and
CurrentEstimator.h (made at run time, never kept in this repository) -- and compiles the file with the reference's own
options (CMakeLists.txt:4-6) under -Wall -Wextra -Werror.  `"controller.h"` then resolves to the reference's header,
not to this repository's mirror of it, so the check fails if the drop-in uses a member the reference's class does not
declare.  The un-vendored catkin headers those files include (path_planner_common/State.h, DubinsPlan.h, dubins.h) come
from mpc_amd/host/compat, as a maintainer's catkin workspace would provide the real ones.

    python tools/check_drop_in.py [reference root]      # default: $MPC_REFERENCE_ROOT, else /root/reference

Exit status 0 = compiled clean, 77 = the reference tree is not there (skipped), anything else = the compiler's.
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_FLAGS = ["-std=c++11", "-pthread", "-funsafe-math-optimizations"]
REFERENCE_HEADERS = ("controller.h", "control_receiver.h", "VehicleState.h", "CurrentEstimator.h")
SKIPPED = 77


def main(argv):
    ref = argv[1] if len(argv) > 1 else os.environ.get("MPC_REFERENCE_ROOT", "/root/reference")
    src = os.path.join(ref, "src")
    if not all(os.path.exists(os.path.join(src, f)) for f in REFERENCE_HEADERS):
        print("check_drop_in: no reference tree at %s: skipped" % ref)
        return SKIPPED
    host = os.path.join(ROOT, "mpc_amd", "host")
    with tempfile.TemporaryDirectory(prefix="drop_in_") as tmp:
        for f in REFERENCE_HEADERS:
            shutil.copy(os.path.join(src, f), tmp)
        shutil.copy(os.path.join(host, "controller_mpc_hip.cpp"), tmp)
        cmd = (["g++"] + REFERENCE_FLAGS + ["-Wall", "-Wextra", "-Werror", "-c", "controller_mpc_hip.cpp", "-o", "drop_in.o",
                                            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(host, "compat")])
        p = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode:
            print(p.stdout)
            return p.returncode
        # the object must define Controller::mpc and nothing else of Controller (the rest stays the reference's)
        nm = subprocess.run(["nm", "-C", "--defined-only", "drop_in.o"], cwd=tmp, stdout=subprocess.PIPE, text=True).stdout
        defined = sorted({ln.split(" ", 2)[2] for ln in nm.splitlines() if " T " in ln and "Controller::" in ln})
        if defined != ["Controller::mpc(State, DubinsPlan const&, double, long)"]:
            print("check_drop_in: unexpected Controller symbols: %s" % defined)
            return 1
    print("check_drop_in: controller_mpc_hip.cpp compiles against %s/src/controller.h with %s -Wall -Wextra -Werror; "
          "defines Controller::mpc only" % (ref, " ".join(REFERENCE_FLAGS)))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
