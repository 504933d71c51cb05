"""In-tree build of libso101_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "libso101_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def sources():
    root = os.path.dirname(_HERE)
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
                  + [os.path.join(root, "include", "so101.h")])


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        # -fno-hip-fp32-correctly-rounded-divide-sqrt: v_rcp/v_sqrt based fp32 division and sqrt (<= ~2.5 ulp)
        # instead of the 10-15 instruction IEEE expansions; the solver is VALU-issue bound and full of both
        # (profiles/README.md).  Parity tolerances in tests/parity_cases.py are stated for this build.
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fno-hip-fp32-correctly-rounded-divide-sqrt",
               "-o", LIB, os.path.join(CSRC, "so101_hip.hip")]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
