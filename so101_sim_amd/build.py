"""In-tree build of libso101_hip.so for gfx950 (hipcc cross-compiles without a GPU).

The kernels are spread over several translation units (csrc/tu_*.hip, one or two heavy kernels each) that are
compiled in parallel and linked into one shared object; a kernel is launched from the file it is compiled in
(csrc/so101_launch.hpp), so no relocatable device code is needed.
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libso101_hip.so")
LIB_CLOCKS = os.path.join(CSRC, "libso101_hip_clocks.so")     # -DSO101_DEBUG_CLOCKS profiling build: its own file (SO101_HIP_LIB selects it)
LIB_EPA = os.path.join(CSRC, "libso101_hip_epa.so")           # -DSO101_EPA: MPR portals expanded to the nearest face by EPA (so101_device.hpp); selected by SO101_HIP_LIB too
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-hip-fp32-correctly-rounded-divide-sqrt: v_rcp/v_sqrt based fp32 division and sqrt (<= ~2.5 ulp) instead of
# the 10-15 instruction IEEE expansions; the solver is latency-bound and full of both (profiles/README.md).
# Parity tolerances in tests/parity_cases.py are stated for this build.
# -ffp-contract=on: a*b+c is fused where the SOURCE writes it in one expression (frontend fmuladd), never across
# statements by the backend.  With the default (fast) the backend picks which product of a*b + c*d to fuse from the
# surrounding code, so the same device function rounds differently in two kernels and the bit-identity tests
# (pipelined vs fused step, prefetch on/off) hold only by luck.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=on"]


def translation_units():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def sources():
    root = os.path.dirname(_HERE)
    return translation_units() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(root, "include", "so101.h")]


def source_hash(clocks: bool = False) -> str:
    """Hash of every source AND the compiler flags the library is built from (bench.py keys PMC traffic files by it; the
    settled-state cache and the bench line carry it)."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + (["-DSO101_DEBUG_CLOCKS"] if clocks else [])).encode())
    for p in sources():
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def needs_build(clocks: bool = False, epa: bool = False) -> bool:
    lib = LIB_EPA if epa else (LIB_CLOCKS if clocks else LIB)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False, clocks: bool = False, epa: bool = False) -> str:
    lib = LIB_EPA if epa else (LIB_CLOCKS if clocks else LIB)
    if not (force or needs_build(clocks, epa)):
        return lib
    os.makedirs(OBJ, exist_ok=True)
    flags = list(FLAGS)
    if clocks:
        flags.append("-DSO101_DEBUG_CLOCKS")      # stage clocks + SO101_DEBUG_* env vars for scripts/gpu_*.py
    if epa:
        flags.append("-DSO101_EPA")
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    newest_header = max(os.path.getmtime(s) for s in sources() if not s.endswith(".hip"))

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + (".clk" if clocks else "") + (".epa" if epa else "") + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        subprocess.check_call([HIPCC, *flags, "-c", "-o", obj, src])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, translation_units()))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, clocks="--clocks" in sys.argv, epa="--epa" in sys.argv))
