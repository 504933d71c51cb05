"""In-tree build of libso101_hip.so for gfx950 (hipcc cross-compiles without a GPU).

The kernels are spread over several translation units (csrc/tu_*.hip, one or two heavy kernels each) that are
compiled in parallel and linked into one shared object; a kernel is launched from the file it is compiled in
(csrc/so101_launch.hpp), so no relocatable device code is needed.
"""
from __future__ import annotations

import glob
import hashlib
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libso101_hip.so")
LIB_CLOCKS = os.path.join(CSRC, "libso101_hip_clocks.so")     # -DSO101_DEBUG_CLOCKS profiling build: its own file (SO101_HIP_LIB selects it)
LIB_EXP = os.path.join(CSRC, "libso101_hip_exp.so")           # -DSO101_EXPERIMENTAL_PIPELINES: the default library plus the two step paths that were built, proven bit-identical and measured
                                                              # slower than the launch chains (pipeline = 2 per-env chaining, 3 merged launches); built on demand
LIB_MPR = os.path.join(CSRC, "libso101_hip_mpr.so")           # -DSO101_MPR: the narrowphase="mpr" option - MPR's own portal depth instead of the EPA expansion to the nearest face that the
                                                              # default library runs (so101_device.hpp); built on demand, not by __graft_entry__.build()
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


def variant_flags(clocks: bool = False, mpr: bool = False, exp: bool = False):
    return (["-DSO101_DEBUG_CLOCKS"] if clocks else []) + (["-DSO101_MPR"] if mpr else []) + (["-DSO101_EXPERIMENTAL_PIPELINES"] if exp else [])


def lib_path(clocks: bool = False, mpr: bool = False, exp: bool = False) -> str:
    if exp:
        return LIB_EXP if not (clocks or mpr) else os.path.join(CSRC, "libso101_hip_exp" + ("_mpr" if mpr else "") + ("_clocks" if clocks else "") + ".so")
    if clocks and mpr:
        return os.path.join(CSRC, "libso101_hip_mpr_clocks.so")
    return LIB_MPR if mpr else (LIB_CLOCKS if clocks else LIB)


def source_hash(clocks: bool = False, mpr: bool = False, exp: bool = False) -> str:
    """Hash of every source AND the compiler flags the library is built from - the variant defines included, so that the
    MPR option and the default library never share a hash (bench.py keys PMC traffic files by it; the settled-state cache
    and the bench line carry it)."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + variant_flags(clocks, mpr, exp)).encode())
    for p in sources():
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def needs_build(clocks: bool = False, mpr: bool = False, exp: bool = False) -> bool:
    lib = lib_path(clocks, mpr, exp)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False, clocks: bool = False, mpr: bool = False, exp: bool = False) -> str:
    lib = lib_path(clocks, mpr, exp)
    if not (force or needs_build(clocks, mpr, exp)):
        return lib
    os.makedirs(OBJ, exist_ok=True)
    # -DSO101_DEBUG_CLOCKS: stage clocks + SO101_DEBUG_* env vars for scripts/gpu_*.py; -DSO101_MPR: see LIB_MPR
    flags = list(FLAGS) + variant_flags(clocks, mpr, exp)
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    newest_header = max(os.path.getmtime(s) for s in sources() if not s.endswith(".hip"))

    def newest_input(src):          # the file, every header, and any .hip it includes (tu_tree64.hip is tu_tree.hip under another variant)
        included = re.findall(r'#include\s+"([^"]+\.hip)"', open(src).read())
        return max([os.path.getmtime(src), newest_header] + [os.path.getmtime(os.path.join(os.path.dirname(src), i)) for i in included])

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + (".clk" if clocks else "") + (".mpr" if mpr else "") + (".exp" if exp else "") + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest_input(src):
            return obj
        subprocess.check_call([HIPCC, *flags, "-c", "-o", obj, src])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, translation_units()))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, clocks="--clocks" in sys.argv, mpr="--mpr" in sys.argv, exp="--experimental" in sys.argv))
