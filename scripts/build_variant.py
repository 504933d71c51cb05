"""Kernel experiments: build ab/lib_<name>.so = the default library with some translation units recompiled under extra -D flags.
   python scripts/build_variant.py <name> [--tus tu_narrow,tu_pipe_solve] -- -DFOO=1 ...
The variant is selected at run time with SO101_HIP_LIB=ab/lib_<name>.so (so101_sim_amd/native.py); ab/ is git-ignored and travels with gpurun."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from so101_sim_amd import build as B

def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        k = argv.index("--"); extra = argv[k + 1:]; argv = argv[:k]
    name = argv[0]
    tus = ["tu_narrow"]
    if "--tus" in argv:
        tus = argv[argv.index("--tus") + 1].split(",")
    B.build()
    root = os.path.dirname(B._HERE)
    out = os.path.join(root, "ab"); os.makedirs(out, exist_ok=True)
    objs = []
    for src in B.translation_units():
        base = os.path.basename(src)[:-4]
        if base in tus or "all" in tus:
            obj = os.path.join(out, f"{base}.{name}.o")
            cmd = [B.HIPCC, *B.FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", "-o", obj, src]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                print(r.stderr); sys.exit(1)
            cur = None
            for line in r.stderr.splitlines():
                if "Function Name" in line: cur = line.split("Function Name:")[1].split("[")[0].strip()
                for key in ("VGPRs:", "VGPRs Spill", "ScratchSize", "Occupancy", "LDS Size"):
                    if key in line and cur and ("k_narrow" in cur or "k_pipe_solve" in cur or "k_tree" in cur):
                        print(f"  {cur[:40]:40s} {line.split('remark:')[1].split('[-R')[0].strip()}")
        else:
            obj = os.path.join(B.OBJ, base + ".o")
        objs.append(obj)
    lib = os.path.join(out, f"lib_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    print(lib)

if __name__ == "__main__":
    main()
