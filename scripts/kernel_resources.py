"""Per-kernel resources of the built library, from the code objects themselves (llvm-objcopy --dump-section .hip_fatbin, clang-offload-bundler,
llvm-readelf --notes): VGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, LDS bytes per workgroup, waves per SIMD by registers.
   python scripts/kernel_resources.py [> profiles/rNN_kernel_resources.txt]"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
rows = []
for obj in sorted(glob.glob(os.path.join(ROOT, "so101_sim_amd", "csrc", "build", "tu_*.o"))):
    if re.search(r"\.(clk|mpr|exp)\.o$", obj):
        continue
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        if subprocess.call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], stderr=subprocess.DEVNULL) != 0:
            continue                       # no device code in this object (the experimental step paths in the default build)
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    # one map per kernel, keys in alphabetical order: a kernel's entry starts at its .agpr_count
    for block in re.split(r"\n\s+- \.agpr_count:", "\n" + notes)[1:]:
        block = ".agpr_count:" + block
        cur = {}
        for k, v in re.findall(r"\.(\w+):\s+(\S+)", block):
            if k in ("group_segment_fixed_size", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "agpr_count") and k not in cur:
                cur[k] = int(v)
            elif k == "name" and "name" not in cur and v.startswith("_Z") or (k == "name" and v.startswith("k_")):
                cur.setdefault("name", v)
        if "name" in cur:
            rows.append((os.path.basename(obj), cur))
demangle = lambda n: subprocess.check_output(["c++filt", n], text=True).strip().split("(")[0]
print("%-18s %-34s %5s %5s %7s %7s %8s %7s %6s" % ("object", "kernel", "VGPR", "AGPR", "spillV", "spillS", "scratchB", "LDS B", "w/SIMD"))
for obj, r in rows:
    tot = r.get("vgpr_count", 0)                 # (unified register count: AGPRs included)
    waves = 8 if tot <= 64 else (512 // (-(-max(tot, 1) // 8) * 8))
    print("%-18s %-34s %5d %5d %7d %7d %8d %7d %6d" % (obj, demangle(r["name"])[-34:], r.get("vgpr_count", 0), r.get("agpr_count", 0), r.get("vgpr_spill_count", 0),
                                                          r.get("sgpr_spill_count", 0), r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0), min(8, waves)))
