"""Kernel experiments: builds the library with extra compiler flags into ab/libso101_<tag>.so (git-ignored; selected with SO101_HIP_LIB).
    usage: python scripts/build_variant.py TAG -DNAME=VALUE ..."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from so101_sim_amd import build as b

tag, extra = sys.argv[1], sys.argv[2:]
obj_dir = f"/tmp/so101_objs_{tag}"
os.makedirs(obj_dir, exist_ok=True)
os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)


def one(src):
    obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
    subprocess.check_call([b.HIPCC, *b.FLAGS, *extra, "-c", "-o", obj, src])
    return obj


with ThreadPoolExecutor(max_workers=8) as pool:
    objs = list(pool.map(one, b.translation_units()))
out = os.path.join(ROOT, "ab", f"libso101_{tag}.so")
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print(out)
