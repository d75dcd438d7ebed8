#!/usr/bin/env python3
"""A second build of libpcs_hip.so with extra compiler flags for ONE source file (the other objects are the tree's):
    python tools/build_variant.py NAME pcs_icp.hip -DPCS_NN_WPE=6 ...   ->  build/ab/NAME.so   (for tools/ab_multi.sh)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pcseqlearning_amd import build as B  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build_hip(verbose=False)  # the tree's own objects are current
    out_dir = os.path.join(ROOT, "build", "ab")
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, f"{name}_{src[:-4]}.o")
    subprocess.run([B.HIPCC, *flags, *B.HIP_FLAGS, "-I", os.path.join(ROOT, "include"), "-c", os.path.join(B.CSRC, src),
                    "-o", obj], check=True)
    objs = [obj if f == src else os.path.join(B.OBJ_DIR, f[:-4] + ".o")
            for f in sorted(os.listdir(B.CSRC)) if f.endswith(".hip")]
    lib = os.path.join(out_dir, name + ".so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
    print(lib)


if __name__ == "__main__":
    main()
