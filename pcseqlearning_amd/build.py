"""Build recipe for the gfx950 HIP library (`libpcs_hip.so`) and the CPU oracle.

`python -m pcseqlearning_amd.build` compiles every `csrc/*.hip` with hipcc for gfx950 (hipcc
cross-compiles without a GPU) and links one C-ABI shared library next to this file, in-tree, so it
travels to the GPU box with the repo snapshot. Objects are cached by mtime under `build/`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(ROOT, "build", "hip")
LIB_PATH = os.path.join(HERE, "libpcs_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off + correctly rounded fp32 divide: voxel / cell coordinates are integers derived
# from fp32 (p - lo) / size and must match the CPU restatement bit for bit.
HIP_FLAGS = [*os.environ.get("PCS_EXTRA_HIP_FLAGS", "").split(),
    "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-munsafe-fp-atomics", "-Wno-unused-result",
]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def build_hip(verbose=True, force=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(ROOT, "include", "pcs.h"))
    headers = [h for h in headers if os.path.exists(h)]
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = []
    objs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ_DIR, f[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            jobs.append([HIPCC, *HIP_FLAGS, "-I", os.path.join(ROOT, "include"), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-3:]), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(LIB_PATH) or any(_newer(o, LIB_PATH) for o in objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs])
    return LIB_PATH


def build_oracle(verbose=True, force=False):
    """Compile the CPU restatement (test infrastructure) and, when /root/reference is present,
    the reference's own CPU sources (oracle/_ref)."""
    odir = os.path.join(ROOT, "oracle")
    cmd = ["make", "-C", odir, "-s"]
    if force:
        cmd.append("-B")
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def main():
    force = "--force" in sys.argv
    build_hip(force=force)
    build_oracle(force=force)


if __name__ == "__main__":
    main()
