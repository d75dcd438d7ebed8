#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy of every kernel of one csrc file, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage with the flags of pcseqlearning_amd/build.py):
    python tools/kernel_resources.py pcseqlearning_amd/csrc/pcs_icp.hip [tree-root] [name filter]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, root):
    sys.path.insert(0, HERE)
    from pcseqlearning_amd import build as B
    cmd = [B.HIPCC, *B.HIP_FLAGS, "-I" + os.path.join(root, "include"), "-c", os.path.join(root, src), "-o", os.devnull,
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for ln in err.splitlines():
        m = re.search(r"remark: (.*)", ln)
        if not m:
            continue
        t = m.group(1).strip()
        f = re.match(r"Function Name: (\S+)", t)
        if f:
            cur = f.group(1)
            rows[cur] = {}
            continue
        kv = re.match(r"([A-Za-z ]*[A-Za-z])[^:]*: (\d+)", t)
        if kv and cur:
            rows[cur][kv.group(1).strip()] = int(kv.group(2))
    return rows


def main():
    src = sys.argv[1]
    root = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else HERE
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    for n, r in resources(src, root).items():
        if flt and flt not in n:
            continue
        print("%-72s vgpr %4d agpr %3d spill %4d scratch %5d occ %2d lds %6d" % (
            n[:72], r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("VGPRs Spill", 0), r.get("ScratchSize", 0),
            r.get("Occupancy", 0), r.get("LDS Size", 0)))


if __name__ == "__main__":
    main()
