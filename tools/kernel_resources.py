#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from the compiler's resource-usage remarks (no GPU needed).
usage: python tools/kernel_resources.py [file.hip ...]   (default: every .hip of plonky2_bn254_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonky2_bn254_amd", "csrc")


def resources(path):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
           "-c", path, "-o", "/dev/null"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in err.splitlines():
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            full = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
            cur = {"name": full.split("(")[0]}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = v
    return rows


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print("%-52s %6s %6s %8s %5s %7s" % ("kernel", "VGPRs", "AGPRs", "scratch", "occ", "LDS"))
    for f in files:
        for r in resources(os.path.abspath(f)):
            print("%-52s %6s %6s %8s %5s %7s" % (r["name"][:52], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
