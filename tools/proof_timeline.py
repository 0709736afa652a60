#!/usr/bin/env python3
"""Kernels of the last proof in a rocprofv3 kernel trace of tools/one_proof.py, in launch order: start (ms from the proof's
first kernel), duration, idle gap before it; consecutive launches of one kernel are merged.  usage: proof_timeline.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
# the last proof starts at the last launch of a chain kernel
starts = [i for i, e in enumerate(ev) if "dbl_chain" in e[2] or "k_fq_chain" in e[2]]
ev = ev[starts[-1]:]
t0, prev_end = ev[0][0], ev[0][0]
out = []
for s, e, n in ev:
    gap = max(0, s - prev_end)
    if out and out[-1][0] == n and gap < 3000:
        out[-1][2] += e - s
        out[-1][3] += 1
    else:
        out.append([n, s - t0, e - s, 1, gap])
    prev_end = max(prev_end, e)
print("%-44s %9s %9s %5s %9s" % ("kernel", "start ms", "dur us", "calls", "gap us"))
tot_gap = 0
for n, s, d, c, g in out:
    print("%-44s %9.3f %9.1f %5d %9.1f" % (n[:44], s / 1e6, d / 1e3, c, g / 1e3))
    tot_gap += g
print("span %.3f ms, idle between kernels %.3f ms" % ((prev_end - t0) / 1e6, tot_gap / 1e6))
