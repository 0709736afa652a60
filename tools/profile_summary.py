#!/usr/bin/env python3
"""Turns a rocprofv3 `*_kernel_stats.csv` + a bench JSON line into profiles/rNN_summary.md.

usage: python tools/profile_summary.py profiles/r01_bench_kernel_stats.csv profiles/r01_bench_n1.json > profiles/r01_summary.md
"""
import csv
import json
import sys


def main():
    stats, bench = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(stats)))
    line = json.loads(open(bench).read().strip().splitlines()[-1])
    print("# rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (1 MI355X)\n")
    print("3 batches of 8 proofs (128 G1 scalar-muls each, 2^16 rows) and the exclusive NTT re-run.\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in rows:
        if float(r["Percentage"]) < 0.02:
            continue
        name = r["Name"].replace("|", "/")
        print("| `%s` | %s | %.2f | %.1f | %s |" % (name[:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print("\nbench line of the same build (un-profiled run):\n")
    print("```json")
    print(json.dumps(line, indent=1))
    print("```")


if __name__ == "__main__":
    main()
