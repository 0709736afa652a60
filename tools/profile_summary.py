#!/usr/bin/env python3
"""Turns a rocprofv3 `*_kernel_stats.csv` + a bench JSON line into profiles/rNN_summary.md.

usage: python tools/profile_summary.py profiles/r01_bench_kernel_stats.csv profiles/r01_bench_n1.json > profiles/r01_summary.md
"""
import csv
import json
import sys


def ntt_cross_check(trace_csv):
    """NTT/LDE stage of one proof from the kernel trace: the launches over 781 (trace) and 456 (aux) columns, to be compared
    with bench.py's HIP-event figure roofline.ms (both measured inside the batches, other proofs' small kernels running)."""
    import collections
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(trace_csv)):
        if "k_ntt_" not in r["Kernel_Name"] or int(r["Grid_Size_Y"]) not in (781, 456):
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_Y"]))
        acc[key][0] += 1
        acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("\nNTT/LDE stage from the kernel trace (launches over the 781 trace / 456 aux columns of the proofs in the batches):\n")
    print("| kernel | columns | launches | avg us |")
    print("|---|---|---|---|")
    tot = 0.0
    for (name, cols), (n, t) in sorted(acc.items()):
        print("| `%s` | %d | %d | %.1f |" % (name, cols, n, t / n))
        tot += t
    proofs = max(sum(n for (name, cols), (n, t) in acc.items() if name.startswith("k_ntt_intt2_lde1") and cols == 781), 1)
    print("\nSum per proof: **%.3f ms** for 3.243 GB algorithmic = %.0f GB/s (bench.py reports `roofline.ms` from HIP events on "
          "the same launches of its own, un-profiled run)." % (tot / proofs / 1e3, 3.242721280 / (tot / proofs / 1e6)))


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
    if len(sys.argv) > 3:
        ntt_cross_check(sys.argv[3])
    print("\nbench line of the same build (un-profiled run):\n")
    print("```json")
    print(json.dumps(line, indent=1))
    print("```")


if __name__ == "__main__":
    main()
