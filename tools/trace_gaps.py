#!/usr/bin/env python3
"""Where a batch run leaves the GPU without a wide (GPU-filling) kernel: reads a rocprofv3 kernel trace of bench.py, takes the
densest cluster of launches (the timed steps) and prints the concurrency histogram, the gaps without a wide kernel and what runs
in them (tuning only).  usage: python tools/trace_gaps.py gpurun_out/.../x_kernel_trace.csv"""
import collections
import csv
import sys

WIDE = ("k_leaf_hash", "k_ntt_", "k_quotient", "k_logup", "k_fri_combine", "k_openings", "k_histogram", "k_ctl_terms")


def is_wide(n):
    return any(w in n for w in WIDE) and "coop" not in n


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
    clusters, cs, ce, items = [], ev[0][0], ev[0][1], [ev[0]]
    for x in ev[1:]:
        if x[0] - ce > 3e6:
            clusters.append((cs, ce, items))
            cs, ce, items = x[0], x[1], []
        items.append(x)
        ce = max(ce, x[1])
    clusters.append((cs, ce, items))
    cs, ce, items = max(clusters, key=lambda c: len(c[2]))
    n_leaf = sum(1 for _, _, n in items if n == "k_leaf_hash")
    print("cluster %.1f ms, %d kernels, %d proofs (leaf-hash launches / 3)" % ((ce - cs) / 1e6, len(items), n_leaf // 3))

    def hist(pred):
        pts = []
        for s, e, n in items:
            if pred(n):
                pts += [(s, 1), (e, -1)]
        pts.sort()
        cur, last, h, gaps = 0, cs, {}, []
        for t, d in pts:
            if t > last:
                h[cur] = h.get(cur, 0) + (t - last)
                if cur == 0:
                    gaps.append((last, t))
            last, cur = t, cur + d
        tot = sum(h.values())
        return {k: round(v / tot, 3) for k, v in sorted(h.items())}, gaps

    print("kernels running at once:", hist(lambda n: True)[0])
    h, gaps = hist(is_wide)
    print("wide kernels running at once:", h)
    print("leaf hash:", hist(lambda n: n == "k_leaf_hash")[0], " NTT:", hist(lambda n: "k_ntt_" in n)[0])
    tot = sum(b - a for a, b in gaps)
    print("no wide kernel: %.1f ms in %d gaps; the eight longest:" % (tot / 1e6, len(gaps)))
    for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:8]:
        print("   at +%.1f ms: %.2f ms" % ((a - cs) / 1e6, (b - a) / 1e6))
    acc = collections.Counter()
    for a, b in gaps:
        for s, e, n in items:
            if e > a and s < b:
                acc[n] += min(e, b) - max(s, a)
    print("kernel time inside those gaps:", ", ".join("%s %.1f ms" % (n, v / 1e6) for n, v in acc.most_common(8)))


if __name__ == "__main__":
    main()
