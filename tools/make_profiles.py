#!/usr/bin/env python3
"""Turns gpurun_out/profiles_rNN/ (tools/gpu_profiles.sh) into the committed profiles/rNN_* files.
usage: python tools/make_profiles.py r02"""
import collections
import csv
import json
import os
import re
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_" + R)
DST = os.path.join(ROOT, "profiles")
N, W, A = 1 << 16, 781, 456


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    bench = json.loads(open(os.path.join(SRC, "bench_n1.json")).read().strip().splitlines()[-1])
    shutil.copy(os.path.join(SRC, "bench_kernel_stats.csv"), os.path.join(DST, R + "_bench_kernel_stats.csv"))
    for f in ("ubench_issue.log", "ubench_poseidon.log", "hash_sweep.log"):
        shutil.copy(os.path.join(SRC, f), os.path.join(DST, R + "_" + f.replace(".log", ".txt")))

    # ---- issue peak from the microbenchmark: half-rate class at 8 waves per SIMD ----
    cyc = {}
    for ln in open(os.path.join(SRC, "ubench_issue.log")):
        m = re.match(r"(.+?)\s+(\d+) waves/SIMD\s+([\d.]+) ms ->\s+([\d.]+) cycles", ln)
        if m:
            cyc[(m.group(1).strip(), int(m.group(2)))] = float(m.group(4))
    mad8 = cyc[("v_mad_u64_u32 -> vcc", 8)]
    peak = 1024 * 2.4e9 / mad8        # wave instructions per second of the half-rate class (the one Poseidon is made of)

    # ---- Poseidon leaf hash: VALU instructions per launch and per permutation ----
    h = counters(os.path.join(SRC, "hash_counter_collection.csv"))
    big = h[("k_leaf_hash", 1 << 20)]
    valu_per_wave = sum(big["SQ_INSTS_VALU"]) / sum(big["SQ_WAVES"])
    salu_per_wave = sum(big["SQ_INSTS_SALU"]) / sum(big["SQ_WAVES"])
    per_perm = valu_per_wave / 98
    alu = {"leaf_hash_valu_wave_insts_per_launch_781x2e17": 2048 * valu_per_wave, "valu_insts_per_permutation": round(per_perm, 1),
           "salu_insts_per_permutation": round(salu_per_wave / 98, 1),
           "full_rate_share": 0.1323,   # tools/gen_poseidon_asm.py: 1576 of the 11908 instructions are v_mov_b32 / v_sub_u32
           "valu_peak_wave_insts_per_s": peak, "peak_definition": "1024 SIMDs x 2.4 GHz / %.2f cycles: issue cost of v_mad_u64_u32 (and of "
           "every other half-rate instruction) measured with 8 waves per SIMD, tools/ubench/sgpr_ops.hip" % mad8,
           "source": "profiles/%s_valu_insts.md" % R}
    json.dump(alu, open(os.path.join(DST, R + "_alu.json"), "w"), indent=1)

    # ---- VALU instructions per kernel of single proofs ----
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(os.path.join(SRC, "insts_counter_collection.csv"))):
        n = short(r["Kernel_Name"])
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[n] += 1
    tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
    # executed VALU instructions per wave of the four launches of a commitment (bench.py: roofline.valu_floor)
    pw = {n: v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1) for n, v in acc.items()}
    ntt_stage = pw["k_ntt_pass1<true>"] + pw["k_ntt_intt2_lde1<8>"] + 2 * pw["k_ntt_pass2<false, true>"]
    json.dump({"valu_insts_per_wave_stage": round(ntt_stage, 1),
               "per_kernel": {k: round(v, 1) for k, v in pw.items() if k.startswith("k_ntt")},
               "source": "profiles/%s_valu_insts.md (SQ_INSTS_VALU / SQ_WAVES; stage = pass1<true> + intt2_lde1<8> + 2 x pass2<false,true>)" % R},
              open(os.path.join(DST, R + "_ntt_valu.json"), "w"), indent=1)
    with open(os.path.join(DST, R + "_valu_insts.md"), "w") as f:
        f.write("# %s - where the VALU issue slots go: `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -- python3 tools/run_proofs.py 2 single`\n\n" % R)
        f.write("Three single G1 proofs (128 instances, 2^16 rows; `tools/gpu_profiles.sh`).  Wave-level VALU instructions per kernel, "
                "summed over the run.\n\n| kernel | launches | VALU instructions | share | per wave |\n|---|---|---|---|---|\n")
        for n, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:22]:
            f.write("| `%s` | %d | %.3g | %.1f %% | %.0f |\n" % (n[:40], calls[n], v["SQ_INSTS_VALU"], 100 * v["SQ_INSTS_VALU"] / tot,
                                                               v["SQ_INSTS_VALU"] / max(v["SQ_WAVES"], 1)))
        f.write("\nPoseidon leaf hashing alone (`tools/bench_hash.py` under `--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES`, 781 columns x 2^20 "
                "leaves): %.0f VALU + %.0f SALU instructions per wave = **%.0f VALU per permutation** (98 permutations per leaf; the "
                "generator counts 12 846 for the permutation itself, the rest is the sponge loop: loads, round 0's constants, "
                "parking the input).  Round 1: 24.4 k (compiler output).\n" % (valu_per_wave, salu_per_wave, per_perm))
        f.write("\nIssue peak used by bench.py's `roofline_alu`: %.1f G wave-instructions/s = 1024 SIMDs x 2.4 GHz / %.2f cycles "
                "(`%s_ubench_issue.txt`: every half-rate instruction - v_mad_u64_u32, v_lshl_add_u64, carry instructions, "
                "v_cndmask with a scalar mask, three-operand adds - costs that; v_add_u32 / v_mov_b32 cost %.2f).\n"
                % (peak / 1e9, mad8, R, cyc[("v_add_u32", 8)]))

    # ---- HBM traffic of the NTT/LDE stage ----
    fe, wr = counters(os.path.join(SRC, "fetch_counter_collection.csv")), counters(os.path.join(SRC, "write_counter_collection.csv"))
    dur = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(SRC, "ntt_kernel_stats.csv")))}
    rows, total = [], 0.0
    for (name, grid), c in sorted(fe.items()):
        if not (name.startswith("k_ntt") or name == "k_copy_u64"):
            continue
        fkb = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        wkb = sum(wr[(name, grid)]["WRITE_SIZE"]) / len(wr[(name, grid)]["WRITE_SIZE"])
        launches = 2 if "pass2" in name else 1
        rows.append((name, launches, fkb, 2 * fkb * 1.024e-3, wkb, wkb * 1.024e-3, dur.get(name, 0.0)))
        if name.startswith("k_ntt"):
            total += launches * (2 * fkb + wkb) * 1024
    algo = 40 * N * (W + A)
    json.dump({"ntt_stage_traffic_bytes_per_1237_cols": total, "algorithmic_bytes": algo, "source": "profiles/%s_pmc_ntt.md" % R},
              open(os.path.join(DST, R + "_pmc_ntt.json"), "w"))
    with open(os.path.join(DST, R + "_pmc_ntt.md"), "w") as f:
        f.write("# %s - HBM traffic of the NTT/LDE kernels from PMC counters (rocprofv3 --pmc, separate passes)\n\n" % R)
        f.write("Command: `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/pmc_ntt.py` (`tools/gpu_profiles.sh`; 1237 columns = one "
                "proof's trace + aux).  Units: KB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts half of a coalesced streaming "
                "read on gfx950 (the calibration copy k_copy_u64, 648.5 MB read + written, confirms the factor 2); WRITE_SIZE is exact.  "
                "Durations: `rocprofv3 --kernel-trace --stats` of the same command.\n\n")
        f.write("| kernel | launches | FETCH_SIZE (KB) | x2 corrected (MB) | WRITE_SIZE (KB) | written (MB) | avg us |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| `%s` | %d | %.0f | %.1f | %.0f | %.1f | %.1f |\n" % r)
        f.write("\nNTT/LDE stage: measured HBM traffic **%.3f GB** per commitment of 1237 columns against %.3f GB algorithmic (40*N*C): "
                "ratio %.2f.\n" % (total / 1e9, algo / 1e9, total / algo))
    # ---- summary of the bench profile ----
    os.system("python %s %s %s %s > %s" % (os.path.join(ROOT, "tools", "profile_summary.py"), os.path.join(DST, R + "_bench_kernel_stats.csv"),
                                          os.path.join(SRC, "bench_n1.json"), os.path.join(SRC, "bench_kernel_trace.csv"),
                                          os.path.join(DST, R + "_summary.md")))
    shutil.copy(os.path.join(SRC, "bench_n1.json"), os.path.join(DST, R + "_bench_n1.json"))
    print("wrote profiles/%s_*: %.0f VALU per permutation, issue peak %.1f G/s, NTT traffic %.3f GB" % (R, per_perm, peak / 1e9, total / 1e9))


if __name__ == "__main__":
    main()
