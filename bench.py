#!/usr/bin/env python3
"""bench.py - G1 scalar-mul STARK proofs/sec on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input per GPU:
  * N = 1 : BASELINE.json configs[1], "Batch 1024 G1 scalar-muls, full trace-gen+NTT+Merkle+FRI on 1 MI355X" = 8
    independent 2^16-row proofs of 128 instances each (the reference's own test shape, scalar_mul_stark.rs:554,569);
  * N > 1 : configs[3], "Batch 16384 G1 scalar-muls sharded across 8 MI355X" = 2048 instances = 16 proofs per GPU per
    step (N = 8 is literally configs[3]; N = 2, 4 keep the same per-GPU shard so the curve is weak scaling).
Independent proofs shard across GPUs with no data-path collective; RCCL only all-gathers the Merkle caps of every proof.
Every step proves DIFFERENT inputs (fresh 256-bit scalars and a fresh assignment of base points per step and rank).
After the timed region every proof of the last step is verified with bn254s_verify and one of them is re-proven alone
(bn254s_prove_g1) and compared word for word, so nothing is timed that was not checked.

`--workload map_to_g2` runs configs[4] instead ("fq_exp STARK + map_to_g2 pipeline, 4096 Fq2 inputs"): the 4096 inputs are
sharded contiguously over the ranks (64 Fq-exp + 32 G2 proofs in total), caps gathered the same way.

Launch: `python bench.py --gpus N --steps K --warmup W` (N = 1), or for N > 1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W`.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INSTANCES_PER_PROOF = 128
N_ROWS = 1 << 16
W, A = 781, 456              # trace width, auxiliary polynomials
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
NTT_BYTES_PER_COL = 40 * N_ROWS   # iNTT r+w 16N, LDE read 8N + write 16N
SEED0 = 0x706C6F6E6B7932     # SURVEY.md §8(d) "Synthetic inputs"
MAP_TO_G2_INPUTS = 4096      # configs[4]


def proofs_per_gpu(world: int) -> int:
    """configs[1] on one GPU (1024 scalar-muls = 8 proofs), configs[3]'s shard on several (2048 = 16 proofs per GPU)."""
    return 8 if world == 1 else 16


def shard_range(rank: int, world: int, per_rank: int):
    """Instances [lo, hi) of the global synthetic batch that `rank` proves (weak scaling)."""
    return rank * per_rank, (rank + 1) * per_rank


def split_range(rank: int, world: int, total: int):
    """Contiguous strong-scaling split of `total` units (configs[4]: 4096 inputs): rank r gets [lo, hi)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def map_to_g2_proof_counts(n_inputs: int):
    """(Fq-exp proofs, G2 proofs) of n inputs: two Legendre jobs and one cofactor-clearing job per input, 128 per proof."""
    return (2 * n_inputs + 127) // 128, (n_inputs + 127) // 128


def step_inputs(pool, step: int, rank: int):
    """Inputs of one step: scalars are fresh uniform 256-bit values, base points and offsets a fresh permutation of the
    rank's pool of synthetic points (seeded by step and rank: every step proves something different)."""
    xs, offs = pool
    n = xs.shape[0]
    rng = np.random.default_rng([SEED0 & 0xFFFFFFFF, 1 + step, rank])
    scalars = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    px, po = rng.permutation(n), rng.permutation(n)
    return np.ascontiguousarray(scalars), np.ascontiguousarray(xs[px]), np.ascontiguousarray(offs[po])


def caps_of(proofs) -> np.ndarray:
    """[n_proofs, 192] the three Merkle caps (trace, aux, quotient) of every proof."""
    return np.stack([p.caps() for p in proofs]).astype(np.uint64)


def gather_caps(local_caps: np.ndarray, dist, device):
    """All-gather of per-proof Merkle caps (the only collective on this path; 1536 B per proof)."""
    import torch
    t = torch.from_numpy(local_caps.view(np.int64)).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def host_cpu_share() -> int:
    """Cores this process may actually use: the cgroup CPU quota when there is one (a GPU box of the pool gives a one-GPU job 16 of
    its 256 hardware threads: more OpenMP threads than that only contend - 4.2 / 5.2 / 6.5 / 7.7 s per proof with 16 / 32 / 64 /
    128 threads, tools/oracle_threads.py), else the visible CPUs; at most 32."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(txt[0]) // int(txt[1])))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 32))


def cpu_baseline():
    """Times ONE 128-instance proof with the CPU oracle (same algorithm, OpenMP) on the host cores."""
    from tests import oracle_lib
    from tools import synth
    lib = oracle_lib.load()
    threads = int(os.environ.get("BENCH_CPU_THREADS", "0")) or min(16, host_cpu_share())
    if hasattr(lib, "orc_set_num_threads"):
        lib.orc_set_num_threads(threads)
    s, x, o = synth.g1_inputs(INSTANCES_PER_PROOF)
    t0 = time.time()
    oracle_lib.g1_prove(lib, s, x, o)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": int(lib.orc_num_threads()), "kind": "port",
            "sample": "1 proof of 128 G1 scalar-muls (2^16 rows), CPU restatement (oracle/: sparse-partial-round Poseidon, radix-2 "
                      "NTT, OpenMP), %.1f s" % dt}


def other_kinds(ctx, synth):
    """BASELINE.json configs[2] (1024 G2 scalar-muls) and the Fq-exp STARK of configs[4], same step shape as the headline: 8 proofs
    x 128 instances per step on one GPU, four steps timed with two in flight (16 proofs queued: a G2 proof's workspace is 7 GB).
    128 distinct synthetic instances are tiled (python big-int G2 points are slow to make), the scalars differ in every proof;
    reported beside the headline, never part of `value`."""
    res = {}
    for name, kind, gen in (("g2_scalar_mul", 1, synth.g2_inputs), ("fq_exp", 2, synth.fq_inputs)):
        try:
            base = gen(INSTANCES_PER_PROOF)
            rng = np.random.default_rng(kind)
            steps = []
            for _ in range(5):
                sc = rng.integers(0, 1 << 63, size=(8 * INSTANCES_PER_PROOF, 4), dtype=np.uint64) * np.uint64(2) + \
                    rng.integers(0, 2, size=(8 * INSTANCES_PER_PROOF, 4), dtype=np.uint64)
                steps.append([np.ascontiguousarray(sc)] + [np.tile(a, (8, 1)) for a in base[1:]])

            def begin(ins):
                return ctx.prove_batch_begin(kind, ins[0], ins[1], ins[2] if len(ins) > 2 else None)
            warm = [begin(steps[0]), begin(steps[1])]   # warm-up with as many proofs in flight as the timed loop has: the
            for h in warm:                              # workspaces of all 16 slots grow to this kind's size outside the timing
                h.end()
            t0 = time.perf_counter()
            pending, proofs = [begin(steps[1])], None
            for ins in steps[2:]:
                pending.append(begin(ins))
                proofs = pending.pop(0).end()
            proofs = pending.pop(0).end()
            dt = (time.perf_counter() - t0) / 4
            ins, p = steps[4], proofs[-1]
            lo = 128 * (len(proofs) - 1)
            ctx.verify(kind, p.words, p.degree_bits, ins[0][lo:lo + 128], ins[1][lo:lo + 128],
                       None if len(ins) < 3 else ins[2][lo:lo + 128], p.outputs)
            res[name] = {"proofs_per_s": round(8 / dt, 2), "ms_per_step_of_8": round(dt * 1e3, 1), "steps_in_flight": 2,
                         "instances_per_s": round(8 * INSTANCES_PER_PROOF / dt, 1), "last_proof_verified": True}
        except Exception as e:
            res[name] = {"error": str(e)}
    return res


def latest_profile(pattern: str):
    """Newest committed profiles/rNN_* file matching the pattern (the PMC passes cannot run inside a timed bench)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def init_dist(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dist = None
    # BENCH_FORCE_DIST=1: a process group even for one rank (tests: the RCCL branch of the cap gather on a one-GPU box)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29733")
        # BENCH_BACKEND=gloo / BENCH_DEVICE=0 exist only to rehearse the N>1 path on a one-GPU box
        dist_mod.init_process_group(backend=os.environ.get("BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        dist = dist_mod
    return rank, local_rank, world, dist


def run_g1(args, pk, torch, synth, ctx, rank, world, dist, device, gather_device):
    ppg = args.proofs_per_gpu or proofs_per_gpu(world)
    per_rank = INSTANCES_PER_PROOF * ppg
    # pool of synthetic points of this rank (x = k1*G, offset = k2*G, python big integers: made once, outside the timing)
    _, xs, offs = synth.g1_inputs(per_rank, seed=SEED0 + 1 + rank)
    pool = (xs, offs)
    n_steps = args.warmup + args.steps
    inputs = [step_inputs(pool, i, rank) for i in range(n_steps)]

    # A step = one batch (bn254s_prove_batch_begin ... _end + the cap gather).  Up to `--steps-in-flight` (default: 32 proofs' worth) steps are
    # open at a time: the next batch is queued before the current one is collected, so its first proofs fill the GPU while the
    # last proofs of the current batch run their latency-bound tail (FRI folds, proof of work).  All K steps complete inside the
    # timed region; --steps-in-flight 1 is the strictly sequential loop (reported beside the headline as `sequential_steps`).
    import collections

    def begin(i):
        s, x, o = inputs[i]
        return ctx.prove_batch_begin(0, s, x, o, per_proof=INSTANCES_PER_PROOF)

    def finish(h):
        proofs = h.end()
        caps = caps_of(proofs)
        if dist is not None:
            caps = gather_caps(caps, dist, gather_device)
        return proofs, caps

    def run_steps(lo, hi, depth, on_step=None):
        inflight, last = collections.deque(), None
        for i in range(lo, hi):
            inflight.append(begin(i))
            if len(inflight) >= depth:
                last = finish(inflight.popleft())
                if on_step:
                    on_step(last[0])
        while inflight:
            last = finish(inflight.popleft())
            if on_step:
                on_step(last[0])
        return last

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    depth = args.steps_in_flight if args.steps_in_flight > 0 else max(1, 32 // ppg)
    # Set-up, not a step: every slot (stream + ~4 GB workspace of a proof in flight, allocated on first use) that the pipeline will
    # touch is used once - `depth` batches in flight, as in the timed region.  Without it the W warm-up steps of a short run reach
    # only the first slots and the remaining workspaces would be allocated (hipMalloc of gigabytes) inside the timed region.
    prime = [step_inputs(pool, 100000 + i, rank) for i in range(depth)]
    hs = [ctx.prove_batch_begin(0, s, x, o, per_proof=INSTANCES_PER_PROOF) for s, x, o in prime]
    for h in hs:
        h.end()
    del hs, prime
    if args.warmup:
        run_steps(0, args.warmup, depth)
    sync()
    stage_acc, n_acc = {}, [0]

    def account(step_proofs):
        for p in step_proofs:
            for k, v in p.stage_ms.items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
            n_acc[0] += 1

    t0 = time.perf_counter()
    proofs, caps = run_steps(args.warmup, n_steps, depth, account)
    sync()
    dt = time.perf_counter() - t0
    n_acc = n_acc[0]
    if dist is not None:
        t = torch.tensor([dt], device=gather_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- check what was timed (every rank, after the clock stopped): all proofs of the last step verify, the gathered
    # caps are the proofs' caps, and one proof equals the single-proof entry point word for word
    s, x, o = inputs[n_steps - 1]
    checked = {"verified_proofs": 0, "caps_match": False, "batch_equals_single": False}
    for j, p in enumerate(proofs):
        lo = j * INSTANCES_PER_PROOF
        ctx.verify(0, p.words, p.degree_bits, s[lo:lo + 128], x[lo:lo + 128], o[lo:lo + 128], p.outputs)
        checked["verified_proofs"] += 1
    mine = caps[rank] if dist is not None else caps
    checked["caps_match"] = bool(np.array_equal(mine, caps_of(proofs)))
    j = (n_steps - 1) % len(proofs)
    lo = j * INSTANCES_PER_PROOF
    single = ctx.prove_g1(s[lo:lo + 128], x[lo:lo + 128], o[lo:lo + 128])
    checked["batch_equals_single"] = bool(np.array_equal(single.words, proofs[j].words))
    ok = checked["caps_match"] and checked["batch_equals_single"] and checked["verified_proofs"] == len(proofs)
    if dist is not None:
        t = torch.tensor([1.0 if ok else 0.0], device=gather_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item() > 0.5)
    if not ok:
        raise SystemExit(f"rank {rank}: timed proofs failed the post-run check: {checked}")
    if rank != 0:
        return None

    total_proofs = world * ppg * args.steps
    sequential = None
    if depth > 1 and world == 1:  # the same K steps again, one at a time (after the clock of the headline stopped)
        sync()
        ts = time.perf_counter()
        run_steps(args.warmup, n_steps, 1)
        sync()
        ds = time.perf_counter() - ts
        sequential = {"value": round(total_proofs / ds, 3), "unit": "proofs/s", "ms_per_step": round(ds / args.steps * 1e3, 3),
                      "note": "--steps-in-flight 1: every batch is collected before the next one is queued"}
    stage_ms = {k: v / max(n_acc, 1) for k, v in stage_acc.items()}
    # roofline of the NTT/LDE stage (north-star kernel): HIP-event time of the NTT launches of the two commitments of a
    # proof, measured on the proof's own stream inside the timed region.
    ntt_ms = stage_ms.get("trace_ntt", 0.0) + stage_ms.get("aux_ntt", 0.0)
    ntt_bytes = NTT_BYTES_PER_COL * (W + A)
    achieved = ntt_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
    # the same stage alone on the GPU (no other stream), after the timed region, with the shader clock it holds (one probe wave
    # samples the core-clock counter against the 100 MHz counter) ...
    excl_ms, ntt_mhz, ntt_mhz_min = ctx.bench_ntt_clock(W + A, 20)
    excl = ntt_bytes / (excl_ms * 1e-3) / 1e9
    # ... and the limit that actually binds it: vector-instruction issue.  floor = waves per SIMD x executed VALU instructions per
    # wave (SQ_INSTS_VALU / SQ_WAVES of the committed PMC pass: static) x the issue cost of one half-rate instruction, measured
    # live in cycles (bn254s_bench_issue: ns per instruction x its own clock) and converted at the clock the NTT stage holds.
    valu_floor = None
    try:
        ntt_prof, ntt_prof_src = latest_profile("r*_ntt_valu.json")
        issue_ns, issue_mhz = ctx.bench_issue()
        if ntt_prof and ntt_mhz > 0 and issue_mhz > 0:
            cyc = issue_ns * 1e-9 * issue_mhz * 1e6
            n_simd = torch.cuda.get_device_properties(device).multi_processor_count * 4
            waves_per_simd = (W + A) * N_ROWS / 16 / 64 / n_simd          # a lane owns 16 elements of a column
            floor_ms = waves_per_simd * float(ntt_prof["valu_insts_per_wave_stage"]) * cyc / (ntt_mhz * 1e6) * 1e3
            valu_floor = {"floor_ms": round(floor_ms, 4), "frac": round(floor_ms / ntt_ms, 4) if ntt_ms > 0 else None,
                          "frac_exclusive": round(floor_ms / excl_ms, 4),
                          "valu_insts_per_wave_stage": ntt_prof["valu_insts_per_wave_stage"], "valu_insts_source": f"static: {ntt_prof_src}",
                          "waves_per_simd": round(waves_per_simd, 2), "cycles_per_valu_inst": round(cyc, 3),
                          "issue_probe": {"ns_per_inst": round(issue_ns, 4), "mhz": round(issue_mhz, 1)},
                          "ntt_stage_clock_mhz": {"mean": round(ntt_mhz, 1), "slowest_10us": round(ntt_mhz_min, 1)},
                          "note": "the stage is bound by vector-instruction issue, not by HBM: frac = this floor / the measured "
                                  "stage time (1.0 = every issue slot of every SIMD used by the stage's own instructions)"}
    except Exception as e:
        valu_floor = {"error": str(e)}
    pmc, pmc_src = latest_profile("r*_pmc_ntt.json")
    traffic = int(pmc["ntt_stage_traffic_bytes_per_1237_cols"]) if pmc else None
    # integer-ALU roofline of the Poseidon leaf hash (the kernel that bounds proofs/s): wave-level VALU instructions per
    # second against the issue peak measured by tools/ubench/intops.hip; the instruction count per launch comes from the
    # committed SQ_INSTS_VALU pass, the duration is measured here with HIP events
    alu = None
    try:
        lh_ms = ctx.bench_leafhash(W, 17, 5)
        prof, prof_src = latest_profile("r*_alu.json")
        issue_ns2, issue_mhz2 = ctx.bench_issue()
        if prof:
            insts = float(prof["leaf_hash_valu_wave_insts_per_launch_781x2e17"])
            # issue peak of the half-rate class measured live (bn254s_bench_issue, 8 waves per SIMD); the full-rate instructions of
            # the permutation (v_mov_b32, v_sub_u32, v_min_u32: 16 % of it) cost about half an issue slot each
            n_simd = torch.cuda.get_device_properties(device).multi_processor_count * 4
            peak = n_simd / (issue_ns2 * 1e-9)
            full_share = float(prof.get("full_rate_share", 0.158))
            ach = insts * (1.0 - 0.46 * full_share) / (lh_ms * 1e-3)
            alu = {"bound": "valu-issue", "kernel": "k_leaf_hash (781 columns x 2^17 leaves = 12.8 M Poseidon permutations)",
                   "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G half-rate wave-instr/s",
                   "frac": round(ach / peak, 4), "ms": round(lh_ms, 4),
                   "valu_insts_per_permutation": prof.get("valu_insts_per_permutation"),
                   "valu_insts_source": f"static: SQ_INSTS_VALU pass, {prof_src}; a full-rate instruction counted as 0.54 issue slots",
                   "issue_probe": {"ns_per_inst": round(issue_ns2, 4), "mhz": round(issue_mhz2, 1)},
                   "permutations_per_s": round((1 << 17) * 98 / (lh_ms * 1e-3) / 1e9, 3)}
        else:
            alu = {"ms": round(lh_ms, 4), "permutations_per_s": round((1 << 17) * 98 / (lh_ms * 1e-3) / 1e9, 3)}
    except Exception as e:
        alu = {"error": str(e)}
    # the same 1024 scalar multiplications as ONE tall proof (N = 2^19), the shape Bn254Hook::constrain produces for a
    # circuit with 1024 calls; reported next to the headline, not part of `value`
    tall = None
    if world == 1 and not args.no_extras:
        try:
            s8, x8, o8 = s[:1024], x[:1024], o[:1024]
            ctx.prove_g1(s8, x8, o8)
            tt0 = time.perf_counter()
            tp = ctx.prove_g1(s8, x8, o8)
            tdt = time.perf_counter() - tt0
            ctx.verify(0, tp.words, tp.degree_bits, s8, x8, o8, tp.outputs)
            tall = {"rows_log2": tp.degree_bits, "ms_per_proof": round(tdt * 1e3, 2),
                    "scalar_muls_per_s": round(1024 / tdt, 1), "verified": True}
        except Exception as e:
            tall = {"error": str(e)}
    cfg = "configs[1]: batch of 1024 G1 scalar-muls on 1 GPU per step = 8 proofs x 128 instances" if world == 1 and ppg == 8 else \
        (f"configs[3] shard: {per_rank} G1 scalar-muls = {ppg} proofs x 128 instances per GPU per step "
         f"({world * per_rank} per step over {world} GPUs" + ("; N = 8 is the 16384 of configs[3])" if ppg == 16 else ")"))
    out = {
        "metric": "G1 scalar-mul STARK proofs/sec (256-bit scalars)",
        "value": round(total_proofs / dt, 3),
        "unit": "proofs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64 (Goldilocks p = 2^64-2^32+1; BN254 Fq as 10x26-bit Montgomery limbs)",
        "data": "synthetic (fresh random 256-bit scalars and point assignment every step)",
        "config": {"workload": cfg + ", 2^16 rows, W=781, standard_fast_config",
                   "proofs_per_step_per_gpu": ppg, "instances_per_proof": INSTANCES_PER_PROOF, "steps_in_flight": depth,
                   "parallelism": "1 GPU, no collective" if world == 1 else
                                  f"{world} ranks x independent proofs, one RCCL all-gather of the Merkle caps per step"},
        "checked": checked,
        "sequential_steps": sequential,
        "scalar_muls_per_s": round(total_proofs * INSTANCES_PER_PROOF / dt, 1),
        "stage_ms_per_proof": {k: round(v, 3) for k, v in stage_ms.items()},
        "one_tall_proof_of_1024": tall,
        "roofline": {"bound": "hbm", "kernel": "ntt_lde stage = iNTT + both coset NTTs (k_ntt_* launches) over the 781 trace "
                                               "+ 456 aux columns of one proof",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": (f"static: FETCH_SIZE/WRITE_SIZE PMC passes of the same launches, {pmc_src}"
                                        if pmc_src else None),
                     "algorithmic_bytes": ntt_bytes, "ms": round(ntt_ms, 4),
                     "exclusive": {"achieved": round(excl, 1), "frac": round(excl / HBM_PEAK_GBS, 4),
                                   "ms": round(excl_ms, 4),
                                   "note": "same launches with no other stream on the GPU (bn254s_bench_ntt)"},
                     "valu_floor": valu_floor},
        "roofline_alu": alu,
    }
    if world == 1 and not args.no_extras:
        out["other_kinds"] = other_kinds(ctx, synth)
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the bench line must still be produced
            out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    return out


def map_to_g2_inputs(lo: int, hi: int):
    """Inputs [lo, hi) of the global 4096-input batch of configs[4]: uniform Fq2 elements u (xoshiro stream of
    tools/map_to_g2_ref.inputs) and non-infinity G2 offsets (128 distinct synthetic points, tiled by global index)."""
    from tools import synth
    from tools import map_to_g2_ref as m2g
    us = m2g.inputs(hi)[lo:hi]
    u = np.array([synth._to_words(a[0]) + synth._to_words(a[1]) for a in us], dtype=np.uint64).reshape(hi - lo, 8)
    _, _, base_off = synth.g2_inputs(128, seed=SEED0 + 5)
    off = base_off[np.arange(lo, hi) % 128].copy()
    return u, off


def run_map_to_g2(args, pk, torch, synth, ctx, rank, world, dist, device, gather_device):
    total = args.inputs
    lo, hi = split_range(rank, world, total)
    u, off = map_to_g2_inputs(lo, hi)
    n_steps = args.warmup + args.steps

    def step(i):
        # a different slice order every step: rotate the rank's inputs (same multiset, different proofs)
        r = i % max(hi - lo, 1)
        uu, oo = np.roll(u, r, axis=0), np.roll(off, r, axis=0)
        res = ctx.map_to_g2(np.ascontiguousarray(uu), np.ascontiguousarray(oo))
        caps = caps_of(res[3] + res[4])
        if dist is not None:
            # ranks may hold different proof counts when world does not divide the inputs: pad to the maximum
            n_max = sum(map_to_g2_proof_counts(split_range(0, world, total)[1]))
            pad = np.zeros((n_max, 192), np.uint64)
            pad[:caps.shape[0]] = caps
            caps = gather_caps(pad, dist, gather_device)
        return (uu, oo) + tuple(res), caps

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        last, caps = step(i)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=gather_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # check the last step: every proof verifies against the job arrays the library returned
    uu, oo, pts, fq_jobs, g2_jobs, pf, pg = last
    fs, fx = np.ascontiguousarray(fq_jobs[:, :4]), np.ascontiguousarray(fq_jobs[:, 4:])
    gs, gx = np.ascontiguousarray(g2_jobs[:, :4]), np.ascontiguousarray(g2_jobs[:, 4:])
    for i, p in enumerate(pf):
        a = 128 * i
        ctx.verify(2, p.words, p.degree_bits, fs[a:a + 128], fx[a:a + 128], None, p.outputs)
    for i, p in enumerate(pg):
        a = 128 * i
        ctx.verify(1, p.words, p.degree_bits, gs[a:a + 128], gx[a:a + 128], oo[a:a + 128], p.outputs)
    if rank != 0:
        return None
    n_fq, n_g2 = 0, 0
    for r in range(world):
        a, b = split_range(r, world, total)
        f, g = map_to_g2_proof_counts(b - a)
        n_fq, n_g2 = n_fq + f, n_g2 + g
    return {
        "metric": "map_to_g2 inputs/sec (fq_exp STARK + G2 scalar-mul STARK pipeline)",
        "value": round(total * args.steps / dt, 2),
        "unit": "inputs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 (Goldilocks) / BN254 Fq as 10x26-bit Montgomery limbs",
        "data": "synthetic",
        "config": {"workload": f"configs[4]: fq_exp STARK + map_to_g2 pipeline, {total} Fq2 inputs = {n_fq} Fq-exp proofs + "
                               f"{n_g2} G2 proofs of 128 instances, inputs sharded contiguously over {world} GPU(s)",
                   "parallelism": "1 GPU, no collective" if world == 1 else
                                  f"{world} ranks, one RCCL all-gather of the Merkle caps per step"},
        "proofs_per_s": round((n_fq + n_g2) * args.steps / dt, 2),
        "checked": {"verified_proofs_rank0_last_step": len(pf) + len(pg)},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("g1", "map_to_g2"), default="g1",
                    help="g1 = the headline metric (configs[1] / configs[3]); map_to_g2 = configs[4]")
    ap.add_argument("--proofs-per-gpu", type=int, default=0, help="override the per-GPU batch of the g1 workload")
    ap.add_argument("--inputs", type=int, default=MAP_TO_G2_INPUTS, help="map_to_g2 workload: total Fq2 inputs per step")
    ap.add_argument("--steps-in-flight", type=int, default=0,
                    help="batches open at a time (k: step i + k is queued before step i is collected; 1: sequential; default: as "
                         "many as keep 32 proofs queued = the library's slots: 4 steps of 8 proofs on one GPU, 2 steps of 16 on several)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the G2 / Fq-exp / tall-proof figures reported beside the headline")
    args = ap.parse_args()

    import plonky2_bn254_amd as pk   # first: sets GPU_MAX_HW_QUEUES before the HIP runtime initialises
    import torch
    from tools import synth

    rank, local_rank, world, dist = init_dist(args)
    dev_id = int(os.environ.get("BENCH_DEVICE", local_rank))
    if world > 1:
        # several ranks on one host: the 32 proof threads of a rank sleep on an interrupt while they wait for the GPU instead of
        # spinning (one GPU: 86.1-89.5 proofs/s spinning with 33 s of CPU per run, 87.7-88.3 sleeping with 10.5 s; tools/gpu_sync_ab.sh)
        os.environ.setdefault("BN254S_SYNC", "blocking")
    ctx = pk.Context(dev_id)      # before torch touches the device (BN254S_SYNC is applied at the first use of a device)
    torch.cuda.set_device(dev_id)
    device = torch.device("cuda", dev_id)
    gather_device = device if os.environ.get("BENCH_BACKEND", "nccl") == "nccl" else torch.device("cpu")
    fn = run_g1 if args.workload == "g1" else run_map_to_g2
    out = fn(args, pk, torch, synth, ctx, rank, world, dist, device, gather_device)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
