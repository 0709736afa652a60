#!/usr/bin/env python3
"""bench.py - G1 scalar-mul STARK proofs/sec on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input: BASELINE.json configs[1],
"Batch 1024 G1 scalar-muls, full trace-gen+NTT+Merkle+FRI on 1 MI355X", i.e. 8 independent 2^16-row
proofs of 128 instances each (the reference's own test shape, scalar_mul_stark.rs:554,569) per GPU
per step.  Independent proofs shard across GPUs with no data-path collective (weak scaling: every rank
proves its own 1024 instances); RCCL is used only to all-gather the Merkle caps of every proof.

Launch: `python bench.py --gpus N --steps K --warmup W` (N = 1), or for N > 1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W`.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INSTANCES_PER_PROOF = 128
PROOFS_PER_STEP = 8          # 1024 scalar multiplications per GPU per step
N_ROWS = 1 << 16
W, A = 781, 456              # trace width, auxiliary polynomials
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
NTT_BYTES_PER_COL = 40 * N_ROWS   # iNTT r+w 16N, LDE read 8N + write 16N


def shard_range(rank: int, world: int, per_rank: int):
    """Instances [lo, hi) of the global synthetic batch that `rank` proves (weak scaling)."""
    return rank * per_rank, (rank + 1) * per_rank


def caps_of(proofs) -> np.ndarray:
    """[n_proofs, 192] the three Merkle caps (trace, aux, quotient) of every proof."""
    return np.stack([p.words[:192] for p in proofs]).astype(np.uint64)


def gather_caps(local_caps: np.ndarray, dist, device):
    """All-gather of per-proof Merkle caps (the only collective on this path; 1536 B per proof)."""
    import torch
    t = torch.from_numpy(local_caps.view(np.int64)).to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy().view(np.uint64)


def cpu_baseline():
    """Times ONE 128-instance proof with the CPU oracle (same algorithm, OpenMP) on the host cores."""
    from tests import oracle_lib
    from plonky2_bn254_amd import synth
    lib = oracle_lib.load()
    s, x, o = synth.g1_inputs(INSTANCES_PER_PROOF)
    t0 = time.time()
    oracle_lib.g1_prove(lib, s, x, o)
    dt = time.time() - t0
    return {"value": round(1.0 / dt, 5), "unit": "proofs/s", "cores": int(lib.orc_num_threads()), "kind": "port",
            "sample": "1 proof of 128 G1 scalar-muls (2^16 rows), CPU restatement (oracle/), %.1f s" % dt}


def other_kinds(ctx, synth):
    """BASELINE.json configs[2] (1024 G2 scalar-muls) and the Fq-exp STARK of configs[4], same batch shape as the headline:
    8 proofs x 128 instances on one GPU.  128 distinct synthetic instances are tiled 8x (python big-int G2 points are slow to
    make); reported beside the headline, never part of `value`."""
    res = {}
    for name, kind, gen in (("g2_scalar_mul", 1, synth.g2_inputs), ("fq_exp", 2, synth.fq_inputs)):
        try:
            ins = [np.tile(a, (PROOFS_PER_STEP, 1)) for a in gen(INSTANCES_PER_PROOF)]
            off = ins[2] if len(ins) > 2 else None
            ctx.prove_batch(kind, ins[0], ins[1], off)
            t0 = time.perf_counter()
            reps = 2
            for _ in range(reps):
                ctx.prove_batch(kind, ins[0], ins[1], off)
            dt = (time.perf_counter() - t0) / reps
            res[name] = {"proofs_per_s": round(PROOFS_PER_STEP / dt, 2), "ms_per_batch_of_8": round(dt * 1e3, 1),
                         "instances_per_s": round(PROOFS_PER_STEP * INSTANCES_PER_PROOF / dt, 1)}
        except Exception as e:
            res[name] = {"error": str(e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the G2 / Fq-exp / tall-proof figures reported beside the headline")
    args = ap.parse_args()

    import plonky2_bn254_amd as pk   # first: sets GPU_MAX_HW_QUEUES before the HIP runtime initialises
    import torch
    from plonky2_bn254_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_BACKEND=gloo / BENCH_DEVICE=0 exist only to rehearse the N>1 path on a one-GPU box
        dist_mod.init_process_group(backend=os.environ.get("BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
        dist = dist_mod
    dev_id = int(os.environ.get("BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_id)
    device = torch.device("cuda", dev_id)
    gather_device = device if os.environ.get("BENCH_BACKEND", "nccl") == "nccl" else torch.device("cpu")
    ctx = pk.Context(dev_id)

    per_rank = INSTANCES_PER_PROOF * PROOFS_PER_STEP
    lo, hi = shard_range(rank, world, per_rank)
    # synthetic inputs: every rank derives its own shard deterministically (seed + rank)
    s, x, o = synth.g1_inputs(per_rank, seed=0x706C6F6E6B7932 + 1 + rank)

    def step():
        proofs = ctx.prove_g1_batch(s, x, o, per_proof=INSTANCES_PER_PROOF)
        caps = caps_of(proofs)
        if dist is not None:
            caps = gather_caps(caps, dist, gather_device)
        return proofs, caps

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    stage_acc, n_acc = {}, 0
    for _ in range(args.steps):
        proofs, caps = step()
        for p in proofs:
            for k, v in p.stage_ms.items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
            n_acc += 1
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=gather_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_proofs = world * PROOFS_PER_STEP * args.steps
        stage_ms = {k: v / max(n_acc, 1) for k, v in stage_acc.items()}
        # roofline of the NTT/LDE stage (north-star kernel): HIP-event time of the four NTT launches
        # of one commitment, measured on the proof's own stream inside the timed region.
        ntt_ms = stage_ms.get("trace_ntt", 0.0) + stage_ms.get("aux_ntt", 0.0)
        ntt_bytes = NTT_BYTES_PER_COL * (W + A)
        achieved = ntt_bytes / (ntt_ms * 1e-3) / 1e9 if ntt_ms > 0 else 0.0
        # the same stage alone on the GPU (no other stream), after the timed region
        excl_ms = ctx.bench_ntt(W + A, 5)
        excl = ntt_bytes / (excl_ms * 1e-3) / 1e9
        # HBM traffic of the same four launches from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_ntt.md)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_ntt.json")) as f:
                traffic = int(json.load(f)["ntt_stage_traffic_bytes_per_1237_cols"])
        except Exception:
            pass
        # the same 1024 scalar multiplications as ONE tall proof (N = 2^19), the shape Bn254Hook::constrain produces for a
        # circuit with 1024 calls; reported next to the headline, not part of `value`
        tall = None
        try:
            if args.no_extras:
                raise RuntimeError("skipped (--no-extras)")
            ctx.prove_g1(s, x, o)
            tt0 = time.perf_counter()
            ctx.prove_g1(s, x, o)
            tdt = time.perf_counter() - tt0
            tall = {"rows_log2": 19, "ms_per_proof": round(tdt * 1e3, 2), "scalar_muls_per_s": round(per_rank / tdt, 1)}
        except Exception as e:
            tall = {"error": str(e)}
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
            "data": "synthetic",
            "config": {"workload": "configs[1]: batch of 1024 G1 scalar-muls per GPU per step = 8 proofs x 128 instances, "
                                   "2^16 rows, W=781, standard_fast_config",
                       "proofs_per_step_per_gpu": PROOFS_PER_STEP, "instances_per_proof": INSTANCES_PER_PROOF,
                       "parallelism": f"{world} x independent proofs, RCCL all-gather of Merkle caps"},
            "scalar_muls_per_s": round(total_proofs * INSTANCES_PER_PROOF / dt, 1),
            "stage_ms_per_proof": {k: round(v, 3) for k, v in stage_ms.items()},
            "one_tall_proof_of_1024": tall,
            "roofline": {"bound": "hbm", "kernel": "ntt_lde stage = k_ntt_pass1 (iNTT) + k_ntt_intt2_lde1 (fused iNTT pass 2 / pass 1 of "
                                                   "both cosets) + k_ntt_pass2 x {coset g, coset g*w_2N} over the 781 trace + 456 aux "
                                                   "columns of one proof",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes": ntt_bytes, "ms": round(ntt_ms, 4),
                         "exclusive": {"achieved": round(excl, 1), "frac": round(excl / HBM_PEAK_GBS, 4),
                                       "ms": round(excl_ms, 4),
                                       "note": "same four launches with no other stream on the GPU (bn254s_bench_ntt)"}},
        }
        if world == 1 and not args.no_extras:
            out["other_kinds"] = other_kinds(ctx, synth)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the bench line must still be produced
                out["cpu_baseline"] = {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
