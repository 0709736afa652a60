"""BASELINE.json configs[3] and configs[4] at their stated per-GPU size, the N > 1 branch of bench.py as a real child-process
run (gloo ranks sharing the one GPU of the box), and a G1 proof at 2^23 rows - what the first 8-GPU run of the driver executes,
exercised here so that it cannot die on a code path nobody ran.
Reference: hook.rs:63-89 (all calls of a circuit are proven together), utils/hash_to_g2.rs:113-148,184-201 (map_to_g2)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import bench
import plonky2_bn254_amd as pk
from tests import oracle_lib
from tools import map_to_g2_ref as m2g
from tools import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_shard_2048_g1_is_16_verified_proofs(gpu_ctx, oracle):
    """configs[3] = 16384 G1 scalar-muls over 8 GPUs: one rank's shard (2048 instances = 16 proofs, the shape bench.py runs per
    GPU and step for N > 1), inputs drawn exactly as bench.py draws them for rank 3, through bn254s_prove_batch_begin / _end."""
    ppg = bench.proofs_per_gpu(8)
    per_rank = bench.INSTANCES_PER_PROOF * ppg
    assert (ppg, per_rank) == (16, 2048)
    _, xs, offs = synth.g1_inputs(per_rank, seed=bench.SEED0 + 1 + 3)
    s, x, o = bench.step_inputs((xs, offs), 2, 3)
    h = gpu_ctx.prove_batch_begin(0, s, x, o, per_proof=bench.INSTANCES_PER_PROOF)
    proofs = h.end()
    assert len(proofs) == 16 and all(p.degree_bits == 16 for p in proofs)
    caps = bench.caps_of(proofs)
    assert caps.shape == (16, 192) and len({c.tobytes() for c in caps}) == 16
    for j, p in enumerate(proofs):
        sl = slice(128 * j, 128 * (j + 1))
        gpu_ctx.verify(0, p.words, 16, s[sl], x[sl], o[sl], p.outputs)
        assert np.array_equal(p.section("trace_cap"), caps[j, :64])
    for j in (0, 15):  # the oracle's restatement of the reference's verify() (common/verifier.rs:32-98)
        sl = slice(128 * j, 128 * (j + 1))
        rc, msg = oracle_lib.g1_verify(oracle, proofs[j].words, 16, s[sl], x[sl], o[sl])
        assert rc == 0, (j, msg)
    single = gpu_ctx.prove_g1(s[128 * 11:128 * 12], x[128 * 11:128 * 12], o[128 * 11:128 * 12])
    assert np.array_equal(single.words, proofs[11].words)
    i = 2047
    want = synth.g1_scalar_mul_offset(synth.words_to_int(s[i]), (synth.words_to_int(x[i, :4]), synth.words_to_int(x[i, 4:])),
                                      (synth.words_to_int(o[i, :4]), synth.words_to_int(o[i, 4:])))
    got = proofs[15].outputs.reshape(128, 8)[127]
    assert (synth.words_to_int(got[:4]), synth.words_to_int(got[4:])) == want


def test_config5_map_to_g2_4096_inputs_every_proof_verified(gpu_ctx):
    """configs[4] at full size on one GPU: 4096 Fq2 inputs -> 8192 Legendre jobs (64 Fq-exp proofs) + 4096 cofactor-clearing jobs
    (32 G2 proofs).  Job arrays equal the Python front-end's (tools/map_to_g2_ref.py: SvdW candidates, norms, selected point with
    its signed square root - all 4096), every one of the 96 proofs passes bn254s_verify against those Python-made job arrays."""
    n = bench.MAP_TO_G2_INPUTS
    u, off = bench.map_to_g2_inputs(0, n)
    us = m2g.inputs(n)
    assert all(synth._to_words(a[0]) + synth._to_words(a[1]) == list(u[k]) for k, a in enumerate(us[:64]))
    pts, fq_jobs, g2_jobs, pf, pg = gpu_ctx.map_to_g2(u, off)
    assert (len(pf), len(pg)) == bench.map_to_g2_proof_counts(n) == (64, 32)
    fs, fx = m2g.fq_exp_jobs(us)
    assert np.array_equal(fq_jobs[:, :4], fs) and np.array_equal(fq_jobs[:, 4:], fx)
    legendre = [pow(synth.words_to_int(w), (synth.P - 1) // 2, synth.P) for w in fx]
    branches = set()
    gx = np.zeros((n, 16), np.uint64)
    for k in range(n):
        a, b = legendre[2 * k] == 1, legendre[2 * k + 1] == 1
        gx[k] = m2g._pt_words(m2g.select_point(us[k], a, b))
        branches.add(0 if a else 1 if b else 2)
    assert branches == {0, 1, 2}
    gs = np.tile(np.array(synth._to_words(m2g.COFACTOR), dtype=np.uint64), (n, 1))
    assert np.array_equal(g2_jobs[:, :4], gs) and np.array_equal(g2_jobs[:, 4:], gx)
    for i, p in enumerate(pf):
        sl = slice(128 * i, 128 * (i + 1))
        gpu_ctx.verify(2, p.words, p.degree_bits, np.ascontiguousarray(fs[sl]), np.ascontiguousarray(fx[sl]), None, p.outputs)
        assert [synth.words_to_int(w) for w in p.outputs.reshape(-1, 4)] == legendre[sl]
    for i, p in enumerate(pg):
        sl = slice(128 * i, 128 * (i + 1))
        gpu_ctx.verify(1, p.words, p.degree_bits, np.ascontiguousarray(gs[sl]), np.ascontiguousarray(gx[sl]), off[sl], p.outputs)
    for k in (0, 1777, 4095):  # python G2 arithmetic is slow: spot-check the images cofactor * point
        assert synth.g2_from_words(pts[k]) == synth.g2_mul(m2g.COFACTOR, synth.g2_from_words(gx[k]))
        assert m2g.finish(pg[k // 128].outputs.reshape(-1, 16)[k % 128], synth.g2_from_words(off[k])) == synth.g2_from_words(pts[k])


def test_g1_2pow23_rows_one_proof(gpu_ctx):
    """16384 G1 scalar multiplications in ONE proof (N = 2^23: what Bn254Hook::constrain produces for configs[3] as a single
    circuit, hook.rs:63-71, scalar_mul_stark.rs:60): the radix-2 level and the compact 245 GB workspace at their real size.
    128 distinct points tiled, 16384 distinct scalars.  Checked by both of the library's verifiers (GPU constraint sum, and the
    independent host statement of the AIR), outputs by Python big integers, a corrupted opening rejected by both."""
    n = 16384
    _, x0, o0 = synth.g1_inputs(128, seed=91)
    rng = np.random.default_rng(2023)
    s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    x = np.ascontiguousarray(x0[rng.integers(0, 128, size=n)])
    o = np.ascontiguousarray(o0[rng.integers(0, 128, size=n)])
    pr = gpu_ctx.prove_g1(s, x, o)
    assert pr.degree_bits == 23
    outs = pr.outputs.reshape(n, 8)
    for i in (0, 8191, 16383):
        want = synth.g1_scalar_mul_offset(synth.words_to_int(s[i]), (synth.words_to_int(x[i, :4]), synth.words_to_int(x[i, 4:])),
                                          (synth.words_to_int(o[i, :4]), synth.words_to_int(o[i, 4:])))
        assert (synth.words_to_int(outs[i, :4]), synth.words_to_int(outs[i, 4:])) == want
    gpu_ctx.verify(0, pr.words, 23, s, x, o, pr.outputs)
    pk.verify_host(0, pr.words, 23, s, x, o, pr.outputs)
    bad = pr.words.copy()
    bad[64 * 3 + 2 * 781 + 2 * 781 + 5] ^= np.uint64(1)  # an auxiliary opening
    with pytest.raises(pk.VerifyError):
        gpu_ctx.verify(0, bad, 23, s, x, o, pr.outputs)
    with pytest.raises(pk.VerifyError):
        pk.verify_host(0, bad, 23, s, x, o, pr.outputs)
    print("G1 2^23 rows: stage ms", {k: round(v, 1) for k, v in pr.stage_ms.items()})
    del pr
    # the context returns to short proofs: with the tall workspace still in slot 0, and after bn254s_ctx_trim gave it back
    s2, x2, o2 = synth.g1_inputs(3, seed=92)
    p2 = gpu_ctx.prove_g1(s2, x2, o2)
    gpu_ctx.verify(0, p2.words, 16, s2, x2, o2, p2.outputs)
    import torch
    free0 = torch.cuda.mem_get_info(0)[0]
    gpu_ctx.trim()
    assert torch.cuda.mem_get_info(0)[0] - free0 > 150e9      # ~245 GB went back to the driver
    p3 = gpu_ctx.prove_g1(s2, x2, o2)
    assert np.array_equal(p3.words, p2.words)


def _run_bench_2_ranks(extra, port, timeout):
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` as a fresh child process: two gloo ranks on
    the one GPU of this box (BENCH_BACKEND=gloo BENCH_DEVICE=0; the driver's 8-GPU run uses nccl and one GPU per rank - the rest of
    the code path is the same)."""
    # (two ranks share this box's one GPU and its memory: 16 slots and one step in flight per rank instead of the 32 / 2 a rank
    # has to itself on its own GPU)
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               BN254S_SLOTS="16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]     # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_g1_child_process(gpu_ctx):
    gpu_ctx.trim()        # the two child processes need the device memory this process's idle workspaces hold
    out = _run_bench_2_ranks(["--steps", "2", "--warmup", "1", "--steps-in-flight", "1", "--no-extras", "--no-cpu-baseline"], 29711, 900)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "proofs/s" and out["value"] > 0 and out["higher_is_better"] is True
    assert out["config"]["proofs_per_step_per_gpu"] == 16 and "configs[3] shard: 2048 G1 scalar-muls" in out["config"]["workload"]
    assert out["checked"] == {"verified_proofs": 16, "caps_match": True, "batch_equals_single": True}
    assert abs(out["value"] - 2 * 16 * 2 / (out["ms_per_step"] * 2e-3)) < 0.05 * out["value"]
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1


def test_bench_two_ranks_map_to_g2_child_process(gpu_ctx):
    gpu_ctx.trim()
    out = _run_bench_2_ranks(["--steps", "1", "--warmup", "1", "--workload", "map_to_g2"], 29713, 900)
    assert out["n_gpus"] == 2 and out["unit"] == "inputs/s" and out["value"] > 0 and out["scaling"] == "strong"
    assert "configs[4]" in out["config"]["workload"] and "4096 Fq2 inputs = 64 Fq-exp proofs + 32 G2 proofs" in out["config"]["workload"]
    assert out["checked"]["verified_proofs_rank0_last_step"] == 32 + 16


def test_bench_cap_gather_over_rccl_one_rank(gpu_ctx):
    """The nccl (= RCCL) branch of bench.py's cap gather: device tensors, int64 view of the u64 caps, all_gather, barrier and the
    MAX / MIN all-reduces - with a process group of ONE rank (two ranks cannot share a GPU under RCCL), so that the first multi-GPU
    run of the driver does not meet this code for the first time."""
    gpu_ctx.trim()
    env = dict(os.environ, BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29717", HSA_ENABLE_IPC_MODE_LEGACY="0", BN254S_SLOTS="16")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--steps-in-flight", "2",
           "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["checked"] == {"verified_proofs": 8, "caps_match": True, "batch_equals_single": True}
    assert out["value"] > 0
