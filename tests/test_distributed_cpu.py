"""world_size-2 gloo tests of the multi-GPU plumbing of bench.py: input sharding of both workloads (configs[3]: weak
shards of 16 proofs per GPU; configs[4]: 4096 map_to_g2 inputs split contiguously) and the Merkle-cap all-gather (the only
collective on this path).  Runs on CPU; proofs are replaced by synthetic caps."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _caps(rank, n):
    return (np.arange(n * 192, dtype=np.uint64).reshape(n, 192) + np.uint64(1000003 * rank)) | np.uint64(1 << 63)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ppg = bench.proofs_per_gpu(world)
    lo, hi = bench.shard_range(rank, world, 128 * ppg)
    allcaps = bench.gather_caps(_caps(rank, ppg), dist, torch.device("cpu"))
    # configs[4]: an input count the world does not divide -> ragged proof counts, padded gather
    total = 4096 + 130
    a, b = bench.split_range(rank, world, total)
    n_fq, n_g2 = bench.map_to_g2_proof_counts(b - a)
    n_max = sum(bench.map_to_g2_proof_counts(bench.split_range(0, world, total)[1]))
    pad = np.zeros((n_max, 192), np.uint64)
    pad[:n_fq + n_g2] = _caps(rank, n_fq + n_g2)
    allcaps5 = bench.gather_caps(pad, dist, torch.device("cpu"))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ret[rank] = (lo, hi, allcaps.copy(), float(t.item()), (a, b, n_fq, n_g2), allcaps5.copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_cap_gather_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29671, ret), nprocs=world, join=True)
    assert ret[0][:2] == (0, 2048) and ret[1][:2] == (2048, 4096)   # 16 proofs x 128 instances per rank (configs[3] shard)
    for r in range(world):
        allcaps = ret[r][2]
        assert allcaps.shape == (2, 16, 192) and allcaps.dtype == np.uint64
        for src in range(world):
            assert np.array_equal(allcaps[src], _caps(src, 16))   # u64 values above 2^63 survive the int64 view
        assert ret[r][3] == 2.0
    # configs[4] sharding: contiguous, covering, ordered; every rank sees every rank's caps in rank order
    assert ret[0][4][:2] == (0, 2113) and ret[1][4][:2] == (2113, 4226)
    for r in range(world):
        for src in range(world):
            a, b, n_fq, n_g2 = ret[src][4]
            assert (n_fq, n_g2) == ((2 * (b - a) + 127) // 128, (b - a + 127) // 128)
            got = ret[r][5][src]
            assert np.array_equal(got[:n_fq + n_g2], _caps(src, n_fq + n_g2)) and not got[n_fq + n_g2:].any()


def test_bench_shapes_follow_baseline_configs():
    import bench
    assert bench.proofs_per_gpu(1) * 128 == 1024                       # configs[1]
    assert bench.proofs_per_gpu(8) * 128 * 8 == 16384                  # configs[3]
    assert bench.proofs_per_gpu(2) == bench.proofs_per_gpu(4) == bench.proofs_per_gpu(8)   # weak scaling for N > 1
    # configs[4]: 4096 inputs = 64 Fq-exp + 32 G2 proofs, for every world size the driver uses
    for world in (1, 2, 4, 8):
        parts = [bench.split_range(r, world, 4096) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == 4096 and all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        counts = [bench.map_to_g2_proof_counts(b - a) for a, b in parts]
        assert (sum(c[0] for c in counts), sum(c[1] for c in counts)) == (64, 32)
    # ragged split
    assert [bench.split_range(r, 3, 10) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


def test_step_inputs_change_every_step_and_rank():
    import bench
    xs = np.arange(8 * 8, dtype=np.uint64).reshape(8, 8)
    offs = xs + np.uint64(1000)
    a = bench.step_inputs((xs, offs), 0, 0)
    b = bench.step_inputs((xs, offs), 1, 0)
    c = bench.step_inputs((xs, offs), 0, 1)
    a2 = bench.step_inputs((xs, offs), 0, 0)
    assert all(np.array_equal(u, v) for u, v in zip(a, a2))            # deterministic
    assert not np.array_equal(a[0], b[0]) and not np.array_equal(a[0], c[0])
    for s, x, o in (a, b, c):
        assert s.shape == (8, 4) and s.dtype == np.uint64 and s.flags["C_CONTIGUOUS"]
        assert sorted(map(tuple, x)) == sorted(map(tuple, xs)) and sorted(map(tuple, o)) == sorted(map(tuple, offs))
    assert (a[0] >> np.uint64(63)).any()                                # scalars use all 256 bits
