"""world_size-2 gloo test of the multi-GPU plumbing of bench.py: input sharding and the Merkle-cap
all-gather (the only collective on this path).  Runs on CPU; proofs are replaced by synthetic caps."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_range(rank, world, 1024)
    caps = (np.arange(8 * 192, dtype=np.uint64).reshape(8, 192) + np.uint64(1000003 * rank)) | np.uint64(1 << 63)
    allcaps = bench.gather_caps(caps, dist, torch.device("cpu"))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ret[rank] = (lo, hi, allcaps.copy(), float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_cap_gather_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29671, ret), nprocs=world, join=True)
    assert ret[0][:2] == (0, 1024) and ret[1][:2] == (1024, 2048)
    for r in range(world):
        allcaps = ret[r][2]
        assert allcaps.shape == (2, 8, 192) and allcaps.dtype == np.uint64
        for src in range(world):
            want = (np.arange(8 * 192, dtype=np.uint64).reshape(8, 192) + np.uint64(1000003 * src)) | np.uint64(1 << 63)
            assert np.array_equal(allcaps[src], want)   # u64 values above 2^63 survive the int64 view
        assert ret[r][3] == 2.0
