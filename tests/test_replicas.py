"""N>1 path on CPU: two gloo ranks run the replica aggregation bench.py uses (no data-path collective exists)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from blazr_amd import replicas
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert replicas.local_device_index(os.environ) == rank
    dist.barrier()
    tok_s, wall = replicas.aggregate_tokens_per_s(100.0 * (rank + 1), 50, dist)   # rank 1 is the slow replica
    out[rank] = (tok_s, wall)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_aggregation():
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, 29571, out), nprocs=2, join=True)
    assert len(out) == 2
    for r in (0, 1):
        tok_s, wall = out[r]
        assert wall == pytest.approx(200.0)                 # max over ranks
        assert tok_s == pytest.approx(2 * 50 / 0.2)         # both replicas' tokens / slowest time


def test_single_process_and_router():
    sys.path.insert(0, ROOT)
    from blazr_amd import replicas
    assert replicas.aggregate_tokens_per_s(50.0, 100) == (2000.0, 50.0)
    assert [replicas.round_robin(i, 3) for i in range(5)] == [0, 1, 2, 0, 1]
