"""Multi-GPU = independent replicas: no data-path collective, only the timing reduction of bench.py.

Mirrors the reference's DataParallelGroup (N full executors behind a router, /root/reference/src/engine/data_parallel.rs:22-29,84-92):
each rank owns one GPU and one decode stream; aggregate throughput = all ranks' tokens / the slowest rank's time.
"""


def local_device_index(env):
    """LOCAL_RANK -> device ordinal (one process per GPU under torch.distributed.run)."""
    return int(env.get("LOCAL_RANK", "0"))


def aggregate_tokens_per_s(local_ms, steps, dist=None, device=None):
    """(tokens/s over all replicas, max-over-ranks milliseconds). `dist` is an initialised torch.distributed module or None."""
    world = 1
    wall_ms = float(local_ms)
    if dist is not None and dist.is_initialized():
        import torch
        world = dist.get_world_size()
        t = torch.tensor([wall_ms], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall_ms = float(t.item())
    return world * steps / (wall_ms / 1e3), wall_ms


def round_robin(counter, n_replicas):
    """DataParallelGroup::select (data_parallel.rs:84-92): replica index for the next request."""
    return counter % n_replicas
