"""Multi-GPU sharding of the hot path: independent frames per rank, no data-path collective.

One process per GPU.  Rank r owns the frame pairs with global indices r*F .. r*F+F-1 (independent streams / GOP
segments); the only communication is the barrier around the timed region and a MAX reduction of the elapsed time.
"""


def spawn_ranks(script, argv, n, port=None, env=None):
    """`python script --gpus N ...` started without a launcher: run the N ranks (one process per GPU) the way the driver does --
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P script argv -- and return
    its exit status.  The rendezvous is on 127.0.0.1 (a container hostname may not resolve)."""
    import os
    import socket
    import subprocess
    import sys
    if port is None:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # this pool's driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=e)


def init_ranks(gpus_arg, backend, device_count=None):
    """Rank set-up of a process started by torch.distributed.run (or alone): returns (rank, local_rank, world).  WORLD_SIZE must
    agree with --gpus, and with backend nccl (= RCCL) every rank needs its own GPU."""
    import os
    rank, local_rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if gpus_arg != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (gpus_arg, world))
    if device_count is not None and device_count < world:
        raise SystemExit("%d ranks but only %d GPU(s) visible: one rank per GPU, no oversubscription" % (world, device_count))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == world
    return rank, local_rank, world


def rank_frame_seeds(rank, frames_per_rank):
    """Global frame indices (also the synthetic-frame seeds) owned by `rank`."""
    return list(range(rank * frames_per_rank, (rank + 1) * frames_per_rank))


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX of a python float over all ranks (identity when not distributed)."""
    if dist is None or not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_mpixels_per_s(world, pixels_per_step_per_rank, steps, seconds):
    """Aggregate throughput of the whole job: every rank processed the same amount (weak scaling)."""
    return world * pixels_per_step_per_rank * steps / seconds / 1e6


def rank_estimates(rank, world, estimates):
    """Lookahead frame-cost estimates (p0, b, p1) are independent given the pictures (which every rank holds -- a lookahead window of
    half-resolution pictures is a few MB): rank r takes every world-th estimate, no exchange.  Estimates that share a (b, list,
    distance) search are kept on one rank so that the reference's search reuse (bDoSearch) stays local: the key is (b, b - p0)."""
    keys = sorted({(b, b - p0) for (p0, b, p1) in estimates})
    owner = {k: i % world for i, k in enumerate(keys)}
    return [e for e in estimates if owner[(e[1], e[1] - e[0])] == rank]
