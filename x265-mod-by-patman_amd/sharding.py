"""Multi-GPU sharding of the hot path: independent frames per rank, no data-path collective.

One process per GPU.  Rank r owns the frame pairs with global indices r*F .. r*F+F-1 (independent streams / GOP
segments); the only communication is the barrier around the timed region and a MAX reduction of the elapsed time.
"""


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
