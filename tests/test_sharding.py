"""N>1 path on CPU: world_size-2 gloo.  Ranks own disjoint frames, generate different deterministic pictures,
need no data exchange, and the bench bookkeeping (barrier, MAX-over-ranks time, whole-job aggregate) is consistent."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import x265hip  # noqa: F401
from x265hip_pkg.sharding import rank_frame_seeds, max_over_ranks, whole_job_mpixels_per_s, rank_estimates
from x265hip_pkg.synth import frame_pair


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, frames, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from oracle_py import Oracle
    seeds = rank_frame_seeds(rank, frames)
    ora = Oracle(8)
    row = ora.mvcost_row(28, 4096)
    sums = []
    for s in seeds:                                    # each rank's own frames through the (CPU) oracle: no peer data needed
        cur, ref, stride, (dx, dy) = frame_pair(64, 64, 8, s, margin=40, max_shift=6)
        off = 40 * stride + 40
        mv = ora.me(64, 64, cur.reshape(-1), stride, off, ref.reshape(-1), stride, off, [-16, -16, 16, 16], (0, 0), [], 16, 1, 2, row)
        sums.append((s, int(cur.sum()), mv[0], mv[1], 4 * dx, 4 * dy))
    dist.barrier()
    dt = max_over_ranks(0.5 + 0.25 * rank, dist)       # slowest rank defines the job time
    gathered = [None] * world
    dist.all_gather_object(gathered, sums)             # test-only: collect for the assertions (not a data-path collective)
    if rank == 0:
        np.save(os.path.join(out_dir, "res.npy"), np.array([dt, whole_job_mpixels_per_s(world, 64 * 64 * frames, 3, dt)]))
        import json
        json.dump(gathered, open(os.path.join(out_dir, "g.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    world, frames = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), frames, str(tmp_path)), nprocs=world, join=True)
    dt, value = np.load(tmp_path / "res.npy")
    assert dt == 0.75                                   # MAX over ranks
    assert abs(value - 2 * 64 * 64 * frames * 3 / 0.75 / 1e6) < 1e-9
    import json
    g = json.load(open(tmp_path / "g.json"))
    seeds = [r[0] for part in g for r in part]
    assert sorted(seeds) == list(range(world * frames))            # disjoint and complete
    assert len({r[1] for part in g for r in part}) == world * frames   # every frame is a different picture
    for part in g:
        for (_, _, mvx, mvy, tx, ty) in part:                       # each rank found its own frames' motion
            assert abs(mvx - tx) <= 4 and abs(mvy - ty) <= 4


def test_rank_seeds_are_disjoint_for_eight_gpus():
    allseeds = [s for r in range(8) for s in rank_frame_seeds(r, 8)]
    assert sorted(allseeds) == list(range(64))
    assert max_over_ranks(1.25) == 1.25


def _la_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
    from oracle_py import Oracle
    from lookahead_util import Geometry, lowres_planes_oracle, oracle_frame_cost, oracle_intra, synth_clip
    from x265hip_pkg.lookahead import minigop_estimates
    ora = Oracle(8)
    frames = synth_clip(96, 64, 5, 8, seed=5)             # every rank holds the (small) window of pictures
    g = Geometry(96, 64)
    planes = [lowres_planes_oracle(ora, f, g) for f in frames]
    intra = [oracle_intra(ora, p, g) for p in planes]
    est = minigop_estimates(5, 2)
    mine = rank_estimates(rank, world, est) if world > 1 else est
    res = {}
    for (p0, b, p1) in mine:                               # the rank's own estimates through the (CPU) oracle: no peer data needed
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], None)
        res[(p0, b, p1)] = (o["costEst"], o["costEstAq"], o["intraMbs"], int(o["mvs0"].astype(np.int64).sum()))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)                  # test-only collection
    if rank == 0:
        import pickle
        pickle.dump((est, gathered), open(os.path.join(out_dir, "la.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


def test_lookahead_estimates_shard_without_exchange(tmp_path):
    import pickle
    mp.spawn(_la_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    est, two = pickle.load(open(tmp_path / "la.pkl", "rb"))
    assert not (set(two[0]) & set(two[1])) and set(two[0]) | set(two[1]) == set(est) and two[0] and two[1]
    mp.spawn(_la_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    _, one = pickle.load(open(tmp_path / "la.pkl", "rb"))
    merged = dict(two[0]); merged.update(two[1])
    assert merged == one[0]


def _bench(*argv, env=None):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), capture_output=True, text=True, env=e, timeout=600)


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` (no launcher around it) starts two ranks through torch.distributed.run; on the CPU they run the
    dry-run step over gloo: same rank set-up, barrier, MAX-over-ranks time and whole-job aggregate as the GPU run."""
    import json
    r = _bench("--gpus", "2", "--steps", "3", "--warmup", "0", "--cpu-dry-run", "--workload", "1080p8_medium", "--frames", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["data"] == "dry-run"
    assert d["config"]["frames_owned"] == [[0, 1], [2, 3]]             # disjoint frames per rank
    # whole-job value = 2 ranks x their pixels over the SLOWEST rank's time (rank 1 sleeps twice as long as rank 0)
    assert d["ms_per_step"] >= 4.0
    assert abs(d["value"] - 2 * 2 * 1920 * 1088 * 3 / (d["ms_per_step"] * 3e-3) / 1e6) / d["value"] < 1e-3


def test_bench_gpus_flag_fails_loudly_without_enough_gpus():
    """The real (non dry-run) path must refuse N ranks on fewer than N GPUs instead of running one rank and printing n_gpus 1."""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", env={"HIP_VISIBLE_DEVICES": "0", "CUDA_VISIBLE_DEVICES": "0"})
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)


def test_world_size_must_match_gpus_flag():
    r = _bench("--gpus", "1", "--steps", "1", "--warmup", "0", "--cpu-dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
