"""N>1 path on CPU: two gloo ranks shard independent windows, solve them (with the oracle standing in for the GPU solver —
tests may use it as the checker) and gather the newest poses; the gathered result must equal the single-process run."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _uneven_worker(rank, world, port, n_seq, q):
    sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import shard
    plan = shard.Plan(n_seq, world)
    mine = torch.tensor([[float(k)] * 7 for k in range(plan.first(rank), plan.first(rank) + plan.count(rank))], dtype=torch.float64).reshape(-1, 7)
    g = shard.gather_poses(mine, dist, world, plan.counts())
    # the per-step form on persistent buffers (what bench.py runs inside every timed step): same result, step after step, no new receive buffer
    pg = shard.PoseGather(dist, world, plan.counts(), "cpu")
    ptr = pg.recv.data_ptr()
    for step in range(3):
        gs = pg(mine + step)
        assert torch.equal(gs, g + step) and pg.recv.data_ptr() == ptr
    if rank == 0:
        q.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_gather_in_global_order():
    """5 sequences on 2 ranks (3 + 2): the owner table and the padded gather agree with the partition"""
    sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
    import shard
    plan = shard.Plan(5, 2)
    assert plan.parts == [(0, 3), (3, 2)] and [plan.owner_of(k) for k in range(5)] == [0, 0, 0, 1, 1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == (5, 7) and np.array_equal(got[:, 0], np.arange(5.0))


def _worker(rank, world, port, per_gpu, q):
    for p in (ROOT, os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py
    import shard
    import synth_window as SW
    s0 = shard.first_sequence(rank, per_gpu)
    newest = []
    for k in range(s0, s0 + per_gpu):
        assert shard.owner_of(k, per_gpu) == rank
        w = SW.make_window(100 + k, oracle_py, max_features=40, n_landmarks=60)
        oracle_py.ba_solve(w, 3)
        newest.append(w["para_Pose"].reshape(-1, 7)[-1])
    g = shard.gather_poses(torch.tensor(np.stack(newest)), dist, world)
    if rank == 0:
        q.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_pose_gather():
    sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    import shard
    import synth_window as SW
    oracle_py.build()
    world, per_gpu = 2, 2
    assert shard.partition(5, 2) == [(0, 3), (3, 2)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_gpu, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = []
    for k in range(world * per_gpu):
        w = SW.make_window(100 + k, oracle_py, max_features=40, n_landmarks=60)
        oracle_py.ba_solve(w, 3)
        ref.append(w["para_Pose"].reshape(-1, 7)[-1])
    assert got.shape == (4, 7) and np.array_equal(got, np.stack(ref))
