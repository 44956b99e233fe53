"""world_size-2 test of the multi-GPU host path on CPU (gloo): row-interleaved slices + one gather == the full frame.

The partition / pack / gather / interleave code is the product's (raytracing-in-one-weekend_amd/multigpu.py, the same
functions bench.py calls under RCCL); the CPU checker stands in for the kernel because the HIP path needs a GPU."""
import pytest
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH = 48, 27, 2, 6   # odd height: ranks own a different number of rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    import importlib
    sys.path.insert(0, ROOT)
    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.scenes.cover_scene()
    osc = ob.OracleScene(scene.desc())
    params = rt.scenes.make_params(scene, W, H, spp=SPP, trace_depth=DEPTH)

    def render_slice(p):
        assert (p.sliceOffset, p.sliceDivider) == (rank, world)
        ins = ob.zero_buffers(W * H)
        ins["color"][:] = -5.0                      # pixels this rank does not own must stay untouched
        r = osc.sample_batch(p, ins, nthreads=2)
        rows = np.repeat(np.arange(H) % world == rank, W)
        assert np.all(r["color"][~rows] == -5.0)
        r["color"][rows] -= 0.0
        return torch.from_numpy(r["color"])

    # inputs are -5 everywhere, so owned pixels hold (-5 + sum); undo the offset after the gather
    frame = mg.render_partitioned(render_slice, params, H, W, rank, world)
    if rank == 0:
        assert frame.shape == (H, W, 4)
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    assert list(mg.owned_rows(rank, world, H)) == list(range(rank, H, world))
    dist.barrier()
    dist.destroy_process_group()
    osc.close()


def test_two_rank_partition_and_gather_reproduces_the_full_frame(rt, oracle, tmp_path):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    frame = np.load(out).reshape(W * H, 4)
    scene = rt.scenes.cover_scene()
    osc = oracle.OracleScene(scene.desc())
    ins = oracle.zero_buffers(W * H)
    ins["color"][:] = -5.0
    full = osc.sample_batch(rt.scenes.make_params(scene, W, H, spp=SPP, trace_depth=DEPTH), ins)
    osc.close()
    assert np.array_equal(frame.view(np.uint32), full["color"].view(np.uint32))   # bit-identical to the single-rank frame


def test_pack_and_gather_single_rank_identity(rt):
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    x = torch.arange(5 * 3 * 4, dtype=torch.float32).view(5, 3, 4)
    assert torch.equal(mg.gather_frame(mg.pack_owned(x, 0, 1), 5, 0, 1), x)
    assert [list(mg.owned_rows(r, 3, 7)) for r in range(3)] == [[0, 3, 6], [1, 4], [2, 5]]
    p = rt.abi.SampleParams()
    mg.slice_params(p, 2, 8)
    assert (p.sliceOffset, p.sliceDivider) == (2, 8)


def _batch_worker(rank, world, port, out_path):
    import importlib
    sys.path.insert(0, ROOT)
    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.scenes.cover_scene()
    osc = ob.OracleScene(scene.desc())
    n = W * H
    spp_total, step_seed = 5, 3                       # 5 samples over 2 ranks: 3 + 2 (ragged split)

    m, padded = mg.slice_floats(n, world), mg.padded_floats(n, world)

    def render_full():
        p = rt.scenes.make_params(scene, W, H, spp=mg.batch_split(spp_total, rank, world), trace_depth=DEPTH, seed=mg.batch_seed(step_seed, rank, world))
        r = osc.sample_batch(p, ob.zero_buffers(n), nthreads=2)
        flat = torch.zeros(padded)
        flat[:mg.ACCUM_FLOATS * n] = torch.from_numpy(np.concatenate([r["color"].ravel(), r["normal"].ravel(), r["albedo"].ravel(), r["scw"].ravel()]))
        return flat

    def add_flat(dst, src):                            # stands in for rtowAddAccumDevice on CPU
        dst += src

    acc_slice = torch.full((m,), 0.25)                 # this rank's slice of a non-trivial running accumulation
    res = mg.render_batches(render_full, acc_slice, n, rank, world, add_flat)
    if rank == 0:
        res = res[:mg.ACCUM_FLOATS * n]
    if rank == 0:
        np.save(out_path, res.numpy())
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()
    osc.close()


@pytest.mark.parametrize("world", [2, 3])
def test_batch_parallel_equals_ordered_sum_of_batches(rt, oracle, tmp_path, world):
    """All-to-all + per-rank ordered fold of one slice + gather == folding the whole partials in rank order on one rank; world 3 does
    not divide the pixel count (padded slices)."""
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    out = str(tmp_path / "acc.npy")
    mp.spawn(_batch_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    n = W * H
    scene = rt.scenes.cover_scene()
    osc = oracle.OracleScene(scene.desc())
    want = np.full(mg.ACCUM_FLOATS * n, 0.25, np.float32)
    total = 0
    for r in range(world):                             # rank order, float32 adds: the order every slice is folded in
        spp = mg.batch_split(5, r, world)
        total += spp
        p = rt.scenes.make_params(scene, W, H, spp=spp, trace_depth=DEPTH, seed=mg.batch_seed(3, r, world))
        b = osc.sample_batch(p, oracle.zero_buffers(n))
        want = want + np.concatenate([b["color"].ravel(), b["normal"].ravel(), b["albedo"].ravel(), b["scw"].ravel()])
    osc.close()
    assert total == 5 and [mg.batch_seed(3, r, world) for r in range(world)] == [2 * world + 1 + r for r in range(world)]
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    counts = got[:4 * n].reshape(n, 4)[:, 3] - 0.25
    assert counts.max() == 5                                             # every sub-batch landed in every sky pixel
    assert mg.slice_floats(n, world) % mg.ACCUM_FLOATS == 0 and mg.padded_floats(n, world) >= mg.ACCUM_FLOATS * n


def test_accum_views_are_one_contiguous_block(rt):
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    flat = torch.arange(mg.ACCUM_FLOATS * 6, dtype=torch.float32)
    c, nn, a, s = mg.accum_views(flat, 6)
    assert c.shape == (6, 4) and nn.shape == (6, 3) and a.shape == (6, 3) and s.shape == (6,)
    assert c.data_ptr() == flat.data_ptr() and s[-1] == flat[-1]
    c[0, 0] = -1
    assert flat[0] == -1
    assert [mg.batch_split(256, r, 8) for r in range(8)] == [32] * 8 and sum(mg.batch_split(10, r, 4) for r in range(4)) == 10


# ---------------------------------------------------------------------------------------------------
# tiles x batches (rtowHybridPlan / rtowExchangeAccumDevice): the host logic over gloo, the CPU checker standing in for the kernel
# ---------------------------------------------------------------------------------------------------
HYBRID_SPP, HYBRID_STEPS = 5, 2


def _hybrid_worker(rank, world, port, tiles, out_path):
    import importlib
    sys.path.insert(0, ROOT)
    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.scenes.cover_scene()
    osc = ob.OracleScene(scene.desc())
    n = W * H
    accum = [torch.full((n, 4), 0.5), torch.full((n, 3), 0.25), torch.full((n, 3), 0.125), torch.full((n,), 2.0)]   # a non-trivial running accumulation
    accum[0][:, 3] = 3.0                                   # ... whose success count is a whole number, as every real one is (the job reads it back as an int)

    def render_partial(plan):
        p = rt.scenes.make_params(scene, W, H, spp=plan["samples"], trace_depth=DEPTH, seed=plan["seed"], slice_offset=plan["slice_offset"], slice_divider=plan["slice_divider"])
        ins = ob.zero_buffers(n)
        r = osc.sample_batch(p, ins, nthreads=2)
        out = [torch.from_numpy(r[k].copy()) for k in ("color", "normal", "albedo", "scw")]
        rows = torch.from_numpy(np.repeat(np.arange(H) % plan["slice_divider"] != plan["slice_offset"], W))
        for t in out:
            t[rows] = float("nan")                         # rows outside this rank's tile must never be read by the exchange
        return out

    frame = None
    for step in range(1, HYBRID_STEPS + 1):
        frame = mg.render_hybrid(render_partial, accum, H, W, rank, world, tiles, HYBRID_SPP, step)
    # every rank's own rows of all four buffers, for the check of the distributed accumulation
    np.savez(out_path + ".rank%d.npz" % rank, **{k: t.numpy() for k, t in zip(("color", "normal", "albedo", "scw"), accum)})
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()
    osc.close()


@pytest.mark.parametrize("world,tiles", [(2, 1), (2, 2), (3, 1), (4, 2), (6, 2), (6, 3)])
def test_hybrid_tiles_x_batches_equals_the_ordered_fold_of_the_reference_batches(rt, oracle, tmp_path, world, tiles):
    """G = T x B ranks, two steps: the gathered colour frame and every rank's rows of all four accumulators equal - bit for bit - the oracle's B
    sub-batches per step folded in group order on top of the running accumulation, and agree with the reference's own SEQUENTIAL accumulation of the
    same batches (each batch on top of its predecessor, UNITY/Raytracer.cs:798-802) to 1e-4 on the mean colour.  T = 2, B = 1 is the pure tile
    partition (nothing travels), T = 1 the pure batch partition; 5 samples over 2 or 3 groups split raggedly."""
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    out = str(tmp_path / "frame.npy")
    mp.spawn(_hybrid_worker, args=(world, _free_port(), tiles, out), nprocs=world, join=True)
    n = W * H
    groups = world // tiles
    scene = rt.scenes.cover_scene()
    osc = oracle.OracleScene(scene.desc())
    keys = ("color", "normal", "albedo", "scw")
    want = {"color": np.full((n, 4), 0.5, np.float32), "normal": np.full((n, 3), 0.25, np.float32), "albedo": np.full((n, 3), 0.125, np.float32), "scw": np.full(n, 2.0, np.float32)}
    want["color"][:, 3] = 3.0
    seq = {k: v.copy() for k, v in want.items()}
    seeds = []
    for step in range(1, HYBRID_STEPS + 1):
        for g in range(groups):
            plan = mg.hybrid_plan(world, g * tiles, tiles, HYBRID_SPP, step)
            seeds.append(plan["seed"])
            p = rt.scenes.make_params(scene, W, H, spp=plan["samples"], trace_depth=DEPTH, seed=plan["seed"])
            b = osc.sample_batch(p, oracle.zero_buffers(n))
            for k in keys:
                want[k] = want[k] + b[k].reshape(want[k].shape)                       # group order, one float32 rounding per add
            seq = {k: v for k, v in osc.sample_batch(p, {k: seq[k] for k in keys}).items() if k in keys}   # the reference: this batch on top of its predecessor
    osc.close()
    assert seeds == list(range(1, HYBRID_STEPS * groups + 1))                         # consecutive Seeds over groups and steps, like frameSeed
    assert sum(mg.hybrid_plan(world, g * tiles, tiles, HYBRID_SPP, 1)["samples"] for g in range(groups)) == HYBRID_SPP
    frame = np.load(out).reshape(n, 4)
    assert np.array_equal(frame.view(np.uint32), want["color"].view(np.uint32))
    rows = np.arange(n) // W
    for r in range(world):
        got = np.load(out + ".rank%d.npz" % r)
        mine = rows % world == r
        for k in keys:
            assert np.array_equal(got[k][mine].view(np.uint32), want[k][mine].view(np.uint32)), (r, k)
    # against the sequential accumulation: same samples, another association of the float sums
    cnt = np.maximum(frame[:, 3:4] - 3.0, 1)
    assert np.array_equal(frame[:, 3], seq["color"][:, 3])                            # success counts are integers: exact either way
    assert np.abs((frame[:, :3] - 0.5) / cnt - (seq["color"][:, :3] - 0.5) / cnt).max() <= 1e-4


def test_hybrid_plan_matches_the_c_abi_contract(rt):
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    assert mg.default_tiles(8, 256) == 1 and mg.default_tiles(8, 4) == 2 and mg.default_tiles(8, 1) == 8 and mg.default_tiles(6, 4) == 2 and mg.default_tiles(1, 256) == 1
    p = mg.hybrid_plan(8, 5, 2, 10, 3)                                                # rank 5 = tile 1 + 2 * group 2 of T = 2, B = 4
    assert (p["tile"], p["group"], p["slice_offset"], p["slice_divider"], p["groups"]) == (1, 2, 1, 2, 4)
    assert p["samples"] == 2 and p["seed"] == 2 * 4 + 2 + 1                           # 10 samples over 4 groups: 3 3 2 2
    with pytest.raises(ValueError):
        mg.hybrid_plan(8, 0, 3, 10, 1)
