"""GPU: which pixels share a wave (SampleKernelArgs.ticketMap, regroup_tickets_kernel; RtowContextOptions.schedulerTune[7]).

The reference hands pixels to its workers in no defined order (Schedule(W * H, 1), UNITY/Raytracer.cs:730), so the library is free to choose which 64 pixels
a wave traces together: inside super-tiles of side x side 8 x 8 tiles the pixels are sorted by the ray count of the previous launch and dealt out 64 at a
time, and the map is re-sorted behind every launch.  None of it may change a result: whole frames against the oracle - every pixel, so a map that is not a
permutation (a pixel rendered twice, or never) shows - on frames whose tile counts are no multiple of the super-tile side, with rows behind the last tile row,
slices, in-place accumulation, chains and groups, for every side, launch after launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))


def _render(rt, ctx, p, n, ins=None, stride=4):
    src = [rt.DeviceBuffer(ctx).upload(ins[k]) for k, _ in KEYS] if ins else [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
    outs = [rt.DeviceBuffer(ctx, n * c * 4) for _, c in KEYS]
    for o in outs:
        assert rt.lib.load().rtowDeviceMemset(ctx.handle, o.handle, 0xFF, o.nbytes) == 0          # NaN pattern: a pixel no ticket stands for would show
    diag = rt.DeviceBuffer(ctx, n * stride).zero()
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = src
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
    job.OutputDiagnostics = diag
    assert job.Schedule().Complete() == 0
    ctx.synchronize()
    res = {k: b.download(np.float32, (n, c)) for (k, c), b in zip(KEYS, outs)}
    res["diag"] = diag.download(np.float32, (n, stride // 4))
    for b in src + outs + [diag]:
        b.free()
    return res


def _same(a, b, what):
    for k, _ in KEYS:
        assert np.array_equal(a[k].reshape(-1).view(np.uint32), b[k].reshape(-1).view(np.uint32)), (what, k, int((a[k].reshape(len(a["scw"].reshape(-1)), -1).view(np.uint32) != b[k].reshape(len(a["scw"].reshape(-1)), -1).view(np.uint32)).any(axis=1).sum()))
    assert np.array_equal(a["diag"][:, 0], b["diag"][:, 0]), (what, "RayCount")


# (scene, width, height, spp, depth, slice offset, slice divider): 520 x 264 = 65 x 33 tiles (no multiple of 2, 4 or 8); 528 x 262: six rows behind the last tile row;
# 776 x 344 sliced in three: 97 x 14 tiles of owned rows + 2 rows behind them
FRAMES = [("cover", 520, 264, 3, 8, 0, 1), ("cover", 528, 262, 2, 8, 0, 1), ("moving", 776, 344, 2, 6, 1, 3), ("mixed", 520, 264, 2, 6, 0, 1)]


@pytest.mark.parametrize("side", [2, 3, 4, 8])      # 3: the tiles as they are, each tile's tickets most expensive first
@pytest.mark.parametrize("name,w,h,spp,depth,off,div", FRAMES)
def test_regrouped_frames_equal_the_oracle_launch_after_launch(rt, oracle, name, w, h, spp, depth, off, div, side):
    S = rt.scenes
    scene = {"cover": S.cover_scene, "moving": S.moving_scene, "mixed": S.mixed_scene}[name]()
    desc = scene.desc()
    n = w * h
    plist = [S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=s, slice_offset=off, slice_divider=div) for s in (1, 2, 3)]
    osc = oracle.OracleScene(desc)
    refs = [osc.sample_batch(p) for p in plist]
    osc.close()
    with rt.Context(0, scheduler_tune=(0, 0, 0, 0, 0, 0, 0, side, 0)) as ctx:
        ctx.upload_scene(desc)
        assert ctx.scene_info().schedulerTune[7] == side
        for i, (p, ref) in enumerate(zip(plist, refs)):      # launch 0: map from the 1-sample probe; launches 1, 2: re-sorted from the ray counts of the launch before
            gpu = _render(rt, ctx, p, n)
            owned = (np.arange(h) % div == off).repeat(w)
            assert not np.isnan(gpu["scw"][owned]).any(), "an owned pixel was not written"
            assert np.isnan(gpu["scw"][~owned]).all(), "a pixel of another slice was written"
            for k, c in KEYS:
                assert np.array_equal(gpu[k].reshape(n, -1)[owned].view(np.uint32), ref[k].reshape(n, -1)[owned].view(np.uint32)), (name, side, "launch", i, k)
            assert np.array_equal(gpu["diag"][owned, 0], ref["diag"][owned, 0])


@pytest.mark.parametrize("side", [2, 3, 8])
def test_regrouped_chains_and_groups_equal_the_separate_batches(rt, side):
    """Chains hand a chunk's pixels from batch to batch through per-chunk counters (per TICKET chunk: the map only says which pixels those are); groups hand out
    (chunk, batch) pairs.  Both against the same batches as separate launches of a context that keeps the tiles as they are."""
    S = rt.scenes
    scene = S.cover_scene()
    desc = scene.desc()
    w, h = 520, 264
    n = w * h
    plist = [S.make_params(scene, w, h, spp=2, trace_depth=8, seed=s) for s in (5, 6, 7, 8)]
    with rt.Context(0, scheduler_tune=(0, 0, 0, 0, 0, 0, 0, 1, 0)) as plain:
        plain.upload_scene(desc)
        separate = []
        acc = None
        for p in plist:
            acc = _render(rt, plain, p, n, ins=acc)
            separate.append(acc)
        from_zero = [_render(rt, plain, p, n) for p in plist]
    with rt.Context(0, scheduler_tune=(0, 0, 0, 0, 0, 0, 0, side, 0)) as ctx:
        ctx.upload_scene(desc)
        for rep in range(2):                                  # the second chain / group runs under a map sorted from the first one's ray counts
            bufs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
            diags = [rt.DeviceBuffer(ctx, n * 4).zero() for _ in plist]
            rt.lib.check(rt.sample_batch_chain_device(ctx, plist, bufs, bufs, diags), "rtowSampleBatchChainDevice")
            ctx.synchronize()
            chained = {k: b.download(np.float32, (n, c)) for (k, c), b in zip(KEYS, bufs)}
            chained["diag"] = diags[-1].download(np.float32, (n, 1))
            _same(chained, separate[-1], ("chain", side, rep))
            src = [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
            outs = [[rt.DeviceBuffer(ctx, n * c * 4) for _, c in KEYS] for _ in plist]
            assert rt.sample_batch_group_device(ctx, plist, src, outs, diags) == 0
            ctx.synchronize()
            for b, (o, d) in enumerate(zip(outs, diags)):
                g = {k: x.download(np.float32, (n, c)) for (k, c), x in zip(KEYS, o)}
                g["diag"] = d.download(np.float32, (n, 1))
                _same(g, from_zero[b], ("group", side, rep, b))
            for x in bufs + diags + src + [y for o in outs for y in o]:
                x.free()


def test_adaptive_sample_counts_under_a_regrouped_map(rt, oracle):
    """sampleCountRange (1, 12) with the weights of a first batch: pixels differ in how many samples they take (JOBS/SampleBatchJob.cs:118-126); the map sorts by rays."""
    S = rt.scenes
    scene = S.cover_scene()
    desc = scene.desc()
    w, h = 520, 264
    n = w * h
    p0 = S.make_params(scene, w, h, spp=4, trace_depth=8, seed=1)
    osc = oracle.OracleScene(desc)
    first = osc.sample_batch(p0)
    weights = first["scw"].reshape(-1) / np.maximum(first["color"].reshape(n, 4)[:, 3], 1)
    p1 = S.make_params(scene, w, h, spp=1, spp_max=12, trace_depth=8, seed=2, extrema=(float(weights.min()), float(weights.max())))
    ref = osc.sample_batch(p1, {k: first[k] for k, _ in KEYS})
    osc.close()
    with rt.Context(0, scheduler_tune=(0, 0, 0, 0, 0, 0, 0, 4, 0)) as ctx:
        ctx.upload_scene(desc)
        a = _render(rt, ctx, p0, n)
        b = _render(rt, ctx, p1, n, ins=a)
        _same(b, ref, "adaptive")
        counts = b["color"][:, 3] - a["color"][:, 3]
        assert counts.min() < counts.max()                    # the batch really was non-uniform
