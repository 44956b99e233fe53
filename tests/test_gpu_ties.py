"""GPU: the nearest-hit tie fix-up of the rank-rule sphere kernels.

Hits at bit-identical distance are ordered by the reference's sort of the whole hit list (JOBS/SampleBatchJob.cs:473-474: an unstable introsort once the list is longer
than 16, RT/HitRecord.cs:22-25).  The sphere kernels keep only the nearest hit and break a tie by leaf rank - exact for rays of at most 16 hits.  Since round 4 a lane
that meets such a tie between two DIFFERENT spheres marks its pixel in a bitmap, and after the launch the exact-tie kernel of the same kind renders the marked pixels
again - every batch of a chain or group - from the launch's inputs (a copy of them when the launch accumulates in place), over what the fast kernel stored.  Scene: two spheres mirrored about the plane x = 0 - every
camera ray in that plane (the centre column of an odd-width frame, jitter off) meets both at bit-identical distance - in front of a row of spheres the same rays thread,
so that the hit list has more than 16 entries."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))
W, H = 33, 33


def _scene(rt, behind):
    S = rt.scenes
    s = S.Scene("mirrored pair in front of a row")
    s.add_sphere((-0.3, 0.0, 5.0), 0.5, S.lambertian((0.9, 0.1, 0.1)))
    s.add_sphere((0.3, 0.0, 5.0), 0.5, S.metal((0.2, 0.9, 0.3), 0.0))
    for k in range(behind):
        s.add_sphere((0.0, 0.0, 3.5 - 1.0 * k), 0.3, S.lambertian((0.2 + 0.03 * k, 0.4, 0.8 - 0.03 * k)))
    s.camera = {"position": [0.0, 0.0, 10.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 12.0, "aperture": 0.0}
    return s


def _oracle_batches(oracle, scene, plist, start):
    osc = oracle.OracleScene(scene.desc())
    acc = {k: v.copy() for k, v in start.items()}
    diags = []
    for p in plist:
        r = osc.sample_batch(p, acc)
        acc = {k: r[k] for k, _ in KEYS}
        diags.append(r["diag"])
    osc.close()
    return acc, diags


def _start(n, seed=3):
    rng = np.random.default_rng(seed)
    ins = {"color": rng.random((n, 4)).astype(np.float32), "normal": rng.normal(size=(n, 3)).astype(np.float32),
           "albedo": rng.random((n, 3)).astype(np.float32), "scw": rng.random(n).astype(np.float32)}
    ins["color"][:, 3] = rng.integers(0, 4, n)
    return ins


def _differs(a, b):
    return int(sum((a[k].reshape(b[k].shape).view(np.uint32) != b[k].view(np.uint32)).any(axis=-1).sum() if b[k].ndim > 1 else (a[k].reshape(b[k].shape).view(np.uint32) != b[k].view(np.uint32)).sum()
                   for k, _ in KEYS))


def test_tied_pixels_are_rendered_by_the_exact_kernel_and_equal_the_oracle(rt, oracle):
    """Plain batches in place (inputs = outputs, non-zero), default flags: every output equals the oracle for hit lists of 19 .. 26 entries; with the fix-up switched off
    (RTOW_CONTEXT_EXACT_TIES_NEVER: the rank rule everywhere) at least one of those scenes differs - the scene does exercise the exception."""
    n = W * H
    teeth = 0
    for behind in range(17, 25):
        scene = _scene(rt, behind)
        desc = scene.desc()
        plist = [rt.scenes.make_params(scene, W, H, spp=5, trace_depth=6, seed=s, jitter=False, focus=5.0, diagnostics_stride=4) for s in (11, 12)]
        start = _start(n)
        want, wdiag = _oracle_batches(oracle, scene, plist, start)
        for flags, exact in ((0, True), (rt.abi.CONTEXT_EXACT_TIES_NEVER, False)):
            with rt.Context(0, flags=flags) as ctx:
                ctx.upload_scene(desc)
                info = ctx.scene_info()
                assert info.hitSpillBytes == 0                                   # the rank-rule kernels: not the exact-tie variants (which keep whole hit lists)
                # what a device-resident caller compares across RTOW_ERROR_CAPACITY: the capacity of the fix-up pass's lists where that pass exists
                assert info.hitListCapacity == (0 if flags else min(info.entityCount, 128))
                acc = {k: v.copy() for k, v in start.items()}
                diags = []
                for p in plist:
                    r = rt.sample_batch_host(ctx, p, acc)
                    acc = {k: r[k] for k, _ in KEYS}
                    diags.append(r["diag"])
                d = _differs(acc, want)
                if exact:
                    assert d == 0, (behind, "default flags", d)
                    for got, wd in zip(diags, wdiag):
                        assert np.array_equal(got[:, 0], wd[:, 0]), (behind, "ray counts")
                else:
                    teeth += 1 if d else 0
    assert teeth >= 1, "the rank rule alone agreed with the reference's sort in every scene: the test has lost its teeth"


@pytest.mark.parametrize("mode", ["chain", "group"])
def test_tied_pixels_in_chains_and_groups(rt, oracle, mode):
    """A chain of four batches in place: a pixel listed by batch b is left alone by the later batches of the launch and carried through all of them by the fix-up launch;
    a group of four: each (pixel, batch) on its own.  Device-resident buffers, 16-byte records."""
    n = W * H
    for behind in (18, 21, 23):
        scene = _scene(rt, behind)
        desc = scene.desc()
        plist = [rt.scenes.make_params(scene, W, H, spp=3 + (s % 2), trace_depth=5, seed=s, jitter=False, focus=5.0, diagnostics_stride=16) for s in (21, 22, 23, 24)]
        if mode == "chain":
            for p in plist:
                p.sampleCountRange[0] = p.sampleCountRange[1] = 4                    # a chain's batches differ in nothing but Seed
        start = _start(n, 8)
        with rt.Context(0) as ctx:
            ctx.upload_scene(desc)
            if mode == "chain":
                bufs = [rt.DeviceBuffer(ctx).upload(start[k]) for k, _ in KEYS]
                diags = [rt.DeviceBuffer(ctx, n * 16).zero() for _ in plist]
                rt.lib.check(rt.sample_batch_chain_device(ctx, plist, bufs, bufs, diags), "rtowSampleBatchChainDevice")
                ctx.synchronize()
                got = {k: b.download(np.float32, (n, c)) for (k, c), b in zip(KEYS, bufs)}
                want, wdiag = _oracle_batches(oracle, scene, plist, start)
                assert _differs(got, want) == 0, (behind, "chain")
                for d, wd in zip(diags, wdiag):
                    assert np.array_equal(d.download(np.float32, (n, 4))[:, 0], wd[:, 0]), (behind, "chain ray counts")
            else:
                for p in plist:
                    p.sampleCountRange[0] = p.sampleCountRange[1] = 4
                src = [rt.DeviceBuffer(ctx).upload(start[k]) for k, _ in KEYS]
                outs = [[rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS] for _ in plist]
                rt.lib.check(rt.sample_batch_group_device(ctx, plist, src, outs), "rtowSampleBatchGroupDevice")
                ctx.synchronize()
                for k, p in enumerate(plist):
                    want, _ = _oracle_batches(oracle, scene, [p], start)
                    got = {key: b.download(np.float32, (n, c)) for (key, c), b in zip(KEYS, outs[k])}
                    assert _differs(got, want) == 0, (behind, "group batch", k)
            ctx.batch_status()


def test_fallback_aovs_with_and_without_earlier_successes(rt, oracle):
    """trace depth 1: every path that meets a surface fails, so pixels over geometry end a first batch with no successful sample and the AOVs of their failed
    sample 0 as fallback (JOBS/SampleBatchJob.cs:152-156,160-161); the early fallback store is skipped where earlier batches already succeeded (the record is
    overwritten at the end of the pixel anyway).  Cover scene (486 spheres: a watched scene), in place, two batches, then a third on top of successes."""
    scene = rt.scenes.cover_scene()
    w, h = 120, 68
    n = w * h
    plist = [rt.scenes.make_params(scene, w, h, spp=3, trace_depth=1, seed=s) for s in (5, 6)] + [rt.scenes.make_params(scene, w, h, spp=3, trace_depth=8, seed=7),
                                                                                                  rt.scenes.make_params(scene, w, h, spp=3, trace_depth=1, seed=8)]
    start = {k: np.zeros((n, c) if c > 1 else (n,), np.float32) for k, c in KEYS}
    want, _ = _oracle_batches(oracle, scene, plist, start)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene.desc())
        acc = {k: v.copy() for k, v in start.items()}
        for p in plist:
            r = rt.sample_batch_host(ctx, p, acc)
            acc = {k: r[k] for k, _ in KEYS}
    assert _differs(acc, want) == 0
    first = _oracle_batches(oracle, scene, plist[:2], start)[0]
    assert (first["color"][:, 3] == 0).sum() > n // 4 and np.abs(first["normal"]).sum() > 0          # after two batches: many pixels without a success, their fallback normals stored


@pytest.mark.parametrize("flags", [0, "always"])
def test_tie_lists_longer_than_the_default_capacity_grow(rt, oracle, flags):
    """142 spheres on the centre column's rays: longer than the 128 entries the exact-tie procedure's lists start with outside volume scenes.  The fix-up launch (default
    flags) or the exact-tie kernels themselves (RTOW_CONTEXT_EXACT_TIES_ALWAYS) flag the ray, the lists double up to the scene's 142 entities, the host-buffer call runs
    the batch again by itself - the reference's list simply grows (UTIL/HybridCollections.cs:65-71)."""
    flags = rt.abi.CONTEXT_EXACT_TIES_ALWAYS if flags == "always" else flags
    scene = _scene(rt, 140)
    desc = scene.desc()
    n = W * H
    p = rt.scenes.make_params(scene, W, H, spp=3, trace_depth=4, seed=21, jitter=False, focus=5.0, diagnostics_stride=4)
    start = _start(n)
    want, wdiag = _oracle_batches(oracle, scene, [p], start)
    with rt.Context(0, flags=flags) as ctx:
        ctx.upload_scene(desc)
        r = rt.sample_batch_host(ctx, p, {k: v.copy() for k, v in start.items()})
        assert _differs({k: r[k] for k, _ in KEYS}, want) == 0
        assert np.array_equal(r["diag"][:, 0], wdiag[0][:, 0])
        if flags:
            assert ctx.scene_info().hitListCapacity == 142


# ---- round 6: all-triangle scenes are watched the same way (the reference's live host makes one entity per mesh triangle, UNITY/Raytracer.cs:1193-1198) ----
TW, TH = 96, 64


def _triangle_frames(rt, oracle, scene, flags, mode, seeds=(31, 32, 33), spp=3, depth=6):
    n = TW * TH
    desc = scene.desc()
    plist = [rt.scenes.make_params(scene, TW, TH, spp=spp, trace_depth=depth, seed=s, diagnostics_stride=4) for s in seeds]
    start = _start(n, 5)
    with rt.Context(0, flags=flags) as ctx:
        ctx.upload_scene(desc)
        if mode == "plain":
            acc = {k: v.copy() for k, v in start.items()}
            for p in plist:
                r = rt.sample_batch_host(ctx, p, acc)
                acc = {k: r[k] for k, _ in KEYS}
            got = [acc]
            want = [_oracle_batches(oracle, scene, plist, start)[0]]
        elif mode == "chain":
            bufs = [rt.DeviceBuffer(ctx).upload(start[k]) for k, _ in KEYS]
            rt.lib.check(rt.sample_batch_chain_device(ctx, plist, bufs, bufs), "rtowSampleBatchChainDevice")
            ctx.synchronize()
            got = [{k: b.download(np.float32, (n, c)) for (k, c), b in zip(KEYS, bufs)}]
            want = [_oracle_batches(oracle, scene, plist, start)[0]]
        else:
            src = [rt.DeviceBuffer(ctx).upload(start[k]) for k, _ in KEYS]
            outs = [[rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS] for _ in plist]
            rt.lib.check(rt.sample_batch_group_device(ctx, plist, src, outs), "rtowSampleBatchGroupDevice")
            ctx.synchronize()
            got = [{key: b.download(np.float32, (n, c)) for (key, c), b in zip(KEYS, outs[k])} for k in range(len(plist))]
            want = [_oracle_batches(oracle, scene, [p], start)[0] for p in plist]
        ctx.batch_status()
    return sum(_differs(g, w) for g, w in zip(got, want))


@pytest.mark.parametrize("mode", ["plain", "chain", "group"])
def test_watched_triangle_scenes_equal_the_oracle(rt, oracle, mode):
    """Walls of triangles, 23 hits per ray, two different pairs of triangles in the nearest plane: by default the rank-rule triangle kernels trace the frame and the
    exact-tie kernels only the pixels that met a tie (tieWatchOk: no triangle twice); with every pixel on the exact-tie kernels (RTOW_CONTEXT_EXACT_TIES_ALWAYS) the
    same frames; with the rank rule alone (RTOW_CONTEXT_EXACT_TIES_NEVER) the scene must differ somewhere - or it does not exercise the fix-up."""
    scene = rt.scenes.triangle_layers_scene()
    assert _triangle_frames(rt, oracle, scene, 0, mode) == 0
    assert _triangle_frames(rt, oracle, scene, rt.abi.CONTEXT_EXACT_TIES_ALWAYS, mode) == 0
    if mode == "plain":
        assert _triangle_frames(rt, oracle, scene, rt.abi.CONTEXT_EXACT_TIES_NEVER, mode) > 0, "the rank rule alone agreed with the reference's sort: the test has lost its teeth"


def test_a_triangle_scene_with_the_same_triangle_twice_keeps_the_exact_kernels(rt, oracle):
    """Coinciding triangles tie over whole regions: no watch (SceneLayout.tieWatchOk = 0), every pixel through the exact-tie kernels, as before round 6."""
    assert _triangle_frames(rt, oracle, rt.scenes.triangle_layers_scene(duplicate=True), 0, "plain") == 0


def test_a_triangle_scene_that_ties_often_moves_to_the_exact_kernels(rt, oracle):
    """The first watched launch of a frame in which thousands of pixels tie (more than 8 workgroups should render: kTieWatchBusy) is still correct, and sends the
    scene to its exact-tie kernels for the launches after it; frames equal the oracle before and after."""
    scene = rt.scenes.triangle_layers_scene()
    w, h = 320, 200
    n = w * h
    desc = scene.desc()
    plist = [rt.scenes.make_params(scene, w, h, spp=1, trace_depth=3, seed=s, diagnostics_stride=4) for s in (41, 42, 43)]
    start = {k: np.zeros((n, c) if c > 1 else (n,), np.float32) for k, c in KEYS}
    want, _ = _oracle_batches(oracle, scene, plist, start)
    with rt.Context(0) as ctx:
        ctx.upload_scene(desc)
        acc = {k: v.copy() for k, v in start.items()}
        for p in plist:
            r = rt.sample_batch_host(ctx, p, acc)
            acc = {k: r[k] for k, _ in KEYS}
    assert _differs(acc, want) == 0
