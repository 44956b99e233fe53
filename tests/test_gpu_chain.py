"""GPU: chained batches (rtowSampleBatchChainDevice, include/rtow.h).

The reference keeps two sample batches in flight, the second depending on the first (UNITY/Raytracer.cs:586-593); the chain entry point
takes `count` such batches at once and - when they differ only in Seed - runs them as ONE launch in which batch k + 1 of a 64-pixel chunk
starts as soon as batch k of that chunk is stored.  Defined result: the batches one after the other.  Every test compares the chain, bit
for bit, with that sequence (and one with the oracle), across workgroups / XCDs handing accumulators to each other inside the kernel."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))


def _zero_bufs(rt, ctx, n):
    return [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]


def _download(bufs, n):
    return {k: b.download(np.float32, (n, c)) for (k, c), b in zip(KEYS, bufs)}


def _params(rt, scene, w, h, spp, depth, seeds, **kw):
    return [rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=s, **kw) for s in seeds]


def _sequential(rt, ctx, plist, n, stride, start=None):
    bufs = _zero_bufs(rt, ctx, n) if start is None else start
    diags = []
    for p in plist:
        d = rt.DeviceBuffer(ctx, n * stride).zero()
        job = rt.SampleBatchJob(ctx, p)
        job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
        job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = bufs
        job.OutputDiagnostics = d
        assert job.Schedule().Complete() == 0
        diags.append(d)
    ctx.synchronize()
    out = _download(bufs, n)
    out["diag"] = [d.download(np.float32, (n, stride // 4)) for d in diags]
    for b in bufs + diags:
        b.free()
    return out


def _chained(rt, ctx, plist, n, stride, separate_input=False):
    outs = _zero_bufs(rt, ctx, n)
    ins = _zero_bufs(rt, ctx, n) if separate_input else outs
    diags = [rt.DeviceBuffer(ctx, n * stride).zero() for _ in plist]
    assert rt.sample_batch_chain_device(ctx, plist, ins, outs, diags) == 0
    ctx.synchronize()
    out = _download(outs, n)
    out["diag"] = [d.download(np.float32, (n, stride // 4)) for d in diags]
    for b in set(outs + ins + diags):
        b.free()
    return out


def _same(a, b, what):
    for k, _ in KEYS:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), (what, k, int((a[k].view(np.uint32) != b[k].view(np.uint32)).any(axis=-1).sum()))
    assert len(a["diag"]) == len(b["diag"])
    for i, (x, y) in enumerate(zip(a["diag"], b["diag"])):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (what, "diagnostics of batch", i)


@pytest.mark.parametrize("name,w,h,spp,depth,count", [("cover", 1920, 1080, 4, 8, 3), ("cover", 96, 54, 6, 8, 16), ("moving", 640, 360, 5, 8, 4),
                                                       ("mixed", 320, 200, 4, 6, 5), ("volumes", 256, 144, 3, 10, 4), ("textured", 256, 144, 3, 6, 3),
                                                       ("stress", 960, 540, 3, 8, 3), ("twins", 320, 180, 4, 8, 3), ("mesh", 320, 200, 3, 6, 3), ("decals", 256, 256, 3, 6, 3)])
def test_chain_equals_the_batches_one_after_the_other(rt, gpu_context, name, w, h, spp, depth, count):
    S = rt.scenes
    scene = {"cover": S.cover_scene, "moving": S.moving_scene, "mixed": S.mixed_scene, "volumes": S.volume_scene, "textured": S.textured_scene,
             "stress": lambda: S.stress_scene(count=6000, max_tentatives=30000),
             "twins": S.twin_spheres_scene, "mesh": S.mesh_scene, "decals": lambda: S.decal_stack_scene(20)}[name]()       # the last three: exact-tie kernels
    ctx = gpu_context
    ctx.upload_scene(scene.desc())
    n = w * h
    plist = _params(rt, scene, w, h, spp, depth, [100 + 7 * k for k in range(count)], diagnostics_stride=16, focus=6.0 if name != "cover" else None)
    seq = _sequential(rt, ctx, plist, n, 16)
    for attempt in range(2):                                      # second attempt: chunk order from the measured cost map
        _same(_chained(rt, ctx, plist, n, 16, separate_input=attempt == 1), seq, (name, attempt))
    assert seq["color"][:, 3].max() == spp * count


def test_chain_of_slices_and_the_oracle(rt, oracle, gpu_context):
    """Interlaced slice (rows the chain must not touch), 4-byte diagnostics, and the CPU oracle as the referee of the accumulated result."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    desc = scene.desc()
    ctx.upload_scene(desc)
    w, h, spp, depth = 128, 72, 3, 8
    n = w * h
    plist = _params(rt, scene, w, h, spp, depth, [5, 6, 7, 8], slice_offset=2, slice_divider=3)
    got = _chained(rt, ctx, plist, n, 4)
    osc = oracle.OracleScene(desc)
    ref = None
    for i, p in enumerate(plist):
        ref = osc.sample_batch(p, None if ref is None else {k: ref[k] for k, _ in KEYS})
        assert np.array_equal(got["diag"][i][(np.arange(n) // w) % 3 == 2, 0], ref["diag"][(np.arange(n) // w) % 3 == 2, 0]), i
    osc.close()
    for k, _ in KEYS:
        assert np.array_equal(got[k].reshape(-1).view(np.uint32), ref[k].reshape(-1).view(np.uint32)), k


@pytest.mark.parametrize("name,stride,depth", [("cover", 16, 32), ("cover", 4, 64), ("moving", 16, 32)])
def test_deep_plain_and_chained_launches_with_lanes_in_a_hurry(rt, oracle, gpu_context, name, stride, depth):
    """The reference host's committed configuration (trace depth 32, 50 samples per batch; 16-byte records or RayCount only) as plain launches and as a chain.  At this depth a
    pixel in a few thousand - rays trapped in glass - takes ten times the rays of the mean; plain and chained launches of a static-sphere scene let a pixel that runs at more
    than 18 rays per sample stop waiting for company, and use thresholds of their own (csrc/rtow_sample_kernel.hip.h HURRY, csrc/rtow_api.hip) - scheduling only: every form
    equals the oracle's batches in sequence.  The frame must hold such a pixel for the test to mean anything: a first sample of more than 18 x 2 rays is in a hurry at once
    (the oracle's 1-sample batch with the same Seed traces that very sample).  Moving spheres have no such variants and run the same test."""
    ctx = gpu_context
    scene = {"cover": rt.scenes.cover_scene, "moving": rt.scenes.moving_scene}[name]()
    desc = scene.desc()
    ctx.upload_scene(desc)
    w, h, spp, count = 320, 180, 50, 3
    n = w * h
    focus = 6.0 if name != "cover" else None
    plist = _params(rt, scene, w, h, spp, depth, [31, 32, 33], diagnostics_stride=stride, focus=focus)
    osc = oracle.OracleScene(desc)
    ref, ref_diag = None, []
    for p in plist:
        ref = osc.sample_batch(p, None if ref is None else {k: ref[k] for k, _ in KEYS})
        ref_diag.append(ref["diag"])
    first = max(osc.sample_batch(q)["diag"][:, 0].max() for q in _params(rt, scene, w, h, 1, depth, [31, 32, 33], diagnostics_stride=stride, focus=focus))
    osc.close()
    if depth == 64:
        assert first > 36, "no first sample beyond 36 rays: nothing was in a hurry from its first sample on"

    def check(got, what):
        for k, c in KEYS:
            assert np.array_equal(got[k].view(np.uint32), ref[k].reshape(n, c).view(np.uint32)), (name, what, k)
        for b, d in enumerate(got["diag"]):
            for col in ((0, 3) if stride == 16 else (0,)):          # RayCount and the sample count weight; columns 1 and 2 count visits of this library's own tree
                assert np.array_equal(d[:, col].view(np.uint32), ref_diag[b][:, col].view(np.uint32)), (name, what, "diagnostics of batch", b, col)

    check(_sequential(rt, ctx, plist, n, stride), "plain launches")
    for attempt in range(2):                                      # second attempt: chunk order from the measured cost map
        check(_chained(rt, ctx, plist, n, stride), ("chain", attempt))


def test_chain_falls_back_when_batches_differ_or_are_too_many(rt, gpu_context):
    """Batches that differ in more than Seed (here: samples per pixel), more batches than one launch holds (16), and the per-sample RNG policy
    all run as the sequence the chain is defined to equal."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 160, 90
    n = w * h
    mixed = [rt.scenes.make_params(scene, w, h, spp=s, trace_depth=6, seed=30 + s) for s in (2, 3, 2)]
    _same(_chained(rt, ctx, mixed, n, 4), _sequential(rt, ctx, mixed, n, 4), "differing spp")
    many = _params(rt, scene, w, h, 1, 6, list(range(1, 20)))
    _same(_chained(rt, ctx, many, n, 4), _sequential(rt, ctx, many, n, 4), "19 batches")
    per_sample = _params(rt, scene, w, h, 20, 6, [3, 4], rng_policy=rt.abi.RNG_PER_SAMPLE)
    _same(_chained(rt, ctx, per_sample, n, 4), _sequential(rt, ctx, per_sample, n, 4), "per-sample policy")


def test_chain_with_adaptive_sample_counts(rt, gpu_context):
    """SampleCountRange.x != .y: batch k + 1's per-pixel sample count depends on what batch k accumulated (JOBS/SampleBatchJob.cs:118-126) -
    the hand-off must deliver exactly those accumulators."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 200, 120
    n = w * h
    plist = _params(rt, scene, w, h, 2, 8, [11, 12, 13, 14, 15], spp_max=9, extrema=(0.2, 1.4), diagnostics_stride=16)
    _same(_chained(rt, ctx, plist, n, 16), _sequential(rt, ctx, plist, n, 16), "adaptive")


def test_chain_under_texture_driven_noise_cubemap_sky_and_reference_diagnostics(rt, oracle):
    """The per-batch Seed also seeds the blue / STBN texture walks (RT/PerPixelNoise.cs), the sky may be the cubemap, and with
    RTOW_CONTEXT_REFERENCE_DIAGNOSTICS every batch of the chain writes the reference's own FULL_DIAGNOSTICS record to its own buffer:
    chain == sequence, and the last batch's counters == the oracle's."""
    a, S = rt.abi, rt.scenes
    scene = S.cover_scene()
    desc = scene.desc()
    noise = S.NoiseTextures(row_stride=16, count=2, seed=4)
    sky = S.synthetic_sky(size=32)
    w, h, spp, depth = 112, 63, 3, 6
    n = w * h
    with rt.Context(0, flags=a.CONTEXT_REFERENCE_DIAGNOSTICS) as ctx:
        ctx.upload_scene(desc)
        ctx.upload_blue_noise(noise.blue_desc())
        ctx.upload_stb_noise(noise.stb_desc())
        ctx.upload_sky_cubemap(sky.desc())
        for what, kw in (("blue", dict(noise_color=a.NOISE_BLUE, noise_texture_index=1)), ("stbn", dict(noise_color=a.NOISE_SPATIOTEMPORAL_BLUE)),
                         ("white + cubemap", dict(sky_type=a.SKY_CUBEMAP))):
            plist = _params(rt, scene, w, h, spp, depth, [3, 4, 5], diagnostics_stride=16, **kw)
            seq = _sequential(rt, ctx, plist, n, 16)
            got = _chained(rt, ctx, plist, n, 16)
            _same(got, seq, what)
            osc = oracle.OracleScene(desc)
            osc.set_blue_noise(noise.blue_desc())
            osc.set_stb_noise(noise.stb_desc())
            osc.set_cubemap(sky.desc())
            ref = None
            for p in plist:
                ref = osc.sample_batch(p, None if ref is None else {k: ref[k] for k, _ in KEYS})
            osc.close()
            for k, _ in KEYS:
                assert np.array_equal(got[k].reshape(-1).view(np.uint32), ref[k].reshape(-1).view(np.uint32)), (what, k)
            assert np.array_equal(got["diag"][-1].view(np.uint32), ref["diag"].view(np.uint32)), (what, "FULL_DIAGNOSTICS of the last batch")


def test_chain_of_a_slice_that_owns_no_row(rt, gpu_context):
    ctx = gpu_context
    scene = rt.scenes.tiny_scene()
    ctx.upload_scene(scene.desc())
    w, h = 16, 4
    plist = _params(rt, scene, w, h, 2, 4, [1, 2, 3], slice_offset=6, slice_divider=8)
    got = _chained(rt, ctx, plist, w * h, 4)
    for k, _ in KEYS:
        assert not got[k].any(), k


def test_host_buffer_chain_equals_host_batches_one_after_the_other(rt, gpu_context):
    """rtowSampleBatchChain (host arrays in, final accumulators + per-batch diagnostics out, one blocking call) against rtowSampleBatch called
    per batch with the outputs fed back as inputs - the reference's own accumulation (UNITY/Raytracer.cs:798-802); whole frame and a slice."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 144, 81
    n = w * h
    rng = np.random.default_rng(8)
    start = {"color": rng.random((n, 4)).astype(np.float32), "normal": rng.normal(size=(n, 3)).astype(np.float32),
             "albedo": rng.random((n, 3)).astype(np.float32), "scw": rng.random(n).astype(np.float32)}
    start["color"][:, 3] = rng.integers(0, 5, n)
    for kw in ({}, {"slice_offset": 1, "slice_divider": 4}):
        plist = _params(rt, scene, w, h, 3, 6, [9, 10, 11, 12], diagnostics_stride=16, **kw)
        acc, diags = start, []
        for p in plist:
            acc = rt.sample_batch_host(ctx, p, inputs={k: acc[k] for k, _ in KEYS})
            diags.append(acc["diag"])
        got = rt.sample_batch_chain_host(ctx, plist, inputs=start)
        for k, _ in KEYS:
            assert np.array_equal(got[k].reshape(-1).view(np.uint32), acc[k].reshape(-1).view(np.uint32)), (kw, k)
        rows = (np.arange(n) // w) % kw.get("slice_divider", 1) == kw.get("slice_offset", 0)
        for i, (x, y) in enumerate(zip(got["diag"], diags)):
            assert np.array_equal(x[rows].view(np.uint32), y[rows].view(np.uint32)), (kw, "diagnostics of batch", i)


def test_chain_after_the_frame_grows(rt):
    """A context's first chain on a small frame, the next on a larger one: the per-chunk hand-off counters are reallocated, the table
    of per-batch seeds must survive that (it was once freed with them and written again: a use after free that a fresh context never shows)."""
    scene = rt.scenes.cover_scene()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene.desc())
        for w, h in ((64, 36), (320, 180), (96, 54), (640, 360)):
            plist = _params(rt, scene, w, h, 3, 8, (5, 6, 7))
            _same(_chained(rt, ctx, plist, w * h, 4), _sequential(rt, ctx, plist, w * h, 4), "%dx%d" % (w, h))


@pytest.mark.parametrize("stride", [4, 16])
@pytest.mark.parametrize("pattern", [(False, True, True), (True, False, True), (False, False, True)])
def test_chain_with_diagnostics_for_some_batches_only(rt, gpu_context, stride, pattern):
    """Per-batch diagnostics pointers may be NULL (include/rtow.h).  One launch is one kernel variant and the variant follows the record
    format, so a chain whose FIRST batch has no buffer while a later one wants 16-byte FULL_DIAGNOSTICS records must not run the short-record
    variant over all of them (4-byte RayCount records at pix * 4 inside a buffer read as 16-byte records, BoundsHitCount /
    CandidateCount never written): every batch that has a buffer gets exactly the records it gets when run on its own."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h, spp, depth = 160, 90, 3, 8
    n = w * h
    plist = _params(rt, scene, w, h, spp, depth, [41, 42, 43], diagnostics_stride=stride)
    seq = _sequential(rt, ctx, plist, n, stride)
    outs = _zero_bufs(rt, ctx, n)
    diags = [rt.DeviceBuffer(ctx, n * stride).zero() if want else None for want in pattern]
    assert rt.sample_batch_chain_device(ctx, plist, outs, outs, diags) == 0
    ctx.synchronize()
    got = _download(outs, n)
    for k, _ in KEYS:
        assert np.array_equal(got[k].view(np.uint32), seq[k].view(np.uint32)), k
    for i, d in enumerate(diags):
        if d is None:
            continue
        rec = d.download(np.float32, (n, stride // 4))
        assert np.array_equal(rec.view(np.uint32), seq["diag"][i].view(np.uint32)), ("diagnostics of batch", i)
        if stride == 16:
            assert rec[:, 1].max() > 0 and rec[:, 2].max() > 0        # BoundsHitCount / CandidateCount were really counted
    for b in outs + [d for d in diags if d is not None]:
        b.free()


def test_cancelled_chain_returns_promptly_and_leaves_the_context_usable(rt, gpu_context):
    """The cancellation token inside ONE chained launch (JOBS/SampleBatchJob.cs:61-62): waves stop taking chunks at the next refill - in the device-wide
    queue of batch 0 and in the per-XCD queues of the later batches alike - lanes parked behind a chunk's previous batch are released when that batch's
    pixels (already handed out) finish, and the call returns RTOW_ERROR_CANCELLED well before the chain would have ended.  The next chain on the same
    context equals its batches in sequence."""
    import ctypes as C
    import threading
    import time
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 1920, 1080
    n = w * h
    plist = _params(rt, scene, w, h, 256, 8, [1, 2, 3, 4, 5, 6, 7, 8])          # ~0.5 s of GPU work if not cancelled
    bufs = _zero_bufs(rt, ctx, n)
    for delay in (0.03, 0.2):                                                      # during batch 0; in the middle of the chain
        token = C.c_uint8(0)
        threading.Timer(delay, lambda t=token: setattr(t, "value", 1)).start()
        t0 = time.perf_counter()
        rc = rt.sample_batch_chain_device(ctx, plist, bufs, bufs, None, cancel=C.addressof(token))
        dt = time.perf_counter() - t0
        assert rc == rt.abi.RTOW_ERROR_CANCELLED, rc
        assert dt < delay + 0.25, "a cancelled chain took %.3f s" % dt
    for b in bufs:
        b.free()
    small = _params(rt, scene, 160, 90, 3, 8, [11, 12, 13])
    _same(_chained(rt, ctx, small, 160 * 90, 4), _sequential(rt, ctx, small, 160 * 90, 4), "after the cancelled chains")


def test_same_xcd_handover_litmus_runs_at_context_creation_and_chains_fall_back_without_it(rt):
    """the chained launch's hand-over (plain stores + sc1 loads inside one XCD) is measured behaviour, so every context measures it on its own device
    (rtowCreateContext: pairs of workgroups per XCD, 48 rounds) and logs the outcome; RTOW_CONTEXT_NO_CHAIN_FUSION is what a context does when the litmus
    fails - chains run one launch per batch, same result."""
    log = []
    S = rt.scenes
    scene = S.cover_scene()
    w, h, n = 320, 180, 320 * 180
    plist = _params(rt, scene, w, h, 4, 8, [5, 6, 7], diagnostics_stride=4)
    with rt.Context(0, log=lambda lvl, tag, msg, ud: log.append((lvl, msg.decode())), log_level=4) as ctx:
        lines = [m for _, m in log if "same-XCD hand-over litmus" in m]
        assert len(lines) == 1, log
        pairs, stale, timeouts = (int(x) for x in __import__("re").findall(r"(\d+) pairs, (\d+) stale dwords, (\d+) timeouts", lines[0])[0])
        assert pairs >= 64 and stale == 0 and timeouts == 0, lines[0]          # 512 workgroups over 8 XCDs: 256 pairs when they all find a partner
        ctx.upload_scene(scene.desc())
        fused = _chained(rt, ctx, plist, n, 4)
    with rt.Context(0, flags=rt.abi.CONTEXT_NO_CHAIN_FUSION) as ctx:
        ctx.upload_scene(scene.desc())
        unfused = _chained(rt, ctx, plist, n, 4)
        seq = _sequential(rt, ctx, plist, n, 4)
    _same(fused, seq, "fused chain")
    _same(unfused, seq, "chain run batch by batch")
