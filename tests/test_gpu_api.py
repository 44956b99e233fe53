"""GPU tests of the rest of the boundary: device-resident form, post passes, error behaviour, cancellation, big scenes."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(rt, ctx, arr):
    return rt.DeviceBuffer(ctx).upload(arr)


def test_device_resident_form_matches_host_form(rt, oracle, gpu_context):
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 80, 45
    n = w * h
    p = rt.scenes.make_params(scene, w, h, spp=4, trace_depth=8, diagnostics_stride=16)
    host = rt.sample_batch_host(ctx, p)
    ins = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
    outs = [rt.DeviceBuffer(ctx, n * k * 4) for k in (4, 3, 3, 1)]
    diag = rt.DeviceBuffer(ctx, n * 16)
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = ins
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
    job.OutputDiagnostics = diag
    assert job.Schedule(n, 1).Complete() == 0
    ctx.synchronize()
    assert ctx.last_sample_kernel_ms() > 0
    for key, buf, k in zip(("color", "normal", "albedo", "scw"), outs, (4, 3, 3, 1)):
        got = buf.download(np.float32, (n, k) if k > 1 else (n,))
        assert np.array_equal(got.view(np.uint32), host[key].view(np.uint32)), key
    assert np.array_equal(diag.download(np.float32, (n, 4))[:, 0], host["diag"][:, 0])

    # in-place accumulation (in == out) over two more batches equals ping-pong accumulation in the oracle
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = outs
    ref = {k: host[k] for k in ("color", "normal", "albedo", "scw")}
    osc = oracle.OracleScene(scene.desc())
    for seed in (2, 3):
        job.Seed = seed
        assert job.Schedule().Complete() == 0
        p2 = rt.scenes.make_params(scene, w, h, spp=4, trace_depth=8, seed=seed)
        ref = osc.sample_batch(p2, {k: ref[k] for k in ("color", "normal", "albedo", "scw")})
    osc.close()
    ctx.synchronize()
    got = outs[0].download(np.float32, (n, 4))
    assert np.array_equal(got.view(np.uint32), ref["color"].view(np.uint32))

    # post passes on the resident buffers
    c3, n3, a3 = (rt.DeviceBuffer(ctx, n * 12) for _ in range(3))
    cj = rt.CombineJob(ctx, (w, h), DebugMode=False, LdrAlbedo=True)
    cj.InputColor, cj.InputNormal, cj.InputAlbedo = outs[0], outs[1], outs[2]
    cj.OutputColor, cj.OutputNormal, cj.OutputAlbedo = c3, n3, a3
    assert cj.Schedule().Complete() == 0
    r8 = [rt.DeviceBuffer(ctx, n * 4) for _ in range(3)]
    fj = rt.FinalizeTexturesJob(ctx, n)
    fj.InputColor, fj.InputNormal, fj.InputAlbedo = c3, n3, a3
    fj.OutputColor, fj.OutputNormal, fj.OutputAlbedo = r8
    assert fj.Schedule().Complete() == 0
    ctx.synchronize()
    oc, on, oa = oracle.combine(w, h, ref["color"], outs[1].download(np.float32, (n, 3)), outs[2].download(np.float32, (n, 3)), False, True)
    assert np.array_equal(c3.download(np.float32, (n, 3)).view(np.uint32), oc.view(np.uint32))
    assert np.array_equal(n3.download(np.float32, (n, 3)).view(np.uint32), on.view(np.uint32))
    assert np.array_equal(a3.download(np.float32, (n, 3)).view(np.uint32), oa.view(np.uint32))
    fc, fn, fa = oracle.finalize(oc, on, oa)
    assert np.array_equal(r8[0].download(np.uint8, (n, 4)), fc)
    assert np.array_equal(r8[1].download(np.uint8, (n, 4)), fn)
    assert np.array_equal(r8[2].download(np.uint8, (n, 4)), fa)

    rj = rt.ReduceMetricsJob(ctx, n, 16)
    rj.Diagnostics, rj.AccumulatedColor, rj.AccumulatedSampleCountWeight = diag, outs[0], outs[3]
    assert rj.Schedule().Complete() == 0
    m = oracle.reduce_metrics(diag.download(np.float32, (n, 4)), ref["color"], outs[3].download(np.float32, (n,)))
    assert (rj.TotalRayCount, rj.TotalSamples) == (m.totalRayCount, m.totalSamples)
    assert rj.SampleCountExtrema == (m.sampleCountExtrema[0], m.sampleCountExtrema[1])
    assert rj.SampleCountWeightExtrema == (m.sampleCountWeightExtrema.x, m.sampleCountWeightExtrema.y)


def test_combine_interlace_lookaround_and_debug_colours(rt, oracle, gpu_context):
    ctx = gpu_context
    rng = np.random.default_rng(0)
    w, h = 40, 30
    n = w * h
    color = np.concatenate([rng.uniform(0, 40, (n, 3)), rng.integers(0, 6, (n, 1))], axis=1).astype(np.float32)
    color[np.repeat(np.arange(h) % 3 != 0, w), 3] = 0      # interlaced buffer: two of three rows have no samples yet
    color[7, 1] = np.nan
    normal = rng.normal(size=(n, 3)).astype(np.float32)
    albedo = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    for debug in (False, True):
        ins = [_dev(rt, ctx, a) for a in (color, normal, albedo)]
        outs = [rt.DeviceBuffer(ctx, n * 12) for _ in range(3)]
        cj = rt.CombineJob(ctx, (w, h), DebugMode=debug)
        cj.InputColor, cj.InputNormal, cj.InputAlbedo = ins
        cj.OutputColor, cj.OutputNormal, cj.OutputAlbedo = outs
        assert cj.Schedule().Complete() == 0
        ctx.synchronize()
        want = oracle.combine(w, h, color, normal, albedo, debug, False)
        for o, wv in zip(outs, want):
            assert np.array_equal(o.download(np.float32, (n, 3)).view(np.uint32), wv.view(np.uint32))


def test_hit_list_overflow_is_reported(rt):
    """The reference's hit list grows without bound; the kernel's grows up to RtowContextOptions.hitListCapacity (24 = what a lane holds
    itself, no spill area).  A ray that meets more must not pass silently: the batch reports RTOW_ERROR_CAPACITY, once, and the context
    stays usable."""
    with rt.Context(0, hit_list_capacity=24) as ctx:
        _hit_list_overflow_is_reported(rt, ctx)
    with rt.Context(0, hit_list_capacity=26) as ctx:                    # two spilled entries: still one short of the 27 hits
        ctx.upload_scene(rt.scenes.volume_stack_scene(slabs=13).desc())
        with pytest.raises(rt.lib.RtowError) as e:
            rt.sample_batch_host(ctx, rt.scenes.make_params(rt.scenes.volume_stack_scene(slabs=13), 32, 32, spp=1, trace_depth=4))
        assert e.value.code == rt.abi.RTOW_ERROR_CAPACITY


def test_context_options_are_validated(rt):
    a = rt.abi
    for kw in (dict(hit_list_capacity=-1), dict(flags=a.CONTEXT_EXACT_TIES_ALWAYS | a.CONTEXT_EXACT_TIES_NEVER), dict(device_ordinal=-1), dict(device_ordinal=4096)):
        with pytest.raises(rt.lib.RtowError) as e:
            rt.Context(**kw)
        assert e.value.code == a.RTOW_ERROR_INVALID_VALUE, kw


def _hit_list_overflow_is_reported(rt, ctx):
    a = rt.abi
    deep = rt.scenes.volume_stack_scene(slabs=13)                       # 13 hulls x (entry + exit) + the wall = 27 hits per camera ray
    ctx.upload_scene(deep.desc())
    p = rt.scenes.make_params(deep, 32, 32, spp=1, trace_depth=4)
    with pytest.raises(rt.lib.RtowError) as e:
        rt.sample_batch_host(ctx, p)
    assert e.value.code == a.RTOW_ERROR_CAPACITY
    fits = rt.scenes.volume_stack_scene(slabs=10)
    ctx.upload_scene(fits.desc())
    out = rt.sample_batch_host(ctx, rt.scenes.make_params(fits, 32, 32, spp=1, trace_depth=4))
    assert out["color"][:, 3].sum() > 0
    # device-buffer path: the flag surfaces at the next rtowSynchronize
    ctx.upload_scene(deep.desc())
    n = 32 * 32
    bufs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
    acc = a.AccumBuffers(*[b.ptr for b in bufs])
    lib = rt.lib.load()
    assert lib.rtowSampleBatchDevice(ctx.handle, C.byref(p), C.byref(acc), C.byref(acc), None, None, None) == a.RTOW_SUCCESS
    assert lib.rtowSynchronize(ctx.handle) == a.RTOW_ERROR_CAPACITY
    assert lib.rtowSynchronize(ctx.handle) == a.RTOW_SUCCESS
    # a batch enqueued on a CALLER's stream: the status query waits for that batch (not for the context's own, idle stream), reports it
    # once, and a cancelled batch does not leave its flag behind for the next one
    hip = C.CDLL("libamdhip64.so")                                     # a caller-owned stream, created through the HIP runtime the library itself uses
    side = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(side)) == 0
    assert lib.rtowSampleBatchDevice(ctx.handle, C.byref(p), C.byref(acc), C.byref(acc), None, side, None) == a.RTOW_SUCCESS
    assert lib.rtowGetBatchStatus(ctx.handle) == a.RTOW_ERROR_CAPACITY
    assert lib.rtowGetBatchStatus(ctx.handle) == a.RTOW_SUCCESS
    token = C.c_uint8(1)                                                # already cancelled: whatever the kernel flagged is discarded with the batch
    big = rt.scenes.make_params(deep, 256, 256, spp=8, trace_depth=4)
    nb = 256 * 256
    bb = [rt.DeviceBuffer(ctx, nb * c * 4).zero() for c in (4, 3, 3, 1)]
    accb = a.AccumBuffers(*[b.ptr for b in bb])
    assert lib.rtowSampleBatchDevice(ctx.handle, C.byref(big), C.byref(accb), C.byref(accb), None, None, C.addressof(token)) == a.RTOW_ERROR_CANCELLED
    assert lib.rtowGetBatchStatus(ctx.handle) == a.RTOW_SUCCESS
    for b in bufs + bb:
        b.free()
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    assert hip.hipStreamDestroy(side) == 0


def test_slice_that_owns_no_row_does_nothing(rt, gpu_context):
    """SliceOffset >= height (tile-parallel with more ranks than rows): the reference's Execute returns for every index
    (JOBS/SampleBatchJob.cs:69-70); the batch succeeds, writes nothing and can be timed."""
    ctx = gpu_context
    scene = rt.scenes.tiny_scene()
    ctx.upload_scene(scene.desc())
    w, h = 16, 4
    p = rt.scenes.make_params(scene, w, h, spp=2, trace_depth=4, slice_offset=5, slice_divider=8)
    ins = {"color": np.full((w * h, 4), 3.0, np.float32), "normal": np.full((w * h, 3), 4.0, np.float32),
           "albedo": np.full((w * h, 3), 5.0, np.float32), "scw": np.full(w * h, 6.0, np.float32)}
    out = rt.sample_batch_host(ctx, p, inputs=ins)
    for k in ins:
        assert np.array_equal(out[k], ins[k]), k
    assert ctx.last_sample_kernel_ms() >= 0.0
    # per-sample policy too (its fold kernel would be a zero-sized launch)
    p.rngPolicy = rt.abi.RNG_PER_SAMPLE
    out = rt.sample_batch_host(ctx, p, inputs=ins)
    for k in ins:
        assert np.array_equal(out[k], ins[k]), k


def test_cancelled_per_sample_batch_leaves_the_accumulators_alone(rt, gpu_context):
    """RTOW_RNG_PER_SAMPLE folds unit records into the accumulators after the sample kernel; after a cancellation the records of units
    that never ran are stale, so nothing may be folded: in-place accumulators keep their input values (the reference's cancelled Execute
    returns before any write, JOBS/SampleBatchJob.cs:61-62)."""
    a = rt.abi
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 640, 360
    n = w * h
    p = rt.scenes.make_params(scene, w, h, spp=64, trace_depth=8, rng_policy=a.RNG_PER_SAMPLE)
    bufs = [rt.DeviceBuffer(ctx, n * c * 4) for c in (4, 3, 3, 1)]
    lib = rt.lib.load()
    for b in bufs:
        assert lib.rtowDeviceMemset(ctx.handle, b.handle, 0, b.nbytes) == 0
    acc = a.AccumBuffers(*[b.ptr for b in bufs])
    token = C.c_uint8(1)
    assert lib.rtowSampleBatchDevice(ctx.handle, C.byref(p), C.byref(acc), C.byref(acc), None, None, C.addressof(token)) == a.RTOW_ERROR_CANCELLED
    ctx.synchronize()
    for b, c in zip(bufs, (4, 3, 3, 1)):
        assert not b.download(np.float32, (n, c)).any(), "a cancelled batch folded stale unit records into the accumulators"
        b.free()


def test_exact_tie_kernels_can_be_forced(rt, oracle):
    """RTOW_CONTEXT_EXACT_TIES_ALWAYS (a context option, applied at upload) selects the exact-tie kernels for a scene without duplicates;
    same image, bit for bit."""
    scene = rt.scenes.mixed_scene()
    desc = scene.desc()
    p = rt.scenes.make_params(scene, 64, 40, spp=4, trace_depth=8)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p)
    osc.close()
    for forced in (False, True):
        log = []
        ctx = rt.Context(0, log=lambda lvl, tag, msg, ud: log.append(msg.decode()), log_level=4, flags=rt.abi.CONTEXT_EXACT_TIES_ALWAYS if forced else 0)
        ctx.upload_scene(desc)
        assert any("exact-tie kernels" in m for m in log) == forced, log
        got = rt.sample_batch_host(ctx, p)
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (forced, k)
        ctx.close()
    with pytest.raises(rt.lib.RtowError):                                # contradictory switches
        rt.Context(0, flags=rt.abi.CONTEXT_EXACT_TIES_ALWAYS | rt.abi.CONTEXT_EXACT_TIES_NEVER)


def test_error_codes(rt, gpu_context):
    lib = rt.lib.load()
    a = rt.abi
    fresh = rt.Context(0)
    scene = rt.scenes.tiny_scene()
    p = rt.scenes.make_params(scene, 8, 8, spp=1, trace_depth=4)
    z = {k: np.zeros((64, c), np.float32) for k, c in (("color", 4), ("normal", 3), ("albedo", 3))}
    z["scw"] = np.zeros(64, np.float32)
    with pytest.raises(rt.lib.RtowError) as e:
        rt.sample_batch_host(fresh, p, z)
    assert e.value.code == a.RTOW_ERROR_NO_SCENE
    info = a.SceneInfo()
    assert lib.rtowGetSceneInfo(fresh.handle, C.byref(info)) == a.RTOW_ERROR_NO_SCENE

    fresh.upload_scene(scene.desc())
    for field, value, code in (("traceDepth", 0, a.RTOW_ERROR_INVALID_VALUE), ("traceDepth", 65, a.RTOW_ERROR_CAPACITY),
                               ("sliceDivider", 0, a.RTOW_ERROR_INVALID_VALUE), ("noiseColor", a.NOISE_BLUE, a.RTOW_ERROR_INVALID_VALUE),   # no blue-noise set uploaded
                               ("noiseColor", 3, a.RTOW_ERROR_INVALID_VALUE),
                               ("diagnosticsStride", 8, a.RTOW_ERROR_INVALID_VALUE)):
        q = rt.scenes.make_params(scene, 8, 8, spp=1, trace_depth=4)
        setattr(q, field, value)
        with pytest.raises(rt.lib.RtowError) as e:
            rt.sample_batch_host(fresh, q, z)
        assert e.value.code == code, field

    bad = scene.desc()
    bad.entities[0].type = 9
    assert lib.rtowUploadScene(fresh.handle, C.byref(bad)) == a.RTOW_ERROR_INVALID_VALUE
    bad = scene.desc()
    bad.entities[0].type = a.ENTITY_TRIANGLE           # no triangle payloads supplied
    assert lib.rtowUploadScene(fresh.handle, C.byref(bad)) == a.RTOW_ERROR_INVALID_VALUE
    bad = scene.desc()
    bad.entities[0].materialIndex = 99
    assert lib.rtowUploadScene(fresh.handle, C.byref(bad)) == a.RTOW_ERROR_INVALID_VALUE
    bad = scene.desc()
    bad.materials[0].type = 7
    assert lib.rtowUploadScene(fresh.handle, C.byref(bad)) == a.RTOW_ERROR_INVALID_VALUE
    bad = scene.desc()
    bad.materials[0].albedo.type = a.TEXTURE_CHECKER_PATTERN       # dead code in the reference (RT/Texture.cs:61-78)
    assert lib.rtowUploadScene(fresh.handle, C.byref(bad)) == a.RTOW_ERROR_UNSUPPORTED
    bad = scene.desc()
    bad.materials[0].albedo.type = a.TEXTURE_IMAGE
    bad.materials[0].albedo.imageIndex = 3                          # no such image
    assert lib.rtowUploadScene(fresh.handle, C.byref(bad)) == a.RTOW_ERROR_INVALID_VALUE
    # a failed upload leaves the previous scene usable
    out = rt.sample_batch_host(fresh, p, z)
    assert out["color"][:, 3].sum() > 0
    fresh.close()


def test_sky_cubemap_upload_validation(rt, gpu_context):
    a = rt.abi
    lib = rt.lib.load()
    px = (C.c_uint16 * (6 * 4 * 4 * 4))()
    ok = a.CubemapDesc(4, 4, a.CUBEMAP_SIGNED_HALF, 8, C.addressof(px))
    assert lib.rtowUploadSkyCubemap(gpu_context.handle, C.byref(ok)) == 0
    for field, value in (("faceWidth", 0), ("faceHeight", -1), ("channelType", 7), ("pixelStride", 4), ("pixelStride", 7)):
        bad = a.CubemapDesc(4, 4, a.CUBEMAP_SIGNED_HALF, 8, C.addressof(px))
        setattr(bad, field, value)
        assert lib.rtowUploadSkyCubemap(gpu_context.handle, C.byref(bad)) == a.RTOW_ERROR_INVALID_VALUE, field
    assert lib.rtowUploadSkyCubemap(None, C.byref(ok)) == a.RTOW_ERROR_INVALID_VALUE
    assert lib.rtowUploadSkyCubemap(gpu_context.handle, None) == 0           # drop


def test_cancellation_token(rt, gpu_context):
    """NativeReference<bool> CancellationToken (JOBS/SampleBatchJob.cs:23,61-62): set mid-batch -> RTOW_ERROR_CANCELLED, promptly."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    w, h = 1920, 1080
    n = w * h
    p = rt.scenes.make_params(scene, w, h, spp=2048, trace_depth=8)     # ~1.4 s of GPU work if not cancelled
    bufs = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = bufs
    token = C.c_uint8(0)
    job.CancellationToken = token
    threading.Timer(0.05, lambda: setattr(token, "value", 1)).start()
    t = time.perf_counter()
    rc = job.Schedule().Complete()
    dt = time.perf_counter() - t
    assert rc == rt.abi.RTOW_ERROR_CANCELLED
    full = None
    token.value = 0
    t = time.perf_counter()
    assert job.Schedule().Complete() == 0
    full = time.perf_counter() - t
    assert dt < 0.6 * full, "cancelled batch took %.3f s, a full one %.3f s" % (dt, full)


def test_stress_scene_too_large_for_lds(rt, oracle, gpu_context):
    """BASELINE.json configs[3] at a reduced size: a scene whose image exceeds LDS (BVH top levels staged, rest through L2)."""
    ctx = gpu_context
    scene = rt.scenes.stress_scene(count=3000, max_tentatives=12000)
    desc = scene.desc()
    ctx.upload_scene(desc)
    info = ctx.scene_info()
    assert info.sceneInLds == 0 and info.bvhNodeCount == scene.entity_count - 1 and 0 < info.ldsBytesScene < info.sceneBytesDevice
    p = rt.scenes.make_params(scene, 96, 54, spp=4, trace_depth=8)
    gpu = rt.sample_batch_host(ctx, p)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p)
    osc.close()
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), k
    assert np.array_equal(gpu["diag"][:, 0], ref["diag"][:, 0])


def test_deep_trace_depths_use_wider_history(rt, oracle, gpu_context):
    ctx = gpu_context
    scene = rt.scenes.tiny_scene()
    desc = scene.desc()
    ctx.upload_scene(desc)
    osc = oracle.OracleScene(desc)
    for depth in (9, 16, 17, 33, 64):
        p = rt.scenes.make_params(scene, 48, 27, spp=3, trace_depth=depth)
        gpu = rt.sample_batch_host(ctx, p)
        ref = osc.sample_batch(p)
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (depth, k)
    osc.close()


def test_single_sphere_scene(rt, oracle, gpu_context):
    scene = rt.scenes.Scene("one")
    scene.add_sphere((0, 0, -3), 1.0, rt.scenes.lambertian((0.7, 0.3, 0.3)))
    scene.camera = {"position": [0, 0, 0], "target": [0, 0, -1], "up": [0, 1, 0], "vfov": 60.0, "aperture": 0.0}
    desc = scene.desc()
    gpu_context.upload_scene(desc)
    p = rt.scenes.make_params(scene, 32, 32, spp=4, trace_depth=5)
    gpu = rt.sample_batch_host(gpu_context, p)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p)
    osc.close()
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), k


def test_add_accum_device(rt, gpu_context):
    """rtowAddAccumDevice: dst += src on the four accumulators (the fold step of the batch-parallel multi-GPU path)."""
    ctx = gpu_context
    rng = np.random.default_rng(4)
    n = 12345                                            # not a multiple of 4: exercises the scalar tail of the float3 arrays
    dst = [rng.normal(size=(n, k)).astype(np.float32) for k in (4, 3, 3, 1)]
    src = [rng.normal(size=(n, k)).astype(np.float32) for k in (4, 3, 3, 1)]
    dd = [_dev(rt, ctx, a) for a in dst]
    ds = [_dev(rt, ctx, a) for a in src]
    bd = rt.abi.AccumBuffers(*[b.ptr for b in dd])
    bs = rt.abi.AccumBuffers(*[b.ptr for b in ds])
    assert rt.lib.load().rtowAddAccumDevice(ctx.handle, n, C.byref(bd), C.byref(bs), None) == 0
    ctx.synchronize()
    for a, b, buf, k in zip(dst, src, dd, (4, 3, 3, 1)):
        assert np.array_equal(buf.download(np.float32, (n, k)), a + b)
    assert rt.lib.load().rtowAddAccumDevice(ctx.handle, 0, C.byref(bd), C.byref(bs), None) == rt.abi.RTOW_ERROR_INVALID_VALUE


def test_add_accum_device_on_a_flat_slice(rt, gpu_context):
    """The distributed fold (multigpu.render_batches) adds SLICES of the flat [colour | normal | albedo | weight] accumulator that
    straddle the section boundaries: a slice of 11 * k floats goes through the 4-buffer add as k "pixels" at offsets 0, 4k, 7k, 10k."""
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    ctx = gpu_context
    rng = np.random.default_rng(9)
    n, world = 1000, 3
    m, padded = mg.slice_floats(n, world), mg.padded_floats(n, world)
    assert m == 11 * 334 and padded == 3 * m
    acc = rng.normal(size=padded).astype(np.float32)
    part = rng.normal(size=padded).astype(np.float32)
    dacc, dpart = _dev(rt, ctx, acc), _dev(rt, ctx, part)
    k = m // mg.ACCUM_FLOATS
    for r in range(world):
        bd = rt.abi.AccumBuffers(*[dacc.ptr + 4 * (r * m + o * k) for o in (0, 4, 7, 10)])
        bs = rt.abi.AccumBuffers(*[dpart.ptr + 4 * (r * m + o * k) for o in (0, 4, 7, 10)])
        assert rt.lib.load().rtowAddAccumDevice(ctx.handle, k, C.byref(bd), C.byref(bs), None) == 0
    ctx.synchronize()
    assert np.array_equal(dacc.download(np.float32, (padded,)), acc + part)


def test_adaptive_sample_counts_match_oracle(rt, oracle, gpu_context):
    """SampleCountRange.x != .y with weight extrema (JOBS/SampleBatchJob.cs:118-126), first batch (0/0 -> max) and a second one."""
    ctx = gpu_context
    scene = rt.scenes.tiny_scene()
    desc = scene.desc()
    ctx.upload_scene(desc)
    osc = oracle.OracleScene(desc)
    p = rt.scenes.make_params(scene, 40, 24, spp=2, spp_max=7, trace_depth=5, extrema=(0.5, 1.5), diagnostics_stride=16)
    g1, r1 = rt.sample_batch_host(ctx, p), osc.sample_batch(p)
    ins = {k: r1[k] for k in ("color", "normal", "albedo", "scw")}
    p.seed = 9
    g2, r2 = rt.sample_batch_host(ctx, p, ins), osc.sample_batch(p, ins)
    osc.close()
    for g, r in ((g1, r1), (g2, r2)):
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(g[k].view(np.uint32), r[k].view(np.uint32)), k
        assert np.array_equal(g["diag"][:, 0], r["diag"][:, 0])
        assert np.array_equal(g["diag"][:, 3].view(np.uint32), r["diag"][:, 3].view(np.uint32))   # FULL_DIAGNOSTICS SampleCountWeight
    assert len(np.unique(r2["diag"][:, 0])) > 3


def test_camera_ray_candidate_lists_are_conservative(rt, oracle, gpu_context):
    """Depth-0 rays take their candidates from a per-pixel list of leaf-parent nodes built by one conservative beam walk
    (primary_candidates_kernel) instead of walking the tree.  The frame must not change by a bit when the lists are switched off, and must equal the oracle, for the cases that
    stretch the beam: huge pixels (tiny frames), a wide lens, no jitter, interlaced slices, camera inside the geometry, a tree outside LDS."""
    ctx = gpu_context
    cases = []
    cover = rt.scenes.cover_scene()
    cases.append((cover, dict(width=24, height=14, spp=16, trace_depth=4)))
    cases.append((cover, dict(width=97, height=31, spp=8, trace_depth=3, jitter=False)))
    cases.append((cover, dict(width=64, height=36, spp=8, trace_depth=4, slice_offset=1, slice_divider=3)))
    wide = rt.scenes.moving_scene()
    wide.camera = dict(wide.camera, aperture=1.5)                       # lens radius 0.75: beams as wide as the spheres
    cases.append((wide, dict(width=40, height=24, spp=16, trace_depth=3)))
    wide2 = rt.scenes.cover_scene()
    wide2.camera = dict(wide2.camera, aperture=0.3, vfov=70.0)
    cases.append((wide2, dict(width=33, height=19, spp=16, trace_depth=3)))
    inside = rt.scenes.tiny_scene()
    inside.camera = {"position": [-1.0, 0.05, -1.0], "target": [1.0, 0.2, -1.0], "up": [0.0, 1.0, 0.0], "vfov": 90.0, "aperture": 0.2}   # inside the glass ball
    cases.append((inside, dict(width=32, height=32, spp=8, trace_depth=6)))
    cases.append((rt.scenes.mixed_scene(), dict(width=48, height=32, spp=8, trace_depth=4)))
    cases.append((rt.scenes.stress_scene(count=3000, max_tentatives=12000), dict(width=60, height=34, spp=4, trace_depth=3)))
    walker = rt.Context(0, flags=rt.abi.CONTEXT_NO_CAMERA_RAY_LISTS)     # a context that walks the tree for camera rays too
    for scene, kw in cases:
        desc = scene.desc()
        ctx.upload_scene(desc)
        walker.upload_scene(desc)
        p = rt.scenes.make_params(scene, **kw)
        with_lists = rt.sample_batch_host(ctx, p)
        without = rt.sample_batch_host(walker, p)
        osc = oracle.OracleScene(desc)
        ref = osc.sample_batch(p)
        osc.close()
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(with_lists[k].view(np.uint32), without[k].view(np.uint32)), (scene.name, kw, k)
            assert np.array_equal(with_lists[k].view(np.uint32), ref[k].view(np.uint32)), (scene.name, kw, k)
        assert np.array_equal(with_lists["diag"][:, 0], ref["diag"][:, 0])
    walker.close()


def test_registered_host_buffers_give_the_same_results(rt, oracle, gpu_context):
    """rtowRegisterHostBuffer: with the host's pools pinned, rtowSampleBatch lets the kernel store outputs and diagnostics straight into host
    memory (no copy-back).  Same bits as the staged path and the oracle; in-place (in == out) accumulation over two batches; a slice
    leaves the rows it does not own alone; unregistered memory keeps working next to registered memory."""
    a = rt.abi
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    desc = scene.desc()
    ctx.upload_scene(desc)
    w, h = 80, 45
    n = w * h
    osc = oracle.OracleScene(desc)
    # one pool per buffer kind, like UNITY/Raytracer.cs:279-288; page-unaligned on purpose (numpy gives 64-byte alignment at best)
    pool = {k: np.zeros((n, c), np.float32) for k, c in (("color", 4), ("normal", 3), ("albedo", 3))}
    pool["scw"] = np.zeros(n, np.float32)
    diag = np.zeros((n, 4), np.float32)
    ctx.register_host_buffers(pool["color"], pool["normal"], pool["albedo"], pool["scw"], diag)
    lib = rt.lib.load()
    assert lib.rtowRegisterHostBuffer(ctx.handle, pool["color"].ctypes.data + 16, 64) == a.RTOW_ERROR_INVALID_VALUE      # overlaps a live registration
    ref = None
    for batch in range(2):
        p = rt.scenes.make_params(scene, w, h, spp=3, trace_depth=6, seed=21 + batch, diagnostics_stride=16)
        job = rt.SampleBatchJob(ctx, p)
        job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
        job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
        job.OutputDiagnostics = diag
        assert job.Schedule(n, 1).Complete() == 0
        ref = osc.sample_batch(p, ref if ref is None else {k: ref[k] for k in ("color", "normal", "albedo", "scw")})
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(pool[k].reshape(-1).view(np.uint32), ref[k].reshape(-1).view(np.uint32)), (batch, k)
        assert np.array_equal(diag[:, 0], ref["diag"][:, 0])
    # a slice through the zero-copy path: rows it does not own keep their bytes
    before = {k: v.copy() for k, v in pool.items()}
    p = rt.scenes.make_params(scene, w, h, spp=2, trace_depth=6, seed=40, slice_offset=1, slice_divider=3)
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
    assert job.Schedule(n, 1).Complete() == 0
    rows = np.arange(n) // w
    refs = osc.sample_batch(p, before)
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(pool[k][rows % 3 != 1], before[k][rows % 3 != 1]), k
        assert np.array_equal(pool[k].reshape(n, -1)[rows % 3 == 1].view(np.uint32), refs[k].reshape(n, -1)[rows % 3 == 1].view(np.uint32)), k
    # mixed: registered inputs, unregistered outputs -> staged copy-back
    outs = {k: np.zeros_like(v) for k, v in pool.items()}
    p = rt.scenes.make_params(scene, w, h, spp=2, trace_depth=6, seed=41)
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs["color"], outs["normal"], outs["albedo"], outs["scw"]
    assert job.Schedule(n, 1).Complete() == 0
    refm = osc.sample_batch(p, pool)
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(outs[k].reshape(-1).view(np.uint32), refm[k].reshape(-1).view(np.uint32)), k
    # the per-sample RNG policy folds its unit records with a second kernel: that one stores into the registered arrays too
    before = {k: v.copy() for k, v in pool.items()}
    p = rt.scenes.make_params(scene, w, h, spp=20, trace_depth=6, seed=43, rng_policy=a.RNG_PER_SAMPLE)
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = pool["color"], pool["normal"], pool["albedo"], pool["scw"]
    assert job.Schedule(n, 1).Complete() == 0
    refp = osc.sample_batch(p, before)
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(pool[k].reshape(-1).view(np.uint32), refp[k].reshape(-1).view(np.uint32)), ("per-sample", k)
    osc.close()
    ctx.unregister_host_buffers()
    assert lib.rtowUnregisterHostBuffer(ctx.handle, pool["color"].ctypes.data) == a.RTOW_ERROR_INVALID_VALUE             # already dropped


@pytest.mark.parametrize("name,w,h,spp,depth,max_bvh_depth", [("cover", 160, 90, 4, 8, 32), ("cover", 96, 54, 3, 6, 5), ("moving", 120, 68, 3, 6, 32),
                                                               ("mixed", 96, 64, 3, 6, 32), ("volumes", 96, 54, 3, 8, 32), ("stress", 120, 68, 2, 5, 32)])
def test_reference_identical_full_diagnostics(rt, oracle, name, w, h, spp, depth, max_bvh_depth):
    """RTOW_CONTEXT_REFERENCE_DIAGNOSTICS: the FULL_DIAGNOSTICS record {RayCount, BoundsHitCount, CandidateCount, SampleCountWeight}
    (UNITY/Raytracer.cs:54-64) as the REFERENCE produces it - BoundsHitCount / CandidateCount count the boxes and leaf entities of the tree
    RebuildBvh builds (JOBS/SampleBatchJob.cs:427-440), including the backwards containment probes of volume scenes (:495) and leaves
    forced at MaxBvhDepth.  All four columns against the oracle, which walks that very tree; colours unchanged."""
    S = rt.scenes
    scene = {"cover": S.cover_scene, "moving": S.moving_scene, "mixed": S.mixed_scene, "volumes": S.volume_scene,
             "stress": lambda: S.stress_scene(count=3000, max_tentatives=12000)}[name]()
    desc = scene.desc(max_bvh_depth=max_bvh_depth)
    p = S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=17, diagnostics_stride=16, focus=6.0 if name != "cover" else None)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p)
    osc.close()
    with rt.Context(0, flags=rt.abi.CONTEXT_REFERENCE_DIAGNOSTICS) as ctx:
        ctx.upload_scene(desc)
        got = rt.sample_batch_host(ctx, p)
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
        for col, what in enumerate(("RayCount", "BoundsHitCount", "CandidateCount", "SampleCountWeight")):
            assert np.array_equal(got["diag"][:, col].view(np.uint32), ref["diag"][:, col].view(np.uint32)), (name, what, got["diag"][:4], ref["diag"][:4])
        # the per-sample RNG policy sums the same counters through its unit records
        p.rngPolicy = rt.abi.RNG_PER_SAMPLE
        got2 = rt.sample_batch_host(ctx, p)
        assert got2["diag"][:, 1].sum() > 0 and got2["diag"][:, 2].sum() > 0
    with rt.Context(0) as plain:                                        # without the option: the library's own tree, different numbers, same RayCount
        plain.upload_scene(desc)
        own = rt.sample_batch_host(plain, S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=17, diagnostics_stride=16, focus=6.0 if name != "cover" else None))
        assert np.array_equal(own["diag"][:, 0], ref["diag"][:, 0])
        assert not np.array_equal(own["diag"][:, 1], ref["diag"][:, 1])


def test_fused_combine_finalize_equals_the_two_passes_and_the_oracle(rt, oracle, gpu_context):
    """rtowCombineFinalizeDevice (the reference's default chain, denoiseMode 0: CombineJob -> FinalizeTexturesJob) writes the bytes of rtowCombineDevice followed by
    rtowFinalizeDevice and of oracle.combine -> oracle.finalize: rendered accumulators, an interlaced buffer (look-around), NaNs, zero-sample pixels,
    debug colours, LDR albedo, sizes that are no multiple of the block."""
    ctx = gpu_context
    lib = rt.lib.load()
    a = rt.abi
    rng = np.random.default_rng(4)
    scene = rt.scenes.cover_scene()
    ctx.upload_scene(scene.desc())
    cases = []
    w, h = 96, 54
    r = rt.sample_batch_host(ctx, rt.scenes.make_params(scene, w, h, spp=5, trace_depth=8))
    cases.append((w, h, r["color"], r["normal"], r["albedo"]))
    w, h = 41, 29
    n = w * h
    color = np.concatenate([rng.uniform(0, 40, (n, 3)), rng.integers(0, 6, (n, 1))], axis=1).astype(np.float32)
    color[np.repeat(np.arange(h) % 3 != 0, w), 3] = 0
    color[7, 1] = np.nan
    cases.append((w, h, color, rng.normal(size=(n, 3)).astype(np.float32), rng.uniform(0, 3, (n, 3)).astype(np.float32)))
    for w, h, color, normal, albedo in cases:
        n = w * h
        for debug, ldr in ((False, True), (True, False), (False, False)):
            ins = [_dev(rt, ctx, x) for x in (color, normal, albedo)]
            r8 = [rt.DeviceBuffer(ctx, n * 4).zero() for _ in range(3)]
            cp = a.CombineParams(w, h, int(debug), int(ldr))
            rt.lib.check(lib.rtowCombineFinalizeDevice(ctx.handle, C.byref(cp), ins[0].ptr, ins[1].ptr, ins[2].ptr, r8[0].ptr, r8[1].ptr, r8[2].ptr, None), "rtowCombineFinalizeDevice")
            ctx.synchronize()
            oc, on, oa = oracle.combine(w, h, color, normal, albedo, debug, ldr)
            want = oracle.finalize(oc, on, oa)
            for got, wv, name in zip(r8, want, ("color", "normal", "albedo")):
                assert np.array_equal(got.download(np.uint8, (n, 4)), wv), (w, h, debug, ldr, name)
            # and the two separate passes
            f3 = [rt.DeviceBuffer(ctx, n * 12) for _ in range(3)]
            s8 = [rt.DeviceBuffer(ctx, n * 4).zero() for _ in range(3)]
            rt.lib.check(lib.rtowCombineDevice(ctx.handle, C.byref(cp), ins[0].ptr, ins[1].ptr, ins[2].ptr, f3[0].ptr, f3[1].ptr, f3[2].ptr, None), "rtowCombineDevice")
            rt.lib.check(lib.rtowFinalizeDevice(ctx.handle, n, f3[0].ptr, f3[1].ptr, f3[2].ptr, s8[0].ptr, s8[1].ptr, s8[2].ptr, None), "rtowFinalizeDevice")
            ctx.synchronize()
            for got, sep in zip(r8, s8):
                assert np.array_equal(got.download(np.uint8, (n, 4)), sep.download(np.uint8, (n, 4)))
            for b in ins + r8 + f3 + s8:
                b.free()
    cp = a.CombineParams(0, 4, 0, 0)
    assert lib.rtowCombineFinalizeDevice(ctx.handle, C.byref(cp), 1, 1, 1, 1, 1, 1, None) == a.RTOW_ERROR_INVALID_VALUE


@pytest.mark.parametrize("n", [1, 3, 255, 256, 1000, 96 * 54, 1920 * 1080])
def test_add_accum_is_one_exact_pass_over_the_four_buffers(rt, gpu_context, n):
    """rtowAddAccumDevice: dst += src, element for element one float32 addition, whatever the sizes and alignments of the four buffers (views into larger
    allocations at odd float offsets take the unaligned path)."""
    ctx = gpu_context
    a = rt.abi
    lib = rt.lib.load()
    rng = np.random.default_rng(n)
    comps = (4, 3, 3, 1)
    for shift in (0, 1):                                      # shift 1: every buffer starts 4 bytes into its allocation
        dst = [rng.normal(size=n * c + shift).astype(np.float32) for c in comps]
        src = [rng.normal(size=n * c + shift).astype(np.float32) for c in comps]
        dd = [_dev(rt, ctx, x) for x in dst]
        ds = [_dev(rt, ctx, x) for x in src]
        bd = a.AccumBuffers(*[b.ptr + 4 * shift for b in dd])
        bs = a.AccumBuffers(*[b.ptr + 4 * shift for b in ds])
        rt.lib.check(lib.rtowAddAccumDevice(ctx.handle, n, C.byref(bd), C.byref(bs), None), "rtowAddAccumDevice")
        ctx.synchronize()
        for d, s_, b, c in zip(dst, src, dd, comps):
            got = b.download(np.float32, (n * c + shift,))
            want = d.copy()
            want[shift:] = d[shift:] + s_[shift:]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, shift, c)
        for b in dd + ds:
            b.free()


def test_reduce_metrics_async_writes_the_record_in_stream_order(rt, oracle, gpu_context):
    """rtowReduceMetricsDeviceAsync: the same numbers as the blocking form and the oracle, written by the device into registered host memory and into device
    memory; pageable host memory is refused."""
    ctx = gpu_context
    a = rt.abi
    lib = rt.lib.load()
    rng = np.random.default_rng(8)
    n = 320 * 180
    diag = rng.integers(0, 3000, (n, 4)).astype(np.float32)
    color = np.concatenate([rng.uniform(0, 40, (n, 3)), rng.integers(0, 300, (n, 1))], axis=1).astype(np.float32)
    scw = rng.uniform(0, 50, n).astype(np.float32)
    dd, dc, ds = _dev(rt, ctx, diag), _dev(rt, ctx, color), _dev(rt, ctx, scw)
    want = oracle.reduce_metrics(diag, color, scw)

    def same(m):
        return ((m.totalRayCount, m.totalSamples, m.totalRayCount64, m.totalSamples64, m.sampleCountExtrema[0], m.sampleCountExtrema[1]) ==
                (want.totalRayCount, want.totalSamples, want.totalRayCount64, want.totalSamples64, want.sampleCountExtrema[0], want.sampleCountExtrema[1]) and
                np.float32(m.sampleCountWeightExtrema.x) == np.float32(want.sampleCountWeightExtrema.x) and np.float32(m.sampleCountWeightExtrema.y) == np.float32(want.sampleCountWeightExtrema.y))

    blocking = a.Metrics()
    rt.lib.check(lib.rtowReduceMetricsDevice(ctx.handle, n, dd.ptr, 16, dc.ptr, ds.ptr, None, C.byref(blocking)), "rtowReduceMetricsDevice")
    assert same(blocking)
    # registered host memory: a numpy array that holds the record
    host = np.zeros(C.sizeof(a.Metrics) // 4 + 16, np.uint32)
    ctx.register_host_buffers(host)
    for rep in range(3):                                      # back to back: the per-context partials are ordered by the library
        rt.lib.check(lib.rtowReduceMetricsDeviceAsync(ctx.handle, n, dd.ptr, 16, dc.ptr, ds.ptr, None, host.ctypes.data), "rtowReduceMetricsDeviceAsync")
    ctx.synchronize()
    assert same(a.Metrics.from_buffer_copy(host.tobytes()[:C.sizeof(a.Metrics)]))
    ctx.unregister_host_buffers()
    # device memory
    dm = rt.DeviceBuffer(ctx, C.sizeof(a.Metrics)).zero()
    rt.lib.check(lib.rtowReduceMetricsDeviceAsync(ctx.handle, n, dd.ptr, 16, dc.ptr, ds.ptr, None, dm.ptr), "rtowReduceMetricsDeviceAsync")
    ctx.synchronize()
    assert same(a.Metrics.from_buffer_copy(dm.download(np.uint8, (C.sizeof(a.Metrics),)).tobytes()))
    pageable = np.zeros(64, np.uint32)
    assert lib.rtowReduceMetricsDeviceAsync(ctx.handle, n, dd.ptr, 16, dc.ptr, ds.ptr, None, pageable.ctypes.data) == a.RTOW_ERROR_INVALID_VALUE
    for b in (dd, dc, ds, dm):
        b.free()
