"""GPU: batch groups (rtowSampleBatchGroupDevice, include/rtow.h).

`count` INDEPENDENT batches of one frame - same inputs, own outputs, own Seed - in one launch whose work queue holds (pixel chunk, batch) pairs.
Defined result: the same batches as `count` separate rtowSampleBatchDevice calls.  Every test compares the group with those calls bit for bit:
accumulators, fallback AOVs, diagnostics; non-zero inputs (a group's batches all read them), slices, frames that are no multiple of the 8 x 8 ticket
tiles, every scene kind, 4- and 16-byte records, more batches than one launch holds, batches that differ in more than Seed (run one by one)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEYS = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))


def _inputs(n, seed):
    rng = np.random.default_rng(seed)
    ins = {"color": rng.random((n, 4)).astype(np.float32), "normal": rng.normal(size=(n, 3)).astype(np.float32),
           "albedo": rng.random((n, 3)).astype(np.float32), "scw": rng.random(n).astype(np.float32)}
    ins["color"][:, 3] = rng.integers(0, 5, n)                     # success counts are whole numbers; some pixels have none yet
    return ins


def _upload(rt, ctx, arrays):
    return [rt.DeviceBuffer(ctx).upload(arrays[k]) for k, _ in KEYS]


def _download(bufs, n):
    return {k: b.download(np.float32, (n, c)) for (k, c), b in zip(KEYS, bufs)}


def _separate(rt, ctx, plist, ins, n, stride, fill):
    res = []
    src = _upload(rt, ctx, ins)
    for p in plist:
        outs = _upload(rt, ctx, fill)
        d = rt.DeviceBuffer(ctx, n * stride).zero()
        job = rt.SampleBatchJob(ctx, p)
        job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = src
        job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
        job.OutputDiagnostics = d
        assert job.Schedule().Complete() == 0
        ctx.synchronize()
        r = _download(outs, n)
        r["diag"] = d.download(np.float32, (n, stride // 4))
        res.append(r)
        for b in outs + [d]:
            b.free()
    for b in src:
        b.free()
    return res


def _grouped(rt, ctx, plist, ins, n, stride, fill, with_diag=True):
    src = _upload(rt, ctx, ins)
    outs = [_upload(rt, ctx, fill) for _ in plist]
    diags = [rt.DeviceBuffer(ctx, n * stride).zero() for _ in plist] if with_diag else None
    assert rt.sample_batch_group_device(ctx, plist, src, outs, diags) == 0
    ctx.synchronize()
    res = []
    for k in range(len(plist)):
        r = _download(outs[k], n)
        r["diag"] = diags[k].download(np.float32, (n, stride // 4)) if with_diag else None
        res.append(r)
    for b in src + [x for o in outs for x in o] + (diags or []):
        b.free()
    return res


def _same(a, b, what):
    for i, (x, y) in enumerate(zip(a, b)):
        for k, _ in KEYS:
            assert np.array_equal(x[k].view(np.uint32), y[k].view(np.uint32)), (what, "batch", i, k, int((x[k].view(np.uint32) != y[k].view(np.uint32)).any(axis=-1).sum()))
        if x["diag"] is not None and y["diag"] is not None:
            assert np.array_equal(x["diag"].view(np.uint32), y["diag"].view(np.uint32)), (what, "diagnostics of batch", i)


CASES = [("cover", 1920, 1080, 3, 8, 3, 4, 1, 0), ("cover", 96, 54, 5, 8, 16, 16, 1, 0), ("cover", 100, 37, 4, 12, 5, 4, 3, 1), ("moving", 640, 360, 4, 8, 4, 16, 1, 0),
         ("mixed", 320, 200, 3, 6, 5, 4, 2, 1), ("volumes", 256, 144, 3, 10, 4, 16, 1, 0), ("textured", 256, 144, 3, 6, 3, 4, 1, 0), ("twins", 320, 180, 4, 8, 3, 16, 1, 0),
         ("mesh", 320, 200, 3, 6, 3, 4, 1, 0), ("cover", 64, 40, 2, 20, 19, 4, 1, 0)]


@pytest.mark.parametrize("name,w,h,spp,depth,count,stride,divider,offset", CASES)
def test_group_equals_the_batches_launched_separately(rt, gpu_context, name, w, h, spp, depth, count, stride, divider, offset):
    S = rt.scenes
    scene = {"cover": S.cover_scene, "moving": S.moving_scene, "mixed": S.mixed_scene, "volumes": S.volume_scene, "textured": S.textured_scene,
             "twins": S.twin_spheres_scene, "mesh": S.mesh_scene}[name]()
    ctx = gpu_context
    ctx.upload_scene(scene.desc())
    n = w * h
    plist = [rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=300 + 11 * k, diagnostics_stride=stride, slice_offset=offset, slice_divider=divider,
                                   focus=6.0 if name != "cover" else None) for k in range(count)]
    ins = _inputs(n, 5)
    fill = {k: np.full((n, c), -7.0, np.float32) for k, c in KEYS}              # rows a slice does not own must stay untouched in every output
    sep = _separate(rt, ctx, plist, ins, n, stride, fill)
    grp = _grouped(rt, ctx, plist, ins, n, stride, fill)
    _same(grp, sep, (name, w, h, count))
    assert not np.array_equal(grp[0]["color"], grp[1]["color"])                  # different Seeds, different batches
    if divider > 1:
        rows = (np.arange(n) // w) % divider != offset
        assert np.all(grp[-1]["color"][rows] == -7.0)
    ctx.batch_status()


def test_group_without_diagnostics_and_with_batches_that_differ_in_more_than_seed(rt, gpu_context):
    """No diagnostics at all (NULL array); and a group whose batches differ in sample count and trace depth: not one launch, the same result."""
    scene = rt.scenes.cover_scene()
    ctx = gpu_context
    ctx.upload_scene(scene.desc())
    w, h = 160, 90
    n = w * h
    ins = _inputs(n, 9)
    fill = {k: np.zeros((n, c), np.float32) for k, c in KEYS}
    plist = [rt.scenes.make_params(scene, w, h, spp=3, trace_depth=8, seed=40 + k) for k in range(4)]
    sep = _separate(rt, ctx, plist, ins, n, 4, fill)
    grp = _grouped(rt, ctx, plist, ins, n, 4, fill, with_diag=False)
    _same(grp, sep, "no diagnostics")
    mixed = [rt.scenes.make_params(scene, w, h, spp=2 + k, trace_depth=6 + 2 * k, seed=70 + k) for k in range(3)]
    _same(_grouped(rt, ctx, mixed, ins, n, 4, fill), _separate(rt, ctx, mixed, ins, n, 4, fill), "batches that differ in more than Seed")


def test_group_rejects_shared_outputs(rt, gpu_context):
    a = rt.abi
    scene = rt.scenes.cover_scene()
    ctx = gpu_context
    ctx.upload_scene(scene.desc())
    w, h = 64, 36
    n = w * h
    src = [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
    o1 = [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
    plist = [rt.scenes.make_params(scene, w, h, spp=2, trace_depth=6, seed=s) for s in (1, 2)]
    assert rt.sample_batch_group_device(ctx, plist, src, [o1, o1]) == a.RTOW_ERROR_INVALID_VALUE        # two batches, one output
    assert rt.sample_batch_group_device(ctx, plist, src, [o1, src]) == a.RTOW_ERROR_INVALID_VALUE       # an output that is the shared input
    assert rt.sample_batch_group_device(ctx, plist[:1], src, [src]) == 0                                 # alone, a batch may accumulate in place like rtowSampleBatchDevice
    ctx.synchronize()
    for b in src + o1:
        b.free()


def test_group_can_be_cancelled(rt, gpu_context):
    scene = rt.scenes.cover_scene()
    ctx = gpu_context
    ctx.upload_scene(scene.desc())
    w, h = 1920, 1080
    n = w * h
    src = [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
    outs = [[rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS] for _ in range(4)]
    plist = [rt.scenes.make_params(scene, w, h, spp=512, trace_depth=8, seed=s) for s in range(1, 5)]
    token = C.c_uint8(1)                                                             # already cancelled: the launch drains at its first ticket requests
    import time
    t = time.perf_counter()
    rc = rt.sample_batch_group_device(ctx, plist, src, outs, None, None, C.addressof(token))
    dt = time.perf_counter() - t
    assert rc == rt.abi.RTOW_ERROR_CANCELLED and dt < 0.4, (rc, dt)                  # 4 x 512 spp at 1080p would take 0.9 s (the bound leaves room for a GPU shared with other test processes)
    for b in src + [x for o in outs for x in o]:
        b.free()


@pytest.mark.parametrize("slots", [1, 2, 4, 7, 15])
@pytest.mark.parametrize("w,h,count,divider,offset", [(520, 264, 5, 1, 0), (100, 37, 3, 3, 1), (1032, 520, 10, 1, 0)])
def test_slot_tickets_any_number_of_slots_per_pull(rt, w, h, count, divider, offset, slots):
    """A group's tickets are slot << 6 | pixel of the chunk; a wave reserves `slots` (chunk, batch) slots with one atomic (schedulerTune[7] + 256 x slots; 4 built in) and finds chunk
    and batch from the ticket.  Whatever the number - also one that does not divide the slot count, with a last chunk that is not full (100 x 37 sliced in three: 1 300 owned pixels)
    and with more chunks than the order needs (1 032 x 520: 8 385 chunks, the cost-ordered hand-out is on) - the group equals the batches launched separately, bit for bit."""
    scene = rt.scenes.cover_scene()
    n = w * h
    plist = [rt.scenes.make_params(scene, w, h, spp=2, trace_depth=8, seed=900 + 7 * k, slice_offset=offset, slice_divider=divider) for k in range(count)]
    ins = _inputs(n, 9)
    fill = {k: np.full((n, c), -3.0, np.float32) for k, c in KEYS}
    with rt.Context(0, scheduler_tune=(0, 0, 0, 0, 0, 0, 0, 3 + 256 * slots, 0)) as ctx:
        ctx.upload_scene(scene.desc())
        sep = _separate(rt, ctx, plist, ins, n, 4, fill)
        for _ in range(2):                                        # second launch: order and ticket maps come from the first one's ray counts
            grp = _grouped(rt, ctx, plist, ins, n, 4, fill)
            _same(grp, sep, (w, h, count, slots))
