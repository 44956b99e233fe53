"""GPU: rtowProbeNearestHit against the oracle's Raytracer.HitWorld (oracle_hit_world: the recursive HitTests.Hit(BvhNode), RT/HitTests.cs:152-196) - the host's
auto-focus probe (UNITY/Raytracer.cs:608-609).  Distance bit for bit, hit / miss, and the entity wherever the nearest distance is not shared."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.float32(x).view(np.uint32)


def _rays(scene, count, seed):
    rng = np.random.default_rng(seed)
    cam = np.asarray(scene.camera["position"], dtype=np.float32)
    target = np.asarray(scene.camera["target"], dtype=np.float32)
    yield cam, (target - cam).astype(np.float32)                                   # the view axis itself, unnormalised
    for k in range(count):
        o = cam if k % 2 == 0 else (cam + rng.normal(size=3) * 2.0).astype(np.float32)
        d = (target - o + rng.normal(size=3) * (0.05 if k % 4 == 0 else 1.5)).astype(np.float32)
        if k % 3 == 0: d = (d / np.linalg.norm(d)).astype(np.float32)
        if k % 17 == 0: d = -d                                                     # mostly misses
        yield o, d


SCENES = ["cover", "moving", "mixed", "volumes", "mesh", "textured", "twins", "tiny", "stress", "coplanar"]


@pytest.mark.parametrize("name", SCENES)
def test_probe_equals_hit_world(rt, oracle, name):
    S = rt.scenes
    scene = {"cover": S.cover_scene, "moving": S.moving_scene, "mixed": S.mixed_scene, "volumes": S.volume_scene, "mesh": lambda: S.mesh_scene(3), "textured": S.textured_scene,
             "twins": lambda: S.twin_spheres_scene(True), "tiny": S.tiny_scene, "stress": lambda: S.stress_scene(count=3000, max_tentatives=12000), "coplanar": S.coplanar_scene}[name]()
    desc = scene.desc()
    osc = oracle.OracleScene(desc)
    rng = np.random.default_rng(3)
    hits = misses = entities = 0
    with rt.Context(0) as ctx:
        ctx.upload_scene(desc)
        for k, (o, d) in enumerate(_rays(scene, 300, 7)):
            time = float(np.float32(rng.random())) if name in ("moving", "twins") and k % 2 else 0.0
            ref_hit, ref = osc.hit_world(o, d, time)
            hit, dist, ent = ctx.hit_world(o, d, time)
            assert hit == ref_hit, (name, k)
            if not hit:
                misses += 1
                assert ent == -1 and np.isposinf(dist)
                continue
            hits += 1
            assert _bits(dist) == _bits(ref[0]), (name, k, dist, ref[0])
            n, job = osc.nearest_hit(o, d, time)                                  # the job's sorted hit list: is the nearest distance shared?
            if int(ref[7]) == int(job[7]):                                        # both reference procedures name the same entity: no tie to argue about
                entities += 1
                assert ent == int(ref[7]), (name, k)
    osc.close()
    assert hits > 50 and misses > 5 and entities > 40, (hits, misses, entities)


def test_probe_before_upload_and_null_outputs(rt):
    import ctypes as C
    a = rt.abi
    lib = rt.lib.load()
    with rt.Context(0) as ctx:
        o, d = a.Float3(0, 0, 5), a.Float3(0, 0, -1)
        assert lib.rtowProbeNearestHit(ctx.handle, C.byref(o), C.byref(d), 0.0, None, None) == a.RTOW_ERROR_NO_SCENE
        ctx.upload_scene(rt.scenes.tiny_scene().desc())
        assert lib.rtowProbeNearestHit(ctx.handle, C.byref(o), C.byref(d), 0.0, None, None) == 0
        assert lib.rtowProbeNearestHit(ctx.handle, None, C.byref(d), 0.0, None, None) == a.RTOW_ERROR_INVALID_VALUE


def test_probe_between_batches_in_flight(rt, oracle):
    """The host calls it from ScheduleSample while the previous batch is still running (UNITY/Raytracer.cs:586-611): the probe is ordered behind the context's own
    stream and must neither disturb the batch nor be disturbed."""
    S = rt.scenes
    scene = S.cover_scene()
    desc = scene.desc()
    w, h = 640, 360
    n = w * h
    p = S.make_params(scene, w, h, spp=16, trace_depth=8)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p)
    o = np.asarray(scene.camera["position"], dtype=np.float32)
    fwd = (np.asarray(scene.camera["target"], dtype=np.float32) - o).astype(np.float32)
    ref_hit, rh = osc.hit_world(o, fwd, 0.0)
    osc.close()
    with rt.Context(0) as ctx:
        ctx.upload_scene(desc)
        bufs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
        job = rt.SampleBatchJob(ctx, p)
        job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
        job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = bufs
        handle = job.Schedule()
        hit, dist, ent = ctx.hit_world(o, fwd, 0.0)
        assert hit == ref_hit and _bits(dist) == _bits(rh[0]) and ent == int(rh[7])
        assert handle.Complete() == 0
        ctx.synchronize()
        for k, b, c in zip(("color", "normal", "albedo", "scw"), bufs, (4, 3, 3, 1)):
            assert np.array_equal(b.download(np.float32, (n, c)).reshape(-1).view(np.uint32), ref[k].reshape(-1).view(np.uint32)), k
