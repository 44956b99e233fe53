"""GPU: seeded random scenes - random mixes of every entity type, transform, material class, texture kind, camera (also inside geometry),
lens, noise colour, RNG policy, slice and trace depth, and kernel family: forced 32-bit candidate codes,
a chain of two batches, (round 4) a group of two independent batches - rendered small and compared with the oracle bit for bit.  Catches the
combinations nobody thought of writing a scene for."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _random_scene(rt, seed, only_triangles=False):
    """only_triangles: every entity becomes a triangle and there is no ground sphere (the all-triangle scene kinds); the random stream is consumed
    exactly as without it, so every other seed keeps its scene."""
    S, abi = rt.scenes, rt.abi
    rng = np.random.default_rng(seed)
    s = S.Scene("fuzz%d" % seed)
    use_volumes = rng.random() < 0.35
    use_textures = rng.random() < 0.35
    if use_textures:
        s.images = [rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(3, 5))), dtype=np.uint8) for _ in range(2)]

    def tex_or(const):
        if use_textures and rng.random() < 0.4:
            return S.image_tex(int(rng.integers(-1, 2)), tuple(rng.random(3) * 1.2), channel=int(rng.integers(0, 3)))
        return const

    def material():
        k = rng.random()
        if use_volumes and k < 0.25:
            return S.volume(tuple(rng.random(3)), float(rng.uniform(0.2, 3.0)))
        if k < 0.45:
            m = S.lambertian(tuple(rng.random(3)))
        elif k < 0.7:
            m = S.standard(tuple(rng.uniform(0.3, 1.0, 3)), float(rng.choice([0.0, 0.4, 1.0])), float(rng.choice([0.0, 0.6, 1.0])),
                           emission=tuple(rng.random(3) * 3) if rng.random() < 0.3 else None)
        elif k < 0.85:
            m = S.metal(tuple(rng.uniform(0.5, 1.0, 3)), float(rng.uniform(0.0, 0.5)))
        else:
            m = S.dielectric(float(rng.uniform(1.1, 2.0)))
            m.glossiness = tex_or(S._const_tex(float(rng.choice([1.0, 0.7]))))
            return m
        m.albedo = tex_or(m.albedo)
        if m.type == abi.MATERIAL_STANDARD and rng.random() < 0.5:
            m.glossiness = tex_or(m.glossiness)
            m.metallic = tex_or(m.metallic)
        return m

    def quat():
        if rng.random() < 0.4:
            return (0.0, 0.0, 0.0, 1.0)
        axis = rng.normal(size=3)
        return S.quat_axis_angle(tuple(axis / np.linalg.norm(axis)), float(rng.uniform(-180, 180)))

    n = int(rng.integers(1, 14))
    if os.environ.get("RTOW_FUZZ_HEAVY") and rng.random() < 0.5:
        n = int(rng.integers(14, 120))                                    # soak runs: deeper trees, candidate-list flushes, longer hit lists
    for _ in range(n):
        pos = tuple(rng.uniform(-2.5, 2.5, 3))
        moving = rng.random() < 0.25
        kw = dict(moving=moving, dest_offset=tuple(rng.uniform(-0.6, 0.6, 3)) if moving else (0, 0, 0), time_range=(0.0, 1.0) if moving else (0, 0))
        t = rng.random()
        if only_triangles:
            t = 0.9
        copies = 2 if rng.random() < 0.06 else 1            # now and then the same primitive twice (other material): nearest-hit ties everywhere on it
        if t < 0.4:
            r = float(rng.uniform(0.2, 1.1)) * (-1 if rng.random() < 0.1 else 1)
            for c in range(copies):
                s.add_sphere(pos, r if c == 0 or rng.random() < 0.5 else -r, material(), **kw)
        elif t < 0.6:
            size, q = tuple(rng.uniform(0.5, 3.0, 2)), quat()
            for _c in range(copies):
                s.add_rect(pos, size, material(), rotation=q, **kw)
        elif t < 0.8:
            size, q = tuple(rng.uniform(0.3, 1.8, 3)), quat()
            for _c in range(copies):
                s.add_box(pos, size, material(), rotation=q, **kw)
        else:
            v = [np.array(pos) + rng.uniform(-1.5, 1.5, 3) for _ in range(3)]
            uvs = tuple(tuple(rng.uniform(-0.2, 1.3, 2)) for _ in range(3))
            for _c in range(copies):
                s.add_triangle(v[0], v[1], v[2], material(), uvs=uvs)
    if rng.random() < 0.6 and not only_triangles:
        s.add_sphere((0, -101.5, 0), 100.0, S.lambertian((0.5, 0.5, 0.5)))
    cam = rng.uniform(-4, 4, 3)
    if rng.random() < 0.2 and n > 0:
        cam = np.asarray(s.positions[0], dtype=np.float64) + rng.uniform(-0.05, 0.05, 3)      # inside / on the first entity
    s.camera = {"position": [float(c) for c in cam], "target": [float(c) for c in rng.uniform(-1, 1, 3)], "up": [0.0, 1.0, 0.0],
                "vfov": float(rng.uniform(20, 100)), "aperture": float(rng.choice([0.0, 0.0, 0.1, 0.8]))}
    s.sky_bottom, s.sky_top = tuple(rng.random(3)), tuple(rng.random(3))
    return s, rng


@pytest.fixture(scope="module")
def geometry_contexts(rt):
    """Contexts that force the kernels a small scene would not pick by itself: 32-bit candidate codes (what scenes beyond 65 535 entities use) and
    the 512- / 256-lane launch geometries (what a tile-split launch may pick)."""
    ctxs = {"wide": rt.Context(0, flags=rt.abi.CONTEXT_FORCE_WIDE_CODES)}
    yield ctxs
    for c in ctxs.values():
        c.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("RTOW_FUZZ_SEEDS", "24")))))   # RTOW_FUZZ_SEEDS=1000 for a soak run
def test_random_scene(rt, oracle, gpu_context, geometry_contexts, seed):
    _run_seed(rt, oracle, gpu_context, geometry_contexts, seed)


@pytest.mark.parametrize("seed", [115, 417, 611, 666, 3, 58])
def test_heavy_seeds(rt, oracle, gpu_context, geometry_contexts, seed, monkeypatch):
    """The soak generator (RTOW_FUZZ_HEAVY: up to 120 entities, larger frames) on the seeds whose launches round 6's first LDS plan refused - 32-bit stack rows of a deep
    tree next to the 32 path-history rows of trace depth 40 had no room, RTOW_ERROR_CAPACITY with hit lists of 6 to 20 entries (profiles/r06_runs/run_r06l.sh) - and two more."""
    monkeypatch.setenv("RTOW_FUZZ_HEAVY", "1")
    _run_seed(rt, oracle, gpu_context, geometry_contexts, seed)


def _run_seed(rt, oracle, gpu_context, geometry_contexts, seed):
    abi = rt.abi
    # (drawn from a generator of its own so that every seed keeps the scene and parameters it always had)
    extra = np.random.default_rng(77000 + seed)
    only_triangles = bool(extra.random() < 0.12)
    scene, rng = _random_scene(rt, 1000 + seed, only_triangles)
    desc = scene.desc(max_bvh_depth=int(rng.choice([32, 32, 3])))
    which = str(extra.choice(["default", "default", "default", "wide"]))
    chain = bool(extra.random() < 0.25)
    group = (not chain) and bool(extra.random() < 0.2)      # (drawn last: every seed keeps what it had)
    ctx = gpu_context if which == "default" else geometry_contexts[which]
    ctx.upload_scene(desc)
    noise_color = int(rng.choice([abi.NOISE_WHITE, abi.NOISE_WHITE, abi.NOISE_BLUE, abi.NOISE_SPATIOTEMPORAL_BLUE]))
    policy = int(rng.choice([abi.RNG_REFERENCE, abi.RNG_REFERENCE, abi.RNG_PER_SAMPLE, abi.RNG_PER_SAMPLE_XOROSHIRO])) if noise_color == abi.NOISE_WHITE else abi.RNG_REFERENCE
    div = int(rng.choice([1, 1, 2, 3]))
    big = 3 if os.environ.get("RTOW_FUZZ_HEAVY") else 1
    p = rt.scenes.make_params(scene, int(rng.integers(8, 48 * big)), int(rng.integers(8, 36 * big)), spp=int(rng.integers(1, 20)), trace_depth=int(rng.choice([1, 2, 5, 8, 12, 17, 40])),
                              seed=int(rng.integers(1, 1 << 30)), jitter=bool(rng.random() < 0.8), slice_offset=int(rng.integers(0, div)), slice_divider=div,
                              diagnostics_stride=int(rng.choice([4, 16])), focus=float(rng.uniform(1.0, 8.0)), noise_color=noise_color,
                              noise_texture_index=int(rng.integers(0, 2)), rng_policy=policy,
                              sky_type=int(rng.choice([abi.SKY_GRADIENT, abi.SKY_GRADIENT, abi.SKY_CUBEMAP, abi.SKY_NONE])))
    noise = rt.scenes.NoiseTextures(row_stride=8, count=2, seed=seed)
    sky = rt.scenes.synthetic_sky(size=8, half=bool(rng.random() < 0.5), seed=seed)
    osc = oracle.OracleScene(desc)
    try:
        ctx.upload_blue_noise(noise.blue_desc()); ctx.upload_stb_noise(noise.stb_desc()); ctx.upload_sky_cubemap(sky.desc())
        osc.set_blue_noise(noise.blue_desc()); osc.set_stb_noise(noise.stb_desc()); osc.set_cubemap(sky.desc())
        n = int(p.size.x) * int(p.size.y)
        ins = {"color": rng.random((n, 4)).astype(np.float32), "normal": rng.normal(size=(n, 3)).astype(np.float32),
               "albedo": rng.random((n, 3)).astype(np.float32), "scw": rng.random(n).astype(np.float32)}
        ins["color"][:, 3] = rng.integers(0, 5, n)
        p2 = abi.SampleParams.from_buffer_copy(p)
        p2.seed = (p.seed * 7 + 1) & 0x3fffffff
        try:
            if chain:                                    # two successive batches as one rtowSampleBatchChain call: defined as the batches in sequence
                gpu = rt.sample_batch_chain_host(ctx, [p, p2], ins)
                gpu["diag"] = gpu["diag"][-1]
            elif group:                                  # two independent batches as one rtowSampleBatchGroupDevice call: defined as the separate calls on the same inputs
                keys = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))
                src = [rt.DeviceBuffer(ctx).upload(ins[k]) for k, _ in keys]
                outs = [[rt.DeviceBuffer(ctx).upload(ins[k]) for k, _ in keys] for _ in range(2)]      # rows a slice does not own keep the inputs, like the host form
                dg = [rt.DeviceBuffer(ctx, n * p.diagnosticsStride).zero() for _ in range(2)]
                rc = rt.sample_batch_group_device(ctx, [p, p2], src, outs, dg)
                if rc == abi.RTOW_SUCCESS:
                    try:
                        ctx.synchronize()
                    except rt.lib.RtowError as e:
                        rc = e.code
                gpus = []
                for o, d in zip(outs, dg):
                    g = {k: b.download(np.float32, (n, c) if c > 1 else (n,)) for (k, c), b in zip(keys, o)}
                    g["diag"] = d.download(np.float32, (n, p.diagnosticsStride // 4))
                    gpus.append(g)
                for b in src + outs[0] + outs[1] + dg:
                    b.free()
                if rc != abi.RTOW_SUCCESS:
                    raise rt.lib.RtowError(rc, "rtowSampleBatchGroupDevice")
                ref2 = osc.sample_batch(p2, ins)
                for k in ("color", "normal", "albedo", "scw"):
                    assert np.array_equal(gpus[1][k].reshape(ref2[k].shape).view(np.uint32), ref2[k].view(np.uint32)), (seed, which, "group batch 1", k)
                assert np.array_equal(gpus[1]["diag"][:, 0], ref2["diag"][:, 0]), (seed, which, "group batch 1")
                gpu = {k: gpus[0][k].reshape(ins[k].shape) for k in ("color", "normal", "albedo", "scw")}
                gpu["diag"] = gpus[0]["diag"]
            else:
                gpu = rt.sample_batch_host(ctx, p, ins)
        except rt.lib.RtowError as e:
            # the one legitimate refusal: a ray whose whole hit list is needed (volume scenes; nearest-hit ties in scenes with duplicate
            # primitives) met more surfaces than the context's hitListCapacity (default 1024; lists beyond 24 entries spill to HBM) - the
            # oracle must confirm that such a ray exists
            assert e.code == abi.RTOW_ERROR_CAPACITY, e
            _, counters = osc.sample_batch(p, ins, want_counters=True)
            most = counters.maxHits
            if chain or group:
                _, c2 = osc.sample_batch(p2, ins, want_counters=True)
                most = max(most, c2.maxHits)
            assert most > 1024, (seed, most)
            return
        ref = osc.sample_batch(p, ins)
        if chain:
            ref = osc.sample_batch(p2, {k: ref[k] for k in ("color", "normal", "albedo", "scw")})
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (seed, which, chain, k)
        assert np.array_equal(gpu["diag"][:, 0], ref["diag"][:, 0]), (seed, which, chain)
    finally:
        ctx.upload_blue_noise(None); ctx.upload_stb_noise(None); ctx.upload_sky_cubemap(None)
        osc.close()
