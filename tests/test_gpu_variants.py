"""GPU: every compiled variant of the sample kernel, one by one.

The kernel is instantiated per (tree in LDS | in HBM) x scene kind x the (history width, diagnostics, noise source, RNG policy) combinations
`launchByDiag` (end of csrc/rtow_sample_kernel.hip.h) dispatches to - 338 kernels at the end of round 6, the spilling ones compiled under heavy register pressure.  During development one of them (VOLUMES, spatio-temporal noise,
short diagnostics) was once MISCOMPILED by hipcc (ROCm 7.2): a VGPR spill store was scheduled in front of the `s_or_b64 exec` of a join
block, so the lanes that had skipped the region kept a stale spill slot and later reloaded it - their ray-count diagnostic came out 0
while the colours were right.  Any source change reshuffles the register allocation of all of them, so every build
renders a small, divergent frame through each of them and compares every output with the oracle, bit for bit."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = ["spheres", "spheres_ties", "spheres_motion", "spheres_motion_ties", "general", "general_ties", "volumes", "textured", "textured_ties", "volumes_textured",
         "triangles", "triangles_ties", "triangles_textured", "triangles_textured_ties"]


def _scene(rt, kind):
    S = rt.scenes
    # the *_ties kernels (duplicate spheres: nearest-hit ties go through resolve_nearest_tie) are chosen at upload when the scene holds duplicates
    def textured_with_twins():
        s = S.textured_scene()
        s.add_sphere((0.4, 0.6, 0.3), 0.35, S.lambertian((0.8, 0.3, 0.2)))
        s.add_sphere((0.4, 0.6, 0.3), 0.35, S.metal((0.9, 0.9, 0.9), 0.1))
        return s

    def few_triangles():
        # 14 triangles (a tetrahedron-ish fan, a slanted pane of two, a floor of two, glass and metal among them): all-triangle scene of at most 16
        # entities -> SCENE_KIND_TRIANGLES without the exact-tie resolver
        s = S.Scene("few triangles")
        mats = [S.lambertian((0.7, 0.3, 0.3)), S.metal((0.8, 0.8, 0.9), 0.1), S.dielectric(1.5), S.standard((0.4, 0.7, 0.3), 0.3, 0.7, emission=(0.3, 0.2, 0.1))]
        apex = (0.0, 1.6, 0.0)
        ring = [(math.cos(a) * 1.1, 0.2, math.sin(a) * 1.1) for a in (0.0, 1.3, 2.5, 3.8, 5.0)]
        for k in range(5):
            s.add_triangle(apex, ring[k], ring[(k + 1) % 5], mats[k % 4])
        for k in range(3):
            s.add_triangle((0.0, 0.2, 0.0), ring[(k + 1) % 5], ring[k], mats[(k + 1) % 4])
        S._quad(s, (-1.8, 0.1, -1.5), (-0.9, 0.1, -1.9), (-0.9, 1.7, -1.9), (-1.8, 1.7, -1.5), mats[1])
        S._quad(s, (1.0, 0.3, 1.2), (1.9, 0.3, 0.6), (1.9, 1.4, 0.6), (1.0, 1.4, 1.2), mats[2])
        S._quad(s, (-20, 0, -20), (20, 0, -20), (20, 0, 20), (-20, 0, 20), S.lambertian((0.5, 0.5, 0.5)))
        assert s.entity_count == 14
        s.camera = {"position": [0.4, 1.9, 5.5], "target": [0.0, 0.7, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.05}
        return s

    return {"triangles": few_triangles, "triangles_ties": lambda: S.mesh_scene(1),
            "triangles_textured": lambda: S.textured_scene(triangles_only=True), "triangles_textured_ties": S.textured_mesh_scene,
            "spheres": lambda: S.cover_scene(60, 600), "spheres_ties": S.twin_spheres_scene, "spheres_motion": S.tiny_scene,
            "spheres_motion_ties": lambda: S.twin_spheres_scene(True), "general": S.mixed_scene, "general_ties": S.coplanar_scene,
            "volumes": S.volume_tie_scene, "textured": S.textured_scene, "textured_ties": textured_with_twins,
            "volumes_textured": S.textured_volume_scene}[kind]()


# (trace depth -> history width 4 / 8 / 32, noise, rng policy); the texture-driven noise sources only exist with history width 32
def _modes(abi):
    out = []
    for depth in (5, 12, 20):
        out.append((depth, abi.NOISE_WHITE, abi.RNG_REFERENCE))
        out.append((depth, abi.NOISE_WHITE, abi.RNG_PER_SAMPLE))
    out.append((40, abi.NOISE_WHITE, abi.RNG_REFERENCE))      # a path deeper than 32 segments through the generic 32-word history (64 is the limit)
    out.append((5, abi.NOISE_WHITE, abi.RNG_PER_SAMPLE_XOROSHIRO))
    out.append((12, abi.NOISE_WHITE, abi.RNG_PER_SAMPLE_XOROSHIRO))
    out.append((6, abi.NOISE_BLUE, abi.RNG_REFERENCE))
    out.append((6, abi.NOISE_SPATIOTEMPORAL_BLUE, abi.RNG_REFERENCE))
    return out


@pytest.mark.parametrize("in_lds", [True, False], ids=["lds", "hbm"])
@pytest.mark.parametrize("kind", KINDS)
def test_every_kernel_variant(rt, oracle, kind, in_lds):
    abi = rt.abi
    scene = _scene(rt, kind)
    desc = scene.desc()
    log = []
    # RtowContextOptions.ldsSceneBudgetBytes = 1024: 16 nodes in LDS, everything else read through L2
    ctx = rt.Context(0, log=lambda lvl, tag, msg, ud: log.append(msg.decode()), log_level=4, lds_scene_budget=0 if in_lds else 1024)
    ctx.upload_scene(desc)
    assert bool(ctx.scene_info().sceneInLds) == in_lds
    assert any("exact-tie kernels" in m for m in log) == kind.endswith("_ties"), log        # the scene really selects the variant it is meant to cover
    noise = rt.scenes.NoiseTextures(row_stride=8, count=2, seed=3)
    ctx.upload_blue_noise(noise.blue_desc())
    ctx.upload_stb_noise(noise.stb_desc())
    osc = oracle.OracleScene(desc)
    osc.set_blue_noise(noise.blue_desc())
    osc.set_stb_noise(noise.stb_desc())
    w, h = 40, 24
    rng = np.random.default_rng(11)
    ins = {"color": rng.random((w * h, 4)).astype(np.float32), "normal": rng.normal(size=(w * h, 3)).astype(np.float32),
           "albedo": rng.random((w * h, 3)).astype(np.float32), "scw": rng.random(w * h).astype(np.float32)}
    ins["color"][:, 3] = rng.integers(0, 4, w * h)
    try:
        for depth, noise_color, policy in _modes(abi):
            for stride in (4, 16):
                p = rt.scenes.make_params(scene, w, h, spp=3, trace_depth=depth, seed=77, diagnostics_stride=stride, noise_color=noise_color,
                                          noise_texture_index=1, rng_policy=policy)
                gpu = rt.sample_batch_host(ctx, p, ins)
                ref = osc.sample_batch(p, ins)
                where = (kind, in_lds, depth, noise_color, policy, stride)
                for k in ("color", "normal", "albedo", "scw"):
                    assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (where, k)
                assert np.array_equal(gpu["diag"][:, 0], ref["diag"][:, 0]), (where, "ray count")        # columns 1, 2 count visits of the (different) tree
                if stride == 16:
                    assert np.array_equal(gpu["diag"][:, 3].view(np.uint32), ref["diag"][:, 3].view(np.uint32)), (where, "sample count weight")
    finally:
        osc.close()


@pytest.mark.parametrize("in_lds", [True, False], ids=["lds", "hbm"])
@pytest.mark.parametrize("kind", ["spheres", "spheres_motion"])
def test_generic_variants_with_and_without_the_lanes_in_a_hurry(rt, oracle, kind, in_lds):
    """The static-sphere kind's generic reference-stream variants (paths deeper than 16 segments, or 16-byte records) exist twice since round 6: plain and chained launches run the
    twins in which a pixel far beyond the mean ray count stops waiting for company (GEO bit 4), batch groups the variants without that code.  test_every_kernel_variant's plain
    launches reach the twins; here the same batches as a batch group reach the others - both against the oracle, tree in LDS and beyond.  (Moving spheres have no twin: there the
    two forms run the same kernels, and must agree all the same.)"""
    scene = _scene(rt, kind)
    desc = scene.desc()
    w, h = 160, 96
    n = w * h
    keys = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))
    rng = np.random.default_rng(12)
    ins = {"color": rng.random((n, 4)).astype(np.float32), "normal": rng.normal(size=(n, 3)).astype(np.float32),
           "albedo": rng.random((n, 3)).astype(np.float32), "scw": rng.random(n).astype(np.float32)}
    ins["color"][:, 3] = rng.integers(0, 4, n)
    osc = oracle.OracleScene(desc)
    try:
        with rt.Context(0, lds_scene_budget=0 if in_lds else 1024) as ctx:
            ctx.upload_scene(desc)
            assert bool(ctx.scene_info().sceneInLds) == in_lds
            for depth, stride in ((64, 4), (20, 16), (5, 16)):
                plist = [rt.scenes.make_params(scene, w, h, spp=12, trace_depth=depth, seed=90 + k, diagnostics_stride=stride) for k in range(2)]
                refs = [osc.sample_batch(p, ins) for p in plist]
                src = [rt.DeviceBuffer(ctx).upload(ins[k]) for k, _ in keys]
                outs = [[rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in keys] for _ in plist]
                diags = [rt.DeviceBuffer(ctx, n * stride).zero() for _ in plist]
                assert rt.sample_batch_group_device(ctx, plist, src, outs, diags) == 0
                ctx.synchronize()
                for b, ref in enumerate(refs):
                    plain = rt.sample_batch_host(ctx, plist[b], ins)
                    for (k, c), buf in zip(keys, outs[b]):
                        got = buf.download(np.float32, (n, c)).reshape(ref[k].shape)
                        assert np.array_equal(got.view(np.uint32), ref[k].view(np.uint32)), (kind, in_lds, depth, stride, b, k, "group")
                        assert np.array_equal(plain[k].view(np.uint32), ref[k].view(np.uint32)), (kind, in_lds, depth, stride, b, k, "plain")
                    assert np.array_equal(diags[b].download(np.float32, (n, stride // 4))[:, 0], ref["diag"][:, 0]), (kind, in_lds, depth, stride, b, "ray count")
                if depth == 64:
                    # a first sample of more than 18 x 2 (tree beyond LDS: 14 x 2) rays puts its lane in a hurry at once; the oracle's 1-sample batch with the same Seed traces it
                    first = max(osc.sample_batch(rt.scenes.make_params(scene, w, h, spp=1, trace_depth=depth, seed=90 + k, diagnostics_stride=stride), ins)["diag"][:, 0].max() for k in range(2))
                    assert first > 36 or kind != "spheres", "no first sample beyond 36 rays: the twins ran like the others"
                for b in src + [x for o in outs for x in o] + diags:
                    b.free()
    finally:
        osc.close()


def _compare_modes(rt, oracle, ctx, scene, desc, modes, where):
    abi = rt.abi
    noise = rt.scenes.NoiseTextures(row_stride=8, count=2, seed=3)
    ctx.upload_blue_noise(noise.blue_desc())
    ctx.upload_stb_noise(noise.stb_desc())
    osc = oracle.OracleScene(desc)
    osc.set_blue_noise(noise.blue_desc())
    osc.set_stb_noise(noise.stb_desc())
    w, h = 40, 24
    rng = np.random.default_rng(11)
    ins = {"color": rng.random((w * h, 4)).astype(np.float32), "normal": rng.normal(size=(w * h, 3)).astype(np.float32),
           "albedo": rng.random((w * h, 3)).astype(np.float32), "scw": rng.random(w * h).astype(np.float32)}
    ins["color"][:, 3] = rng.integers(0, 4, w * h)
    try:
        for depth, noise_color, policy, stride in modes:
            p = rt.scenes.make_params(scene, w, h, spp=3, trace_depth=depth, seed=77, diagnostics_stride=stride, noise_color=noise_color, noise_texture_index=1, rng_policy=policy)
            gpu = rt.sample_batch_host(ctx, p, ins)
            ref = osc.sample_batch(p, ins)
            for k in ("color", "normal", "albedo", "scw"):
                assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (where, depth, noise_color, policy, stride, k)
            assert np.array_equal(gpu["diag"][:, 0], ref["diag"][:, 0]), (where, depth, noise_color, policy, stride, "ray count")
    finally:
        osc.close()


@pytest.mark.parametrize("kind", KINDS)
def test_wide_code_variants(rt, oracle, kind):
    """The kernels with 32-bit candidate / stack codes and 4 x 32-bit camera-ray lists (scenes beyond 65 535 entities or tree nodes), forced onto
    small scenes with RTOW_CONTEXT_FORCE_WIDE_CODES: per scene kind (the volume kinds included) the reference-stream variants of every history width and
    record format, and the generic one per noise source / RNG policy."""
    abi = rt.abi
    scene = _scene(rt, kind)
    desc = scene.desc()
    with rt.Context(0, flags=abi.CONTEXT_FORCE_WIDE_CODES) as ctx:
        ctx.upload_scene(desc)
        info = ctx.scene_info()
        assert info.wideCodes == 1 and not info.sceneInLds
        modes = [(5, abi.NOISE_WHITE, abi.RNG_REFERENCE, 4), (12, abi.NOISE_WHITE, abi.RNG_REFERENCE, 4), (20, abi.NOISE_WHITE, abi.RNG_REFERENCE, 4), (5, abi.NOISE_WHITE, abi.RNG_REFERENCE, 16),
                 (5, abi.NOISE_WHITE, abi.RNG_PER_SAMPLE, 4), (12, abi.NOISE_WHITE, abi.RNG_PER_SAMPLE_XOROSHIRO, 16), (6, abi.NOISE_BLUE, abi.RNG_REFERENCE, 4),
                 (6, abi.NOISE_SPATIOTEMPORAL_BLUE, abi.RNG_REFERENCE, 16)]
        _compare_modes(rt, oracle, ctx, scene, desc, modes, (kind, "wide"))


@pytest.mark.parametrize("in_lds", [True, False], ids=["lds", "hbm"])
@pytest.mark.parametrize("kind", KINDS)
def test_reference_diagnostics_variant_of_every_kind(rt, oracle, kind, in_lds):
    """Round 6: the walk that counts the REFERENCE's tree (RTOW_CONTEXT_REFERENCE_DIAGNOSTICS, reference_counts with its 64-entry stack) is a variant of its own
    (DIAG 2); the 16-byte records without the option run on DIAG 1, which has no private segment.  Every kind, tree in LDS and beyond, at a register-history depth
    and at a depth whose history codes live in LDS rows: all four diagnostics columns and every output against the oracle, which walks that very tree."""
    scene = _scene(rt, kind)
    desc = scene.desc()
    w, h = 40, 24
    osc = oracle.OracleScene(desc)
    try:
        with rt.Context(0, flags=rt.abi.CONTEXT_REFERENCE_DIAGNOSTICS, lds_scene_budget=0 if in_lds else 1024) as ctx:
            ctx.upload_scene(desc)
            for depth in (5, 20):
                p = rt.scenes.make_params(scene, w, h, spp=3, trace_depth=depth, seed=78, diagnostics_stride=16)
                gpu = rt.sample_batch_host(ctx, p)
                ref = osc.sample_batch(p)
                for k in ("color", "normal", "albedo", "scw"):
                    assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (kind, in_lds, depth, k)
                for col, what in enumerate(("RayCount", "BoundsHitCount", "CandidateCount", "SampleCountWeight")):
                    assert np.array_equal(gpu["diag"][:, col].view(np.uint32), ref["diag"][:, col].view(np.uint32)), (kind, in_lds, depth, what)
    finally:
        osc.close()


def test_lens_and_pinhole_variants_beyond_lds(rt, oracle):
    """Static spheres whose tree is beyond LDS have a pinhole twin (GEO bit 3: lens code and the view's right / up not compiled in) - taken when the lens radius is 0 and
    no component of the view's origin or lower left corner is exactly zero.  The same scene with a lens, without one, and without one from a camera ON a coordinate plane
    (the general variant must serve it: its zero offset carries the reference's sign) against the oracle, at both register-history depths."""
    S = rt.scenes
    scene = S.cover_scene(60, 600)
    desc = scene.desc()
    osc = oracle.OracleScene(desc)
    try:
        with rt.Context(0, lds_scene_budget=1024) as ctx:
            ctx.upload_scene(desc)
            assert not ctx.scene_info().sceneInLds
            for camera in ({"aperture": 0.0}, {"aperture": 0.08}, {"aperture": 0.0, "position": [0.0, 2.0, -4.0]}, {"aperture": 0.0, "position": [9.0, 0.0, 3.0]}):
                scene.camera = dict(scene.camera, **camera)
                for depth in (6, 14):
                    p = S.make_params(scene, 64, 36, spp=4, trace_depth=depth, seed=5)
                    gpu = rt.sample_batch_host(ctx, p)
                    ref = osc.sample_batch(p)
                    for k in ("color", "normal", "albedo", "scw"):
                        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (camera, depth, k)
    finally:
        osc.close()
