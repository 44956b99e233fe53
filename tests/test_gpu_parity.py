"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): output RGB within 1e-4 per channel on the same seed.  Because both sides run the
same float program, everything is in fact required to be BIT-EXACT here (np.array_equal on the raw float32 buffers);
the 1e-4 tolerance is asserted as well so the stated bar is visible in the test.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-4  # per channel, on the per-pixel mean colour (north_star tolerance)


def _mean(color4):
    c = np.maximum(color4[:, 3:4], 1.0)
    return color4[:, :3] / c


def _compare(gpu, ref, check_diag=True):
    assert np.array_equal(gpu["color"][:, 3], ref["color"][:, 3]), "successful-sample counts differ"
    d = np.abs(_mean(gpu["color"]) - _mean(ref["color"]))
    assert np.nanmax(d) <= TOL, "mean colour differs by %g" % np.nanmax(d)
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), "%s not bit-exact" % k
    if check_diag:
        assert np.array_equal(gpu["diag"][:, 0], ref["diag"][:, 0]), "RayCount differs"


def _run_both(rt, oracle, ctx, scene, w, h, spp, depth, inputs=None, max_bvh_depth=32, **kw):
    desc = scene.desc(max_bvh_depth=max_bvh_depth)
    p = rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, **kw)
    ctx.upload_scene(desc)
    gpu = rt.sample_batch_host(ctx, p, inputs)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p, inputs)
    osc.close()
    return gpu, ref


def test_cover_64x36(rt, oracle, gpu_context):
    scene = rt.scenes.cover_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 64, 36, 8, 8)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0.9 * 64 * 36 * 8


def test_cover_config1_400x225(rt, oracle, gpu_context):
    """BASELINE.json configs[0]: cover scene 400x225, 8 spp, 8 bounces."""
    scene = rt.scenes.cover_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 400, 225, 8, 8)
    _compare(gpu, ref)


def test_tiny_scene_aperture_motion_emission(rt, oracle, gpu_context):
    scene = rt.scenes.tiny_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 96, 54, 16, 12, diagnostics_stride=16)
    _compare(gpu, ref)


def test_moving_scene_defocus(rt, oracle, gpu_context):
    """BASELINE.json configs[4] at a reduced size: moving spheres + aperture 0.05."""
    scene = rt.scenes.moving_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 160, 90, 8, 8)
    _compare(gpu, ref)


def test_no_jitter_depth1_first_hit_aovs(rt, oracle, gpu_context):
    """traceDepth 1: every hit sample fails, so normal/albedo are the sample-0 fallbacks = first-hit AOVs."""
    scene = rt.scenes.cover_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 128, 72, 1, 1, jitter=False)
    _compare(gpu, ref)


def test_second_batch_accumulates(rt, oracle, gpu_context):
    scene = rt.scenes.cover_scene()
    gpu1, ref1 = _run_both(rt, oracle, gpu_context, scene, 64, 36, 4, 8, seed=1)
    _compare(gpu1, ref1)
    ins = {k: ref1[k] for k in ("color", "normal", "albedo", "scw")}
    gpu2, ref2 = _run_both(rt, oracle, gpu_context, scene, 64, 36, 4, 8, inputs=ins, seed=2)
    _compare(gpu2, ref2)
    assert ref2["color"][:, 3].max() > ref1["color"][:, 3].max()


def test_slices_partition_the_frame(rt, oracle, gpu_context):
    """Row-interleaved slices (SliceOffset/SliceDivider, JOBS/SampleBatchJob.cs:69-70) are bit-identical to the full frame."""
    scene = rt.scenes.cover_scene()
    w, h = 64, 36
    full, ref = _run_both(rt, oracle, gpu_context, scene, w, h, 4, 8)
    _compare(full, ref)
    parts = {k: np.full_like(full[k], -7.0) for k in ("color", "normal", "albedo", "scw")}
    for g in range(3):
        p = rt.scenes.make_params(scene, w, h, spp=4, trace_depth=8, slice_offset=g, slice_divider=3)
        marker = {k: np.full_like(full[k], -7.0) for k in ("color", "normal", "albedo", "scw")}
        zero = {k: np.zeros_like(full[k]) for k in ("color", "normal", "albedo", "scw")}
        job = rt.SampleBatchJob(gpu_context, p)
        job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = zero["color"], zero["normal"], zero["albedo"], zero["scw"]
        job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = marker["color"], marker["normal"], marker["albedo"], marker["scw"]
        assert job.Schedule(w * h, 1).Complete() == 0
        rows = np.arange(h) % 3 == g
        mask = np.repeat(rows, w)
        for k in parts:
            assert np.all(marker[k][~mask] == -7.0), "slice wrote a pixel it does not own"
            parts[k][mask] = marker[k][mask]
    for k in parts:
        assert np.array_equal(parts[k], full[k])


@pytest.mark.parametrize("w,h", [(64, 24), (64, 27), (72, 13), (8, 64), (70, 16), (136, 5)])
def test_ticket_numbering_in_tiles_reaches_every_owned_pixel(rt, oracle, gpu_context, w, h):
    """The 64 tickets of a chunk are an 8 x 8 tile of the owned pixels where the width is a multiple of 8 (csrc/rtow_kernels.h: owned_pixel_xy;
    rows behind the last whole tile row, and frames of any other width, are numbered row by row).  Whatever the frame's shape - whole tiles,
    a partial tile row, fewer than 8 rows, a width that is no multiple of 8 - every owned pixel is rendered exactly once: whole frames and
    3-way row slices (the owned rows of a slice form the tiles) against the oracle, under the reference stream, the per-sample policy (units
    are numbered pixel by pixel through the same function, and the fold kernel maps them back) and as a chain of two batches."""
    scene = rt.scenes.cover_scene(60, 600)
    desc = scene.desc()
    ctx = gpu_context
    ctx.upload_scene(desc)
    osc = oracle.OracleScene(desc)
    try:
        for policy in (rt.abi.RNG_REFERENCE, rt.abi.RNG_PER_SAMPLE):
            for div in (1, 3):
                got = {k: np.full((w * h, c), -7.0, np.float32) for k, c in (("color", 4), ("normal", 3), ("albedo", 3))}
                got["scw"] = np.full(w * h, -7.0, np.float32)
                want = {k: v.copy() for k, v in got.items()}
                for g in range(div):
                    p = rt.scenes.make_params(scene, w, h, spp=3, trace_depth=5, seed=5, slice_offset=g, slice_divider=div, rng_policy=policy)
                    mask = np.repeat(np.arange(h) % div == g, w)
                    gpu, ref = rt.sample_batch_host(ctx, p), osc.sample_batch(p)
                    for k in got:
                        got[k][mask], want[k][mask] = gpu[k][mask], ref[k][mask]
                    assert np.array_equal(gpu["diag"][mask, 0], ref["diag"][mask, 0])
                for k in got:
                    assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (k, policy, div)
                    assert not np.any(got[k] == -7.0)
        # two chained batches in one launch (the per-chunk hand-off counts the pixels of a chunk: tiles and the row-major remainder alike)
        plist = [rt.scenes.make_params(scene, w, h, spp=3, trace_depth=5, seed=s) for s in (8, 9)]
        n = w * h
        bufs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
        rt.lib.check(rt.sample_batch_chain_device(ctx, plist, bufs, bufs), "rtowSampleBatchChainDevice")
        ctx.synchronize()
        ref = osc.sample_batch(plist[1], {k: v for k, v in osc.sample_batch(plist[0]).items() if k != "diag"})
        for k, b, c in zip(("color", "normal", "albedo", "scw"), bufs, (4, 3, 3, 1)):
            assert np.array_equal(b.download(np.float32, (n, c)).reshape(ref[k].shape).view(np.uint32), ref[k].view(np.uint32)), ("chain", k)
    finally:
        osc.close()


def test_mixed_primitives_rotated_and_moving(rt, oracle, gpu_context):
    """Rect / Box / Triangle / rotated + moving entities through the general Entity transform (RT/Entity.cs:58-127)."""
    scene = rt.scenes.mixed_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 120, 120, 8, 8, focus=6.5, diagnostics_stride=16)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0


def test_triangle_mesh_scene(rt, oracle, gpu_context):
    """The reference's live host only ingests triangle meshes (UNITY/Raytracer.cs:1185-1304): a tessellated sphere over a quad."""
    import numpy as np
    scene = rt.scenes.Scene("mesh")
    grey = scene.materials.append(rt.scenes.lambertian((0.6, 0.6, 0.6))) or 0
    gold = scene.materials.append(rt.scenes.metal((0.9, 0.7, 0.3), 0.15)) or 1
    scene.add_triangle((-3, 0, -3), (3, 0, 3), (3, 0, -3), grey)
    scene.add_triangle((-3, 0, -3), (-3, 0, 3), (3, 0, 3), grey)
    nu, nv, r, c = 12, 8, 0.8, np.array([0.0, 0.9, 0.0])

    def pt(i, j):
        th, ph = 2 * np.pi * i / nu, np.pi * j / nv
        return c + r * np.array([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)])

    for i in range(nu):
        for j in range(nv):
            a, b, cc, d = pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)
            if j > 0:
                scene.add_triangle(a, b, cc, gold, normals=(a - c, b - c, cc - c))      # vertex normals: smooth shading
            if j < nv - 1:
                scene.add_triangle(a, cc, d, gold, normals=(a - c, cc - c, d - c))
    scene.camera = {"position": [2.5, 1.8, 3.0], "target": [0.0, 0.7, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 35.0, "aperture": 0.0}
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 96, 72, 6, 8, focus=4.0)
    _compare(gpu, ref)


def test_probabilistic_volumes(rt, oracle, gpu_context):
    """ProbabilisticVolume materials: all-hits collection, exit-hit injection, containment probe (camera inside a haze sphere),
    the volume branch of Sample and isotropic scatter (JOBS/SampleBatchJob.cs:194-303,463-524, RT/Material.cs:49-65,163-168)."""
    scene = rt.scenes.volume_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 100, 100, 8, 10, focus=6.5, diagnostics_stride=16)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0


def test_volume_hits_at_identical_distances(rt, oracle, gpu_context):
    """Coplanar hulls / floor / lid: hits at bit-identical distances take the order the reference's unstable hit sort leaves them
    in, starting from its tree's leaf order (csrc/rtow_reforder.h, tests/test_reference_tie_order.py)."""
    scene = rt.scenes.volume_tie_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 96, 96, 8, 12, focus=5.0, diagnostics_stride=16)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0


def test_volume_hit_lists_longer_than_sixteen_with_ties(rt, oracle, gpu_context):
    """21 hits per camera ray, ten pairs of them at identical distances: above 16 elements the reference's sort partitions
    (median of three, Hoare) before it insertion-sorts, and the order that leaves the ties in decides which hull is 'entered' first."""
    scene = rt.scenes.volume_stack_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 96, 96, 8, 12, diagnostics_stride=16)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0


@pytest.mark.parametrize("slabs,thickness", [(13, 0.5), (48, 0.125)])
def test_volume_hit_lists_longer_than_a_lane_holds(rt, oracle, gpu_context, slabs, thickness):
    """27 and 99 hits per camera ray: the reference's hitRecordBuffer grows on the heap (UTIL/HybridCollections.cs:65-71); the kernel keeps
    24 hits per lane and the rest of the list in its spill column in HBM, sorted by the same introsort (with its depth limit and heap sort)."""
    scene = rt.scenes.volume_stack_scene(slabs, thickness)
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 64, 64, 4, 12, diagnostics_stride=16)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0


def test_ties_between_different_surfaces_on_rays_with_more_than_sixteen_hits(rt, oracle, gpu_context):
    """Decals in a wall's plane in front of 20 more panes: no duplicate primitive, but more than 16 entities of general geometry - the scene
    compiler picks the exact-tie kernels by itself, and the frame equals the reference's where the leaf-order rule alone would not have to."""
    scene = rt.scenes.decal_stack_scene(20)
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 128, 128, 4, 8, diagnostics_stride=16)
    _compare(gpu, ref)
    # ... and the rule alone does not: the same frame from a context that was told never to use the exact-tie kernels
    with rt.Context(0, flags=rt.abi.CONTEXT_EXACT_TIES_NEVER) as ctx:
        ctx.upload_scene(scene.desc(max_bvh_depth=32))
        rule = rt.sample_batch_host(ctx, rt.scenes.make_params(scene, 128, 128, spp=4, trace_depth=8, diagnostics_stride=16))
    assert np.any(rule["color"].view(np.uint32) != ref["color"].view(np.uint32), axis=1).sum() > 100


@pytest.mark.parametrize("moving", [False, True])
def test_exact_tie_procedure_with_long_hit_lists(rt, oracle, gpu_context, moving):
    """A row of 30 coinciding sphere pairs seen end-on: every nearest hit is a tie and rays near the axis have up to 60 hits, so the
    reference's sort of the whole list (partitions above 16 elements) decides which twin is shaded - and the list does not fit a lane."""
    scene = rt.scenes.twin_row_scene(30, moving)
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 96, 96, 4, 8, diagnostics_stride=16)
    _compare(gpu, ref)


def test_nearest_hit_ties_between_coplanar_entities(rt, oracle, gpu_context):
    """Decals in a wall's plane, boxes sharing a face, one sphere twice: the nearest hit is shared by two or three entities and the
    one that comes first in the reference tree's leaf order wins (JOBS/SampleBatchJob.cs:450-475, csrc/rtow_reforder.h)."""
    scene = rt.scenes.coplanar_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 96, 64, 8, 8, focus=6.0, diagnostics_stride=16)
    _compare(gpu, ref)


@pytest.mark.parametrize("moving,max_depth", [(False, 32), (True, 32), (False, 3)])
def test_nearest_hit_ties_between_coinciding_spheres(rt, oracle, gpu_context, moving, max_depth):
    """Sphere-only scenes (the SPHERES / SPHERES_MOTION kernels): twins and triplets of the same sphere with different materials - the
    reference's sorted hit list starts with the one first in its tree's leaf order (JOBS/SampleBatchJob.cs:450-475)."""
    scene = rt.scenes.twin_spheres_scene(moving)
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 128, 72, 8, 8, diagnostics_stride=16, max_bvh_depth=max_depth)
    _compare(gpu, ref)


def test_sky_cubemap(rt, oracle, gpu_context):
    """SkyType.CubeMap: Cubemap.Sample(ray.Direction) (RT/Texture.cs:171-210, JOBS/SampleBatchJob.cs:356-358) for both channel decodes,
    an odd face size, and after dropping the cubemap again (black sky, like the reference's null data pointer)."""
    ctx = gpu_context
    scene = rt.scenes.cover_scene()
    desc = scene.desc()
    ctx.upload_scene(desc)
    osc = oracle.OracleScene(desc)
    p = rt.scenes.make_params(scene, 96, 54, spp=8, trace_depth=8, sky_type=rt.abi.SKY_CUBEMAP)
    try:
        for sky in (rt.scenes.synthetic_sky(size=64, half=True), rt.scenes.synthetic_sky(size=37, half=False), rt.scenes.synthetic_sky(size=1, half=True)):
            cd = sky.desc()
            ctx.upload_sky_cubemap(cd)
            osc.set_cubemap(cd)
            gpu = rt.sample_batch_host(ctx, p)
            ref = osc.sample_batch(p)
            _compare(gpu, ref)
            assert gpu["color"][:, :3].max() > 0
        ctx.upload_sky_cubemap(None)
        osc.set_cubemap(None)
        gpu = rt.sample_batch_host(ctx, p)
        _compare(gpu, osc.sample_batch(p))
        assert np.all(gpu["color"][:, :3] == 0)
    finally:
        ctx.upload_sky_cubemap(None)
        osc.close()


@pytest.mark.parametrize("noise_color", ["blue", "stbn"])
def test_texture_driven_noise_sources(rt, oracle, gpu_context, noise_color):
    """NoiseColor.Blue / SpatioTemporalBlue (RT/RandomSource.cs, RT/BlueNoise.cs, RT/SpatioTemporalBlueNoise.cs, RT/PerPixelNoise.cs, RT/R2.cs):
    every branch of the source - jitter, lens disk, time, cosine hemisphere (and the lambert path's unused one), sphere direction, scalar
    draws incl. ProbabilisticHit - on sphere, moving + defocus, general and volume scenes, both diagnostics layouts, several textures."""
    ctx = gpu_context
    color = rt.abi.NOISE_BLUE if noise_color == "blue" else rt.abi.NOISE_SPATIOTEMPORAL_BLUE
    noise = rt.scenes.NoiseTextures(row_stride=16, count=3)
    ctx.upload_blue_noise(noise.blue_desc())
    ctx.upload_stb_noise(noise.stb_desc())
    cases = [(rt.scenes.cover_scene(), dict(w=64, h=36, spp=8, depth=8, noise_texture_index=0)),
             (rt.scenes.moving_scene(), dict(w=48, h=27, spp=8, depth=6, noise_texture_index=2, seed=77)),
             (rt.scenes.mixed_scene(), dict(w=48, h=32, spp=6, depth=5, noise_texture_index=1, diagnostics_stride=16)),
             (rt.scenes.volume_scene(), dict(w=40, h=40, spp=6, depth=12, noise_texture_index=1, focus=6.5)),
             (rt.scenes.tiny_scene(), dict(w=32, h=18, spp=5, depth=20, noise_texture_index=0, jitter=False))]
    try:
        for scene, kw in cases:
            desc = scene.desc()
            kw = dict(kw)
            p = rt.scenes.make_params(scene, kw.pop("w"), kw.pop("h"), spp=kw.pop("spp"), trace_depth=kw.pop("depth"), noise_color=color, **kw)
            ctx.upload_scene(desc)
            gpu = rt.sample_batch_host(ctx, p)
            osc = oracle.OracleScene(desc)
            osc.set_blue_noise(noise.blue_desc())
            osc.set_stb_noise(noise.stb_desc())
            ref = osc.sample_batch(p)
            osc.close()
            _compare(gpu, ref)
            assert gpu["color"][:, 3].sum() > 0
    finally:
        ctx.upload_blue_noise(None)
        ctx.upload_stb_noise(None)


def test_image_textures(rt, oracle, gpu_context):
    """TextureType.Image on albedo / emission / glossiness / metallic, Standard and Dielectric, triangle texture coordinates, the (0, 0)
    coordinates of every other entity type, a null image pointer (RT/Texture.cs:80-89,126-135, RT/Material.cs:71-78,123,176-179) - per-hit
    material evaluation and the per-hit colours the fold needs."""
    scene = rt.scenes.textured_scene()
    for w, h, spp, depth, stride in ((96, 64, 8, 8, 4), (64, 40, 6, 12, 16), (40, 28, 4, 20, 4)):
        gpu, ref = _run_both(rt, oracle, gpu_context, scene, w, h, spp, depth, diagnostics_stride=stride)
        _compare(gpu, ref)
        assert gpu["color"][:, 3].sum() > 0
    assert gpu_context.scene_info().bvhNodeCount == scene.entity_count - 1


def test_image_textures_with_probabilistic_volumes(rt, oracle, gpu_context):
    scene = rt.scenes.textured_volume_scene()
    gpu, ref = _run_both(rt, oracle, gpu_context, scene, 64, 64, 6, 10, focus=6.5, diagnostics_stride=16)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0


@pytest.mark.parametrize("policy", ["xorshift", "xoroshiro"])
def test_per_sample_rng_policy(rt, oracle, gpu_context, policy):
    """RTOW_RNG_PER_SAMPLE / RTOW_RNG_PER_SAMPLE_XOROSHIRO (include/rtow.h): NOT the reference's stream - every sample has its own generator, work units are (pixel, group
    of 16 samples), a fold kernel adds the groups in order.  Defined by the oracle's restatement of that definition; bit-exact against it
    for sample counts that are / are not multiples of 16, adaptive counts, slices, all history widths, both diagnostics layouts, and on
    top of existing accumulators."""
    ctx = gpu_context
    PS = rt.abi.RNG_PER_SAMPLE if policy == "xorshift" else rt.abi.RNG_PER_SAMPLE_XOROSHIRO      # north_star's per-lane xoroshiro generator
    cases = [(rt.scenes.cover_scene(), dict(width=64, height=36, spp=40, trace_depth=8)),
             (rt.scenes.cover_scene(), dict(width=48, height=27, spp=16, trace_depth=12, diagnostics_stride=16)),
             (rt.scenes.cover_scene(), dict(width=48, height=27, spp=5, trace_depth=20, slice_offset=1, slice_divider=2)),
             (rt.scenes.moving_scene(), dict(width=48, height=27, spp=33, trace_depth=6)),
             (rt.scenes.mixed_scene(), dict(width=48, height=32, spp=20, trace_depth=5)),
             (rt.scenes.volume_scene(), dict(width=32, height=32, spp=18, trace_depth=10, focus=6.5)),
             (rt.scenes.textured_scene(), dict(width=40, height=28, spp=17, trace_depth=6)),
             (rt.scenes.tiny_scene(), dict(width=32, height=18, spp=3, spp_max=50, extrema=(0.0, 2.0), trace_depth=6, diagnostics_stride=16))]
    for scene, kw in cases:
        desc = scene.desc()
        ctx.upload_scene(desc)
        osc = oracle.OracleScene(desc)
        p = rt.scenes.make_params(scene, rng_policy=PS, **kw)
        gpu = rt.sample_batch_host(ctx, p)
        ref = osc.sample_batch(p)
        _compare(gpu, ref)
        # a second batch on top of the first (non-zero inputs, adaptive counts now driven by the accumulated weight)
        p.seed = 2
        ins = {k: ref[k] for k in ("color", "normal", "albedo", "scw")}
        gpu2 = rt.sample_batch_host(ctx, p, ins)
        ref2 = osc.sample_batch(p, ins)
        _compare(gpu2, ref2)
        # and it IS a different stream: the reference policy gives another image
        q = rt.scenes.make_params(scene, **kw)
        assert not np.array_equal(osc.sample_batch(q)["color"], ref["color"])
        osc.close()


def test_hit_lists_grow_like_the_references(rt, oracle):
    """The reference's hit list grows on the heap without bound (UTIL/HybridCollections.cs:22-36,65-71).  With RtowContextOptions.hitListCapacity left at 0 the library's
    lists start at 1024 entries in a scene with volumes; 560 stacked hulls put 1121 surfaces on a camera ray.  The host-buffer call notices, doubles the lists (up to the
    2 hits per entity a scene can produce at all) and runs the batch again by itself: the caller sees the reference's frame, not RTOW_ERROR_CAPACITY."""
    scene = rt.scenes.volume_stack_scene(560, 1.0 / 128.0)
    with rt.Context(0) as ctx:
        desc = scene.desc()
        ctx.upload_scene(desc)
        assert ctx.scene_info().hitListCapacity == 1024
        p = rt.scenes.make_params(scene, 32, 32, spp=1, trace_depth=4, diagnostics_stride=16)
        gpu = rt.sample_batch_host(ctx, p)
        assert ctx.scene_info().hitListCapacity == 2 * 563                  # 2 rects + 560 boxes + 1 sphere, entry and exit each: nothing in this scene can need more
        osc = oracle.OracleScene(desc)
        ref, counters = osc.sample_batch(p, None, want_counters=True)
        assert counters.maxHits > 1024
        _compare(gpu, ref)
        # the chain's host form likewise (a fresh context: the capacity a context has grown to stays)
    with rt.Context(0) as ctx:
        ctx.upload_scene(desc)
        p2 = rt.scenes.make_params(scene, 32, 32, spp=1, trace_depth=4, diagnostics_stride=16)
        p2.seed = p.seed + 1
        gpu = rt.sample_batch_chain_host(ctx, [p, p2], None)
        ref2 = osc.sample_batch(p2, ref)
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(gpu[k].reshape(ref2[k].shape).view(np.uint32), ref2[k].view(np.uint32)), k
        # device-resident callers are told once, and the batch - issued again from inputs it did not overwrite - then has room
    with rt.Context(0) as ctx:
        a = rt.abi
        ctx.upload_scene(desc)
        n = 32 * 32
        ins = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
        outs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
        dg = rt.DeviceBuffer(ctx, n * 16).zero()
        assert rt.sample_batch_chain_device(ctx, [p], ins, outs, [dg]) == a.RTOW_SUCCESS
        with pytest.raises(rt.lib.RtowError) as e:
            ctx.synchronize()
        assert e.value.code == a.RTOW_ERROR_CAPACITY
        assert rt.sample_batch_chain_device(ctx, [p], ins, outs, [dg]) == a.RTOW_SUCCESS
        ctx.synchronize()
        for (k, c), b in zip((("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1)), outs):
            got = b.download(np.float32, (n, c) if c > 1 else (n,))
            assert np.array_equal(got.reshape(ref[k].shape).view(np.uint32), ref[k].view(np.uint32)), k
        for b in ins + outs + [dg]:
            b.free()
    osc.close()


def test_a_fixed_hit_list_capacity_stays_fixed(rt):
    """A caller that sets RtowContextOptions.hitListCapacity has chosen the memory it spends: the lists do not grow, the batch reports RTOW_ERROR_CAPACITY as before."""
    scene = rt.scenes.volume_stack_scene(48, 0.125)                            # 99 hits per camera ray
    with rt.Context(0, hit_list_capacity=64) as ctx:
        ctx.upload_scene(scene.desc())
        p = rt.scenes.make_params(scene, 32, 32, spp=1, trace_depth=4)
        for _ in range(2):
            with pytest.raises(rt.lib.RtowError) as e:
                rt.sample_batch_host(ctx, p)
            assert e.value.code == rt.abi.RTOW_ERROR_CAPACITY
            assert ctx.scene_info().hitListCapacity == 64


@pytest.mark.parametrize("aperture", [0.0, 0.2])
@pytest.mark.parametrize("w,h", [(6, 6), (12, 8), (48, 32)])
def test_camera_ray_lists_of_many_nodes_and_several_rounds(rt, oracle, gpu_context, w, h, aperture):
    """Few, wide pixels over a dense grid of small spheres: a pixel's beam meets up to eight leaf parents (the list's capacity for the sphere kinds) or more (no list: the
    camera ray walks), and a long list fills the eight candidate slots more than once (continuation rounds in the exact-test stage).  With a lens the beams widen further."""
    S = rt.scenes
    s = S.Scene("dense grid under wide pixels")
    rng = np.random.default_rng(3)
    for ix in range(7):
        for iy in range(5):
            for iz in range(2):
                s.add_sphere((-0.9 + 0.3 * ix + 0.02 * rng.random(), -0.6 + 0.3 * iy + 0.02 * rng.random(), -0.4 * iz), 0.11 + 0.03 * rng.random(),
                             S.lambertian((0.2 + 0.1 * ix, 0.3 + 0.1 * iy, 0.5)) if (ix + iy + iz) % 3 else S.metal((0.8, 0.8, 0.6), 0.1))
    s.add_sphere((0.0, -100.8, 0.0), 100.0, S.lambertian((0.5, 0.5, 0.5)))
    s.camera = {"position": [0.1, 0.05, 3.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": aperture}
    gpu, ref = _run_both(rt, oracle, gpu_context, s, w, h, 16, 6, diagnostics_stride=16, focus=3.0)
    _compare(gpu, ref)
    assert gpu["color"][:, 3].sum() > 0
