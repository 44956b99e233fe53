"""GPU: the rarely combined switches, combined.  Every defect of round 2 sat in a corner no test had walked into (a chain of the exact-tie
kernel with 16-byte diagnostics, a 25-entry hit list, a tie on a ray through a mesh edge), so this file walks the corners systematically:
scenes that live in the slow paths (spilled hit lists, the exact-tie procedure, ties on most rays) x diagnostics record size x tree in
LDS / in HBM x RNG policy x one batch / a chain of three through the host-buffer entry points, each against the oracle's batches one after
the other, on frames small enough for the oracle to finish all 240 in about a minute."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = {
    "volume_stack_48": (lambda S: S.volume_stack_scene(48, 0.125), 10),       # ~100 hits per camera ray: hit lists spill to HBM
    "decal_stack": (lambda S: S.decal_stack_scene(20), 6),                    # exact-tie kernels by scene size, a tie on most rays, 23 hits
    "twin_row": (lambda S: S.twin_row_scene(30, True), 6),                    # exact-tie kernels by duplicates, moving, up to 60 hits per tied ray
    # ... and one scene per remaining kernel family, so that the same corners are walked in the fast paths too
    "cover_60": (lambda S: S.cover_scene(60, 600), 8),                        # spheres
    "tiny_moving": (lambda S: S.tiny_scene(), 8),                             # moving spheres
    "mixed": (lambda S: S.mixed_scene(), 6),                                  # general entities, rank rule (14 entities)
    "mesh": (lambda S: S.mesh_scene(1), 6),                                   # triangles, exact-tie kernels by scene size
    "textured": (lambda S: S.textured_scene(), 6),                            # image textures
    "volumes": (lambda S: S.volume_scene(), 10),                              # fog, short hit lists
    "textured_volumes": (lambda S: S.textured_volume_scene(), 8),
}


@pytest.mark.parametrize("chain", [1, 3])
@pytest.mark.parametrize("policy", ["reference", "per_sample", "xoroshiro"])
@pytest.mark.parametrize("tree", ["lds", "hbm"])
@pytest.mark.parametrize("stride", [4, 16])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_slow_path_scenes_under_every_switch(rt, oracle, name, stride, tree, policy, chain):
    S, a = rt.scenes, rt.abi
    make, depth = SCENES[name]
    scene = make(S)
    desc = scene.desc()
    w, h, spp = 48, 40, 3
    plist = [S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=31 + 5 * k, diagnostics_stride=stride,
                           rng_policy={"reference": a.RNG_REFERENCE, "per_sample": a.RNG_PER_SAMPLE, "xoroshiro": a.RNG_PER_SAMPLE_XOROSHIRO}[policy]) for k in range(chain)]
    osc = oracle.OracleScene(desc)
    ref, ref_diags = None, []
    for p in plist:
        ref = osc.sample_batch(p, None if ref is None else {k: ref[k] for k in ("color", "normal", "albedo", "scw")})
        ref_diags.append(ref["diag"])
    osc.close()
    with rt.Context(0, lds_scene_budget=64 if tree == "hbm" else 0) as ctx:
        ctx.upload_scene(desc)
        assert bool(ctx.scene_info().sceneInLds) == (tree == "lds")
        if chain == 1:
            gpu = rt.sample_batch_host(ctx, plist[0])
            gpu_diags = [gpu["diag"]]
        else:
            gpu = rt.sample_batch_chain_host(ctx, plist)
            gpu_diags = gpu["diag"]
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (k, int(np.any((gpu[k].view(np.uint32) != ref[k].view(np.uint32)).reshape(w * h, -1), axis=1).sum()))
    for b, (g, r) in enumerate(zip(gpu_diags, ref_diags)):
        assert np.array_equal(g[:, 0], r[:, 0]), ("ray counts of batch", b)
    assert ref["color"][:, 3].sum() > 0


NOISE_SCENES = {
    "cover_60": (lambda S: S.cover_scene(60, 600), 8),
    "tiny_moving": (lambda S: S.tiny_scene(), 8),
    "mixed": (lambda S: S.mixed_scene(), 6),
    "decal_stack": (lambda S: S.decal_stack_scene(20), 6),
    "volume_stack_48": (lambda S: S.volume_stack_scene(48, 0.125), 10),
    "textured_volumes": (lambda S: S.textured_volume_scene(), 8),
}


@pytest.mark.parametrize("chain", [1, 3])
@pytest.mark.parametrize("refdiag", [False, True])
@pytest.mark.parametrize("divider", [1, 3])
@pytest.mark.parametrize("sky", ["gradient", "cubemap"])
@pytest.mark.parametrize("noise", ["white", "blue", "stbn"])
@pytest.mark.parametrize("name", sorted(NOISE_SCENES))
def test_noise_sky_slices_and_reference_diagnostics(rt, oracle, name, noise, sky, divider, refdiag, chain):
    """Texture-driven noise (per-pixel walks), the cubemap sky, interlaced slices and the reference-identical FULL_DIAGNOSTICS counters
    (a second, unpruned walk of the reference's own tree) in every combination, single batches and chains."""
    S, a = rt.scenes, rt.abi
    make, depth = NOISE_SCENES[name]
    scene = make(S)
    desc = scene.desc()
    w, h, spp = 40, 36, 2
    textures = S.NoiseTextures(row_stride=8, count=3, seed=7)
    cube = S.synthetic_sky(size=8, half=True, seed=5)
    plist = [S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=77 + 3 * k, diagnostics_stride=16, slice_offset=divider - 1, slice_divider=divider,
                           noise_color={"white": a.NOISE_WHITE, "blue": a.NOISE_BLUE, "stbn": a.NOISE_SPATIOTEMPORAL_BLUE}[noise], noise_texture_index=k % 3,     # batches with different textures are not fused: the chain call runs them one after the other
                          
                           sky_type=a.SKY_CUBEMAP if sky == "cubemap" else a.SKY_GRADIENT) for k in range(chain)]
    osc = oracle.OracleScene(desc)
    osc.set_blue_noise(textures.blue_desc()); osc.set_stb_noise(textures.stb_desc()); osc.set_cubemap(cube.desc())
    rng = np.random.default_rng(3)
    n = w * h
    start = {"color": rng.random((n, 4)).astype(np.float32), "normal": rng.normal(size=(n, 3)).astype(np.float32),
             "albedo": rng.random((n, 3)).astype(np.float32), "scw": rng.random(n).astype(np.float32)}
    start["color"][:, 3] = rng.integers(0, 4, n)
    ref, ref_diags = start, []
    for p in plist:
        ref = osc.sample_batch(p, {k: ref[k] for k in ("color", "normal", "albedo", "scw")})
        ref_diags.append(ref["diag"])
    osc.close()
    with rt.Context(0, flags=a.CONTEXT_REFERENCE_DIAGNOSTICS if refdiag else 0) as ctx:
        ctx.upload_blue_noise(textures.blue_desc()); ctx.upload_stb_noise(textures.stb_desc()); ctx.upload_sky_cubemap(cube.desc())
        ctx.upload_scene(desc)
        if chain == 1:
            gpu = rt.sample_batch_host(ctx, plist[0], start)
            gpu_diags = [gpu["diag"]]
        else:
            gpu = rt.sample_batch_chain_host(ctx, plist, start)
            gpu_diags = gpu["diag"]
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (k, int(np.any((gpu[k].view(np.uint32) != ref[k].view(np.uint32)).reshape(n, -1), axis=1).sum()))
    for b, (g, r) in enumerate(zip(gpu_diags, ref_diags)):
        assert np.array_equal(g[:, 0], r[:, 0]), ("ray counts of batch", b)
        assert np.array_equal(g[:, 3].view(np.uint32), r[:, 3].view(np.uint32)), ("sample count weight of batch", b)
        if refdiag:
            assert np.array_equal(g[:, 1:3], r[:, 1:3]), ("reference tree counters of batch", b)
