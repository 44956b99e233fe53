"""Small renders, one per feature of the path beyond the cover scene, whose oracle outputs are frozen as SHA-256 digests in golden.json
(`features`).  Shared by make_golden.py (writes) and tests/test_oracle_kat.py (checks)."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def feature_cases(rt):
    """name -> (scene, make_params kwargs, oracle set-up callback or None[, options]).  options (round 3):
         context_flags   RtowContextOptions.flags the GPU side needs (e.g. reference-identical FULL_DIAGNOSTICS)
         full_diag       digest all four columns of the 16-byte diagnostics record, not only RayCount
         chain_seeds     successive batches of one frame (Seed per batch): the digests are of the final accumulators and of every batch's RayCount
                         (the GPU side runs them as ONE chained launch, rtowSampleBatchChain)
         sparse          (count, seed): the frame is too large for the oracle in test time - digest only these randomly chosen pixels
         max_bvh_depth   the host's MaxBvhDepth (leaf order of the reference tree = tie order)"""
    S, abi = rt.scenes, rt.abi
    noise = S.NoiseTextures(row_stride=16, count=2)
    sky = S.synthetic_sky(size=32)

    def with_noise(osc):
        osc.set_blue_noise(noise.blue_desc())
        osc.set_stb_noise(noise.stb_desc())

    return {
        "mixed_primitives": (S.mixed_scene(), dict(width=48, height=32, spp=4, trace_depth=6), None),
        "probabilistic_volumes": (S.volume_scene(), dict(width=40, height=40, spp=4, trace_depth=10, focus=6.5), None),
        "volume_ties": (S.volume_tie_scene(), dict(width=40, height=40, spp=4, trace_depth=10, focus=5.0), None),
        "coplanar_ties": (S.coplanar_scene(), dict(width=48, height=32, spp=4, trace_depth=6, focus=6.0), None),
        "twin_sphere_ties": (S.twin_spheres_scene(), dict(width=48, height=27, spp=4, trace_depth=6), None),
        "twin_sphere_ties_moving": (S.twin_spheres_scene(True), dict(width=48, height=27, spp=4, trace_depth=6), None),
        "long_hit_lists_with_ties": (S.volume_stack_scene(), dict(width=32, height=32, spp=4, trace_depth=8), None),
        "image_textures": (S.textured_scene(), dict(width=48, height=32, spp=4, trace_depth=8), None),
        "textured_volumes": (S.textured_volume_scene(), dict(width=32, height=32, spp=4, trace_depth=10, focus=6.5), None),
        "cubemap_sky": (S.cover_scene(), dict(width=48, height=27, spp=4, trace_depth=8, sky_type=abi.SKY_CUBEMAP), lambda osc: osc.set_cubemap(sky.desc())),
        "blue_noise": (S.tiny_scene(), dict(width=32, height=18, spp=4, trace_depth=8, noise_color=abi.NOISE_BLUE, noise_texture_index=1), with_noise),
        "stb_noise": (S.tiny_scene(), dict(width=32, height=18, spp=4, trace_depth=8, noise_color=abi.NOISE_SPATIOTEMPORAL_BLUE), with_noise),
        "adaptive_samples": (S.tiny_scene(), dict(width=32, height=18, spp=2, spp_max=9, extrema=(0.0, 2.0), trace_depth=6), None),
        # ---- round 2 / 3 features ----
        "decal_stack_exact_ties_by_size": (S.decal_stack_scene(20), dict(width=64, height=64, spp=3, trace_depth=6, diagnostics_stride=16), None, dict(full_diag=False)),
        "fog_slabs_27_hits_spilled_list": (S.volume_stack_scene(13, 0.5), dict(width=32, height=32, spp=3, trace_depth=10), None),
        "fog_slabs_99_hits_spilled_list": (S.volume_stack_scene(48, 0.125), dict(width=24, height=24, spp=2, trace_depth=10), None),
        "twin_row_long_tie_lists": (S.twin_row_scene(30), dict(width=48, height=48, spp=3, trace_depth=8), None),
        "rng_per_sample_xorshift": (S.cover_scene(), dict(width=48, height=27, spp=40, trace_depth=8, rng_policy=abi.RNG_PER_SAMPLE), None),
        "rng_per_sample_xoroshiro": (S.cover_scene(), dict(width=48, height=27, spp=40, trace_depth=8, rng_policy=abi.RNG_PER_SAMPLE_XOROSHIRO), None),
        "reference_diagnostics_cover": (S.cover_scene(), dict(width=48, height=27, spp=3, trace_depth=8, diagnostics_stride=16), None,
                                        dict(context_flags=abi.CONTEXT_REFERENCE_DIAGNOSTICS, full_diag=True)),
        "reference_diagnostics_volumes_forced_leaves": (S.volume_scene(), dict(width=32, height=32, spp=3, trace_depth=8, focus=6.5, diagnostics_stride=16), None,
                                                        dict(context_flags=abi.CONTEXT_REFERENCE_DIAGNOSTICS, full_diag=True, max_bvh_depth=3)),
        "chain_of_three_batches": (S.cover_scene(), dict(width=64, height=36, spp=4, trace_depth=8), None, dict(chain_seeds=[21, 22, 23])),
        "chain_of_three_batches_moving_slice": (S.moving_scene(), dict(width=64, height=36, spp=3, trace_depth=8, slice_offset=1, slice_divider=2), None, dict(chain_seeds=[5, 6, 7])),
        "mesh_grid_250k_triangles_wide_codes": (S.mesh_grid_scene(), dict(width=1280, height=720, spp=4, trace_depth=8, focus=None), None, dict(sparse=(400, 3), focus_from_meta=True)),
        # ---- round 3, late: volume kinds with 32-bit codes; frame shapes whose tickets are tiles + a row-major remainder, sliced and chained ----
        "mesh_grid_82k_with_fog_volumes_wide_codes": (S.mesh_grid_fog_scene(), dict(width=1280, height=720, spp=4, trace_depth=8, focus=None), None, dict(sparse=(300, 5), focus_from_meta=True)),
        "chain_of_three_batches_tiles_and_remainder": (S.cover_scene(), dict(width=72, height=29, spp=3, trace_depth=8, slice_offset=1, slice_divider=2), None, dict(chain_seeds=[31, 32, 33])),
    }


def unpack(case):
    scene, kw, setup = case[:3]
    opts = dict(case[3]) if len(case) > 3 else {}
    kw = dict(kw)
    if opts.get("focus_from_meta"):
        kw["focus"] = scene.meta["focus"]
    return scene, kw, setup, opts


def sparse_indices(kw, opts):
    count, seed = opts["sparse"]
    n = kw["width"] * kw["height"]
    return np.unique(np.random.default_rng(seed).integers(0, n, count)).astype(np.int32)


def digests_of(r, opts, batch_raycounts=None):
    """r: dict of arrays (whole frame, or the sparse pixels in index order)."""
    d = {k: sha(r[k]) for k in ("color", "normal", "albedo", "scw")}
    d["raycount"] = sha(r["diag"][:, 0].copy())
    if opts.get("full_diag"):
        d["diagnostics"] = sha(r["diag"])
    if batch_raycounts is not None:
        d["batch_raycounts"] = [sha(x) for x in batch_raycounts]
    d["successful_samples"] = float(r["color"][:, 3].sum())
    return d


def render_digests(rt, ob, name, case):
    scene, kw, setup, opts = unpack(case)
    osc = ob.OracleScene(scene.desc(max_bvh_depth=opts["max_bvh_depth"]) if "max_bvh_depth" in opts else scene.desc())
    if setup:
        setup(osc)
    p = rt.scenes.make_params(scene, **kw)
    batch_raycounts = None
    if "chain_seeds" in opts:
        r, batch_raycounts = None, []
        for seed in opts["chain_seeds"]:
            p.seed = seed
            r = osc.sample_batch(p, None if r is None else {k: r[k] for k in ("color", "normal", "albedo", "scw")})
            batch_raycounts.append(r["diag"][:, 0].copy())
    elif "sparse" in opts:
        r = osc.sample_pixels(p, sparse_indices(kw, opts))
    else:
        r = osc.sample_batch(p)
    osc.close()
    return digests_of(r, opts, batch_raycounts)
