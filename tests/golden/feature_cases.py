"""Small renders, one per feature of the path beyond the cover scene, whose oracle outputs are frozen as SHA-256 digests in golden.json
(`features`).  Shared by make_golden.py (writes) and tests/test_oracle_kat.py (checks)."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def feature_cases(rt):
    """name -> (scene, make_params kwargs, oracle set-up callback or None)"""
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
    }


def render_digests(rt, ob, name, case):
    scene, kw, setup = case
    osc = ob.OracleScene(scene.desc())
    if setup:
        setup(osc)
    p = rt.scenes.make_params(scene, **kw)
    r = osc.sample_batch(p)
    osc.close()
    d = {k: sha(r[k]) for k in ("color", "normal", "albedo", "scw")}
    d["raycount"] = sha(r["diag"][:, 0].copy())
    d["successful_samples"] = float(r["color"][:, 3].sum())
    return d
