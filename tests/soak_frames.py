"""Development helper (not a pytest file): whole frames at higher sample counts, GPU against the oracle, every pixel, compared on the GPU
box itself (only counts leave it).  The suite does the same at a few samples per pixel; this is the long version that was used to
find the 1-in-230-M-rays leaf-box issues and the tie order of rays with more than 16 hits (5.1; the twin-sphere case:
20 of its 921 600 pixels differed before scenes with duplicate primitives got the exact-tie kernels).  Every case must print 0.

    python tests/soak_frames.py [scale]        # scale multiplies the sample counts (default 1.0, about two minutes)
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
from oracle import binding as oracle  # noqa: E402


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    S, abi = rt.scenes, rt.abi
    noise = S.NoiseTextures(row_stride=64, count=2, seed=5)
    sky = S.synthetic_sky(size=64)
    cases = [
        ("cover", S.cover_scene, 1920, 1080, 64, 8, {}),
        ("stress 6000", lambda: S.stress_scene(count=6000, max_tentatives=40000), 1920, 1080, 24, 8, {}),
        ("moving", S.moving_scene, 1920, 1080, 32, 8, {}),
        ("mesh", S.mesh_scene, 1920, 1080, 12, 8, {}),
        ("mixed", S.mixed_scene, 1920, 1080, 16, 8, {}),
        ("volumes", S.volume_scene, 1280, 720, 24, 10, {"focus": 6.5}),
        ("volume stack", S.volume_stack_scene, 640, 640, 16, 10, {}),
        ("volume stack 48", lambda: S.volume_stack_scene(48, 0.125), 640, 640, 8, 10, {}),      # ~100 hits per camera ray: spilled hit lists
        ("textured", S.textured_scene, 1280, 720, 24, 8, {}),
        ("twin spheres", S.twin_spheres_scene, 1280, 720, 16, 8, {}),
        ("twin spheres moving", lambda: S.twin_spheres_scene(True), 1280, 720, 8, 8, {}),
        ("twin row", lambda: S.twin_row_scene(30), 640, 640, 8, 8, {}),                           # exact-tie procedure on lists of up to 60 hits
        ("twin row moving", lambda: S.twin_row_scene(30, True), 640, 640, 8, 8, {}),
        ("decal stack", lambda: S.decal_stack_scene(20), 640, 640, 8, 8, {}),                      # ties between different surfaces, > 16 hits per ray
        ("coplanar", S.coplanar_scene, 1280, 720, 12, 8, {"focus": 6.0}),
        ("cover per-sample", S.cover_scene, 1920, 1080, 48, 8, {"rng_policy": abi.RNG_PER_SAMPLE}),
        ("cover blue noise", S.cover_scene, 1920, 1080, 12, 8, {"noise_color": abi.NOISE_BLUE}),
        ("cover stbn", S.cover_scene, 1920, 1080, 12, 8, {"noise_color": abi.NOISE_SPATIOTEMPORAL_BLUE}),
        ("cover cubemap", S.cover_scene, 1920, 1080, 16, 8, {"sky_type": abi.SKY_CUBEMAP}),
        # round 3: 32-bit candidate codes (forced onto small scenes, and the 250 882-triangle mesh that needs them), slice geometries
        ("cover wide codes", S.cover_scene, 1920, 1080, 16, 8, {"_context": dict(flags=abi.CONTEXT_FORCE_WIDE_CODES)}),
        ("mixed wide codes", S.mixed_scene, 1280, 720, 8, 8, {"_context": dict(flags=abi.CONTEXT_FORCE_WIDE_CODES)}),
        ("moving wide codes", S.moving_scene, 1280, 720, 8, 8, {"_context": dict(flags=abi.CONTEXT_FORCE_WIDE_CODES)}),
        ("mesh grid 250k", S.mesh_grid_scene, 1280, 720, 3, 8, {"_focus_from_meta": True}),
        # round 6: path history in LDS rows (depth 32, both record formats; depth 48), the tie watch of the all-triangle kinds, the reference-diagnostics variant
        ("cover depth 32", S.cover_scene, 1920, 1080, 24, 32, {}),
        ("cover depth 32 records", S.cover_scene, 1920, 1080, 24, 32, {"diagnostics_stride": 16}),
        ("moving depth 48 records", S.moving_scene, 1280, 720, 16, 48, {"diagnostics_stride": 16}),
        ("stress 6000 depth 24", lambda: S.stress_scene(count=6000, max_tentatives=40000), 1280, 720, 12, 24, {}),
        ("triangle layers", S.triangle_layers_scene, 960, 640, 12, 10, {}),
        ("mesh depth 20 records", S.mesh_scene, 1280, 720, 8, 20, {"diagnostics_stride": 16}),
        ("cover reference counters", S.cover_scene, 1280, 720, 8, 20, {"diagnostics_stride": 16, "_context": dict(flags=abi.CONTEXT_REFERENCE_DIAGNOSTICS)}),
    ]
    main_ctx = rt.Context(0)
    main_ctx.upload_blue_noise(noise.blue_desc())
    main_ctx.upload_stb_noise(noise.stb_desc())
    main_ctx.upload_sky_cubemap(sky.desc())
    total_rays, bad_total = 0.0, 0
    for name, make, w, h, spp, depth, kw in cases:
        kw = dict(kw)
        spp = max(1, int(round(spp * scale)))
        scene = make()
        desc = scene.desc()
        own = kw.pop("_context", None)
        if kw.pop("_focus_from_meta", False):
            kw["focus"] = scene.meta["focus"]
        ctx = rt.Context(0, **own) if own else main_ctx
        ctx.upload_scene(desc)
        p = S.make_params(scene, w, h, spp=spp, trace_depth=depth, **kw)
        t0 = time.time()
        gpu = rt.sample_batch_host(ctx, p)
        t1 = time.time()
        if own:
            ctx.close()
        osc = oracle.OracleScene(desc)
        osc.set_blue_noise(noise.blue_desc())
        osc.set_stb_noise(noise.stb_desc())
        osc.set_cubemap(sky.desc())
        ref = osc.sample_batch(p)
        osc.close()
        t2 = time.time()
        bad = np.zeros(w * h, dtype=bool)
        for k in ("color", "normal", "albedo", "scw"):
            bad |= np.any(gpu[k].view(np.uint32).reshape(w * h, -1) != ref[k].view(np.uint32).reshape(w * h, -1), axis=1)
        bad |= gpu["diag"][:, 0] != ref["diag"][:, 0]
        rays = float(ref["diag"][:, 0].sum())
        total_rays += rays
        bad_total += int(bad.sum())
        print("%-18s %4dx%-4d %3d spp  %7.1f M rays  gpu %5.2f s  oracle %6.1f s  differing pixels: %d" % (name, w, h, spp, rays / 1e6, t1 - t0, t2 - t1, int(bad.sum())), flush=True)
    print("total %.2f G rays, %d differing pixels" % (total_rays / 1e9, bad_total))
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
