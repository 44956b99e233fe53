"""Development helper (not a pytest file): chained launches against the same batches run one after the other, at full size - GPU against GPU,
every pixel of every accumulator and of every batch's diagnostics.  The suite does this at a few samples per pixel; this is the long version
(about a minute): hundreds of thousands of chunk hand-offs between CUs and XCDs per case, and slices with one pixel per lane, where almost
every hand-off has to wait.  Every case must print 0.

    python tests/soak_chain.py [scale]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
KEYS = (("color", 4), ("normal", 3), ("albedo", 3), ("scw", 1))


def run(ctx, plist, n, chained):
    bufs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for _, c in KEYS]
    stride = max(4, int(plist[0].diagnosticsStride))
    diags = [rt.DeviceBuffer(ctx, n * stride).zero() for _ in plist]
    if chained:
        rt.lib.check(rt.sample_batch_chain_device(ctx, plist, bufs, bufs, diags), "rtowSampleBatchChainDevice")
    else:
        for p, d in zip(plist, diags):
            job = rt.SampleBatchJob(ctx, p)
            job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
            job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = bufs
            job.OutputDiagnostics = d
            rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
    ctx.synchronize()
    out = [b.download(np.uint32, (n, c)) for b, (_, c) in zip(bufs, KEYS)] + [d.download(np.uint32, (n, stride // 4)) for d in diags]
    for b in bufs + diags:
        b.free()
    return out


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    S = rt.scenes
    cases = [("cover", S.cover_scene, 1920, 1080, 48, 8, 16, {}),
             ("cover, slice 3 of 8", S.cover_scene, 1920, 1080, 64, 8, 16, {"slice_offset": 3, "slice_divider": 8}),
             ("cover 4K, slice 0 of 8", S.cover_scene, 3840, 2160, 24, 16, 8, {"slice_offset": 0, "slice_divider": 8}),
             ("moving", S.moving_scene, 1920, 1080, 32, 8, 12, {}),
             ("stress 10000 (tree in HBM)", S.stress_scene, 1920, 1080, 24, 8, 12, {}),
             ("mixed", S.mixed_scene, 1920, 1080, 16, 8, 10, {}),
             ("volumes", S.volume_scene, 1280, 720, 12, 10, 8, {"focus": 6.5}),
             ("volume stack 48 (spilled hit lists)", lambda: S.volume_stack_scene(48, 0.125), 480, 480, 4, 10, 6, {}),
             ("mesh (exact-tie kernels)", S.mesh_scene, 1280, 720, 8, 8, 8, {}),
             ("decal stack (a tie on most rays)", lambda: S.decal_stack_scene(20), 640, 640, 6, 8, 8, {}),
             ("twin spheres moving", lambda: S.twin_spheres_scene(True), 1280, 720, 6, 8, 8, {}),
             ("cover, adaptive counts", S.cover_scene, 1280, 720, 4, 8, 16, {"spp_max": 40, "extrema": (0.2, 1.4)}),
             ("cover, tiny frame", S.cover_scene, 160, 90, 64, 8, 16, {}),
             # round 3: chains through the wide-code kernels (tree in HBM, 32-bit codes) and the mesh that needs them
             ("cover, wide codes", S.cover_scene, 1920, 1080, 16, 8, 8, {"_context": dict(flags=rt.abi.CONTEXT_FORCE_WIDE_CODES)}),
             ("mesh grid 250k", S.mesh_grid_scene, 1280, 720, 2, 8, 4, {"_focus_from_meta": True}),
             # round 6: chains through the variants whose path history lives in LDS rows, and through the tie watch of the all-triangle kinds
             ("cover depth 32", S.cover_scene, 1920, 1080, 16, 32, 10, {}),
             ("moving depth 24", S.moving_scene, 1280, 720, 16, 24, 10, {}),
             # ... and through the twins with the lanes in a hurry: 16-byte records with adaptive counts, tree beyond LDS
             ("cover depth 32, records, adaptive counts", S.cover_scene, 1280, 720, 8, 32, 8, {"spp_max": 50, "extrema": (0.2, 1.4), "diagnostics_stride": 16}),
             ("stress 10000 depth 24", S.stress_scene, 1280, 720, 12, 24, 6, {}),
             ("triangle layers (tie watch)", S.triangle_layers_scene, 960, 640, 4, 10, 8, {})]
    main_ctx = rt.Context(0)
    bad_total = 0
    for name, make, w, h, spp, depth, count, kw in cases:
        kw = dict(kw)
        spp = max(1, int(round(spp * scale)))
        scene = make()
        own = kw.pop("_context", None)
        if kw.pop("_focus_from_meta", False):
            kw["focus"] = scene.meta["focus"]
        ctx = rt.Context(0, **own) if own else main_ctx
        ctx.upload_scene(scene.desc())
        n = w * h
        plist = [S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=1000 + 13 * k, **kw) for k in range(count)]
        seq = run(ctx, plist, n, False)
        bad = 0
        for attempt in range(2):                 # the second chained run takes its chunk order from the first one's cost map
            got = run(ctx, plist, n, True)
            bad += sum(int(np.any(a.reshape(n, -1) != b.reshape(n, -1), axis=1).sum()) for a, b in zip(got, seq))
        bad_total += bad
        if own:
            ctx.close()
        print("%-30s %4dx%-4d %3d spp x %2d batches  differing pixel-buffers: %d" % (name, w, h, spp, count, bad), flush=True)
    print("total differing: %d" % bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
