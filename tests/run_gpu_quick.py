"""Quick GPU sanity + timing driver used during development (not a pytest file)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")


def main():
    w, h, spp, depth = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1920, 1080, 32, 8))]
    name = sys.argv[5] if len(sys.argv) > 5 else "cover"
    scene = {"cover": rt.scenes.cover_scene, "moving": rt.scenes.moving_scene, "stress": rt.scenes.stress_scene, "mixed": rt.scenes.mixed_scene, "mesh": rt.scenes.mesh_scene, "volumes": rt.scenes.volume_scene, "textured": rt.scenes.textured_scene}[name]()
    tune = [int(x) for x in sys.argv[6].split(",")] if len(sys.argv) > 6 else None      # RtowContextOptions.schedulerTune, 9 integers
    ctx = rt.Context(0, log=lambda lvl, tag, msg, ud: print("[rtow]", tag.decode(), msg.decode()), log_level=4, scheduler_tune=tune)
    ctx.upload_scene(scene.desc())
    info = ctx.scene_info()
    print("scene: nodes", info.bvhNodeCount, "depth", info.bvhDepth, "ldsBytes", info.ldsBytesScene, "inLds", info.sceneInLds)
    p = rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, diagnostics_stride=16)
    n = w * h
    bufs = [rt.DeviceBuffer(ctx, n * 16).zero(), rt.DeviceBuffer(ctx, n * 12).zero(), rt.DeviceBuffer(ctx, n * 12).zero(), rt.DeviceBuffer(ctx, n * 4).zero()]
    outs = [rt.DeviceBuffer(ctx, n * 16), rt.DeviceBuffer(ctx, n * 12), rt.DeviceBuffer(ctx, n * 12), rt.DeviceBuffer(ctx, n * 4)]
    diag = rt.DeviceBuffer(ctx, n * 16)
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
    job.OutputDiagnostics = diag
    for it in range(3):
        t = time.time()
        rc = job.Schedule().Complete()
        ctx.synchronize()
        dt = time.time() - t
        ms = ctx.last_sample_kernel_ms()
        print("iter", it, "rc", rc, "wall %.3f s kernel %.2f ms -> %.1f Msamples/s" % (dt, ms, n * spp / ms / 1e3))
    d = diag.download(np.float32, (n, 4))
    c = outs[0].download(np.float32, (n, 4))
    print("rays/sample %.3f nodes(boundsHit)/ray %.2f candidates/ray %.2f success %.4f" % (
        d[:, 0].sum() / (n * spp), d[:, 1].sum() / d[:, 0].sum(), d[:, 2].sum() / d[:, 0].sum(), c[:, 3].sum() / (n * spp)))


if __name__ == "__main__":
    main()
