#!/usr/bin/env python3
"""What the reference host's COMMITTED configuration costs: FULL_DIAGNOSTICS (16-byte records, ProjectSettings.asset:590), traceDepth 32 and up to 50
samples per batch (Assets/Prefabs/Raytracer.prefab:383-391) against the benchmark configuration (4-byte records, depth 8), on the same scene and frame.
Prints one JSON object: Msamples/s, MRays/s and kernel ms for every (depth, record size, spp) combination, chains of `--chain` batches.

  python profiles/host_default_probe.py [--scene cover] [--chain 10]
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cover")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--chain", type=int, default=10)
    ap.add_argument("--cases", default="8:4:256,8:16:256,16:4:256,16:16:256,32:4:256,32:16:256,32:16:50,32:4:50,8:4:50")
    ap.add_argument("--context-flags", type=int, default=0)
    ap.add_argument("--tune", default=None, help="RtowContextOptions.schedulerTune: 9 comma-separated integers")
    args = ap.parse_args()
    scene = getattr(rt.scenes, {"cover": "cover_scene", "stress": "stress_scene", "moving": "moving_scene", "mesh": "mesh_grid_scene", "mixed": "mixed_scene"}[args.scene])()
    w, h = args.width, args.height
    n = w * h
    a = rt.abi
    lib = rt.lib.load()
    out = {"scene": args.scene, "width": w, "height": h, "chain": args.chain, "tune": args.tune, "cases": {}}
    focus = scene.meta.get("focus")
    with rt.Context(0, flags=args.context_flags, scheduler_tune=[int(x) for x in args.tune.split(",")] if args.tune else None) as ctx:
        ctx.upload_scene(scene.desc())
        bufs = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        diags = [rt.DeviceBuffer(ctx, n * 16).zero() for _ in range(args.chain)]
        for case in args.cases.split(","):
            depth, stride, spp = (int(x) for x in case.split(":"))
            plist = []
            for k in range(args.chain):
                plist.append(rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=1 + k, diagnostics_stride=stride, focus=focus))
            ms = None
            for rep in range(3):                              # first: cost probe + threshold measurement; best of the next two
                for b in bufs:
                    b.zero()
                for k in range(args.chain):
                    plist[k].seed = 1 + k + 100 * rep
                rt.lib.check(rt.sample_batch_chain_device(ctx, plist, bufs, bufs, diags), "rtowSampleBatchChainDevice")
                ctx.synchronize()
                t = ctx.last_sample_kernel_ms()
                if rep > 0:
                    ms = t if ms is None else min(ms, t)
            import numpy as np
            rays = 0.0
            for d in diags:
                rays += float(d.download(np.float32, (n, 4))[:, 0].sum()) if stride == 16 else float(d.download(np.float32, (n * 4,))[:n].sum())
            ok = float(bufs[0].download(np.float32, (n, 4))[:, 3].sum())
            out["cases"]["depth %d, %d-byte records, %d spp" % (depth, stride, spp)] = {
                "kernel_ms_per_batch": round(ms / args.chain, 3), "msamples_per_s": round(n * spp * args.chain / ms / 1e3, 1), "mrays_per_s": round(rays / ms / 1e3, 1),
                "rays_per_sample": round(rays / (n * spp * args.chain), 4), "successful_sample_ratio": round(ok / (n * spp * args.chain), 4),
                "threshold_set": int(ctx.scene_info().thresholdSet)}
        for b in bufs + diags:
            b.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
