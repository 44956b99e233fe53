#!/usr/bin/env python3
"""Predicts tile-parallel scaling on ONE GPU: renders slice g of G (SliceOffset = g, SliceDivider = G, the reference's own interlacing
contract, JOBS/SampleBatchJob.cs:69-70) for G = 1, 2, 4, 8 and reports whole-frame kernel time / slowest slice's kernel time.  A G-GPU
node runs the G slices concurrently, one per GPU, and then gathers colour rows (4.1 MB per peer at 1080p, 16.6 MB at 4K: < 0.3 ms over
xGMI), so the slowest slice's kernel time is the batch time the node would see; the gather is added as a stated estimate.

  python profiles/emulate_tile_split.py [--config 2|3] [--rng reference|per-sample] > gpurun_out/tiles_<config>.json

Every slice is rendered twice (the first launch on a new slice configuration includes the cost probe and runs in probe order; the second
uses the measured chunk order, like every batch after the first one of a frame) and the second launch is the one timed.
"""
import argparse
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")

CONFIGS = {2: ("cover", 1920, 1080, 256, 8), 3: ("cover", 3840, 2160, 1024, 16), 4: ("stress", 1920, 1080, 256, 8), 5: ("moving", 1920, 1080, 512, 8)}
XGMI_LINK_GBS = 153.0 * 0.8    # per direct peer link, achievable fraction (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--rng", choices=["reference", "per-sample"], default="reference")
    ap.add_argument("--spp", type=int, default=None, help="override the config's samples per pixel")
    ap.add_argument("--slices", default="1,2,4,8")
    ap.add_argument("--tune", default=None, help="RtowContextOptions.schedulerTune: 8 stage thresholds + the box-walk slice, comma separated")
    ap.add_argument("--block-threads", default="0", help="kept for old command lines: only 0 / 1024 exist since round 4 (the 512- / 256-lane workgroups were measured slower and removed)")
    args = ap.parse_args()
    name, w, h, spp, depth = CONFIGS[args.config]
    if args.spp:
        spp = args.spp
    scene = {"cover": rt.scenes.cover_scene, "stress": rt.scenes.stress_scene, "moving": rt.scenes.moving_scene}[name]()
    n = w * h
    out = {"config": args.config, "scene": name, "width": w, "height": h, "spp": spp, "depth": depth, "rng": args.rng, "slices": {}}
    tune = [int(x) for x in args.tune.split(",")] if args.tune else None
    out["tune"] = tune
    whole_ms = None
    results = {}
    for bt in [int(x) for x in args.block_threads.split(",")]:
      with rt.Context(0, scheduler_tune=tune, slice_block_threads=bt) as ctx:
        ctx.upload_scene(scene.desc())
        bufs = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        outs = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        diag = rt.DeviceBuffer(ctx, n * 4).zero()
        out["slices"] = {}
        for G in [int(x) for x in args.slices.split(",")]:
            if G == 1 and whole_ms is not None:
                continue
            times = []
            for g in range(G):
                p = rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, slice_offset=g, slice_divider=G,
                                          rng_policy=rt.abi.RNG_PER_SAMPLE if args.rng == "per-sample" else rt.abi.RNG_REFERENCE)
                job = rt.SampleBatchJob(ctx, p)
                job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
                job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
                job.OutputDiagnostics = diag
                ms = None
                for _ in range(2):
                    rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
                    ctx.synchronize()
                    ms = ctx.last_sample_kernel_ms()
                times.append(ms)
            if G == 1:
                if bt not in (0, 1024):
                    continue                       # the whole frame is the baseline: default geometry only
                whole_ms = times[0]
            gather_ms = 0.0 if G == 1 else (n // G) * 16 / (XGMI_LINK_GBS * 1e9) * 1e3     # colour rows of one peer over its own link
            out["slices"][str(G)] = {
                "kernel_ms_per_slice": [round(t, 3) for t in times],
                "slowest_ms": round(max(times), 3),
                "predicted_speedup_kernel": round(whole_ms / max(times), 3),
                "gather_ms_estimate": round(gather_ms, 3),
                "predicted_speedup_with_gather": round(whole_ms / (max(times) + gather_ms), 3),
                "msamples_per_s": round(n * spp / (max(times) + gather_ms) / 1e3, 1),
            }
        for b in bufs + outs + [diag]:
            b.free()
        results[str(bt)] = out["slices"]
    out["whole_frame_ms"] = whole_ms
    out["slices"] = results if len(results) > 1 else next(iter(results.values()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
