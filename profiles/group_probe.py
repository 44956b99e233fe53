#!/usr/bin/env python3
"""Batch groups (rtowSampleBatchGroupDevice) against chains and single launches: kernel ms per batch for `count` batches of the cover scene at 1080p.
  python profiles/group_probe.py [--cases depth:spp:count,...]"""
import argparse
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="8:256:10,8:32:8,8:32:4,8:32:2,32:256:10,32:50:10,16:128:8")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    scene = rt.scenes.cover_scene()
    w, h = args.width, args.height
    n = w * h
    out = {"width": w, "height": h, "cases": {}}
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene.desc())
        zero = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        outs = [[rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)] for _ in range(16)]
        for case in args.cases.split(","):
            depth, spp, count = (int(x) for x in case.split(":"))
            plist = [rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=1 + k) for k in range(count)]
            res = {}
            for mode in ("single", "chain", "group"):
                best = None
                for rep in range(3):
                    for k in range(count):
                        plist[k].seed = 1 + k + 50 * rep
                    if mode == "single":
                        ms = 0.0
                        for k in range(count):
                            job = rt.SampleBatchJob(ctx, plist[k])
                            job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = zero
                            job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs[k]
                            rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
                            ctx.synchronize()
                            ms += ctx.last_sample_kernel_ms()
                    elif mode == "chain":
                        for b in outs[0]:
                            b.zero()
                        rt.lib.check(rt.sample_batch_chain_device(ctx, plist, outs[0], outs[0]), "rtowSampleBatchChainDevice")
                        ctx.synchronize()
                        ms = ctx.last_sample_kernel_ms()
                    else:
                        rt.lib.check(rt.sample_batch_group_device(ctx, plist, zero, outs[:count]), "rtowSampleBatchGroupDevice")
                        ctx.synchronize()
                        ms = ctx.last_sample_kernel_ms()
                    if rep > 0:
                        best = ms if best is None else min(best, ms)
                res[mode + "_ms_per_batch"] = round(best / count, 3)
                res[mode + "_msamples_per_s"] = round(n * spp * count / best / 1e3, 1)
            out["cases"]["depth %d, %d spp, %d batches" % (depth, spp, count)] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
