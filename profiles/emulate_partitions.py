#!/usr/bin/env python3
"""Predicts multi-GPU scaling of every reference-stream partition on ONE GPU: tiles (T = G), batches (T = 1) and the tiles x batches hybrids between
them (rtowHybridPlan / rtowExchangeAccumDevice, include/rtow.h).  A G-GPU node runs its G ranks concurrently, one per GPU; here each distinct rank
workload of a partition is rendered on the one GPU, and the step time the node would see is

    slowest rank's render time  +  exchange estimate  +  gather estimate

Rank (tile t, group b) of a T x B partition renders slice t of T with spp / B samples and the Seed of sub-batch b from zeroed inputs; the ranks of a
tile differ only in the Seed (statistically equal work), so group 0 and group B - 1 of every tile are rendered - twice each: the first launch of a
slice configuration includes the cost probe and the per-scene threshold measurement, the second is timed, like every batch after the first of a frame.

Estimates (stated, not measured - one GPU has no xGMI): exchange = every rank sends and receives (B - 1) messages of its fold rows' 44 B / pixel
(n / G pixels), each pair on its own link, in parallel: n / G x 44 B / (153 GB/s x 0.8) + 30 us; packing + fold = 3 x that volume x B through HBM at
4 TB/s; gather = n / G x 16 B on the root's links (colour rows) + 30 us.  Measured on this GPU and reported beside it: the pack + fold + scatter
kernels' actual time for the G = 8 case.

`--pipelined K`: additionally runs K consecutive steps of ONE rank's workload (fresh Seed each, as the rank of a node would) back to back and
reports wall time per step: what a rank sustains when the host keeps its queue full (launch gaps, chunk-order refresh included).

  python profiles/emulate_partitions.py --config 2 > gpurun_out/partitions_c2.json
"""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")

CONFIGS = {2: ("cover", 1920, 1080, 256, 8), 3: ("cover", 3840, 2160, 1024, 16), 4: ("stress", 1920, 1080, 256, 8), 5: ("moving", 1920, 1080, 512, 8)}
XGMI_LINK_GBS = 153.0 * 0.8    # per direct peer link, achievable fraction (MI355X_MICROARCH.md)
COLLECTIVE_LATENCY_MS = 0.03
HBM_EFFECTIVE_GBS = 4000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--pipelined", type=int, default=6, help="steps of one rank's workload run back to back for the sustained per-step time (0 = skip)")
    ap.add_argument("--context-flags", type=int, default=0)
    args = ap.parse_args()
    name, w, h, spp, depth = CONFIGS[args.config]
    if args.spp:
        spp = args.spp
    scene = {"cover": rt.scenes.cover_scene, "stress": rt.scenes.stress_scene, "moving": rt.scenes.moving_scene}[name]()
    n = w * h
    out = {"config": args.config, "scene": name, "width": w, "height": h, "spp": spp, "depth": depth, "rng": "reference", "partitions": {}}
    cache = {}
    with rt.Context(0, flags=args.context_flags) as ctx:
        ctx.upload_scene(scene.desc())
        zero = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        outs = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        diag = rt.DeviceBuffer(ctx, n * 4).zero()

        def render(t, T, samples, seed, repeats=2):
            key = (t, T, samples, seed)
            if key in cache:
                return cache[key]
            p = rt.scenes.make_params(scene, w, h, spp=samples, trace_depth=depth, slice_offset=t, slice_divider=T, seed=seed)
            job = rt.SampleBatchJob(ctx, p)
            job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = zero
            job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
            job.OutputDiagnostics = diag
            ms = None
            for _ in range(repeats):
                rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
                ctx.synchronize()
                ms = ctx.last_sample_kernel_ms()
            cache[key] = ms
            return ms

        def pipelined(t, T, samples, first_seed, steps):
            """`steps` consecutive steps of one rank's workload, enqueued without waiting in between: wall time per step."""
            p = rt.scenes.make_params(scene, w, h, spp=samples, trace_depth=depth, slice_offset=t, slice_divider=T, seed=first_seed)
            job = rt.SampleBatchJob(ctx, p)
            job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = zero
            job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = outs
            job.OutputDiagnostics = diag
            rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
            ctx.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                p.seed = first_seed + 8 * (k + 1)
                rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
            ctx.synchronize()
            return (time.perf_counter() - t0) * 1e3 / steps

        whole = render(0, 1, spp, 1)
        out["whole_frame_ms"] = round(whole, 3)
        if args.pipelined:
            out["whole_frame_pipelined_ms_per_step"] = round(pipelined(0, 1, spp, 1, max(2, args.pipelined // 2)), 3)
        for G in [int(x) for x in args.worlds.split(",")]:
            if G == 1:
                continue
            for T in [d for d in range(1, G + 1) if G % d == 0]:
                B = G // T
                if spp // B < 1:
                    continue
                times = {}
                for t in range(T):
                    for b in sorted({0, B - 1}):
                        plan = rt.Context.hybrid_plan(G, t + T * b, T, spp, 1)
                        times["tile %d group %d" % (t, b)] = round(render(plan.sliceOffset, plan.sliceDivider, plan.samples, plan.seed), 3)
                slowest = max(times.values())
                fold_pixels = n / G
                exchange_ms = 0.0 if B == 1 else fold_pixels * 44 / (XGMI_LINK_GBS * 1e9) * 1e3 + COLLECTIVE_LATENCY_MS + 3 * fold_pixels * 44 * B / (HBM_EFFECTIVE_GBS * 1e9) * 1e3
                gather_ms = fold_pixels * 16 / (XGMI_LINK_GBS * 1e9) * 1e3 + COLLECTIVE_LATENCY_MS
                step = slowest + exchange_ms + gather_ms
                entry = {"tiles": T, "groups": B, "samples_per_rank": spp // B, "render_ms": times, "slowest_render_ms": slowest, "exchange_ms_estimate": round(exchange_ms, 3),
                         "gather_ms_estimate": round(gather_ms, 3), "predicted_step_ms": round(step, 3), "predicted_speedup_vs_whole_frame": round(whole / step, 3),
                         "predicted_msamples_per_s": round(n * spp / step / 1e3, 1)}
                if args.pipelined and (T == 1 or T == G):
                    pm = pipelined(0, T, spp // B, 1, args.pipelined)
                    entry["pipelined_render_ms_per_step"] = round(pm, 3)
                    entry["predicted_speedup_pipelined_vs_pipelined_whole"] = round(out["whole_frame_pipelined_ms_per_step"] / (pm + exchange_ms + gather_ms), 3)
                out["partitions"]["%d GPUs: %d tiles x %d groups" % (G, T, B)] = entry
        # the exchange's own kernels at G = 8, T = 1, measured: pack 7 peers' rows, fold 8 sources, on this GPU (no transport)
        a = rt.abi
        acc = [rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)]
        bp, ba = a.AccumBuffers(*[b.ptr for b in outs]), a.AccumBuffers(*[b.ptr for b in acc])
        ctx.exchange_accum(w, h, 1, bp, ba)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.exchange_accum(w, h, 1, bp, ba)
        ctx.synchronize()
        out["single_rank_fold_ms_measured"] = round((time.perf_counter() - t0) * 100, 4)      # accum += partial over the whole frame: 8 x what one of 8 ranks folds per source
        for b in zero + outs + acc + [diag]:
            b.free()
    best = {}
    for k, v in out["partitions"].items():
        G = int(k.split()[0])
        if G not in best or v["predicted_speedup_vs_whole_frame"] > out["partitions"][best[G]]["predicted_speedup_vs_whole_frame"]:
            best[G] = k
    out["best_partition_per_world"] = {str(G): {"partition": k, "predicted_speedup_vs_whole_frame": out["partitions"][k]["predicted_speedup_vs_whole_frame"]} for G, k in sorted(best.items())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
