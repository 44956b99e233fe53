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

`--group K` (default 8): additionally renders K consecutive steps' sub-batches of ONE rank as one launch (rtowSampleBatchGroupDevice: what bench.py --gpus N
does) and reports kernel time per step, and the whole frame as the chain of 10 batches bench.py times at N = 1 - so that `predicted_speedup_vs_n1_bench`
is the ratio of the two numbers the driver's scaling run would compare.

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
    ap.add_argument("--group", type=int, default=8, help="sub-batches (steps) of one rank per launch for the batch-group figure (0 = skip)")
    ap.add_argument("--pipelined", type=int, default=0, help=argparse.SUPPRESS)
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

        group_outs = [[rt.DeviceBuffer(ctx, n * k * 4).zero() for k in (4, 3, 3, 1)] for _ in range(max(args.group, 1))] if args.group else []

        def grouped(t, T, samples, rank_of_world, world, tiles):
            """kernel ms per step when the rank renders args.group steps' sub-batches in one launch"""
            plist = []
            for step in range(1, args.group + 1):
                plan = rt.Context.hybrid_plan(world, rank_of_world, tiles, spp, step)
                plist.append(rt.scenes.make_params(scene, w, h, spp=samples, trace_depth=depth, slice_offset=t, slice_divider=T, seed=int(plan.seed)))
            ms = None
            for rep in range(2):
                rt.lib.check(rt.sample_batch_group_device(ctx, plist, zero, group_outs[:len(plist)]), "rtowSampleBatchGroupDevice")
                ctx.synchronize()
                ms = ctx.last_sample_kernel_ms()
            return ms / len(plist)

        def whole_chain(count=10):
            plist = [rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=1 + k) for k in range(count)]
            ms = None
            for rep in range(2):
                rt.lib.check(rt.sample_batch_chain_device(ctx, plist, outs, outs), "rtowSampleBatchChainDevice")
                ctx.synchronize()
                ms = ctx.last_sample_kernel_ms()
            return ms / count

        whole = render(0, 1, spp, 1)
        out["whole_frame_ms"] = round(whole, 3)
        if args.group:
            out["whole_frame_chain10_ms_per_step"] = round(whole_chain(10 if args.config != 3 else 2), 3)      # what bench.py times at N = 1 (4K / 1024 spp: chains of 2 keep the run short; the chain gains < 1 % there)
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
                if args.group and (T == 1 or T == G or T * T == G):
                    gm = max(grouped(t, T, spp // B, t + T * (B - 1), G, T) for t in sorted({0, T - 1}))
                    entry["group_render_ms_per_step"] = round(gm, 3)
                    entry["predicted_step_ms_grouped"] = round(gm + exchange_ms + gather_ms, 3)
                    entry["predicted_speedup_vs_n1_bench"] = round(out["whole_frame_chain10_ms_per_step"] / (gm + exchange_ms + gather_ms), 3)
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
        for b in zero + outs + acc + [diag] + [x for o in group_outs for x in o]:
            b.free()
    best = {}
    key = "predicted_speedup_vs_n1_bench" if args.group else "predicted_speedup_vs_whole_frame"
    for k, v in out["partitions"].items():
        G = int(k.split()[0])
        if key in v and (G not in best or v[key] > out["partitions"][best[G]][key]):
            best[G] = k
    out["best_partition_per_world"] = {str(G): {"partition": k, key: out["partitions"][k][key]} for G, k in sorted(best.items())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
