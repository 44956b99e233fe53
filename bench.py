#!/usr/bin/env python3
"""bench.py - headline benchmark of the sample-batch path on MI355X.

Metric (BASELINE.json): Msamples/s on the 486-sphere book-cover scene, 1920x1080, 8 bounces.
A "step" is one sample batch (SampleBatchJob over the whole frame, `--spp` samples per pixel, default 256 =
BASELINE.json configs[1]) with the accumulators already resident in HBM; successive steps accumulate into
ping-pong buffers with a fresh seed, exactly like the reference's successive batches
(Assets/Scripts/Unity/Raytracer.cs:656-661,798-802).

  python bench.py --gpus 1 --steps 8 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N = 1: the K timed steps are enqueued as equally long chains of at most 16 successive batches (`--chain`) (rtowSampleBatchChainDevice; the reference itself keeps
two batches in flight, Raytracer.cs:586-593): one launch per chain, in which a pixel chunk's next batch starts as soon as its previous
batch is stored.  The same K steps as plain one-launch-per-batch calls and the host-buffer form (rtowSampleBatch on pinned host arrays)
are measured after the timed region and reported next to `value` (`plain_batches`, `host_buffer_ms_per_step`).

N > 1: one process per GPU, no data-path collective inside a batch; the batch's results are combined over RCCL afterwards.  The total work is fixed
(same frame, same spp), so scaling is "strong".  `value` is the reference-stream partition that scales (`--partition hybrid`, the default):
  hybrid   tiles x batches behind the C ABI (rtowHybridPlan / rtowExchangeAccumDevice / rtowGatherRowsDevice, include/rtow.h): G GPUs = T row slices x
           B seed groups; rank tile + T * group renders slice `tile` of T (the reference's SliceOffset / SliceDivider, JOBS/SampleBatchJob.cs:69-70) with
           spp / B samples and the Seed of sub-batch `group` from zeroed accumulators - the reference's own successive batches of a frame
           (fresh frameSeed, sums carried on: UNITY/Raytracer.cs:656-661,798-802) run at the same time instead of one after the other; ONE grouped
           ncclSend / ncclRecv exchange folds every row on the rank that owns it (row % G) in group order, ONE gather of colour rows per batch brings
           the frame to rank 0.  T = 1 (samples only) unless spp < G (`--tiles`); bit-identical to the sub-batches folded in group order on one GPU,
           within 1e-4 of the mean of the sequential accumulation (tests/);
and the line also carries, measured in the same run (`partitions`):
  tiles    north_star's partition: the frame row-interleaved over the GPUs (SliceDivider = N), one gather of colour rows per batch
           (rtowCommInit / rtowGatherRowsDevice); bit-identical to the single-GPU frame; limited by lane-per-pixel granularity under
           the reference RNG stream (a GPU with one pixel per lane finishes when its slowest pixel's sequential samples do: DESIGN.md 6);
  tiles under --rng per-sample (RTOW_RNG_PER_SAMPLE, NOT the reference's random stream: a pixel's samples become independent units).

Rank 0 prints ONE JSON line (see the task contract) that also carries `roofline`, `cpu_baseline` (the C2 sample), `cpu_baseline_c1`
(BASELINE.json configs[0] in full: 400x225, 8 spp), `plain_batches` (one launch per batch), `chain2` (two batches per launch: the queue depth of
the unmodified reference host, UNITY/Raytracer.cs:586-593) and `post_passes` (CombineJob / FinalizeTexturesJob / ReduceMetricsJob / accumulator
add on the device: achieved GB/s against the HBM roofline, the kernels of this repository that ARE bandwidth bound).

RTOW_BENCH_DEBUG_SHARED_GPU=1 (development, one-GPU box): every rank uses cuda:0, torch.distributed runs over gloo and the C-ABI communicator
over the tests' stand-in transport (tests/native/fake_rccl.cpp) - the N > 1 code path end to end, never a measurement (the line says so).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def measured_hbm_traffic(workload="c2"):
    """HBM bytes PER BATCH of the sample kernel from the newest committed rocprofv3 PMC summary of THIS workload (profiles/rNN[x]_pmc_summary.json for the
    headline C2 command, profiles/rNN[x]_<workload>_pmc_summary.json - c4, c5, mesh ... - for the others; profiles/collect.sh + summarize.py:
    FETCH_SIZE and WRITE_SIZE from separate --pmc passes of this same bench command, KiB -> bytes, read side doubled as
    MI355X_MICROARCH.md prescribes for gfx950; the profiled launch held `batches_per_launch` batches).  PMC counters cannot be
    collected from inside the timed run."""
    import glob
    import re
    pattern = re.compile(r"^r\d\d[a-z]*_pmc_summary\.json$" if workload == "c2" else r"^r\d\d[a-z]*_%s_pmc_summary\.json$" % re.escape(workload or "-"))
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")) if pattern.match(os.path.basename(f)))
    for f in reversed(files):
        try:
            d = json.load(open(f))
            keys = ("valu_issue_utilisation", "valu_lane_utilisation", "simd_cycles_per_valu_inst", "valu_pipe_busy_estimate", "lds_busy_fraction", "l2_hit_rate", "l2_memory_read_bytes_estimate",
                    "icache_hit_rate")
            secondary = {k: round(float(d["derived"][k]), 4) for k in keys if k in d["derived"]}
            if "valu_issue_utilisation" in secondary and "valu_lane_utilisation" in secondary:
                # what actually bounds the kernel: VALU instructions issued per SIMD cycle against the peak rate x lanes doing useful work in them
                secondary["valu_frac_of_peak_lane_issue"] = round(secondary["valu_issue_utilisation"] * secondary["valu_lane_utilisation"], 4)
            per_launch = int(d.get("bench_line_under_profiler", {}).get("config", {}).get("batches_per_launch", 1) or 1)
            return float(d["derived"]["hbm_traffic_bytes"]) / per_launch, os.path.basename(f), secondary
        except (KeyError, ValueError, OSError):
            continue
    return None, None, {}


def usable_cores():
    """Logical cores this process may actually use: affinity mask capped by the cgroup CPU quota (cpu.max)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


# BASELINE.json configs the driver can time on one GPU (configs[0] is the reference's own CPU-runnable case: the cpu_baseline leg)
CONFIGS = {
    2: dict(scene="cover", width=1920, height=1080, spp=256, depth=8, label="configs[1]: cover scene 1920x1080, 256 spp, 8 bounces"),
    3: dict(scene="cover", width=3840, height=2160, spp=1024, depth=16, label="configs[2]: cover scene 3840x2160, 1024 spp, 16 bounces"),
    4: dict(scene="stress", width=1920, height=1080, spp=256, depth=8, label="configs[3]: 10k-sphere stress scene 1920x1080, 256 spp, 8 bounces"),
    5: dict(scene="moving", width=1920, height=1080, spp=512, depth=8, label="configs[4]: moving spheres + defocus blur 1920x1080, 512 spp, 8 bounces"),
}
SCENE_TEXT = {
    "cover": "cover scene (486 spheres, generated per Final Scene (Book 1).asset, seed 700)",
    "stress": "stress scene (10 000 spheres dart-thrown on 100x100, seed 10000; tree does not fit LDS)",
    "moving": "moving-spheres scene (Random With Movement (Book 2).asset: 80 % of the random spheres move, aperture 0.05)",
    "mesh": "mesh-grid scene (14 x 14 icospheres of 1 280 smooth triangles + floor = 250 882 triangle entities, one per mesh triangle like the reference's live host, "
            "materials blended across the grid like UNITY/GridGenerator.cs; 32-bit candidate codes, tree in HBM, exact-tie kernels)",
    "mixed": "mixed-primitive scene (spheres, rects, boxes, triangles, rotated and moving: the general-entity kernels, scene in LDS)",
    "volumes": "Cornell-with-volumes scene (ProbabilisticVolume boxes and spheres: the volume kernels, every hit of a ray kept and sorted)",
    "textured": "image-textured scene (per-hit albedo / emission / metallic / glossiness: the textured kernels)",
    "meshfog": "mesh grid of 81 922 triangles with three fog volumes (volume kernels with 32-bit codes, tree in HBM, hit lists spilling to HBM)",
}


def cpu_baseline(rt, scene, width, height, depth, budget_s=10.0, scene_name="cover", full_spp=256):
    """Time the CPU restatement of the reference Burst path (oracle, -O3 -ffast-math build) on this host's cores.

    Bounded sample of the SAME workload: the full 1920x1080 frame of the cover scene at a reduced spp, chosen from a
    short calibration run so that each of the two timed runs costs about `budget_s` seconds.  Scheduling mirrors
    Schedule(W*H, 1): one task per pixel, dynamic hand-out, all logical cores (UNITY/Raytracer.cs:730).
    """
    from oracle import binding as ob  # checker / baseline only - never on the product path

    osc = ob.OracleScene(scene.desc(), kind="fast")
    cores = usable_cores()
    # calibration on the workload itself (full frame, 8 spp, ~1 s): small frames under-report the rate (thread start-up, cold caches)
    focus = scene.meta.get("focus")
    cal = rt.scenes.make_params(scene, width, height, spp=8, trace_depth=depth, focus=focus)
    t = time.perf_counter()
    osc.sample_batch(cal, nthreads=cores)
    cal_rate = width * height * 8 / (time.perf_counter() - t)
    spp = int(max(8, min(128, round(budget_s * cal_rate / (width * height)))))
    p = rt.scenes.make_params(scene, width, height, spp=spp, trace_depth=depth, focus=focus)
    best = None
    rays = 0
    for _ in range(2):
        t = time.perf_counter()
        _, counters = osc.sample_batch(p, nthreads=cores, want_counters=True)
        dt = time.perf_counter() - t
        rays = counters.rays
        best = dt if best is None else min(best, dt)
    osc.close()
    return {
        "value": round(width * height * spp / best / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%s scene %dx%d, %d spp (of the %d-spp workload), %d bounces, 1 batch, best of 2; "
                  "C++ restatement of the reference Burst path, -O3 -ffast-math, 1 task/pixel dynamic" % (scene_name, width, height, spp, full_spp, depth),
        "mrays_per_s": round(rays / best / 1e6, 3),
        "seconds": round(best, 3),
    }


def cpu_baseline_c1(rt):
    """BASELINE.json configs[0] in full - cover scene 400x225, 8 spp, 8 bounces - on the CPU restatement ("timed in the same bench
    run on C1 (full)"; the reference's call site is UNITY/Raytracer.cs:730).  0.72 M samples: best of 5 runs of about 50 ms each."""
    from oracle import binding as ob  # checker / baseline only - never on the product path

    scene = rt.scenes.cover_scene()
    osc = ob.OracleScene(scene.desc(), kind="fast")
    cores = usable_cores()
    w, h, spp, depth = 400, 225, 8, 8
    p = rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=depth)
    best, rays = None, 0
    for _ in range(5):
        t = time.perf_counter()
        _, counters = osc.sample_batch(p, nthreads=cores, want_counters=True)
        dt = time.perf_counter() - t
        rays = counters.rays
        best = dt if best is None else min(best, dt)
    osc.close()
    return {"value": round(w * h * spp / best / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "BASELINE.json configs[0] in full: cover scene 400x225, 8 spp, 8 bounces, 1 batch, best of 5; C++ restatement of the reference Burst path, "
                      "-O3 -ffast-math, 1 task/pixel dynamic", "mrays_per_s": round(rays / best / 1e6, 3), "seconds": round(best, 4)}


def post_passes(rt, ctx, lib, torch, dev, stream, sizes=((1920, 1080), (3840, 2160)), iters=20):
    """Achieved HBM GB/s of the post passes next to the sample kernel (JOBS/CombineJob.cs:29-71, JOBS/FinalizeTexturesJob.cs:23-55,
    JOBS/ReduceMetricsJob.cs:22-45, rtowAddAccumDevice), timed with HIP events on the stream they are launched on.  Bytes per pixel are the
    algorithmic ones: combine 44 read + 36 written, finalize 36 + 12, combine_finalize (the two fused: the reference's default chain) 40 + 12,
    metrics 24 read, add 88 read + 44 written.
    Peak 8 TB/s (spec), ~6.3 TB/s is what a float4 copy reaches on this part (MI355X_MICROARCH.md).  A 1080p working set (91-166 MB) partly
    lives in the 256 MB Infinity Cache between iterations - the 4K figures (365-663 MB) are the HBM ones."""
    import ctypes as C

    abi = rt.abi
    out = {"peak_GBps": HBM_PEAK_GBS, "achievable_GBps": 6300.0, "iterations": iters}
    for w, h in sizes:
        n = w * h
        color4 = torch.rand(n, 4, device=dev)
        color4[:, 3] = 8.0
        normal, albedo = torch.rand(n, 3, device=dev), torch.rand(n, 3, device=dev)
        o3 = [torch.empty(n, 3, device=dev) for _ in range(3)]
        rgba = [torch.empty(n, 4, device=dev, dtype=torch.uint8) for _ in range(3)]
        diag, scw = torch.rand(n, device=dev), torch.rand(n, device=dev)
        acc2 = [torch.zeros(n, c, device=dev) for c in (4, 3, 3)] + [torch.zeros(n, device=dev)]
        cp = abi.CombineParams(w, h, 0, 1)
        metrics = abi.Metrics()
        metrics_dev = torch.zeros(16, device=dev, dtype=torch.int32)      # the asynchronous reduction's record (device memory; a host would register a pinned record)
        # a stream of its own with a real handle: the C ABI maps a NULL stream to the CONTEXT's stream, which torch's events would not see
        ps = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        sp = ps.cuda_stream
        assert sp != 0
        dst = abi.AccumBuffers(*[t.data_ptr() for t in acc2])
        src = abi.AccumBuffers(color4.data_ptr(), normal.data_ptr(), albedo.data_ptr(), scw.data_ptr())
        passes = {
            "combine": (80, lambda: lib.rtowCombineDevice(ctx.handle, C.byref(cp), color4.data_ptr(), normal.data_ptr(), albedo.data_ptr(), o3[0].data_ptr(), o3[1].data_ptr(), o3[2].data_ptr(), sp)),
            "finalize": (48, lambda: lib.rtowFinalizeDevice(ctx.handle, n, o3[0].data_ptr(), o3[1].data_ptr(), o3[2].data_ptr(), rgba[0].data_ptr(), rgba[1].data_ptr(), rgba[2].data_ptr(), sp)),
            "reduce_metrics": (24, lambda: lib.rtowReduceMetricsDevice(ctx.handle, n, diag.data_ptr(), 4, color4.data_ptr(), scw.data_ptr(), sp, C.byref(metrics))),
            "add_accum": (132, lambda: lib.rtowAddAccumDevice(ctx.handle, n, C.byref(dst), C.byref(src), sp)),
            # the reference's default post chain (denoiseMode 0) fused: accumulators in, three RGBA32 textures out
            "combine_finalize": (52, lambda: lib.rtowCombineFinalizeDevice(ctx.handle, C.byref(cp), color4.data_ptr(), normal.data_ptr(), albedo.data_ptr(), rgba[0].data_ptr(), rgba[1].data_ptr(), rgba[2].data_ptr(), sp)),
            "reduce_metrics_async": (24, lambda: lib.rtowReduceMetricsDeviceAsync(ctx.handle, n, diag.data_ptr(), 4, color4.data_ptr(), scw.data_ptr(), sp, metrics_dev.data_ptr())),
        }
        res = {}
        for name, (bytes_per_px, fn) in passes.items():
            for _ in range(2):
                rt.lib.check(fn(), name)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ps.synchronize()
            e0.record(ps)
            for _ in range(iters):
                rt.lib.check(fn(), name)
            e1.record(ps)
            e1.synchronize()
            ms = e0.elapsed_time(e1) / iters
            gbs = n * bytes_per_px / (ms * 1e-3) / 1e9
            res[name] = {"ms": round(ms, 4), "bytes_per_pixel": bytes_per_px, "GBps": round(gbs, 1), "frac_of_peak": round(gbs / HBM_PEAK_GBS, 4)}
        res["reduce_metrics"]["note"] = "includes the 8 KB copy of the per-block partials to the host that ends the call"
        out["%dx%d" % (w, h)] = res
        del color4, normal, albedo, o3, rgba, diag, scw, acc2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2,
                    help="BASELINE.json config to time (1-based like SURVEY.md 8: 2 = configs[1], the headline; 3 = 4K/1024 spp/16 bounces; 4 = 10k spheres; 5 = moving + defocus)")
    ap.add_argument("--scene", choices=sorted(SCENE_TEXT), default=None, help="override the config's scene")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--rng", choices=["reference", "per-sample", "per-sample-xoroshiro"], default="reference",
                    help="reference: the reference's per-pixel generator (same seed, same image: the headline); per-sample: RTOW_RNG_PER_SAMPLE, a different stream")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (plain batches, host-buffer form, other partitions)")
    ap.add_argument("--chain", type=int, default=None, help="batches per launch (rtowSampleBatchChainDevice); default: on one GPU the steps split into equal chains of at most 16, on several 1 (one gather per batch)")
    ap.add_argument("--prewarm", type=float, default=2.0, help="seconds of untimed launches of the timed kind in front of the W warmup steps of the main measurement (clocks / power state of a fresh box; 0 = none)")
    ap.add_argument("--tune", default=None, help="development: RtowContextOptions.schedulerTune as 9 comma-separated integers")
    ap.add_argument("--context-flags", type=int, default=0, help="development: RtowContextOptions.flags (e.g. 1 = exact-tie kernels always)")
    ap.add_argument("--only-leg", choices=("group_fold", "host_default_chain", "host_default_group", "host_default_adaptive", "plain_two_in_flight"), default=None,
                    help="profiling aid: run only this secondary measurement (--steps batches, --chain per launch) and print its block")
    ap.add_argument("--post-only", default=None, metavar="WxH", help="profiling aid: run only the post-pass measurement at this frame size and print its block (profiles/collect.sh)")
    ap.add_argument("--partition", choices=("hybrid", "tiles", "batches"), default="hybrid", help="which N > 1 partition `value` reports (the other is reported beside it): hybrid = tiles x batches "
                    "behind the C ABI (the reference stream's scalable split), tiles = rows only (north_star's), batches = hybrid with one tile")
    ap.add_argument("--group", type=int, default=None, help="N > 1, hybrid partition: sub-batches (steps) a rank renders per launch (rtowSampleBatchGroupDevice; 1 .. 16); default min(8, steps)")
    ap.add_argument("--tiles", type=int, default=None, help="T of the hybrid partition (must divide the number of GPUs); default: 1 unless spp < GPUs")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    overridden = [k for k in ("scene", "width", "height", "spp", "depth") if getattr(args, k) is not None]
    for k in ("scene", "width", "height", "spp", "depth"):
        if getattr(args, k) is None:
            setattr(args, k, cfg[k])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if args.chain is None:
        # one GPU: the K timed steps as equally long chains of at most 16 batches (the most one launch holds): 20 steps -> 10 + 10
        launches = -(-args.steps // 16)
        args.chain = -(-args.steps // launches) if world == 1 else 1
    args.chain = max(1, min(16, args.chain))

    import torch  # device memory, streams, torch.distributed (RCCL); loaded before the HIP library on purpose

    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    abi = rt.abi
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # RTOW_BENCH_DEBUG_SHARED_GPU=1: development aid for a 1-GPU box - all ranks share cuda:0 and talk over gloo, to exercise the N > 1
    # host path end to end.  Never a measurement: the line it prints says so in `config.partition`.
    shared_gpu = os.environ.get("RTOW_BENCH_DEBUG_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    W, H, spp, depth = args.width, args.height, args.spp, args.depth
    n = W * H
    scene = {"cover": rt.scenes.cover_scene, "stress": rt.scenes.stress_scene, "moving": rt.scenes.moving_scene, "mesh": rt.scenes.mesh_grid_scene,
             "mixed": rt.scenes.mixed_scene, "volumes": rt.scenes.volume_scene, "textured": rt.scenes.textured_scene, "meshfog": rt.scenes.mesh_grid_fog_scene}[args.scene]()
    focus = scene.meta.get("focus")                       # scenes without spheres carry their focus distance (the host's auto-focus probe, UNITY/Raytracer.cs:608-609)
    ctx = rt.Context(local_rank, flags=args.context_flags, scheduler_tune=[int(x) for x in args.tune.split(",")] if args.tune else None)
    ctx.upload_scene(scene.desc())
    info = ctx.scene_info()
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    if args.post_only:
        pw, ph = (int(x) for x in args.post_only.lower().split("x"))
        print(json.dumps({"post_passes": post_passes(rt, ctx, rt.lib.load(), torch, dev, torch.cuda.current_stream(dev), sizes=((pw, ph),))}), flush=True)
        ctx.close()
        return

    import ctypes as C
    lib = rt.lib.load()
    stream = torch.cuda.current_stream(dev)

    # the C-ABI communicator of the tile partition: rank 0 makes the id, torch.distributed is only the host channel that carries its 128 bytes
    # (if the communicator cannot be made on some rank - every rank learns it - the same rows travel through torch.distributed instead and the
    # line says so in config.gather)
    have_comm = False
    if world > 1:
        if shared_gpu:
            fake = os.path.join(ROOT, "tests", "build", "libfake_rccl.so")
            if not os.path.exists(fake):
                raise SystemExit("RTOW_BENCH_DEBUG_SHARED_GPU=1 needs %s (python __graft_entry__.py builds it)" % fake)
            rt.Context.comm_set_library_path(fake)
        mine_ok = 0
        try:
            box = [rt.Context.comm_unique_id() if rank == 0 else None]
        except Exception as e:                                  # noqa: BLE001 - reported, then the fallback
            box = [None]
            print("[bench] rank 0: rtowCommGetUniqueId failed: %s" % e, file=sys.stderr, flush=True)
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            try:
                ctx.comm_init(box[0], rank, world)
                mine_ok = 1
            except Exception as e:                              # noqa: BLE001
                print("[bench] rank %d: rtowCommInit failed: %s" % (rank, e), file=sys.stderr, flush=True)
        agreed = torch.tensor([mine_ok], device=dev if not shared_gpu else "cpu", dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        have_comm = bool(int(agreed.item()))
        if mine_ok and not have_comm:
            ctx.comm_destroy()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not dist:
            return x
        t = torch.tensor([x], device=dev if not shared_gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def measure(partition, rng, chain, steps, warmup, prewarm_s=0.0):
        """Time `steps` batches (after `warmup` untimed ones) under one partition / RNG policy / chain length; max over ranks.
        Returns wall seconds, mean kernel ms per step, and the buffers of the last batch (for the ray / success statistics)."""
        hybrid = world > 1 and partition in ("hybrid", "batches")
        tiles_t = 1 if partition == "batches" else (args.tiles or mg.default_tiles(world, spp))
        if hybrid and world % tiles_t:
            raise SystemExit("--tiles %d does not divide %d GPUs" % (tiles_t, world))

        def flat():
            return torch.zeros(mg.ACCUM_FLOATS * n, device=dev)

        state = {"ping_flat": flat(), "pong_flat": flat()}
        state["ping"], state["pong"] = mg.accum_views(state["ping_flat"], n), mg.accum_views(state["pong_flat"], n)
        zero_flat = flat() if hybrid else None            # never written: the input of every rank's sub-batch
        group = max(1, min(16, args.group if args.group else min(8, steps))) if hybrid else 1
        partial_views = [mg.accum_views(flat(), n) for _ in range(group)] if hybrid else None     # one set of partial-sum buffers per sub-batch of a launch
        diags = [torch.zeros(n, device=dev) for _ in range(max(1, chain, group))]
        plan0 = rt.Context.hybrid_plan(world, rank, tiles_t, spp, 1) if hybrid else None
        if hybrid:
            base = rt.scenes.make_params(scene, W, H, spp=int(plan0.samples), trace_depth=depth, slice_offset=int(plan0.sliceOffset), slice_divider=int(plan0.sliceDivider), focus=focus)
        else:
            base = rt.scenes.make_params(scene, W, H, spp=spp, trace_depth=depth, slice_offset=rank, slice_divider=world, focus=focus)
        kernel_ms = []

        def params_for(seed):
            p = abi.SampleParams.from_buffer_copy(base)
            p.seed = seed
            p.rngPolicy = {"per-sample": abi.RNG_PER_SAMPLE, "per-sample-xoroshiro": abi.RNG_PER_SAMPLE_XOROSHIRO}.get(rng, abi.RNG_REFERENCE)
            return p

        def launch(plist, src, dst):
            bi = abi.AccumBuffers(*[t.data_ptr() for t in src])
            bo = abi.AccumBuffers(*[t.data_ptr() for t in dst])
            if len(plist) == 1:
                rt.lib.check(lib.rtowSampleBatchDevice(ctx.handle, C.byref(plist[0]), C.byref(bi), C.byref(bo), diags[0].data_ptr(), stream.cuda_stream, None),
                             "rtowSampleBatchDevice")
            else:
                arr = (abi.SampleParams * len(plist))(*plist)
                dptr = (C.c_void_p * len(plist))(*[diags[k].data_ptr() for k in range(len(plist))])
                rt.lib.check(lib.rtowSampleBatchChainDevice(ctx.handle, len(plist), arr, C.byref(bi), C.byref(bo), dptr, stream.cuda_stream, None),
                             "rtowSampleBatchChainDevice")

        def run(first, count, record):
            i = first
            while i < first + count:
                c = min(chain, first + count - i)
                if hybrid:
                    # this rank's sub-batches of steps i + 1 .. i + c, each from zeroed inputs into its own partial-sum buffers, as ONE launch (a batch group: the
                    # launch ends with its slowest pixel-batch, not c of them in a row); then per step the exchange + ordered fold into the running accumulation
                    # (ping: this rank's rows, row % world == rank) and the gather of the colour rows on rank 0 (in place: ping is the frame there)
                    c = min(group, first + count - i)
                    plans = [rt.Context.hybrid_plan(world, rank, tiles_t, spp, i + 1 + k) for k in range(c)]
                    plist = [params_for(int(pl.seed)) for pl in plans]
                    bi = abi.AccumBuffers(*[t.data_ptr() for t in mg.accum_views(zero_flat, n)])
                    bo = (abi.AccumBuffers * c)(*[abi.AccumBuffers(*[t.data_ptr() for t in partial_views[k]]) for k in range(c)])
                    arr = (abi.SampleParams * c)(*plist)
                    dptr = (C.c_void_p * c)(*[diags[k].data_ptr() for k in range(c)])
                    rt.lib.check(lib.rtowSampleBatchGroupDevice(ctx.handle, c, arr, C.byref(bi), bo, dptr, stream.cuda_stream, None), "rtowSampleBatchGroupDevice")
                    acc = abi.AccumBuffers(*[t.data_ptr() for t in state["ping"]])
                    for k in range(c):
                        part = bo[k]
                        if have_comm:
                            ctx.exchange_accum(W, H, tiles_t, part, acc, what=abi.GATHER_ALL, stream=stream.cuda_stream)
                            ctx.gather_rows(W, H, world, acc, acc if rank == 0 else None, what=abi.GATHER_COLOR | abi.GATHER_NO_BATCH_WAIT, root=0, stream=stream.cuda_stream)
                        else:
                            shapes = (4, 3, 3, 1)
                            src = [t.view(H, W, q) for t, q in zip(partial_views[k], shapes)]
                            dst = [t.view(H, W, q) for t, q in zip(state["ping"], shapes)]
                            if shared_gpu:                        # gloo has no device point-to-point: host copies (development only)
                                hs, hd = [t.cpu() for t in src], [t.cpu() for t in dst]
                                mg.exchange_accum(hs, hd, H, rank, world, tiles_t)
                                for t, hcopy in zip(dst, hd):
                                    t.copy_(hcopy)
                                mg.gather_frame(mg.pack_owned(hd[0], rank, world), H, rank, world)
                            else:
                                mg.exchange_accum(src, dst, H, rank, world, tiles_t)
                                mg.gather_frame(mg.pack_owned(dst[0], rank, world), H, rank, world)
                else:
                    launch([params_for(i + 1 + k) for k in range(c)], state["ping"], state["pong"])
                    if world > 1:
                        # the one collective of the tile path: colour rows of every rank -> rank 0
                        if have_comm:
                            mine = abi.AccumBuffers(*[t.data_ptr() for t in state["pong"]])
                            ctx.gather_rows(W, H, world, mine, mine if rank == 0 else None, what=abi.GATHER_COLOR, root=0, stream=stream.cuda_stream)
                        else:
                            mg.gather_frame(mg.pack_owned(state["pong"][0].view(H, W, 4), rank, world), H, rank, world)
                    state["ping"], state["pong"], state["ping_flat"], state["pong_flat"] = state["pong"], state["ping"], state["pong_flat"], state["ping_flat"]
                if record:
                    kernel_ms.append(ctx.last_sample_kernel_ms())  # HIP events on the launch stream (synchronises); one launch = c steps
                i += c

        if prewarm_s > 0:
            # a fresh box runs its first seconds of sustained load 2 - 4 % slower (clocks, power state: profiles/r05_runs/repeatability.json); launches of the same kind as
            # the timed ones, untimed and before the W warmup steps, until the GPU has been busy for prewarm_s - then the accumulators are zeroed again
            tw = time.perf_counter()
            k = 0
            per = max(chain, group if hybrid else 1)
            # (several ranks: the launches carry collectives, so every rank makes the same number of them - 20 steps per second asked for - instead of watching its own clock)
            while (time.perf_counter() - tw < prewarm_s) if world == 1 else (k < int(20 * prewarm_s)):
                run(1000 + k, per, False)
                torch.cuda.synchronize(dev)
                k += per
            for key in ("ping_flat", "pong_flat"):
                state[key].zero_()
        run(0, warmup, False)
        barrier()
        t0 = time.perf_counter()
        run(warmup, steps, True)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        ctx.batch_status()                                  # the asynchronous batches report here (hit-list capacity)
        avg_kernel_ms = max_over_ranks(sum(kernel_ms) / max(steps, 1))
        launches = len(kernel_ms)
        return {"elapsed": elapsed, "kernel_ms_per_step": avg_kernel_ms, "launches": launches, "last": state["ping"], "diag": diags[0 if hybrid else ((steps % chain) or min(chain, steps)) - 1],
                "hybrid": hybrid, "base": base, "tiles": tiles_t if hybrid else world, "groups": world // tiles_t if hybrid else 1, "rank_spp": int(plan0.samples) if hybrid else spp,
                "steps_per_launch": group if hybrid else chain}

    def timed_batches(mode, t_depth, t_spp, t_stride, steps, per_launch, warm_launches=1):
        """`steps` batches of this frame at another (depth, spp, record size), `per_launch` per launch, one GPU: mode "chain" = rtowSampleBatchChainDevice accumulating
        in place (bit-identical to the batches one after the other); mode "group_fold" = rtowSampleBatchGroupDevice from zeroed inputs into per-batch partial sums,
        then the partial sums added to the accumulators in batch order (rtowAddAccumDevice): the same samples, another association of the float sums.
        Wall time between synchronisations, after `warm_launches` untimed launches; returns a summary dict."""
        acc = mg.accum_views(torch.zeros(mg.ACCUM_FLOATS * n, device=dev), n)
        zero = mg.accum_views(torch.zeros(mg.ACCUM_FLOATS * n, device=dev), n)
        parts = [mg.accum_views(torch.zeros(mg.ACCUM_FLOATS * n, device=dev), n) for _ in range(per_launch)] if mode == "group_fold" else None
        dg = [torch.zeros(n * (t_stride // 4), device=dev) for _ in range(per_launch)]
        basep = rt.scenes.make_params(scene, W, H, spp=t_spp, trace_depth=t_depth, diagnostics_stride=t_stride, focus=focus)
        ba = abi.AccumBuffers(*[t.data_ptr() for t in acc])
        bz = abi.AccumBuffers(*[t.data_ptr() for t in zero])
        kms = []

        def go(first, count):
            i = first
            while i < first + count:
                c = min(per_launch, first + count - i)
                arr = (abi.SampleParams * c)()
                for k in range(c):
                    arr[k] = abi.SampleParams.from_buffer_copy(basep)
                    arr[k].seed = i + 1 + k
                dptr = (C.c_void_p * c)(*[dg[k].data_ptr() for k in range(c)])
                if mode == "chain":
                    rt.lib.check(lib.rtowSampleBatchChainDevice(ctx.handle, c, arr, C.byref(ba), C.byref(ba), dptr, stream.cuda_stream, None), "rtowSampleBatchChainDevice")
                else:
                    bo = (abi.AccumBuffers * c)(*[abi.AccumBuffers(*[t.data_ptr() for t in parts[k]]) for k in range(c)])
                    rt.lib.check(lib.rtowSampleBatchGroupDevice(ctx.handle, c, arr, C.byref(bz), bo, dptr, stream.cuda_stream, None), "rtowSampleBatchGroupDevice")
                    for k in range(c):
                        rt.lib.check(lib.rtowAddAccumDevice(ctx.handle, n, C.byref(ba), C.byref(bo[k]), stream.cuda_stream), "rtowAddAccumDevice")
                kms.append(ctx.last_sample_kernel_ms())
                i += c

        go(0, warm_launches * per_launch)
        kms.clear()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        go(warm_launches * per_launch, steps)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        ctx.batch_status()
        rays_last = float(dg[0].view(n, t_stride // 4)[:, 0].sum().item())
        return {"value": round(float(n) * t_spp * steps / dt / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt / steps * 1e3, 3), "kernel_ms_per_step": round(sum(kms) / steps, 3),
                "mrays_per_s": round(rays_last * steps / dt / 1e6, 1), "steps": steps, "batches_per_launch": per_launch}

    def adaptive_batches(steps, warm=4, t_depth=32, t_range=(1, 50), t_stride=16):
        """The reference host AS COMMITTED (Assets/Prefabs/Raytracer.prefab:383-391: samplesPerBatchRange {1, 50}, traceDepth 32; FULL_DIAGNOSTICS records): every batch takes
        between 1 and 50 samples PER PIXEL, decided per pixel from its accumulated sample-count weight against the extrema of the frame (JOBS/SampleBatchJob.cs:118-126), and
        the extrema come from the ReduceMetricsJob of the batch that completed last (UNITY/Raytracer.cs:527-543) - with two batches in flight (:586-596), of batch i - 2.
        Plain rtowSampleBatchDevice launches accumulating in place, each followed by rtowReduceMetricsDeviceAsync into a pinned record; the host waits for record i - 2
        before it enqueues batch i, exactly the dependency the reference host has.  Samples are counted like the reference counts them (TotalSamples = sum of color.w)."""
        acc = mg.accum_views(torch.zeros(mg.ACCUM_FLOATS * n, device=dev), n)
        ba = abi.AccumBuffers(*[t.data_ptr() for t in acc])
        dg = torch.zeros(n * (t_stride // 4), device=dev)
        records = torch.zeros((steps + warm, 16), dtype=torch.int32).pin_memory()          # one 64-byte slot per batch, RtowMetrics (40 bytes) at its start
        events = [torch.cuda.Event() for _ in range(steps + warm)]
        basep = rt.scenes.make_params(scene, W, H, spp=t_range[0], spp_max=t_range[1], trace_depth=t_depth, diagnostics_stride=t_stride, focus=focus)
        extrema = (0.0, 0.0)                                                                # the host's field before any batch has completed
        kms, spans = [], []

        def record_of(i):
            return abi.Metrics.from_buffer_copy(records[i].numpy().tobytes()[:C.sizeof(abi.Metrics)])

        def go(first, count):
            nonlocal extrema
            for i in range(first, first + count):
                if i >= 2:
                    events[i - 2].synchronize()                                             # the batch before the one in flight has completed: its metrics are the host's
                    r = record_of(i - 2)
                    extrema = (float(r.sampleCountWeightExtrema.x), float(r.sampleCountWeightExtrema.y))
                p = abi.SampleParams.from_buffer_copy(basep)
                p.seed = i + 1
                p.sampleCountWeightExtrema = abi.Float2(*extrema)
                rt.lib.check(lib.rtowSampleBatchDevice(ctx.handle, C.byref(p), C.byref(ba), C.byref(ba), dg.data_ptr(), stream.cuda_stream, None), "rtowSampleBatchDevice")
                rt.lib.check(lib.rtowReduceMetricsDeviceAsync(ctx.handle, n, dg.data_ptr(), t_stride, acc[0].data_ptr(), acc[3].data_ptr(), stream.cuda_stream, records[i].data_ptr()),
                             "rtowReduceMetricsDeviceAsync")
                events[i].record(stream)

        go(0, warm)
        torch.cuda.synchronize(dev)
        before = record_of(warm - 1)
        t0 = time.perf_counter()
        go(warm, steps)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        ctx.batch_status()
        after = record_of(warm + steps - 1)
        samples = int(after.totalSamples64) - int(before.totalSamples64)
        rays = sum(int(record_of(i).totalRayCount64) for i in range(warm, warm + steps))
        per_batch = [int(record_of(i).totalSamples64) - int(record_of(i - 1).totalSamples64) for i in range(warm, warm + steps)]
        return {"value": round(samples / dt / 1e6, 2), "unit": "Msamples/s (successful samples, as the reference's TotalSamples counts them)", "ms_per_step": round(dt / steps * 1e3, 3),
                "mrays_per_s": round(rays / dt / 1e6, 1), "steps": steps, "batches_per_launch": 1, "samples_per_pixel_per_batch_mean": round(samples / steps / n, 2),
                "samples_per_pixel_per_batch_min_max": [round(min(per_batch) / n, 2), round(max(per_batch) / n, 2)],
                "sample_count_weight_extrema_last": [float(after.sampleCountWeightExtrema.x), float(after.sampleCountWeightExtrema.y)],
                "samples_per_pixel_extrema_last": [int(after.sampleCountExtrema[0]), int(after.sampleCountExtrema[1])]}

    def partition_self_check(partition):
        """N > 1: BEFORE anything is timed, the chosen partition renders a 64 x 36 frame over the transport the timed run will use, and rank 0 checks the colour frame it
        ends up with (a) bit for bit against the same sub-batches rendered by rank 0 ALONE and folded in the same order, and (b) within north_star's 1e-4 per channel of the mean of
        the reference's sequential accumulation of those sub-batches.  (a) proves that N ranks really took part and that the exchange / gather deliver every row;
        (b) that the image the scaling line is quoted for is the reference's image up to float association.  Fails the run, loudly, otherwise."""
        nonlocal W, H, n, spp
        import numpy as np
        keep = (W, H, n, spp)
        steps_c = 2
        W, H = 64, 36
        n = W * H
        hybrid_c = partition in ("hybrid", "batches")
        tiles_t = 1 if partition == "batches" else (args.tiles or mg.default_tiles(world, keep[3]))
        spp = 4 * (world // tiles_t) if hybrid_c else 4                   # every seed group gets 4 samples
        try:
            mm = measure(partition, "reference", 1, steps_c, 0)
            got = mm["last"][0].view(n, 4).cpu().numpy().copy() if rank == 0 else None
            result = None
            if rank == 0:
                def render(params, src):
                    bufs = [torch.zeros(n * c, device=dev) if src is None else src[k].clone() for k, c in enumerate((4, 3, 3, 1))]
                    b = abi.AccumBuffers(*[t.data_ptr() for t in bufs])
                    rt.lib.check(lib.rtowSampleBatchDevice(ctx.handle, C.byref(params), C.byref(b), C.byref(b), None, stream.cuda_stream, None), "rtowSampleBatchDevice")
                    torch.cuda.synchronize(dev)
                    return bufs
                fold = np.zeros((n, 4), np.float32)
                seq = None
                rows = np.arange(n) // W
                for step in range(1, steps_c + 1):
                    for r in range(world):
                        if hybrid_c:
                            pl = rt.Context.hybrid_plan(world, r, tiles_t, spp, step)
                            p = rt.scenes.make_params(scene, W, H, spp=int(pl.samples), trace_depth=depth, seed=int(pl.seed), slice_offset=int(pl.sliceOffset), slice_divider=int(pl.sliceDivider), focus=focus)
                            own = rows % int(pl.sliceDivider) == int(pl.sliceOffset)
                        else:
                            p = rt.scenes.make_params(scene, W, H, spp=spp, trace_depth=depth, seed=step, slice_offset=r, slice_divider=world, focus=focus)
                            own = rows % world == r
                        part = render(p, None)[0].view(n, 4).cpu().numpy()
                        fold[own] = fold[own] + part[own]                   # group order (ranks of a tile ascend with their group), float32 like fold_rows_kernel
                        seq = render(p, seq)                                # the reference's way: every batch on top of its predecessor's sums
                seqc = seq[0].view(n, 4).cpu().numpy()
                # hybrid: the fold of the sub-batches in group order; tiles: the reference's own sequential accumulation of the batches, which a tile partition reproduces bit for bit
                same = bool(np.array_equal(got.view(np.uint32), (fold if hybrid_c else seqc).view(np.uint32)))
                mean_a = got[:, :3] / np.maximum(got[:, 3:4], 1)
                mean_s = seqc[:, :3] / np.maximum(seqc[:, 3:4], 1)
                counts = bool(np.array_equal(got[:, 3], seqc[:, 3]))
                result = {"frame": "%dx%d, %d steps, %d samples per step" % (W, H, steps_c, spp), "bit_identical_to_the_same_sub_batches_on_one_gpu": same,
                          "compared_with": "the same sub-batches rendered by rank 0 alone from zeroed accumulators and folded in group order" if hybrid_c else "the same batches accumulated one after the other by rank 0 alone (the single-GPU frame)",
                          "success_counts_equal_the_sequential_accumulation": counts, "max_abs_mean_colour_difference_to_the_sequential_accumulation": float(np.abs(mean_a - mean_s).max()),
                          "transport": "RCCL behind the C ABI" if (have_comm and not shared_gpu) else "stand-in transport (debug)" if have_comm else "torch.distributed"}
                if not (same and counts and result["max_abs_mean_colour_difference_to_the_sequential_accumulation"] <= 1e-4):
                    raise SystemExit("bench.py --gpus %d: the %s partition's frame is not the frame of its sub-batches: %s" % (world, partition, json.dumps(result)))
            return result
        finally:
            W, H, n, spp = keep
            ctx.synchronize()

    # N > 1: both partitions check themselves before anything is timed - the one `value` is quoted for, and the other, reported beside it as a first-class block
    self_checks = {}
    if world > 1:
        for part in ("tiles", "hybrid") if args.partition != "batches" else ("tiles", "batches"):
            self_checks[part] = partition_self_check(part)
    self_check = self_checks.get(args.partition)

    if args.only_leg:
        # profiling / A-B aid (profiles/collect.sh, profiles/r05_runs): ONE of the secondary measurements as the whole run
        per = args.chain
        leg = {"group_fold": lambda: timed_batches("group_fold", depth, spp, 4, args.steps, per),
               "host_default_chain": lambda: timed_batches("chain", 32, 50, 16, args.steps, per),
               "host_default_group": lambda: timed_batches("group_fold", 32, 50, 16, args.steps, per),
               "host_default_adaptive": lambda: adaptive_batches(args.steps),
               "plain_two_in_flight": lambda: adaptive_batches(args.steps, warm=2, t_depth=depth, t_range=(spp, spp), t_stride=4)}[args.only_leg]()
        tuned = ctx.scene_info()
        print(json.dumps({"only_leg": args.only_leg, "scene": args.scene, "width": W, "height": H, "config": {"batches_per_launch": leg.get("batches_per_launch"), "scheduler_tune": [int(x) for x in tuned.schedulerTune],
                                                                                                  "threshold_set": int(tuned.thresholdSet), "context_flags": args.context_flags}, **leg}), flush=True)
        ctx.close()
        return

    main_partition = args.partition if world > 1 else "single"
    m = measure(args.partition, args.rng, args.chain, args.steps, args.warmup, prewarm_s=args.prewarm)
    hybrid = m["hybrid"]

    def summary(mm):
        return {"value": round(float(n) * spp * args.steps / mm["elapsed"] / 1e6, 2), "ms_per_step": round(mm["elapsed"] / args.steps * 1e3, 3),
                "kernel_ms_per_step": round(mm["kernel_ms_per_step"], 3)}

    extras = {}
    host_ms = None
    if not args.no_extras:
        if world == 1:
            if args.chain > 1:
                extras["plain_batches"] = dict(summary(measure("tiles", args.rng, 1, args.steps, 1)), batches_per_launch=1, note="the same steps as one launch per batch (rtowSampleBatchDevice): like for like with round 1's `value`")
            if args.chain != 2:
                steps2 = args.steps + (args.steps & 1)
                m2 = measure("tiles", args.rng, 2, steps2, 2)
                extras["chain2"] = {"value": round(float(n) * spp * steps2 / m2["elapsed"] / 1e6, 2), "ms_per_step": round(m2["elapsed"] / steps2 * 1e3, 3), "kernel_ms_per_step": round(m2["kernel_ms_per_step"], 3),
                                    "batches_per_launch": 2, "steps": steps2,
                                    "note": "two batches per launch (rtowSampleBatchChainDevice, count = 2): what a host with the reference's queue depth of two gets (UNITY/Raytracer.cs:586-593); INTEGRATION.md 3 shows the edit that queues more"}
            # the UNMODIFIED host's call sequence on the device API: per batch rtowSampleBatchDevice, then the metrics reduction that reads that batch's outputs
            # (ScheduleSample enqueues SampleBatchJob -> RecordTimeJob -> ReduceMetricsJob, UNITY/Raytracer.cs:729-754), asynchronously, two batches in flight (:586-596)
            extras["plain_two_in_flight"] = dict(adaptive_batches(args.steps, warm=2, t_depth=depth, t_range=(spp, spp), t_stride=4),
                                                 note="one launch per batch + rtowReduceMetricsDeviceAsync after each, the host two batches ahead of the device at most: what the reference host's own call order gets "
                                                      "from the device-resident API without the queue-depth edit.  Between any two of its batches a consumer of the first one's outputs is enqueued (the metrics the "
                                                      "NEXT batches' adaptive sample counts are decided from), so a library that deferred a launch to fuse it with the next call would have to flush it at once - or change "
                                                      "what that reduction sees (DESIGN.md 8)")
            if args.rng == "reference":
                # north_star's lane-per-pixel-sample shape as far as it is built: RTOW_RNG_PER_SAMPLE (one xorshift32 generator per SAMPLE, work units of 16 samples, partial sums folded in
                # group order) and its xoroshiro64** twin - another image by construction (not the reference's stream), bit-exact against the oracle's same policy; plain launches
                for key, pol in (("per_sample", "per-sample"), ("per_sample_xoroshiro", "per-sample-xoroshiro")):
                    mp = measure("tiles", pol, 1, max(4, args.steps // 2), 2)
                    extras[key] = {"value": round(float(n) * spp * max(4, args.steps // 2) / mp["elapsed"] / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(mp["elapsed"] / max(4, args.steps // 2) * 1e3, 3),
                                   "kernel_ms_per_step": round(mp["kernel_ms_per_step"], 3), "batches_per_launch": 1,
                                   "note": "NOT the reference stream: one generator per sample (rngPolicy %d), units of 16 samples, unit records folded by fold_unit_records; one launch + fold per batch" % (1 if pol == "per-sample" else 2)}
            if args.chain > 1:
                extras["group_fold"] = dict(timed_batches("group_fold", depth, spp, 4, args.steps, args.chain),
                                            note="the same steps as batch groups (rtowSampleBatchGroupDevice: every batch from zeroed inputs into its own partial sums, one launch per group) "
                                                 "+ the partial sums added in batch order (rtowAddAccumDevice): the reference's samples, bit-identical to that fold order, within 1e-4 of the mean "
                                                 "of the sequential accumulation `value` computes exactly; a launch ends with its slowest pixel-batch instead of a pixel's batches in a row")
                # the reference host as committed: FULL_DIAGNOSTICS records (ProjectSettings/ProjectSettings.asset:590), traceDepth 32, up to 50 samples per batch
                # (Assets/Prefabs/Raytracer.prefab:383-391), same scene and frame
                hd_steps = 2 * args.chain
                extras["host_default"] = {
                    "config": "16-byte FULL_DIAGNOSTICS records, traceDepth 32 (the reference host's committed defines and prefab), %s %dx%d; `chain` / `group_fold`: every pixel takes the 50 samples that are "
                              "the UPPER end of the prefab's samplesPerBatchRange {1, 50}; `adaptive`: the range as committed, per-pixel counts" % (args.scene, W, H),
                    "chain": timed_batches("chain", 32, 50, 16, hd_steps, args.chain),
                    "group_fold": timed_batches("group_fold", 32, 50, 16, hd_steps, args.chain),
                    "adaptive": dict(adaptive_batches(hd_steps), note="the schedule the committed host actually runs: sampleCountRange (1, 50) decided per pixel from the sample-count weights, extrema fed back from the "
                                                                      "metrics of batch i - 2 (rtowReduceMetricsDeviceAsync), plain launches with two in flight, traceDepth 32, FULL_DIAGNOSTICS records"),
                    "note": "a chain is bound by its slowest pixel's batches in a row (cover scene at depth 32: 6 429 sequential path segments per 256 samples of one pixel, ~13 us each; "
                            "profiles/r04g_ray_count_stats.txt), a group by the slowest pixel-batch: DESIGN.md 4.1"}
            # the drop-in form of INTEGRATION.md: rtowSampleBatch on the host's own (pinned, registered) accumulation arrays
            import numpy as np
            pool = [np.zeros((n, c), np.float32) for c in (4, 3, 3)] + [np.zeros(n, np.float32)]
            hdiag = np.zeros(n, np.float32)
            ctx.register_host_buffers(*pool, hdiag)
            hb = abi.AccumBuffers(*[a.ctypes.data for a in pool])
            times = []
            for i in range(4):
                p = abi.SampleParams.from_buffer_copy(m["base"])
                p.seed = 1000 + i
                t = time.perf_counter()
                rt.lib.check(lib.rtowSampleBatch(ctx.handle, C.byref(p), C.byref(hb), C.byref(hb), hdiag.ctypes.data, None), "rtowSampleBatch")
                times.append(time.perf_counter() - t)
            host_ms = round(min(times[1:]) * 1e3, 3)
            # and the host-buffer chain: the same arrays, `chain` batches per call (inputs travel once, the batches accumulate on the device)
            if args.chain > 1:
                arr = (abi.SampleParams * args.chain)()
                for k in range(args.chain):
                    arr[k] = abi.SampleParams.from_buffer_copy(m["base"])
                    arr[k].seed = 2000 + k
                hdiags = [np.zeros(n, np.float32) for _ in range(args.chain)]
                dptr = (C.c_void_p * args.chain)(*[d.ctypes.data for d in hdiags])
                ctimes = []
                for i in range(2):
                    t = time.perf_counter()
                    rt.lib.check(lib.rtowSampleBatchChain(ctx.handle, args.chain, arr, C.byref(hb), C.byref(hb), dptr, None), "rtowSampleBatchChain")
                    ctimes.append(time.perf_counter() - t)
                extras["host_buffer_chain_ms_per_step"] = round(min(ctimes) / args.chain * 1e3, 3)
            ctx.unregister_host_buffers()
        else:
            other = "hybrid" if args.partition == "tiles" else "tiles"
            measured = {args.partition: m, other: measure(other, args.rng, 1, args.steps, 1)}
            extras["partitions"] = {k: summary(v) for k, v in measured.items()}
            # north_star's partition and the one that scales, each as a block of its own: value, time per step, ranks RCCL connected, collectives per step, what the image is, self-check
            ranks_connected = world if (have_comm and not shared_gpu) else 0
            for k, v in measured.items():
                is_tiles = k == "tiles"
                extras[k] = dict(summary(v), unit="Msamples/s", n_gpus=world, rccl_ranks=ranks_connected, collectives_per_step=1 if is_tiles else 2,
                                 partition=("row-interleaved slices (SliceDivider = %d, JOBS/SampleBatchJob.cs:69-70), ONE gather of colour rows per sample batch (rtowGatherRowsDevice)" % world) if is_tiles else
                                           ("%d row slices x %d seed groups (rtowHybridPlan): grouped send / receive exchange + ordered fold (rtowExchangeAccumDevice), then the tile gather" % (v["tiles"], v["groups"])),
                                 image="the single-GPU frame, bit for bit" if is_tiles else "the reference's samples under another association of the float sums: bit-identical to the same sub-batches on one GPU, within 1e-4 of the sequential accumulation",
                                 batches_per_launch=v["steps_per_launch"], is_value=(k == args.partition), self_check=self_checks.get(k))
            if args.rng == "reference":
                extras["partitions"]["tiles, RTOW_RNG_PER_SAMPLE (not the reference stream)"] = summary(measure("tiles", "per-sample", 1, args.steps, 1))

    elapsed, avg_kernel_ms = m["elapsed"], m["kernel_ms_per_step"]
    if rank == 0:
        total_samples = float(n) * spp * args.steps
        ms_per_step = elapsed / args.steps * 1e3
        # metrics of the last batch (rays per sample, success ratio) - outside the timed region
        last = m["last"]
        rays = float(m["diag"].sum().item()) * (world if world > 1 else 1)  # every rank traces a statistically equal share
        # algorithmic HBM bytes per launch of the sample kernel: 44 B read + 44 B write + 4 B diagnostics per
        # owned pixel AND batch of the launch, plus the scene image once
        owned_pixels = n if world == 1 else len(range(rank % m["tiles"], H, m["tiles"])) * W     # rank 0's launch: its tile's rows
        rank_spp = m["rank_spp"]
        steps_per_launch = args.steps / max(m["launches"], 1)      # batches one launch of the sample kernel holds: a chain's (one GPU) or a batch group's (hybrid partition)
        alg_bytes = int(owned_pixels * 92 * steps_per_launch) + int(info.sceneBytesDevice)
        launch_ms = avg_kernel_ms * steps_per_launch
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        # which committed profile speaks for this workload: the headline command's, or the one collected for this config / scene (profiles/collect.sh <tag>_<workload> ...)
        wkey = ("c%d" % args.config) if not overridden else (args.scene if overridden == ["scene"] and args.config == 2 else None)
        traffic, traffic_src, secondary = measured_hbm_traffic(wkey) if (world == 1 and wkey) else (None, None, {})
        profiled_bpl = None
        if traffic is not None:
            try:
                profiled_bpl = int(json.load(open(os.path.join(ROOT, "profiles", traffic_src))).get("bench_line_under_profiler", {}).get("config", {}).get("batches_per_launch", 0)) or None
            except (OSError, ValueError):
                profiled_bpl = None
            traffic = round(traffic * steps_per_launch)      # per launch, like `achieved`: the committed profile's per-batch traffic x this run's batches per launch
        tuned = ctx.scene_info()
        out = {
            "metric": "Msamples/s, 486-sphere cover scene 1920x1080 8-bounce" if (args.config == 2 and not overridden) else
                      "Msamples/s, %s scene %dx%d %d-bounce" % (args.scene, W, H, depth),
            "value": round(total_samples / elapsed / 1e6, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_s": args.prewarm,
            "ms_per_step": round(ms_per_step, 3),
            "batches_per_launch": m["steps_per_launch"],   # `value` is measured with this many successive batches fused into one launch (1 = plain batches)
            "launches": m["launches"],
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s%s, %dx%d, %d spp per batch, "
                            "%d bounces, white noise, jitter on, %s" % ("BASELINE.json " + cfg["label"] + " = " if not overridden else "", SCENE_TEXT[args.scene], W, H, spp, depth, "reference RNG stream (lane per pixel)" if args.rng == "reference" else "RTOW_RNG_%s (NOT the reference stream; lane per 16-sample group)" % args.rng.upper().replace("-", "_")),
                "partition": ("DEBUG: %d ranks sharing one GPU over gloo - not a measurement; " % world if shared_gpu else "") + ("single GPU" if world == 1 else
                              "hybrid: %d row slices x %d seed groups (rtowHybridPlan): every rank renders its slice with %d of the %d samples and the Seed of its sub-batch from zeroed accumulators "
                              "(the sub-batches of up to %d steps as one launch, rtowSampleBatchGroupDevice); "
                              "one grouped ncclSend / ncclRecv exchange + rank-ordered fold of every row on its owner (rtowExchangeAccumDevice), one RCCL gather of colour rows per batch on rank 0 "
                              "(rtowGatherRowsDevice); the reference's successive batches (UNITY/Raytracer.cs:656-661,798-802) run concurrently" % (m["tiles"], m["groups"], rank_spp, spp, m["steps_per_launch"])
                              if hybrid else "tiles: row-interleaved slices (SliceDivider=%d), one RCCL gather of colour rows per batch behind the C ABI (rtowGatherRowsDevice)" % world),
                "tiles": m["tiles"] if world > 1 else None, "seed_groups": m["groups"] if world > 1 else None,
                "rccl_ranks": (world if (have_comm and not shared_gpu) else 0) if world > 1 else None,
                "collectives_per_step": None if world == 1 else (2 if hybrid else 1),
                "image": None if world == 1 else ("per step %d sub-batches of %d of the %d samples, Seeds (step - 1) * %d + 1 ... step * %d, each rendered from zeroed accumulators and folded in group order: the reference's "
                                                  "samples under another association of the float sums - bit-identical to the same sub-batches on one GPU, within 1e-4 of the sequential accumulation (self_check)"
                                                  % (m["groups"], rank_spp, spp, m["groups"], m["groups"]) if hybrid else "the single-GPU frame, bit for bit: every rank renders its rows of the one batch (Seed = step)"),
                "self_check": self_check,
                "gather": None if world == 1 else ("rtowGatherRowsDevice (DEBUG: the tests' stand-in transport instead of RCCL, ranks share one GPU)" if (have_comm and shared_gpu) else "rtowGatherRowsDevice (RCCL behind the C ABI)" if have_comm else "torch.distributed (debug: ranks share one GPU)" if shared_gpu else "torch.distributed (the C-ABI communicator was not available)"),
                "batches_per_launch": m["steps_per_launch"],
                "launches": m["launches"],
                "bvh_nodes": int(info.bvhNodeCount), "bvh_depth": int(info.bvhDepth), "scene_in_lds": bool(info.sceneInLds), "wide_codes": bool(info.wideCodes),
                "entities": int(info.entityCount), "hit_spill_bytes": int(info.hitSpillBytes),
                # stage thresholds in use: -1 = the kernel kind's built-in ones, 0..5 = the set the first (warm-up) batch measured as fastest for this scene
                "threshold_set": int(tuned.thresholdSet), "scheduler_tune": [int(x) for x in tuned.schedulerTune],
            },
            "kernel_ms_per_step": round(avg_kernel_ms, 3),
            "kernel_ms_per_launch": round(launch_ms, 3),
            "msamples_per_s_kernel_only": round(owned_pixels * rank_spp * world / (avg_kernel_ms * 1e-3) / 1e6, 2),
            "mrays_per_s": round(rays / (avg_kernel_ms * 1e-3) / 1e6, 1),
            "rays_per_sample": round(rays / (float(n) * spp), 4),
            "successful_sample_ratio": round(float(last[0][:, 3].sum().item()) / (float(n) * spp * (args.steps + args.warmup)), 4) if world == 1 else None,
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 8),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_batches_per_launch_profiled": profiled_bpl,
                "traffic_basis": None if traffic is None else ("measured: PMC passes of this command at this chain length" if profiled_bpl == args.chain else
                                                                "extrapolated: per-batch PMC traffic of a %s-batch launch x this run's %d batches per launch" % (profiled_bpl, args.chain)),
                "kernel": "sample_batch_kernel",
                "algorithmic_bytes_per_launch": alg_bytes,
                "secondary": secondary,   # what actually limits the kernel: from the same committed PMC summary as `traffic`
                "valu": None if "valu_frac_of_peak_lane_issue" not in secondary else {
                    "bound": "valu issue x lane utilisation", "issue": secondary["valu_issue_utilisation"], "lanes": secondary["valu_lane_utilisation"],
                    "frac": secondary["valu_frac_of_peak_lane_issue"], "source": traffic_src},
                "note": "graph-traversal path: algorithmic HBM traffic is 92 B/pixel per batch, so the HBM fraction is tiny by construction; "
                        "the kernel is VALU-issue / divergence bound (see DESIGN.md, profiles/)",
            },
        }
        out.update(extras)
        if host_ms is not None:
            out["host_buffer_ms_per_step"] = host_ms
            out["host_buffer_note"] = ("rtowSampleBatch on pinned host arrays registered with rtowRegisterHostBuffer: inputs by one DMA, outputs stored by the kernel straight into "
                                       "host memory (best of 3 after 1 warm-up); host_buffer_chain_ms_per_step: rtowSampleBatchChain on the same arrays, batches_per_launch batches per call")
        if world == 1 and not args.no_extras:
            out["post_passes"] = post_passes(rt, ctx, lib, torch, dev, stream)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rt, scene, W, H, depth, scene_name=args.scene, full_spp=spp)
            out["cpu_baseline_c1"] = cpu_baseline_c1(rt)
        print(json.dumps(out), flush=True)

    if have_comm:
        ctx.comm_destroy()
    ctx.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
