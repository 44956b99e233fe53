#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (raw rocprofv3 CSV) into the committed profiles/<tag>_* summary files.

  python profiles/summarize.py r01
writes profiles/<tag>_kernel_stats.csv   (the rocprofv3 --kernel-trace --stats kernel table, verbatim)
       profiles/<tag>_pmc_summary.json   (per-launch counter sums for sample_batch_kernel + derived figures)
"""
import csv
import glob
import json
import os
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_" + TAG)
KERNEL = "sample_batch_kernel"


def find(pattern):
    hits = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return hits[0] if hits else None


def main():
    stats = find("trace/**/*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(ROOT, "profiles", TAG + "_kernel_stats.csv"))
    summary = {"tag": TAG, "kernel": KERNEL, "counters_per_launch": {}}
    if stats:
        for row in csv.DictReader(open(stats)):
            if KERNEL in row["Name"]:
                summary["kernel_trace"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]), "total_ns": float(row["TotalDurationNs"]),
                                           "percentage": float(row["Percentage"])}
    trace = find("trace/**/*kernel_trace.csv")
    if trace:
        # The first rtowSampleBatch on a (scene, view) launches the same kernel once more with 1 sample per pixel (the cost probe
        # that orders the pixel chunks); rocprofv3's stats table averages it in.  Split the dispatches here.
        durs = []
        for row in csv.DictReader(open(trace)):
            if KERNEL in row["Kernel_Name"]:
                if "dispatch" not in summary:
                    summary["dispatch"] = {k: row[k] for k in ("Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in row}
                    if "VGPR_Count" in row:
                        # rocprofv3 prints HALF the allocation.  It decodes the kernel descriptor's granulated VGPR count with a granule of 4 registers; gfx90a and
                        # later (gfx950 included) allocate wave64 VGPRs in granules of 8.  Checked against -Rpass-analysis=kernel-resource-usage on kernels of known size: the rank-rule
                        # triangle kernel (116 VGPRs -> 120 allocated) is reported as 60, the 128-VGPR kernels as 64.  Occupancy follows from the allocation: 512 / 128 = 4 waves per SIMD.
                        summary["dispatch"]["VGPRs_allocated"] = 2 * int(row["VGPR_Count"])
                        summary["dispatch"]["VGPR_note"] = "VGPR_Count is rocprofv3's figure (descriptor granule taken as 4); gfx950 allocates in granules of 8: allocated = 2 x VGPR_Count"
                durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if durs:
            longest = max(durs)
            batches = [d for d in durs if d > 0.2 * longest]
            probes = [d for d in durs if d <= 0.2 * longest]
            summary["kernel_trace_split"] = {"batch_launches": len(batches), "batch_avg_ns": sum(batches) / len(batches), "batch_min_ns": min(batches), "batch_max_ns": max(batches),
                                             "probe_launches": len(probes), "probe_avg_ns": (sum(probes) / len(probes)) if probes else None,
                                             "note": "batch_* = the timed 256-spp launches (compare with bench.py kernel_ms_per_step); probe = the short launches of the same kernel: the 1-spp cost probe, "
                                                     "the ten 4-spp threshold probes, and the tie fix-up launches (exact-tie variant, 8 workgroups, leave at once)"}
    for f in glob.glob(os.path.join(SRC, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        # counters of the LAST dispatch of the kernel in each pass = the 256-spp batch (the probe launch precedes it)
        rows = [row for row in csv.DictReader(open(f)) if KERNEL in row["Kernel_Name"]]
        if not rows:
            continue
        # (round 4: every launch of a watched scene is followed by the tie fix-up launch - the exact-tie variant of the same kernel on 8 workgroups, which leaves at once when
        # nothing is listed; the dispatch to read is the last one on the FULL grid)
        if "Grid_Size" in rows[0]:
            full = max(int(row["Grid_Size"]) for row in rows)
            rows = [row for row in rows if int(row["Grid_Size"]) == full]
        last = max(int(row["Dispatch_Id"]) for row in rows)
        sums = {}
        for row in rows:
            if int(row["Dispatch_Id"]) == last:
                sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        summary["counters_per_launch"].update(sums)
    c = summary["counters_per_launch"]
    d = {}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced stream
        # (MI355X_MICROARCH.md "HBM"), so the read side is doubled as that guide prescribes (upper bound for narrow reads).
        d["hbm_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        d["hbm_traffic_bytes"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["valu_lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
    if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        cycles = c["GRBM_GUI_ACTIVE"] / 8  # 8 XCDs
        d["gpu_cycles_per_launch"] = cycles
        # Units pinned with pure instruction streams (profiles/calib/valu_calib.hip, r02_valu_calibration.json): SQ_INSTS_VALU and
        # SQ_ACTIVE_INST_VALU both count INSTRUCTIONS (the latter 2 per transcendental), not busy cycles.  A wave64 instruction occupies its
        # SIMD for 2 cycles if it is full rate (mul / add / fma / logic / mov), ~2.8 if half rate (min / max / compare / select / shift /
        # convert, 3-operand encodings, packed fp32) and ~5.3 if transcendental.  So:
        #   valu_issue_utilisation      = INSTS x 2 / SIMD-cycles: the fraction of issue slots used IF every instruction were full rate (a floor)
        #   simd_cycles_per_valu_inst   = SIMD-cycles / INSTS: compare with ~2.7, what this kernel's instruction mix costs at full issue
        #   valu_pipe_busy_estimate     = 2.7 / simd_cycles_per_valu_inst
        d["valu_issue_utilisation"] = c["SQ_INSTS_VALU"] * 2 / (cycles * 1024)  # wave64 on SIMD32: 2 cycles/instr, 1024 SIMDs
        d["simd_cycles_per_valu_inst"] = cycles * 1024 / c["SQ_INSTS_VALU"]
        d["valu_pipe_busy_estimate"] = min(1.0, 2.7 / d["simd_cycles_per_valu_inst"])
    if "SQ_LDS_IDX_ACTIVE" in c and "GRBM_GUI_ACTIVE" in c:
        d["lds_busy_fraction"] = c["SQ_LDS_IDX_ACTIVE"] / (c["GRBM_GUI_ACTIVE"] / 8 * 256)
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c:
        d["lds_bank_conflict_fraction"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    # what the wave cycles that wait for an instruction wait for (collect.sh passes sq4 / sqc / sq5)
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
            if k in c:
                d[k.lower() + "_per_wave_cycle"] = c[k] / wc
    if "SQC_ICACHE_REQ" in c and c["SQC_ICACHE_REQ"] > 0:
        d["icache_hit_rate"] = c.get("SQC_ICACHE_HITS", 0.0) / c["SQC_ICACHE_REQ"]
        d["icache_misses_per_launch"] = c.get("SQC_ICACHE_MISSES", 0.0)
    if "SQC_DCACHE_REQ" in c and c["SQC_DCACHE_REQ"] > 0:
        d["scalar_cache_hit_rate"] = c.get("SQC_DCACHE_HITS", 0.0) / c["SQC_DCACHE_REQ"]
    if "SQ_IFETCH" in c and "SQ_IFETCH_LEVEL" in c and c["SQ_IFETCH"] > 0:
        d["ifetch_level_per_fetch_uncalibrated"] = c["SQ_IFETCH_LEVEL"] / c["SQ_IFETCH"]      # (the LEVEL counters are sampled occupancies: ratios compare runs, they are not cycles)
        if "SQ_INSTS_VALU" in c:
            d["ifetches_per_valu_inst"] = c["SQ_IFETCH"] / c["SQ_INSTS_VALU"]
    for lvl, cnt in (("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS"), ("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM"), ("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM")):
        if lvl in c and cnt in c and c[cnt] > 0:
            d[cnt.lower().replace("sq_insts_", "") + "_level_per_instruction_uncalibrated"] = c[lvl] / c[cnt]
    if "SQ_INSTS_SALU" in c and "SQ_INSTS_VALU" in c:
        d["salu_per_valu_inst"] = c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"]
        d["lds_per_valu_inst"] = c.get("SQ_INSTS_LDS", 0.0) / c["SQ_INSTS_VALU"]
    if "SQ_INSTS_BRANCH" in c and "SQ_INSTS_VALU" in c:
        d["branches_per_valu_inst"] = c["SQ_INSTS_BRANCH"] / c["SQ_INSTS_VALU"]
    # L2 side (collect.sh with L2=1): requests, hit rate, and the reads that went out to memory (32 B / 64 B / 128 B requests)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        d["l2_requests_per_launch"] = c.get("TCC_REQ_sum", c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in c:
        r32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        d["l2_memory_read_bytes_estimate"] = r32 * 32 + (c["TCC_EA0_RDREQ_sum"] - r32) * 64
    if "TCP_TCC_READ_REQ_sum" in c:
        d["l1_to_l2_read_requests_per_launch"] = c["TCP_TCC_READ_REQ_sum"]
    summary["derived"] = d
    bl = os.path.join(SRC, "bench_line.json")
    if os.path.exists(bl) and os.path.getsize(bl):
        summary["bench_line_under_profiler"] = json.loads(open(bl).read())
    json.dump(summary, open(os.path.join(ROOT, "profiles", TAG + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(summary["derived"], indent=1))
    post_passes()


# algorithmic bytes per pixel of the post passes: combine 44 read + 36 written; finalize 36 + 12; reduce_metrics 24 read;
# add_kernel runs once per accumulation buffer (4 / 3 / 3 / 1 floats per pixel, 8 B read + 4 B written per float): 132 B per pixel over its four launches
POST = {"combine_kernel": 80, "finalize_kernel": 48, "reduce_metrics_kernel": 24, "add_kernel": 132}


def post_passes():
    """profiles/<tag>_post_passes.json: rocprofv3 kernel-trace time of the post-pass kernels at 1080p and 4K -> achieved GB/s against the HBM roofline
    (8 TB/s spec, ~6.3 TB/s achievable), next to what bench.py measured with HIP events in the same process."""
    out = {"peak_GBps": 8000.0, "achievable_GBps": 6300.0}
    for size in ("1920x1080", "3840x2160"):
        stats = find("post_%s/**/*kernel_stats.csv" % size)
        if not stats:
            continue
        shutil.copy(stats, os.path.join(ROOT, "profiles", "%s_post_%s_kernel_stats.csv" % (TAG, size)))
        w, h = (int(x) for x in size.split("x"))
        res = {}
        for row in csv.DictReader(open(stats)):
            for kernel, bytes_per_px in POST.items():
                if kernel in row["Name"]:
                    calls, total_ns = int(row["Calls"]), float(row["TotalDurationNs"])
                    per_pass_ns = total_ns / calls * (4 if kernel == "add_kernel" else 1)      # one rtowAddAccumDevice = four add_kernel launches
                    gbs = w * h * bytes_per_px / per_pass_ns
                    e = res.setdefault(kernel, {"calls": 0, "GBps": 0.0})
                    if calls > e["calls"]:                                                      # (copy_rows_kernel<float4> / <float>: take the row with the most calls)
                        res[kernel] = {"calls": calls, "avg_ns_per_launch": round(total_ns / calls, 1), "ns_per_pass": round(per_pass_ns, 1), "bytes_per_pixel": bytes_per_px,
                                       "GBps": round(gbs, 1), "frac_of_peak": round(gbs / 8000.0, 4), "frac_of_achievable": round(gbs / 6300.0, 4)}
        log = os.path.join(SRC, "post_%s.log" % size)
        if os.path.exists(log):
            lines = [l for l in open(log) if l.startswith("{")]
            if lines:
                res["bench_hip_events_same_run"] = json.loads(lines[-1])["post_passes"].get(size)
        out[size] = res
    if len(out) > 2:
        json.dump(out, open(os.path.join(ROOT, "profiles", TAG + "_post_passes.json"), "w"), indent=1, sort_keys=True)
        print(json.dumps({k: {kk: vv.get("GBps") for kk, vv in v.items() if isinstance(vv, dict) and "GBps" in vv} for k, v in out.items() if isinstance(v, dict)}, indent=1))


if __name__ == "__main__":
    main()
