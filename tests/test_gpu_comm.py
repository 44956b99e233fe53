"""GPU: the multi-GPU boundary behind the C ABI (rtowComm*, rtowGatherRowsDevice, include/rtow.h).

One process per GPU, the frame row-interleaved with the reference's slice contract (JOBS/SampleBatchJob.cs:69-70), one gather of the owned
rows to the root per batch.  The box these tests run on has ONE GPU: the two-rank test puts both ranks on it (two processes, two contexts,
one RCCL communicator) - which RCCL 2.27 refuses, so there the test skips with that reason; on a box with two or more GPUs each rank takes
its own and the test runs.  The collective, the packing and the assembly are the same code an 8-GPU node runs."""
import ctypes as C
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gather_rows_single_rank_copies_owned_rows(rt, gpu_context):
    """world = 1 (no communicator): the root owns every row; frame != mine gets a row-by-row device copy, frame == mine is a no-op."""
    a = rt.abi
    ctx = gpu_context
    w, h = 37, 11
    n = w * h
    rng = np.random.default_rng(2)
    src = [rng.random((n, c)).astype(np.float32) for c in (4, 3, 3, 1)]
    mine = [rt.DeviceBuffer(ctx).upload(x) for x in src]
    frame = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
    bm = a.AccumBuffers(*[b.ptr for b in mine])
    bf = a.AccumBuffers(*[b.ptr for b in frame])
    ctx.gather_rows(w, h, 1, bm, bf, what=a.GATHER_COLOR | a.GATHER_ALBEDO)
    ctx.synchronize()
    got = [b.download(np.float32, (n, c)) for b, c in zip(frame, (4, 3, 3, 1))]
    assert np.array_equal(got[0], src[0]) and np.array_equal(got[2], src[2])
    assert not got[1].any() and not got[3].any()                     # buffers outside the mask are not touched
    ctx.gather_rows(w, h, 1, bm, bm, what=a.GATHER_ALL)               # in place: nothing to do
    lib = rt.lib.load()
    assert lib.rtowGatherRowsDevice(ctx.handle, w, h, 2, C.byref(bm), C.byref(bf), a.GATHER_COLOR, 0, None) == a.RTOW_ERROR_INVALID_VALUE   # divider != world size
    assert lib.rtowGatherRowsDevice(ctx.handle, w, h, 1, C.byref(bm), C.byref(bf), 0, 0, None) == a.RTOW_ERROR_INVALID_VALUE
    for b in mine + frame:
        b.free()


RANK_SCRIPT = r'''
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, idfile, outfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
rt = importlib.import_module("raytracing-in-one-weekend_amd")
a = rt.abi
import ctypes
count = ctypes.c_int(0)
ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(count))
device = rank if count.value > rank else 0           # one GPU per rank where the box has them; else every rank on the one GPU
ctx = rt.Context(device, log=lambda lvl, tag, msg, ud: print("[rank %d] %s: %s" % (rank, tag.decode(), msg.decode()), flush=True), log_level=4)
if rank == 0:
    uid = rt.Context.comm_unique_id()
    open(idfile + ".tmp", "wb").write(uid)
    os.rename(idfile + ".tmp", idfile)                # the host's own channel for the 128 bytes: here a file
else:
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
try:
    ctx.comm_init(uid, rank, world)
except rt.lib.RtowError as e:
    print("[rank %d] comm_init failed: %s" % (rank, e), flush=True)
    sys.exit(3)
scene = rt.scenes.cover_scene()
ctx.upload_scene(scene.desc())
w, h, spp = 96, 54, 4
n = w * h
p = rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=6, seed=9, slice_offset=rank, slice_divider=world)
bufs = [rt.DeviceBuffer(ctx, n * c * 4).zero() for c in (4, 3, 3, 1)]
acc = a.AccumBuffers(*[b.ptr for b in bufs])
for batch in range(2):                                # two batches: accumulate in place, gather after each like a frame loop would
    p.seed = 9 + batch
    job = rt.SampleBatchJob(ctx, p)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = bufs
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = bufs
    rt.lib.check(job.Schedule().Complete(), "rtowSampleBatchDevice")
    ctx.gather_rows(w, h, world, acc, acc if rank == 0 else None, what=a.GATHER_ALL, root=0)
ctx.synchronize()
if rank == 0:
    np.savez(outfile, **{k: b.download(np.float32, (n, c)) for k, b, c in zip(("color", "normal", "albedo", "scw"), bufs, (4, 3, 3, 1))})
ctx.comm_destroy()
ctx.close()
'''


def test_two_ranks_gather_the_single_gpu_frame(rt, gpu_context):
    """Two processes, SliceDivider = 2: after rtowGatherRowsDevice the root holds, bit for bit, the frame one process renders alone."""
    world = 2
    count = C.c_int(0)
    C.CDLL("libamdhip64.so").hipGetDeviceCount(C.byref(count))
    if count.value < world:
        # RCCL 2.27 refuses a communicator with two ranks on one device ("duplicate GPU") - and on some boxes the attempt does not return at
        # all (one 240 s hang in round 2) - so a one-GPU box does not try; the 8-GPU node runs this path through bench.py --gpus N
        pytest.skip("the gather needs one GPU per rank: %d device(s) here" % count.value)
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "rank.py")
        open(script, "w").write(RANK_SCRIPT)
        idfile, outfile = os.path.join(tmp, "uid"), os.path.join(tmp, "frame.npz")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
        procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), str(world), idfile, outfile if r == 0 else os.path.join(tmp, "r%d.txt" % r)],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        outs = []
        for pr in procs:
            try:
                outs.append(pr.communicate(timeout=240)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail("rank processes hung: " + "\n".join(outs))
        if any(pr.returncode == 3 for pr in procs) and any("uplicate GPU" in o or "invalid usage" in o for o in outs):
            pytest.skip("this RCCL build refuses two ranks on one device; the gather needs one GPU per rank: " + outs[0][-300:])
        assert all(pr.returncode == 0 for pr in procs), "\n".join(outs)
        got = np.load(outfile)
        scene = rt.scenes.cover_scene()
        gpu_context.upload_scene(scene.desc())
        w, h, spp = 96, 54, 4
        acc = None
        for batch in range(2):
            p = rt.scenes.make_params(scene, w, h, spp=spp, trace_depth=6, seed=9 + batch)
            acc = rt.sample_batch_host(gpu_context, p, inputs=None if acc is None else {k: acc[k] for k in ("color", "normal", "albedo", "scw")})
        for k in ("color", "normal", "albedo", "scw"):
            assert np.array_equal(got[k].reshape(-1).view(np.uint32), acc[k].reshape(-1).view(np.uint32)), k
