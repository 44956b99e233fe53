"""Host-side mirror of the reference's job structs for this path, above the C ABI.

The reference's host is C# (Unity); no C# toolchain exists in this image, so the mirror is Python (ctypes) and keeps
the reference's names and argument meaning so that tests read like the reference's call site
(Assets/Scripts/Unity/Raytracer.cs:674-738):

    job = SampleBatchJob(context, Size=..., View=..., Seed=..., SampleCountRange=..., TraceDepth=..., ...)
    job.InputColor = ...; job.OutputColor = ...
    job.Schedule().Complete()

`Context` owns the device state (scene, staging); `DeviceBuffer` mirrors CudaBuffer
(Assets/ThirdParty/nVidia OptiX Denoiser/OptixApi.cs:226-251).
"""
import ctypes as C

import numpy as np

from . import abi
from .lib import check, load


class DeviceBuffer:
    """Device allocation with Allocate / Copy / EnsureCapacity semantics of CudaBuffer (OptixApi.cs:226-251)."""

    def __init__(self, context, nbytes=0):
        self.context = context
        self.handle = C.c_void_p(0)
        self.nbytes = 0
        if nbytes:
            self.ensure_capacity(nbytes)

    def ensure_capacity(self, nbytes):
        if nbytes <= self.nbytes and self.handle:
            return
        self.free()
        h = C.c_void_p()
        check(load().rtowDeviceAlloc(self.context.handle, nbytes, C.byref(h)), "rtowDeviceAlloc")
        self.handle, self.nbytes = h, nbytes

    def upload(self, array):
        a = np.ascontiguousarray(array)
        self.ensure_capacity(a.nbytes)
        check(load().rtowDeviceCopy(self.context.handle, a.ctypes.data, self.handle, a.nbytes, abi.MEMCPY_HOST_TO_DEVICE), "rtowDeviceCopy")
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        check(load().rtowDeviceCopy(self.context.handle, self.handle, out.ctypes.data, out.nbytes, abi.MEMCPY_DEVICE_TO_HOST), "rtowDeviceCopy")
        return out

    def zero(self):
        check(load().rtowDeviceMemset(self.context.handle, self.handle, 0, self.nbytes), "rtowDeviceMemset")
        return self

    def free(self):
        if self.handle:
            load().rtowDeviceFree(self.context.handle, self.handle)
        self.handle, self.nbytes = C.c_void_p(0), 0

    @property
    def ptr(self):
        return self.handle.value


class Context:
    """RtowContext: one per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device_ordinal=0, log=None, log_level=0, flags=0, lds_scene_budget=0, scheduler_tune=None, hit_list_capacity=0, slice_block_threads=0):
        """flags: abi.CONTEXT_* (RtowContextFlags); lds_scene_budget / scheduler_tune: the development knobs of RtowContextOptions;
        hit_list_capacity: most surfaces one ray may meet where the whole hit list is kept (0 = 1024)."""
        self._cb = abi.LogCallback(log) if log else abi.LogCallback()
        opts = abi.ContextOptions(device_ordinal, self._cb, None, log_level, flags, lds_scene_budget)
        if scheduler_tune is not None:
            if len(scheduler_tune) != 9:
                raise ValueError("scheduler_tune takes 8 stage thresholds + the box-walk slice")
            for i, v in enumerate(scheduler_tune):
                opts.schedulerTune[i] = int(v)
        opts.hitListCapacity = int(hit_list_capacity)
        opts.sliceBlockThreads = int(slice_block_threads)      # reserved: 0 (or 1024); the 512- / 256-lane workgroups of round 3 were removed
        self.handle = C.c_void_p()
        check(load().rtowCreateContext(C.byref(opts), C.byref(self.handle)), "rtowCreateContext")
        self._scene_keepalive = None
        self._registered = []

    def register_host_buffers(self, *arrays):
        """Pin the host's long-lived accumulation pools (UNITY/Raytracer.cs:279-288) so that rtowSampleBatch runs without staging copies
        on the output side; the arrays must stay alive until unregister_host_buffers() / close()."""
        for a in arrays:
            check(load().rtowRegisterHostBuffer(self.handle, a.ctypes.data, a.nbytes), "rtowRegisterHostBuffer")
            self._registered.append(a)

    def unregister_host_buffers(self):
        for a in self._registered:
            load().rtowUnregisterHostBuffer(self.handle, a.ctypes.data)
        self._registered = []

    def batch_status(self):
        """Status of the batches enqueued since the last report (waits for the most recent one, wherever it was enqueued)."""
        check(load().rtowGetBatchStatus(self.handle), "rtowGetBatchStatus")

    # ---- multi-GPU: one process per GPU, rows gathered over RCCL behind the C ABI ----
    @staticmethod
    def comm_set_library_path(path):
        """rtowCommSetLibraryPath: which RCCL build rtowComm* loads (None = the default search); before the first rtowComm* call of the process."""
        check(load().rtowCommSetLibraryPath(path.encode() if path else None), "rtowCommSetLibraryPath")

    @staticmethod
    def comm_unique_id():
        cid = abi.CommId()
        check(load().rtowCommGetUniqueId(C.byref(cid)), "rtowCommGetUniqueId")
        return C.string_at(C.addressof(cid), 128)       # all 128 bytes (a c_char array's .value would stop at the first NUL)

    def comm_init(self, id_bytes, rank, world):
        cid = abi.CommId()
        C.memmove(C.addressof(cid), id_bytes, 128)
        check(load().rtowCommInit(self.handle, C.byref(cid), rank, world), "rtowCommInit")

    def comm_destroy(self):
        check(load().rtowCommDestroy(self.handle), "rtowCommDestroy")

    def gather_rows(self, width, height, divider, mine, frame, what=abi.GATHER_COLOR, root=0, stream=None):
        """mine / frame: abi.AccumBuffers of device pointers (frame may be None on non-root ranks)."""
        check(load().rtowGatherRowsDevice(self.handle, width, height, divider, C.byref(mine), C.byref(frame) if frame is not None else None, what, root, stream),
              "rtowGatherRowsDevice")

    @staticmethod
    def hybrid_plan(world, rank, tiles, samples_per_batch, step):
        """rtowHybridPlan: this rank's slice fields, sample share and Seed in a tiles x batches partition (G = T x B)."""
        plan = abi.HybridPlan()
        check(load().rtowHybridPlan(world, rank, tiles, samples_per_batch, step, C.byref(plan)), "rtowHybridPlan")
        return plan

    def exchange_accum(self, width, height, tiles, partial, accum, what=abi.GATHER_ALL, stream=None):
        """rtowExchangeAccumDevice: partial sums of this rank's tile -> folded into `accum` on the rank that owns each row (row % G)."""
        check(load().rtowExchangeAccumDevice(self.handle, width, height, tiles, C.byref(partial), C.byref(accum), what, stream), "rtowExchangeAccumDevice")

    def close(self):
        if self.handle:
            load().rtowDestroyContext(self.handle)   # drops the registrations (hipHostUnregister) while the arrays are still alive ...
            self._registered = []                    # ... and only then may they go
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, scene_desc):
        """World rebuild hand-off (the tail of Raytracer.RebuildWorld, UNITY/Raytracer.cs:1167-1183)."""
        check(load().rtowUploadScene(self.handle, C.byref(scene_desc)), "rtowUploadScene")

    def upload_sky_cubemap(self, cubemap_desc):
        """`new Cubemap(skyCubemap)` (RT/Texture.cs:150-169): the six faces travel once; None drops the cubemap."""
        check(load().rtowUploadSkyCubemap(self.handle, C.byref(cubemap_desc) if cubemap_desc is not None else None), "rtowUploadSkyCubemap")

    def upload_blue_noise(self, desc):
        """BlueNoiseData's textures (UNITY/BlueNoiseData.cs): half4 texels, `textureCount` square textures; None drops the set."""
        check(load().rtowUploadBlueNoise(self.handle, C.byref(desc) if desc is not None else None), "rtowUploadBlueNoise")

    def upload_stb_noise(self, desc):
        """SpatioTemporalBlueNoiseData's five texture sets (UNITY/SpatioTemporalBlueNoiseData.cs); None drops them."""
        check(load().rtowUploadStbNoise(self.handle, C.byref(desc) if desc is not None else None), "rtowUploadStbNoise")

    def scene_info(self):
        info = abi.SceneInfo()
        check(load().rtowGetSceneInfo(self.handle, C.byref(info)), "rtowGetSceneInfo")
        return info

    def hit_world(self, origin, direction, time=0.0):
        """Raytracer.HitWorld (UNITY/Raytracer.cs:1353; the auto-focus probe of ScheduleSample, :608-609) through rtowProbeNearestHit:
        (hit, distance, entity index) - `if hit: focusDistance = distance`."""
        d, e = C.c_float(), C.c_int32()
        o3, d3 = abi.Float3(*[float(x) for x in origin]), abi.Float3(*[float(x) for x in direction])
        check(load().rtowProbeNearestHit(self.handle, C.byref(o3), C.byref(d3), float(time), C.byref(d), C.byref(e)), "rtowProbeNearestHit")
        return e.value >= 0, d.value, e.value

    def synchronize(self):
        check(load().rtowSynchronize(self.handle), "rtowSynchronize")

    def last_sample_kernel_ms(self):
        ms = C.c_float()
        check(load().rtowGetLastSampleKernelMs(self.handle, C.byref(ms)), "rtowGetLastSampleKernelMs")
        return ms.value


def _buffers(color, normal, albedo, scw):
    return abi.AccumBuffers(color, normal, albedo, scw)


class JobHandle:
    def __init__(self, result):
        self.result = result

    def Complete(self):
        return self.result


class SampleBatchJob:
    """Mirror of `struct SampleBatchJob : IJobParallelFor` (JOBS/SampleBatchJob.cs:17-51): same field names.

    Input*/Output* are numpy float32 arrays (host form -> rtowSampleBatch) or DeviceBuffer objects
    (device-resident form -> rtowSampleBatchDevice).  OutputDiagnostics likewise.  CancellationToken is a
    ctypes c_uint8 (NativeReference<bool>) or None.
    """

    def __init__(self, context, params=None, **fields):
        self.context = context
        self.params = params if params is not None else abi.SampleParams()
        self.CancellationToken = None
        self.InputColor = self.InputNormal = self.InputAlbedo = self.InputSampleCountWeight = None
        self.OutputColor = self.OutputNormal = self.OutputAlbedo = self.OutputSampleCountWeight = None
        self.OutputDiagnostics = None
        for k, v in fields.items():
            setattr(self, k, v)

    # reference field names -> RtowSampleParams
    Size = property(lambda s: (s.params.size.x, s.params.size.y), lambda s, v: setattr(s.params, "size", abi.Float2(float(v[0]), float(v[1]))))
    SliceOffset = property(lambda s: s.params.sliceOffset, lambda s, v: setattr(s.params, "sliceOffset", int(v)))
    SliceDivider = property(lambda s: s.params.sliceDivider, lambda s, v: setattr(s.params, "sliceDivider", int(v)))
    Seed = property(lambda s: s.params.seed, lambda s, v: setattr(s.params, "seed", int(v)))
    View = property(lambda s: s.params.view, lambda s, v: setattr(s.params, "view", v))
    Environment = property(lambda s: s.params.environment, lambda s, v: setattr(s.params, "environment", v))
    TraceDepth = property(lambda s: s.params.traceDepth, lambda s, v: setattr(s.params, "traceDepth", int(v)))
    SubPixelJitter = property(lambda s: bool(s.params.subPixelJitter), lambda s, v: setattr(s.params, "subPixelJitter", int(bool(v))))
    NoiseColor = property(lambda s: s.params.noiseColor, lambda s, v: setattr(s.params, "noiseColor", int(v)))

    @property
    def SampleCountRange(self):
        return (self.params.sampleCountRange[0], self.params.sampleCountRange[1])

    @SampleCountRange.setter
    def SampleCountRange(self, v):
        self.params.sampleCountRange[0], self.params.sampleCountRange[1] = int(v[0]), int(v[1])

    @property
    def SampleCountWeightExtrema(self):
        return (self.params.sampleCountWeightExtrema.x, self.params.sampleCountWeightExtrema.y)

    @SampleCountWeightExtrema.setter
    def SampleCountWeightExtrema(self, v):
        self.params.sampleCountWeightExtrema = abi.Float2(float(v[0]), float(v[1]))

    def Schedule(self, arrayLength=None, innerloopBatchCount=1, dependsOn=None, stream=None):
        """`sampleBatchJob.Schedule(totalBufferSize, 1, dep)` (UNITY/Raytracer.cs:730).  Runs synchronously for host
        buffers (like IJob.Execute on a worker thread); enqueues on `stream` for device buffers."""
        lib = load()
        n = int(self.params.size.x) * int(self.params.size.y)
        if arrayLength is not None and arrayLength != n:
            raise ValueError("arrayLength must equal Size.x * Size.y")
        cancel = C.addressof(self.CancellationToken) if self.CancellationToken is not None else None
        ins = (self.InputColor, self.InputNormal, self.InputAlbedo, self.InputSampleCountWeight)
        outs = (self.OutputColor, self.OutputNormal, self.OutputAlbedo, self.OutputSampleCountWeight)
        if all(isinstance(b, DeviceBuffer) for b in ins + outs):
            bi = _buffers(*[b.ptr for b in ins])
            bo = _buffers(*[b.ptr for b in outs])
            diag = self.OutputDiagnostics.ptr if self.OutputDiagnostics is not None else None
            rc = lib.rtowSampleBatchDevice(self.context.handle, C.byref(self.params), C.byref(bi), C.byref(bo), diag, stream, cancel)
        else:
            for a, width in zip(ins + outs, (4, 3, 3, 1) * 2):
                if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.size == n * width):
                    raise ValueError("host buffers must be C-contiguous float32 arrays of W*H elements")
            bi = _buffers(*[a.ctypes.data for a in ins])
            bo = _buffers(*[a.ctypes.data for a in outs])
            diag = self.OutputDiagnostics.ctypes.data if self.OutputDiagnostics is not None else None
            rc = lib.rtowSampleBatch(self.context.handle, C.byref(self.params), C.byref(bi), C.byref(bo), diag, cancel)
        return JobHandle(rc)


def sample_batch_chain_device(context, params_list, ins, outs, diags=None, stream=None, cancel=None):
    """rtowSampleBatchChainDevice: `len(params_list)` successive batches of one frame enqueued together (the reference keeps two in flight,
    UNITY/Raytracer.cs:586-593); batch 0 reads `ins`, every later batch reads what its predecessor wrote to `outs`.
    ins / outs: four DeviceBuffers each; diags: one DeviceBuffer (or None) per batch."""
    count = len(params_list)
    arr = (abi.SampleParams * count)(*params_list)
    bi = _buffers(*[b.ptr for b in ins])
    bo = _buffers(*[b.ptr for b in outs])
    dptr = None
    if diags is not None:
        dptr = (C.c_void_p * count)(*[d.ptr if d is not None else None for d in diags])
    return load().rtowSampleBatchChainDevice(context.handle, count, arr, C.byref(bi), C.byref(bo), dptr, stream, cancel)


def sample_batch_group_device(context, params_list, ins, outs_list, diags=None, stream=None, cancel=None):
    """rtowSampleBatchGroupDevice: `len(params_list)` INDEPENDENT batches of one frame in one launch; every batch reads `ins` (four DeviceBuffers) and stores
    to its own four DeviceBuffers outs_list[k]; diags: one DeviceBuffer (or None) per batch."""
    count = len(params_list)
    arr = (abi.SampleParams * count)(*params_list)
    bi = _buffers(*[b.ptr for b in ins])
    bo = (abi.AccumBuffers * count)(*[_buffers(*[b.ptr for b in o]) for o in outs_list])
    dptr = None
    if diags is not None:
        dptr = (C.c_void_p * count)(*[d.ptr if d is not None else None for d in diags])
    return load().rtowSampleBatchGroupDevice(context.handle, count, arr, C.byref(bi), bo, dptr, stream, cancel)


def sample_batch_chain_host(context, params_list, inputs=None, want_diag=True):
    """rtowSampleBatchChain: `len(params_list)` successive batches on HOST buffers in one blocking call; returns the final accumulators like
    sample_batch_host, with out["diag"] = one record array per batch."""
    count = len(params_list)
    w, h = int(params_list[0].size.x), int(params_list[0].size.y)
    n = w * h
    if inputs is None:
        inputs = {"color": np.zeros((n, 4), np.float32), "normal": np.zeros((n, 3), np.float32),
                  "albedo": np.zeros((n, 3), np.float32), "scw": np.zeros(n, np.float32)}
    ins = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inputs.items()}
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in ins.items()}
    diags = [np.zeros((n, params_list[0].diagnosticsStride // 4), np.float32) for _ in range(count)] if want_diag else None
    arr = (abi.SampleParams * count)(*params_list)
    bi = _buffers(*[ins[k].ctypes.data for k in ("color", "normal", "albedo", "scw")])
    bo = _buffers(*[out[k].ctypes.data for k in ("color", "normal", "albedo", "scw")])
    dptr = (C.c_void_p * count)(*[d.ctypes.data for d in diags]) if want_diag else None
    check(load().rtowSampleBatchChain(context.handle, count, arr, C.byref(bi), C.byref(bo), dptr, None), "rtowSampleBatchChain")
    out["diag"] = diags
    return out


def sample_batch_host(context, params, inputs=None, want_diag=True):
    """Convenience used by tests/bench: run one batch with host buffers; returns dict like the oracle binding."""
    w, h = int(params.size.x), int(params.size.y)
    n = w * h
    if inputs is None:
        inputs = {"color": np.zeros((n, 4), np.float32), "normal": np.zeros((n, 3), np.float32),
                  "albedo": np.zeros((n, 3), np.float32), "scw": np.zeros(n, np.float32)}
    ins = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inputs.items()}
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in ins.items()}
    diag = np.zeros((n, params.diagnosticsStride // 4), np.float32) if want_diag else None
    job = SampleBatchJob(context, params)
    job.InputColor, job.InputNormal, job.InputAlbedo, job.InputSampleCountWeight = ins["color"], ins["normal"], ins["albedo"], ins["scw"]
    job.OutputColor, job.OutputNormal, job.OutputAlbedo, job.OutputSampleCountWeight = out["color"], out["normal"], out["albedo"], out["scw"]
    job.OutputDiagnostics = diag
    rc = job.Schedule(n, 1).Complete()
    check(rc, "rtowSampleBatch")
    out["diag"] = diag
    return out


class CombineJob:
    """Mirror of `struct CombineJob` (JOBS/CombineJob.cs:10-27); device buffers."""

    def __init__(self, context, Size, DebugMode=False, LdrAlbedo=False):
        self.context, self.Size, self.DebugMode, self.LdrAlbedo = context, Size, DebugMode, LdrAlbedo
        self.InputColor = self.InputNormal = self.InputAlbedo = None
        self.OutputColor = self.OutputNormal = self.OutputAlbedo = None

    def Schedule(self, stream=None):
        p = abi.CombineParams(int(self.Size[0]), int(self.Size[1]), int(self.DebugMode), int(self.LdrAlbedo))
        rc = load().rtowCombineDevice(self.context.handle, C.byref(p), self.InputColor.ptr, self.InputNormal.ptr, self.InputAlbedo.ptr,
                                      self.OutputColor.ptr, self.OutputNormal.ptr, self.OutputAlbedo.ptr, stream)
        return JobHandle(rc)


class FinalizeTexturesJob:
    """Mirror of `struct FinalizeTexturesJob` (JOBS/FinalizeTexturesJob.cs:11-21); device buffers."""

    def __init__(self, context, pixel_count):
        self.context, self.pixel_count = context, pixel_count
        self.InputColor = self.InputNormal = self.InputAlbedo = None
        self.OutputColor = self.OutputNormal = self.OutputAlbedo = None

    def Schedule(self, stream=None):
        rc = load().rtowFinalizeDevice(self.context.handle, self.pixel_count, self.InputColor.ptr, self.InputNormal.ptr, self.InputAlbedo.ptr,
                                       self.OutputColor.ptr, self.OutputNormal.ptr, self.OutputAlbedo.ptr, stream)
        return JobHandle(rc)


class ReduceMetricsJob:
    """Mirror of `struct ReduceMetricsJob` (JOBS/ReduceMetricsJob.cs:10-20); device buffers in, scalars out."""

    def __init__(self, context, pixel_count, diagnostics_stride=4):
        self.context, self.pixel_count, self.stride = context, pixel_count, diagnostics_stride
        self.Diagnostics = self.AccumulatedColor = self.AccumulatedSampleCountWeight = None
        self.metrics = abi.Metrics()

    def Schedule(self, stream=None):
        rc = load().rtowReduceMetricsDevice(self.context.handle, self.pixel_count, self.Diagnostics.ptr, self.stride, self.AccumulatedColor.ptr,
                                            self.AccumulatedSampleCountWeight.ptr, stream, C.byref(self.metrics))
        return JobHandle(rc)

    TotalRayCount = property(lambda s: s.metrics.totalRayCount)
    TotalSamples = property(lambda s: s.metrics.totalSamples)
    SampleCountWeightExtrema = property(lambda s: (s.metrics.sampleCountWeightExtrema.x, s.metrics.sampleCountWeightExtrema.y))
    SampleCountExtrema = property(lambda s: (s.metrics.sampleCountExtrema[0], s.metrics.sampleCountExtrema[1]))
