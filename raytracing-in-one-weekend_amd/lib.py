"""Loader for the C-ABI product library csrc/librtow_hip.so (include/rtow.h).

Fails loudly: if the shared object is missing or does not load there is NO fallback of any kind.
"""
import ctypes as C
import os
import subprocess

from . import abi

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("RTOW_LIB_PATH") or os.path.join(_CSRC, "librtow_hip.so")  # override: development builds only
_lib = None


class RtowError(RuntimeError):
    def __init__(self, code, where, message=""):
        self.code = code
        super().__init__("%s failed: %d (%s)%s" % (where, code, message, ""))


def build(force=False):
    """Compile the HIP extension for gfx950 with csrc/Makefile (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    proc = subprocess.run(["make", "-C", _CSRC], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("building librtow_hip.so failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


def load():
    """dlopen librtow_hip.so and declare every prototype of include/rtow.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            "%s is missing: build it with `make -C %s` (or __graft_entry__.build()). "
            "There is no CPU fallback for the sample-batch path." % (LIB_PATH, _CSRC))
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    AB = C.POINTER(abi.AccumBuffers)
    lib.rtowGetApiVersion.restype = C.c_int
    lib.rtowErrorString.restype = C.c_char_p
    lib.rtowErrorString.argtypes = [C.c_int]
    lib.rtowCreateContext.argtypes = [C.POINTER(abi.ContextOptions), C.POINTER(vp)]
    lib.rtowDestroyContext.argtypes = [vp]
    lib.rtowUploadScene.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    lib.rtowUploadSkyCubemap.argtypes = [vp, C.POINTER(abi.CubemapDesc)]
    lib.rtowUploadBlueNoise.argtypes = [vp, C.POINTER(abi.BlueNoiseDesc)]
    lib.rtowUploadStbNoise.argtypes = [vp, C.POINTER(abi.StbNoiseDesc)]
    lib.rtowGetSceneInfo.argtypes = [vp, C.POINTER(abi.SceneInfo)]
    lib.rtowSampleBatch.argtypes = [vp, C.POINTER(abi.SampleParams), AB, AB, vp, vp]
    lib.rtowSampleBatchDevice.argtypes = [vp, C.POINTER(abi.SampleParams), AB, AB, vp, vp, vp]
    lib.rtowSampleBatchChainDevice.argtypes = [vp, C.c_int32, C.POINTER(abi.SampleParams), AB, AB, C.POINTER(vp), vp, vp]
    lib.rtowSampleBatchGroupDevice.argtypes = [vp, C.c_int32, C.POINTER(abi.SampleParams), AB, AB, C.POINTER(vp), vp, vp]
    lib.rtowSampleBatchChain.argtypes = [vp, C.c_int32, C.POINTER(abi.SampleParams), AB, AB, C.POINTER(vp), vp]
    lib.rtowGetLastSampleKernelMs.argtypes = [vp, C.POINTER(C.c_float)]
    lib.rtowProbeNearestHit.argtypes = [vp, C.POINTER(abi.Float3), C.POINTER(abi.Float3), C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    lib.rtowReduceMetricsDevice.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, vp, C.POINTER(abi.Metrics)]
    lib.rtowCombineDevice.argtypes = [vp, C.POINTER(abi.CombineParams), vp, vp, vp, vp, vp, vp, vp]
    lib.rtowFinalizeDevice.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    lib.rtowCombineFinalizeDevice.argtypes = [vp, C.POINTER(abi.CombineParams), vp, vp, vp, vp, vp, vp, vp]
    lib.rtowReduceMetricsDeviceAsync.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, vp, vp, vp]
    lib.rtowAddAccumDevice.argtypes = [vp, C.c_int32, AB, AB, vp]
    lib.rtowDeviceAlloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.rtowDeviceFree.argtypes = [vp, vp]
    lib.rtowDeviceCopy.argtypes = [vp, vp, vp, C.c_size_t, C.c_int]
    lib.rtowDeviceMemset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    lib.rtowSynchronize.argtypes = [vp]
    lib.rtowGetBatchStatus.argtypes = [vp]
    lib.rtowRegisterHostBuffer.argtypes = [vp, vp, C.c_size_t]
    lib.rtowUnregisterHostBuffer.argtypes = [vp, vp]
    lib.rtowCommSetLibraryPath.argtypes = [C.c_char_p]
    lib.rtowCommGetUniqueId.argtypes = [C.POINTER(abi.CommId)]
    lib.rtowCommInit.argtypes = [vp, C.POINTER(abi.CommId), C.c_int32, C.c_int32]
    lib.rtowCommDestroy.argtypes = [vp]
    lib.rtowGatherRowsDevice.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, AB, AB, C.c_int32, C.c_int32, vp]
    lib.rtowHybridPlan.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(abi.HybridPlan)]
    lib.rtowExchangeAccumDevice.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, AB, AB, C.c_int32, vp]
    for name in abi.EXPORTED_SYMBOLS:
        if name not in ("rtowErrorString",):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(code, where):
    if code != abi.RTOW_SUCCESS:
        raise RtowError(code, where, load().rtowErrorString(code).decode())
