"""ctypes mirror of include/rtow.h (the C ABI of librtow_hip.so).

Field order and types must match the header exactly; tests/test_abi.py checks sizeof() of every
struct against the values the C compiler reports (rtowGetStructSizes is not needed: the oracle
library is compiled from the same header and exposes the sizes it saw).
"""
import ctypes as C

RTOW_API_VERSION = 11

# RtowResult
RTOW_SUCCESS = 0
RTOW_ERROR_INVALID_VALUE = 1
RTOW_ERROR_MEMORY_ALLOCATION = 2
RTOW_ERROR_NO_DEVICE = 3
RTOW_ERROR_NO_SCENE = 4
RTOW_ERROR_UNSUPPORTED = 5
RTOW_ERROR_LAUNCH_FAILURE = 6
RTOW_ERROR_CANCELLED = 7
RTOW_ERROR_CAPACITY = 8
RTOW_ERROR_INTERNAL = 99

# RtowEntityType (RT/Entity.cs:13-20)
ENTITY_NONE, ENTITY_SPHERE, ENTITY_RECT, ENTITY_BOX, ENTITY_TRIANGLE = range(5)
# RtowMaterialType (RT/Material.cs:9-14)
MATERIAL_STANDARD, MATERIAL_DIELECTRIC, MATERIAL_PROBABILISTIC_VOLUME = range(3)
# RtowTextureType (RT/Texture.cs:13-21)
TEXTURE_NONE, TEXTURE_CONSTANT, TEXTURE_CHECKER_PATTERN, TEXTURE_PERLIN_NOISE, TEXTURE_IMAGE, TEXTURE_CONSTANT_SCALAR = range(6)
# RtowSkyType (RT/Environment.cs:5-10)
SKY_NONE, SKY_GRADIENT, SKY_CUBEMAP = range(3)
# RtowNoiseColor (RT/RandomSource.cs:8-13)
NOISE_WHITE, NOISE_BLUE, NOISE_SPATIOTEMPORAL_BLUE = range(3)
# RtowMemcpyKind
MEMCPY_HOST_TO_HOST, MEMCPY_HOST_TO_DEVICE, MEMCPY_DEVICE_TO_HOST, MEMCPY_DEVICE_TO_DEVICE = range(4)


class Float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("mainColor", Float3), ("parameter", C.c_float),
                ("scalarValueChannel", C.c_int32), ("imageIndex", C.c_int32)]


class Image(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("pixelStride", C.c_int32), ("pixels", C.c_void_p)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("albedo", Texture), ("glossiness", Texture), ("emission", Texture),
                ("metallic", Texture), ("parameter", C.c_float)]


class Entity(C.Structure):
    _fields_ = [("type", C.c_int32), ("moving", C.c_int32), ("rotation", Float4), ("position", Float3),
                ("destinationOffset", Float3), ("timeRange", Float2), ("materialIndex", C.c_int32),
                ("size", Float3), ("contentIndex", C.c_int32)]


class Triangle(C.Structure):
    """RT/EntityTypes/Triangle.cs:8-12 byte for byte (96 bytes): Data {v2-v0, v1-v0, v0}, Normals, TextureCoordinates."""
    _fields_ = [("data", Float3 * 3), ("normals", Float3 * 3), ("textureCoordinates", Float2 * 3)]


class SceneDesc(C.Structure):
    _fields_ = [("entities", C.POINTER(Entity)), ("entityCount", C.c_int32),
                ("materials", C.POINTER(Material)), ("materialCount", C.c_int32),
                ("maxBvhDepth", C.c_int32), ("triangles", C.POINTER(Triangle)), ("triangleCount", C.c_int32),
                ("images", C.POINTER(Image)), ("imageCount", C.c_int32)]


class SceneInfo(C.Structure):
    _fields_ = [("entityCount", C.c_int32), ("materialCount", C.c_int32), ("bvhNodeCount", C.c_int32),
                ("bvhDepth", C.c_int32), ("ldsBytesScene", C.c_int32), ("sceneInLds", C.c_int32),
                ("sceneBytesDevice", C.c_uint64), ("hitSpillBytes", C.c_uint64), ("hitListCapacity", C.c_int32), ("wideCodes", C.c_int32),
                ("thresholdSet", C.c_int32), ("schedulerTune", C.c_int32 * 9)]


class View(C.Structure):
    _fields_ = [("origin", Float3), ("lowerLeftCorner", Float3), ("horizontal", Float3), ("vertical", Float3),
                ("forward", Float3), ("up", Float3), ("right", Float3), ("lensRadius", C.c_float)]


CUBEMAP_UNSIGNED_BYTE, CUBEMAP_SIGNED_HALF = 0, 1
RNG_REFERENCE, RNG_PER_SAMPLE, RNG_PER_SAMPLE_XOROSHIRO = 0, 1, 2


class CubemapDesc(C.Structure):
    _fields_ = [("faceWidth", C.c_int32), ("faceHeight", C.c_int32), ("channelType", C.c_int32), ("pixelStride", C.c_int32),
                ("faces", C.c_void_p)]


class BlueNoiseDesc(C.Structure):
    _fields_ = [("rowStride", C.c_uint32), ("textureCount", C.c_uint32), ("texels", C.c_void_p)]


class StbNoiseDesc(C.Structure):
    _fields_ = [("rowStride", C.c_uint32), ("textureCount", C.c_uint32), ("scalar", C.c_void_p), ("vector2", C.c_void_p),
                ("cosineUnitVector3", C.c_void_p), ("unitVector2", C.c_void_p), ("unitVector3", C.c_void_p)]


class Environment(C.Structure):
    _fields_ = [("skyType", C.c_int32), ("skyBottomColor", Float3), ("skyTopColor", Float3)]


class SampleParams(C.Structure):
    _fields_ = [("size", Float2), ("sliceOffset", C.c_int32), ("sliceDivider", C.c_int32), ("seed", C.c_uint32),
                ("view", View), ("environment", Environment), ("sampleCountRange", C.c_uint32 * 2),
                ("traceDepth", C.c_int32), ("subPixelJitter", C.c_int32), ("noiseColor", C.c_int32),
                ("sampleCountWeightExtrema", Float2), ("diagnosticsStride", C.c_int32), ("noiseTextureIndex", C.c_int32),
                ("rngPolicy", C.c_int32)]


class AccumBuffers(C.Structure):
    _fields_ = [("color", C.c_void_p), ("normal", C.c_void_p), ("albedo", C.c_void_p),
                ("sampleCountWeight", C.c_void_p)]


LogCallback = C.CFUNCTYPE(None, C.c_int32, C.c_char_p, C.c_char_p, C.c_void_p)


# RtowContextFlags
CONTEXT_EXACT_TIES_ALWAYS, CONTEXT_EXACT_TIES_NEVER, CONTEXT_REFERENCE_DIAGNOSTICS, CONTEXT_NO_CAMERA_RAY_LISTS, CONTEXT_NO_CHUNK_ORDER, CONTEXT_FORCE_WIDE_CODES, CONTEXT_NO_THRESHOLD_TUNING, CONTEXT_NO_CHAIN_FUSION = 1, 2, 4, 8, 16, 32, 64, 128
# RtowGatherMask
GATHER_COLOR, GATHER_NORMAL, GATHER_ALBEDO, GATHER_SAMPLE_COUNT_WEIGHT, GATHER_ALL, GATHER_NO_BATCH_WAIT, GATHER_LOOPBACK = 1, 2, 4, 8, 15, 16, 32


class ContextOptions(C.Structure):
    _fields_ = [("deviceOrdinal", C.c_int32), ("logCallback", LogCallback), ("logCallbackData", C.c_void_p),
                ("logCallbackLevel", C.c_int32), ("flags", C.c_uint32), ("ldsSceneBudgetBytes", C.c_int32), ("schedulerTune", C.c_int32 * 9),
                ("hitListCapacity", C.c_int32), ("sliceBlockThreads", C.c_int32)]


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


class HybridPlan(C.Structure):
    _fields_ = [("tileCount", C.c_int32), ("groupCount", C.c_int32), ("tile", C.c_int32), ("group", C.c_int32),
                ("sliceOffset", C.c_int32), ("sliceDivider", C.c_int32), ("samples", C.c_uint32), ("seed", C.c_uint32)]


class Metrics(C.Structure):
    _fields_ = [("totalRayCount", C.c_int32), ("totalSamples", C.c_int32), ("sampleCountWeightExtrema", Float2),
                ("sampleCountExtrema", C.c_int32 * 2), ("totalRayCount64", C.c_int64), ("totalSamples64", C.c_int64)]


class CombineParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("debugMode", C.c_int32), ("ldrAlbedo", C.c_int32)]


# every symbol include/rtow.h declares (tests/test_abi.py checks the library exports all of them)
EXPORTED_SYMBOLS = [
    "rtowGetApiVersion", "rtowErrorString", "rtowCreateContext", "rtowDestroyContext", "rtowUploadScene",
    "rtowUploadSkyCubemap", "rtowUploadBlueNoise", "rtowUploadStbNoise", "rtowGetSceneInfo", "rtowSampleBatch", "rtowSampleBatchDevice", "rtowGetLastSampleKernelMs",
    "rtowReduceMetricsDevice", "rtowCombineDevice", "rtowFinalizeDevice", "rtowAddAccumDevice", "rtowDeviceAlloc", "rtowDeviceFree",
    "rtowDeviceCopy", "rtowDeviceMemset", "rtowSynchronize", "rtowGetBatchStatus", "rtowRegisterHostBuffer", "rtowUnregisterHostBuffer",
    "rtowSampleBatchChainDevice", "rtowSampleBatchChain", "rtowCommSetLibraryPath", "rtowCommGetUniqueId", "rtowCommInit", "rtowCommDestroy", "rtowGatherRowsDevice",
    "rtowHybridPlan", "rtowExchangeAccumDevice", "rtowSampleBatchGroupDevice",
    "rtowCombineFinalizeDevice", "rtowReduceMetricsDeviceAsync", "rtowProbeNearestHit",
]
