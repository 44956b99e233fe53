"""ctypes binding of the CPU oracle (oracle/liboracle_{strict,fast}.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
It reuses the *data layout* of the boundary (the ctypes structs of the product package's abi.py, which
mirror include/rtow.h) and nothing else of the product.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
abi = importlib.import_module("raytracing-in-one-weekend_amd.abi")


class CountersOut(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("boundsHit", C.c_uint64), ("candidates", C.c_uint64),
                ("nodesVisited", C.c_uint64), ("hits", C.c_uint64), ("maxNodeStack", C.c_uint32),
                ("maxCandidates", C.c_uint32), ("maxHits", C.c_uint32), ("bvhNodeCount", C.c_uint32),
                ("bvhDepth", C.c_uint32), ("threads", C.c_uint32)]


def build(force=False):
    """Compile the oracle with oracle/Makefile (g++ only)."""
    if force:
        subprocess.run(["make", "-C", _HERE, "clean"], check=True, capture_output=True)
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


_libs = {}


def load(kind="strict"):
    if kind in _libs:
        return _libs[kind]
    path = os.path.join(_HERE, "liboracle_%s.so" % kind)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    lib.oracle_scene_create.restype = C.c_void_p
    lib.oracle_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
    lib.oracle_scene_destroy.argtypes = [C.c_void_p]
    lib.oracle_scene_node_count.argtypes = [C.c_void_p]
    lib.oracle_scene_depth.argtypes = [C.c_void_p]
    lib.oracle_sample_batch.restype = C.c_int
    lib.oracle_sample_batch.argtypes = [C.c_void_p, C.POINTER(abi.SampleParams)] + [C.c_void_p] * 9 + [C.c_int, C.POINTER(CountersOut)]
    lib.oracle_sample_pixels.restype = C.c_int
    lib.oracle_sample_pixels.argtypes = [C.c_void_p, C.POINTER(abi.SampleParams)] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p, C.c_int]
    lib.oracle_combine.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
    lib.oracle_finalize.argtypes = [C.c_int] + [C.c_void_p] * 6
    lib.oracle_reduce_metrics.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(abi.Metrics)]
    lib.oracle_kat_rng.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_kat_xoroshiro.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_kat_pixel_seed.restype = C.c_uint32
    lib.oracle_kat_pixel_seed.argtypes = [C.c_uint32, C.c_int]
    lib.oracle_kat_sincos.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_kat_log.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_kat_pow.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.oracle_kat_schlick.restype = C.c_float
    lib.oracle_kat_schlick.argtypes = [C.c_float, C.c_float]
    lib.oracle_kat_refract.argtypes = [fp, fp, C.c_float, fp]
    lib.oracle_kat_lambda.restype = C.c_float
    lib.oracle_kat_lambda.argtypes = [fp, fp, C.c_float]
    lib.oracle_kat_roughness_to_alpha.restype = C.c_float
    lib.oracle_kat_roughness_to_alpha.argtypes = [C.c_float]
    lib.oracle_kat_linear_to_gamma.restype = C.c_float
    lib.oracle_kat_linear_to_gamma.argtypes = [C.c_float]
    lib.oracle_kat_basis.argtypes = [fp, fp, fp]
    lib.oracle_kat_aabb_hit.argtypes = [fp, fp, fp, fp]
    lib.oracle_kat_entity_hit.argtypes = [C.POINTER(abi.Entity), C.POINTER(abi.Triangle), C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, fp]
    lib.oracle_kat_entity_bounds.argtypes = [C.POINTER(abi.Entity), C.POINTER(abi.Triangle), C.c_int, fp]
    lib.oracle_scene_set_cubemap.argtypes = [C.c_void_p, C.POINTER(abi.CubemapDesc)]
    lib.oracle_scene_set_blue_noise.argtypes = [C.c_void_p, C.POINTER(abi.BlueNoiseDesc)]
    lib.oracle_scene_set_stb_noise.argtypes = [C.c_void_p, C.POINTER(abi.StbNoiseDesc)]
    lib.oracle_kat_r2.argtypes = [C.c_uint32, fp]
    lib.oracle_kat_r2.restype = None
    lib.oracle_kat_per_pixel_noise.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
    lib.oracle_kat_per_pixel_noise.restype = None
    lib.oracle_kat_cubemap_sample.argtypes = [C.POINTER(abi.CubemapDesc), fp, fp]
    lib.oracle_kat_cubemap_sample.restype = None
    lib.oracle_kat_half_to_float.argtypes = [C.c_uint16]
    lib.oracle_kat_half_to_float.restype = C.c_float
    lib.oracle_kat_unity_sort.argtypes = [fp, C.POINTER(C.c_int), C.c_int]
    lib.oracle_kat_unity_sort.restype = None
    lib.oracle_kat_unity_sort_heapsorts.argtypes = []
    lib.oracle_kat_unity_sort_heapsorts.restype = C.c_int
    lib.oracle_kat_unity_sort_killer.argtypes = [C.c_int, fp]
    lib.oracle_kat_unity_sort_killer.restype = None
    lib.oracle_kat_hit_tie_order.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_int)]
    lib.oracle_kat_hit_tie_order.restype = C.c_int
    lib.oracle_kat_leaf_boxes.argtypes = [C.POINTER(abi.SceneDesc), fp]
    lib.oracle_kat_leaf_boxes.restype = C.c_int
    lib.oracle_kat_scatter.argtypes = [C.POINTER(abi.Material), fp, fp, C.c_float, fp, fp, C.c_float, C.POINTER(C.c_uint32), fp]
    lib.oracle_kat_get_ray.argtypes = [C.POINTER(abi.View), C.c_float, C.c_float, C.POINTER(C.c_uint32), fp]
    lib.oracle_kat_nearest_hit.argtypes = [C.c_void_p, fp, fp, C.c_float, fp]
    lib.oracle_hit_world.argtypes = [C.c_void_p, fp, fp, C.c_float, fp]
    lib.oracle_hit_world.restype = C.c_int
    _libs[kind] = lib
    return lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class OracleScene:
    def __init__(self, scene_desc, kind="strict"):
        self.lib = load(kind)
        self.handle = self.lib.oracle_scene_create(C.byref(scene_desc))
        if not self.handle:
            raise ValueError("oracle: scene rejected (unsupported entity/material kind or bad indices)")

    def close(self):
        if self.handle:
            self.lib.oracle_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cubemap(self, cubemap_desc):
        """Environment.SkyCubemap (RT/Environment.cs:16); None drops it."""
        rc = self.lib.oracle_scene_set_cubemap(self.handle, C.byref(cubemap_desc) if cubemap_desc is not None else None)
        if rc != 0:
            raise ValueError("oracle_scene_set_cubemap failed: %d" % rc)

    def set_blue_noise(self, desc):
        if self.lib.oracle_scene_set_blue_noise(self.handle, C.byref(desc) if desc is not None else None) != 0:
            raise ValueError("oracle_scene_set_blue_noise failed")

    def set_stb_noise(self, desc):
        if self.lib.oracle_scene_set_stb_noise(self.handle, C.byref(desc) if desc is not None else None) != 0:
            raise ValueError("oracle_scene_set_stb_noise failed")

    @property
    def node_count(self):
        return self.lib.oracle_scene_node_count(self.handle)

    @property
    def depth(self):
        return self.lib.oracle_scene_depth(self.handle)

    def sample_batch(self, params, inputs=None, nthreads=0, want_counters=False):
        """Run SampleBatchJob over every pixel.  inputs/outputs: dict of float32 arrays
        color[N,4], normal[N,3], albedo[N,3], scw[N]; diag[N, stride/4]."""
        w, h = int(params.size.x), int(params.size.y)
        n = w * h
        if inputs is None:
            inputs = zero_buffers(n)
        out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in inputs.items()}  # skipped pixels keep input
        diag = np.zeros((n, max(params.diagnosticsStride, 4) // 4), dtype=np.float32)
        counters = CountersOut()
        ins = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inputs.items()}
        rc = self.lib.oracle_sample_batch(
            self.handle, C.byref(params),
            ins["color"].ctypes.data, ins["normal"].ctypes.data, ins["albedo"].ctypes.data, ins["scw"].ctypes.data,
            out["color"].ctypes.data, out["normal"].ctypes.data, out["albedo"].ctypes.data, out["scw"].ctypes.data,
            diag.ctypes.data, nthreads, C.byref(counters))
        if rc != 0:
            raise RuntimeError("oracle_sample_batch failed: %d" % rc)
        out["diag"] = diag
        if want_counters:
            return out, counters
        return out

    def sample_pixels(self, params, indices, inputs=None, nthreads=0):
        """SampleBatchJob.Execute for the given pixel indices only; returns compact arrays ordered like `indices`."""
        w, h = int(params.size.x), int(params.size.y)
        n = w * h
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        ins = zero_buffers(n) if inputs is None else {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in inputs.items()}
        out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in ins.items()}
        diag = np.zeros((n, max(params.diagnosticsStride, 4) // 4), dtype=np.float32)
        rc = self.lib.oracle_sample_pixels(
            self.handle, C.byref(params),
            ins["color"].ctypes.data, ins["normal"].ctypes.data, ins["albedo"].ctypes.data, ins["scw"].ctypes.data,
            out["color"].ctypes.data, out["normal"].ctypes.data, out["albedo"].ctypes.data, out["scw"].ctypes.data,
            diag.ctypes.data, nthreads, idx.ctypes.data, len(idx))
        if rc != 0:
            raise RuntimeError("oracle_sample_pixels failed: %d" % rc)
        res = {k: v[idx] for k, v in out.items()}
        res["diag"] = diag[idx]
        return res

    def hit_world(self, origin, direction, time=0.0):
        """Raytracer.HitWorld (the recursive HitTests.Hit(BvhNode), RT/HitTests.cs:152-196): (hit, [distance, point, normal, entity index])."""
        o = (C.c_float * 8)()
        hit = self.lib.oracle_hit_world(self.handle, _f3(origin), _f3(direction), float(time), o)
        return bool(hit), list(o)

    def nearest_hit(self, origin, direction, time=0.0):
        o = (C.c_float * 8)()
        n = self.lib.oracle_kat_nearest_hit(self.handle, _f3(origin), _f3(direction), float(time), o)
        return n, list(o)


def zero_buffers(n):
    return {"color": np.zeros((n, 4), np.float32), "normal": np.zeros((n, 3), np.float32),
            "albedo": np.zeros((n, 3), np.float32), "scw": np.zeros(n, np.float32)}


def combine(width, height, color4, normal, albedo, debug_mode=False, ldr_albedo=False, kind="strict"):
    lib = load(kind)
    n = width * height
    oc = np.zeros((n, 3), np.float32)
    on = np.zeros((n, 3), np.float32)
    oa = np.zeros((n, 3), np.float32)
    c4 = np.ascontiguousarray(color4, np.float32)
    nn = np.ascontiguousarray(normal, np.float32)
    aa = np.ascontiguousarray(albedo, np.float32)
    lib.oracle_combine(width, height, int(debug_mode), int(ldr_albedo), c4.ctypes.data, nn.ctypes.data, aa.ctypes.data,
                       oc.ctypes.data, on.ctypes.data, oa.ctypes.data)
    return oc, on, oa


def finalize(color, normal, albedo, kind="strict"):
    lib = load(kind)
    n = color.shape[0]
    outs = [np.zeros((n, 4), np.uint8) for _ in range(3)]
    c = np.ascontiguousarray(color, np.float32)
    nn = np.ascontiguousarray(normal, np.float32)
    a = np.ascontiguousarray(albedo, np.float32)
    lib.oracle_finalize(n, c.ctypes.data, nn.ctypes.data, a.ctypes.data, outs[0].ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data)
    return outs


def reduce_metrics(diag, color4, scw, kind="strict"):
    lib = load(kind)
    d = np.ascontiguousarray(diag, np.float32)
    c4 = np.ascontiguousarray(color4, np.float32)
    s = np.ascontiguousarray(scw, np.float32)
    m = abi.Metrics()
    lib.oracle_reduce_metrics(c4.shape[0], d.ctypes.data, d.shape[1] * 4, c4.ctypes.data, s.ctypes.data, C.byref(m))
    return m
