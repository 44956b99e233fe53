"""GPU: the HIP path against the COMMITTED golden vectors (tests/golden/, written by make_golden.py from the strict oracle) - no oracle
in the loop, so this also holds on a box where the checker was not built: raw arrays of the 64 x 36 cover render and of the first-hit
AOVs, SHA-256 digests of BASELINE.json's config 1 (400 x 225), of the moving-sphere scene and of one render per feature."""
import importlib.util
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _feature_module():
    spec = importlib.util.spec_from_file_location("feature_cases", os.path.join(GOLDEN, "feature_cases.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _digests(fc, r):
    d = {k: fc.sha(r[k]) for k in ("color", "normal", "albedo", "scw")}
    d["raycount"] = fc.sha(r["diag"][:, 0].copy())
    return d


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLDEN, "golden.json")))


def test_cover_64x36_equals_the_committed_arrays(rt, gpu_context):
    want = np.load(os.path.join(GOLDEN, "cover_64x36_8spp_d8.npz"))
    scene = rt.scenes.cover_scene()
    gpu_context.upload_scene(scene.desc())
    p = rt.scenes.make_params(scene, 64, 36, spp=8, trace_depth=8, diagnostics_stride=16)
    got = rt.sample_batch_host(gpu_context, p)
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert np.array_equal(got["diag"][:, 0], want["raycount"])


def test_first_hit_aovs_equal_the_committed_arrays(rt, gpu_context):
    want = np.load(os.path.join(GOLDEN, "cover_96x54_firsthit.npz"))
    scene = rt.scenes.cover_scene()
    gpu_context.upload_scene(scene.desc())
    got = rt.sample_batch_host(gpu_context, rt.scenes.make_params(scene, 96, 54, spp=1, trace_depth=1, jitter=False))
    for k in ("normal", "albedo"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k


def test_config1_and_moving_scene_digests(rt, gpu_context, golden):
    fc = _feature_module()
    cover = rt.scenes.cover_scene()
    gpu_context.upload_scene(cover.desc())
    got = rt.sample_batch_host(gpu_context, rt.scenes.make_params(cover, 400, 225, spp=8, trace_depth=8))
    want = golden["config1_400x225_8spp_d8"]
    d = _digests(fc, got)
    for k in d:
        assert d[k] == want[k], k
    assert float(got["diag"][:, 0].sum()) == want["total_rays"]
    assert float(got["color"][:, 3].sum()) == want["successful_samples"]
    moving = rt.scenes.moving_scene()
    gpu_context.upload_scene(moving.desc())
    got = rt.sample_batch_host(gpu_context, rt.scenes.make_params(moving, 96, 54, spp=4, trace_depth=8))
    for k in ("color", "normal", "albedo", "scw"):
        assert fc.sha(got[k]) == golden["moving_96x54_4spp_d8"][k], k


class _ContextAsOracleScene:
    """feature_cases' set-up callbacks talk to an OracleScene; forward them to the device context."""

    def __init__(self, ctx):
        self.ctx = ctx

    def set_blue_noise(self, desc):
        self.ctx.upload_blue_noise(desc)

    def set_stb_noise(self, desc):
        self.ctx.upload_stb_noise(desc)

    def set_cubemap(self, desc):
        self.ctx.upload_sky_cubemap(desc)


def test_feature_renders_equal_the_committed_digests(rt, gpu_context, golden):
    """One render per feature - round 1's (primitives, volumes, ties, textures, sky, noise, adaptive counts) and round 2 / 3's: exact ties by
    size (decal stack), hit lists of 27 and 99 entries (spilled to HBM), the long tie lists of the twin row, both per-sample RNG policies, the
    reference's own FULL_DIAGNOSTICS columns (also with leaves forced at MaxBvhDepth), three batches as one chained launch (whole frame and an
    interlaced slice), and sparse pixels of the 250 882-triangle mesh (32-bit candidate codes) - against digests committed by
    tests/golden/make_golden.py, no oracle in the loop."""
    fc = _feature_module()
    cases = fc.feature_cases(rt)
    assert sorted(cases) == sorted(golden["features"])
    keys = ("color", "normal", "albedo", "scw")
    for name, case in cases.items():
        scene, kw, setup, opts = fc.unpack(case)
        own = "context_flags" in opts
        ctx = rt.Context(0, flags=opts["context_flags"]) if own else gpu_context
        try:
            ctx.upload_scene(scene.desc(max_bvh_depth=opts["max_bvh_depth"]) if "max_bvh_depth" in opts else scene.desc())
            if setup:
                setup(_ContextAsOracleScene(ctx))
            p = rt.scenes.make_params(scene, **kw)
            batch_raycounts = None
            if "chain_seeds" in opts:
                plist = []
                for seed in opts["chain_seeds"]:
                    q = rt.abi.SampleParams.from_buffer_copy(p)
                    q.seed = seed
                    plist.append(q)
                got = rt.sample_batch_chain_host(ctx, plist)
                batch_raycounts = [d[:, 0].copy() for d in got["diag"]]
                got["diag"] = got["diag"][-1]
            else:
                got = rt.sample_batch_host(ctx, p)
            if "sparse" in opts:
                idx = fc.sparse_indices(kw, opts)
                got = {k: got[k][idx] for k in keys + ("diag",)}
                assert ctx.scene_info().wideCodes == 1
            want = golden["features"][name]
            d = fc.digests_of(got, opts, batch_raycounts)
            assert sorted(d) == sorted(want), name
            for k in d:
                assert d[k] == want[k], (name, k)
        finally:
            if own:
                ctx.close()
            else:
                ctx.upload_blue_noise(None)
                ctx.upload_stb_noise(None)
                ctx.upload_sky_cubemap(None)
