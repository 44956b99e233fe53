"""CPU tests of the host-side logic: scene generation, the View constructor, the job mirror's field mapping."""
import ctypes as C
import math

import numpy as np


def test_view_matches_reference_constructor_in_float64(rt):
    """RT/View.cs:16-36 evaluated in float64 numpy vs the float32 host mirror."""
    scene = rt.scenes.cover_scene()
    w, h = 1920, 1080
    v = rt.scenes.make_view(scene, w, h, focus=10.0)
    cam = scene.camera
    origin, look = np.array(cam["position"]), np.array(cam["target"])
    up = np.array(cam["up"])
    theta = cam["vfov"] * math.pi / 180
    hh = math.tan(theta / 2)
    hw = (w / h) * hh
    fwd = (origin - look) / np.linalg.norm(origin - look)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    llc = hw * 10 * -right + hh * 10 * -upv + 10 * -fwd
    assert np.allclose(v.lowerLeftCorner.tuple(), llc, atol=2e-5)
    assert np.allclose(v.horizontal.tuple(), 2 * hw * 10 * right, atol=2e-5)
    assert np.allclose(v.vertical.tuple(), 2 * hh * 10 * upv, atol=2e-5)
    assert np.allclose(v.forward.tuple(), fwd, atol=1e-6) and v.lensRadius == 0.0
    scene.camera["aperture"] = 0.05
    assert rt.scenes.make_view(scene, w, h).lensRadius == np.float32(0.025)


def test_auto_focus_distance_equals_oracle_nearest_hit(rt, oracle):
    """focusDistance = first hit along the view axis (UNITY/Raytracer.cs:608-609)."""
    scene = rt.scenes.cover_scene()
    o = np.array(scene.camera["position"], np.float32)
    d = np.array(scene.camera["target"], np.float32) - o
    d = rt.scenes._normalize(d)
    f = rt.scenes.focus_distance(scene, o, d)
    osc = oracle.OracleScene(scene.desc())
    n, hit = osc.nearest_hit(o, d)
    osc.close()
    assert n > 0 and abs(hit[0] - f) < 1e-4 * f


def test_scene_generator_is_deterministic_and_non_overlapping(rt):
    a, b = rt.scenes.cover_scene(), rt.scenes.cover_scene()
    pa, pb = np.stack(a.positions), np.stack(b.positions)
    assert np.array_equal(pa, pb) and a.entity_count == 486
    small = pa[4:]
    d = np.linalg.norm(small[:, None, :] - small[None, :, :], axis=2) + np.eye(len(small)) * 10
    assert d.min() >= 0.2 + 0.2 + 0.15 - 1e-5          # radius + radius + MinDistance (asset :106)
    assert np.all(np.abs(small[:, 0]) <= 11) and np.all(np.abs(small[:, 2]) <= 11) and np.allclose(small[:, 1], 0.2)
    for big in pa[1:4]:
        assert np.all(np.linalg.norm(small - big, axis=1) >= 1.0 + 0.2 + 0.15 - 1e-5)


def test_stress_scene_counts(rt):
    s = rt.scenes.stress_scene(count=1500, max_tentatives=5000)
    assert s.entity_count == 1500
    r = np.array(s.radii[4:])
    assert r.min() >= 0.05 and r.max() <= 0.2


def test_job_mirror_field_names_map_onto_params(rt):
    job = rt.SampleBatchJob(None)
    job.Size = (400, 225)
    job.SliceOffset, job.SliceDivider, job.Seed = 1, 4, 77
    job.SampleCountRange = (8, 50)
    job.TraceDepth = 35
    job.SubPixelJitter = True
    job.SampleCountWeightExtrema = (0.25, 2.5)
    p = job.params
    assert (p.size.x, p.size.y, p.sliceOffset, p.sliceDivider, p.seed) == (400.0, 225.0, 1, 4, 77)
    assert (p.sampleCountRange[0], p.sampleCountRange[1], p.traceDepth, p.subPixelJitter) == (8, 50, 35, 1)
    assert (p.sampleCountWeightExtrema.x, p.sampleCountWeightExtrema.y) == (0.25, 2.5)
    assert job.Size == (400.0, 225.0) and job.SampleCountRange == (8, 50)


def test_make_params_defaults_follow_the_benchmark_contract(rt):
    scene = rt.scenes.cover_scene()
    p = rt.scenes.make_params(scene, 1920, 1080, spp=256, trace_depth=8)
    assert (p.seed, p.sliceOffset, p.sliceDivider, p.subPixelJitter, p.noiseColor) == (1, 0, 1, 1, rt.abi.NOISE_WHITE)
    assert p.sampleCountRange[0] == p.sampleCountRange[1] == 256
    assert p.environment.skyType == rt.abi.SKY_GRADIENT
    assert p.environment.skyBottomColor.tuple() == (1.0, 1.0, 1.0)
    assert np.allclose(p.environment.skyTopColor.tuple(), (0.5, 0.7, 1.0))


def test_bench_configs_are_the_baseline_configs():
    """bench.py --config N times BASELINE.json configs[N-1]: the sizes, sample counts and bounce counts quoted there."""
    import importlib.util
    import json
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert sorted(bench.CONFIGS) == [2, 3, 4, 5]
    for n, cfg in bench.CONFIGS.items():
        text = base["configs"][n - 1]
        m = re.search(r"(\d+)×(\d+)", text)
        assert (int(m.group(1)), int(m.group(2))) == (cfg["width"], cfg["height"]), text
        assert int(re.search(r"(\d+) spp", text).group(1)) == cfg["spp"], text
        b = re.search(r"(\d+) bounces", text)
        assert cfg["depth"] == (int(b.group(1)) if b else 8), text             # configs 4 and 5 do not name a bounce count: the metric's 8
    assert "10k-sphere" in base["configs"][3] and bench.CONFIGS[4]["scene"] == "stress"
    assert "Moving spheres" in base["configs"][4] and bench.CONFIGS[5]["scene"] == "moving"
