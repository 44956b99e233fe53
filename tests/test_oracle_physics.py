"""Physics pins for the CPU oracle.

The reference holds no vectors and cannot run here (parity unpinned), so the oracle's goldens are its own.  These tests pin it to
things that do not come from this repository: closed-form light transport.  A mis-weighted lobe, a wrong tangent frame, a swapped draw or a
broken generator changes the answers below by far more than their sampling error, whatever the KAT and golden tests (which restate the same
reading of the reference) say.

  * white furnace: in a uniform sky of 1 with non-absorbing, non-emitting surfaces every completed path carries exactly 1 - any extra cos / 1/pi /
    Fresnel weight on a lobe of Material.Scatter (RT/Material.cs:68-173) shows as a colour sum that differs from the success count;
  * one Lambert sphere under the gradient sky: a convex body alone is never hit twice, so radiance = albedo x E[sky(w)] over the cosine-weighted
    hemisphere about N, and E[w] = 2/3 N, E[w w^T] = (I + N N^T) / 4 give the mean AND the variance of every pixel in closed form
    (JOBS/SampleBatchJob.cs:350-359 sky, RT/RandomSource.cs:63-89 hemisphere, UTIL/Tools.cs:19-37 basis);
  * Beer-Lambert: a black ProbabilisticVolume slab in a white sky transmits exp(-density x path length) (RT/Material.cs:49-65);
  * Unity.Mathematics.Random as restated (xorshift32, 23-bit floats): equidistribution of NextFloat in one and two dimensions, chi-squared.
Statistical bounds are 5 sigma per pixel and a chi-squared over all pixels with a one-in-a-million false-alarm rate; seeds are fixed, so the tests are
deterministic all the same."""
import math

import numpy as np
import pytest

S = None  # scenes module, set by the fixture


@pytest.fixture(autouse=True)
def _scenes(rt):
    global S
    S = rt.scenes


def _render(oracle, scene, w, h, spp, depth, seed=1, jitter=True, focus=None, nthreads=8):
    osc = oracle.OracleScene(scene.desc())
    p = S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=seed, jitter=jitter, focus=focus)
    out = osc.sample_batch(p, nthreads=nthreads)
    osc.close()
    return out, p


def test_white_furnace_every_completed_path_carries_exactly_one(rt, oracle):
    """Sky = 1 everywhere, albedo 1, no emission, every material class of the path (lambert, glossy plastic, rough and polished metal, clear and
    frosted glass, moving spheres): each successful sample's colour is a product of ones, so colour sum == success count EXACTLY, pixel by pixel."""
    s = S.Scene("furnace")
    white = (1.0, 1.0, 1.0)
    s.add_sphere((0, -1000, 0), 1000, S.lambertian(white), exclude=True)
    mats = [S.lambertian(white), S.standard(white, 0.0, 0.6), S.standard(white, 1.0, 1.0), S.standard(white, 1.0, 0.4), S.standard(white, 0.5, 0.8),
            S.dielectric(1.5), S.dielectric(1.5, gloss=0.7), S.dielectric(2.4, gloss=1.0)]
    for k, m in enumerate(mats):
        s.add_sphere((-3.5 + k, 0.45, 0.3 * (k % 3)), 0.45, m, moving=(k % 4 == 1), dest_offset=(0, 0.3, 0), time_range=(0, 1))
    s.camera = {"position": [0.0, 2.0, 7.0], "target": [0.0, 0.4, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.1}
    s.sky_bottom = s.sky_top = white
    out, _ = _render(oracle, s, 96, 54, spp=64, depth=48)
    c = out["color"]
    assert c[:, 3].min() >= 1 and c[:, 3].mean() > 60                       # nearly every sample completes at depth 48
    for ch in range(3):
        assert np.array_equal(c[:, ch], c[:, 3]), "channel %d: a lobe is weighted (sum != count)" % ch
    # the albedo AOV of a sample is emission + reflectance of its first non-specular hit, or the sky: 1 either way (JOBS/SampleBatchJob.cs:316-328,366-370)
    assert np.array_equal(out["albedo"][:, 0], c[:, 3])


def _sphere_normal_at_pixel_centres(p, w, h, centre, radius):
    """Float64 camera rays through the pixel centres (RT/View.cs:38-48 without lens / jitter) against one sphere: hit mask, unit normals, directions."""
    v = p.view
    o = np.array([v.origin.x, v.origin.y, v.origin.z], np.float64)
    llc = np.array([v.lowerLeftCorner.x, v.lowerLeftCorner.y, v.lowerLeftCorner.z], np.float64)
    hor = np.array([v.horizontal.x, v.horizontal.y, v.horizontal.z], np.float64)
    ver = np.array([v.vertical.x, v.vertical.y, v.vertical.z], np.float64)
    ys, xs = np.divmod(np.arange(w * h), w)
    u, vv = (xs + 0.5) / w, (ys + 0.5) / h
    d = llc[None, :] + u[:, None] * hor[None, :] + vv[:, None] * ver[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    oc = o - np.asarray(centre, np.float64)
    b = d @ oc
    disc = b * b - (oc @ oc - radius * radius)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0))
    n = (oc[None, :] + t[:, None] * d) / radius
    return hit, n, d


def test_lambert_sphere_under_gradient_sky_matches_the_closed_form_radiance(rt, oracle):
    """One Lambert sphere, nothing else: path = camera -> sphere -> sky.  With cosine-weighted scattering about N (pdf cos / pi, throughput = albedo)
    E[w] = 2/3 N and Cov[w] = (I + N N^T) / 4 - 4/9 N N^T, and the sky is affine in w.y: sky(w) = bottom + (top - bottom) (w.y + 1) / 2.  So for a
    pixel whose centre ray meets the sphere at normal N
        mean   = albedo * (bottom + (top - bottom) * (2/3 N.y + 1) / 2)
        var    = (albedo * (top - bottom) / 2)^2 * (1/4 + N.y^2 / 4 - 4/9 N.y^2)
    per channel.  Jitter off: every sample of a pixel meets the same N.  4096 samples per pixel."""
    s = S.Scene("lambert_ball")
    albedo = (0.8, 0.5, 0.3)
    centre, radius = (0.0, 0.0, 0.0), 1.0
    s.add_sphere(centre, radius, S.lambertian(albedo))
    s.camera = {"position": [0.0, 1.5, 4.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 35.0, "aperture": 0.0}
    s.sky_bottom, s.sky_top = (1.0, 0.9, 0.2), (0.1, 0.3, 1.0)
    w, h, spp = 32, 32, 4096
    out, p = _render(oracle, s, w, h, spp=spp, depth=4, jitter=False)
    hit, n, d = _sphere_normal_at_pixel_centres(p, w, h, centre, radius)
    edge = hit & (np.abs(np.einsum("ij,ij->i", n, d)) < 0.05)               # grazing pixel centres: float32 hit / miss may differ from float64 - excluded
    inner, sky_px = hit & ~edge, ~hit
    assert inner.sum() > 150 and sky_px.sum() > 300
    c = out["color"].astype(np.float64)
    assert np.array_equal(out["color"][:, 3], np.full(w * h, spp, np.float32))   # no path can fail: at most two segments
    mean = c[:, :3] / spp
    bottom, top, alb = (np.array(x, np.float64) for x in (s.sky_bottom, s.sky_top, albedo))
    # pixels that miss: the sky in the ray's own direction (up to the rounding of 4096 float32 additions of the same value: <= 4096 half-ulps of the sum)
    want_sky = bottom[None, :] + (top - bottom)[None, :] * (0.5 * (d[:, 1:2] + 1))
    assert np.abs(mean[sky_px] - want_sky[sky_px]).max() < 1.5e-4
    ny = n[:, 1:2]
    want = alb[None, :] * (bottom[None, :] + (top - bottom)[None, :] * 0.5 * (2.0 / 3.0 * ny + 1))
    var = (alb * (top - bottom) / 2)[None, :] ** 2 * (0.25 + 0.25 * ny ** 2 - 4.0 / 9.0 * ny ** 2)
    z = (mean - want)[inner] / np.sqrt(var[inner] / spp)
    assert np.abs(z).max() < 5.0, "a pixel is %.1f sigma off the closed form" % np.abs(z).max()
    # all pixels together: one channel's z-scores are independent N(0, 1) (the channels of a pixel share their samples)
    chi2 = (z[:, 2] ** 2).sum()
    k = z.shape[0]
    assert abs(chi2 - k) < 5.0 * math.sqrt(2 * k), (chi2, k)
    assert abs(z[:, 2].mean()) < 5.0 / math.sqrt(k)                           # no common bias (a wrong constant factor shows here first)
    # the second moment too: sample variance across pixels of equal N.y cannot be had, but the mean squared z is 1 +- a few percent
    assert 0.8 < (z ** 2).mean() < 1.2
    # normal AOV: the normal of the first hit, summed over samples
    got_n = out["normal"].astype(np.float64)[inner] / spp
    assert np.abs(got_n - n[inner]).max() < 1.5e-4
    # and the same image under other seeds moves by what the variance says (guards against a generator stuck on a short cycle)
    out2, _ = _render(oracle, s, w, h, spp=256, depth=4, jitter=False, seed=77)
    z2 = (out2["color"].astype(np.float64)[:, :3] / 256 - want)[inner] / np.sqrt(var[inner] / 256)
    assert np.abs(z2).max() < 5.5 and 0.8 < (z2 ** 2).mean() < 1.2


def test_black_fog_ball_transmits_by_beer_lambert(rt, oracle):
    """A ProbabilisticVolume SPHERE of density rho and albedo 0 in a uniform sky of 1: a path that crosses the ball unscattered sees 1, a path that
    scatters inside carries 0 from then on.  Pixel mean = P(no scattering) = exp(-rho * chord of the centre ray) (Beer-Lambert; the reference
    draws the free path as -log(u) / rho, RT/Material.cs:56).  Jitter off."""
    s = S.Scene("fog_ball")
    rho, radius = 0.7, 1.5
    s.add_sphere((0.0, 0.0, 0.0), radius, S.volume((0.0, 0.0, 0.0), rho))
    s.camera = {"position": [0.0, 0.0, 6.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.0}
    s.sky_bottom = s.sky_top = (1.0, 1.0, 1.0)
    w, h, spp = 24, 24, 4096

    def chords(p):
        v = p.view
        o = np.array([v.origin.x, v.origin.y, v.origin.z], np.float64)
        _, _, d = _sphere_normal_at_pixel_centres(p, w, h, (0, 0, 0), radius)
        b = d @ o
        disc = b * b - (o @ o - radius * radius)
        return np.where(disc > 0, 2 * np.sqrt(np.maximum(disc, 0)), 0.0)

    out, p = _render(oracle, s, w, h, spp=spp, depth=64, jitter=False)
    chord = chords(p)
    inner, outside = chord > 0.3, chord == 0
    assert inner.sum() > 100 and outside.sum() > 50
    want = np.exp(-rho * chord)
    cnt = out["color"][:, 3].astype(np.float64)
    assert cnt.min() >= spp - 8                                              # a scattered path random-walks out of the ball well within 64 segments
    got = out["color"][:, 0].astype(np.float64) / cnt
    assert np.array_equal(out["color"][outside, 0], out["color"][outside, 3])   # past the ball: the sky, exactly
    with np.errstate(invalid="ignore", divide="ignore"):
        z = ((got - want) / np.sqrt(want * (1 - want) / cnt))[inner]
    assert np.abs(z).max() < 5.0, "transmission is %.1f sigma off exp(-rho d)" % np.abs(z).max()
    assert abs((z ** 2).sum() - z.size) < 5.0 * math.sqrt(2 * z.size)
    assert abs(z.mean()) < 5.0 / math.sqrt(z.size)
    # twice the density: the transmission squares
    s2 = S.Scene("fog_ball2")
    s2.add_sphere((0.0, 0.0, 0.0), radius, S.volume((0.0, 0.0, 0.0), 2 * rho))
    s2.camera, s2.sky_bottom, s2.sky_top = s.camera, s.sky_bottom, s.sky_top
    out2, _ = _render(oracle, s2, w, h, spp=spp, depth=64, jitter=False)
    got2 = out2["color"][:, 0].astype(np.float64) / out2["color"][:, 3]
    with np.errstate(invalid="ignore", divide="ignore"):
        z2 = ((got2 - want ** 2) / np.sqrt(want ** 2 * (1 - want ** 2) / spp))[inner]
    assert np.abs(z2).max() < 5.0


def test_fog_box_entered_from_outside_is_transparent_like_the_reference_says(rt, oracle):
    """Found by the Beer-Lambert test above when it was first written with a slab: HitTests.Hit(Box) reports the normal `sgn = -sign(direction)` for
    hits from inside too ("TODO: Normal is wrong when ray origin is inside the box", RT/HitTests.cs:109-110), so the injected EXIT hit of a volume box
    (JOBS/SampleBatchJob.cs:463-469) faces the ray like an entry, the exit search of the volume branch (:224-247) counts two entries and no exit, and
    the path takes the "volume has holes" way out (:296-302): straight to the sky.  A fog box a camera ray enters from outside is invisible on that
    segment.  The oracle restates the source, quirk included; this pins the quirk to the reference's own comment rather than to the oracle's say-so."""
    s = S.Scene("fog_slab")
    s.add_box((0.0, 0.0, 0.0), (400.0, 400.0, 1.5), S.volume((0.0, 0.0, 0.0), 5.0))
    s.camera = {"position": [0.0, 0.0, 6.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 50.0, "aperture": 0.0}
    s.sky_bottom = s.sky_top = (1.0, 1.0, 1.0)
    out, _ = _render(oracle, s, 16, 16, spp=64, depth=16, focus=6.0)
    assert np.array_equal(out["color"], np.full((256, 4), 64.0, np.float32))   # optical depth 7.5 and yet every sample sees the sky ...
    assert np.array_equal(out["diag"][:, 0], np.full(256, 64.0, np.float32))   # ... on its first segment
    osc = oracle.OracleScene(s.desc())
    n, hit = osc.nearest_hit((0.1, 0.2, 6.0), (0.0, 0.0, -1.0))
    osc.close()
    assert n == 2 and abs(hit[0] - 5.25) < 1e-6                               # the box IS hit (entry at 5.25, exit behind it): it is the exit's normal that hides it


def test_white_fog_ball_conserves_energy(rt, oracle):
    """A fog ball with albedo 1 around a white solid: scattering redirects but never absorbs, and the sky is 1 in every direction, so every completed path carries
    exactly 1 (isotropic phase function with throughput = albedo, RT/Material.cs:163-168)."""
    s = S.Scene("white_fog")
    s.add_sphere((0.0, 0.0, 0.0), 2.0, S.volume((1.0, 1.0, 1.0), 1.3))
    s.add_sphere((0.0, 0.0, 0.0), 0.5, S.lambertian((1.0, 1.0, 1.0)))          # a solid inside the fog
    s.camera = {"position": [0.0, 0.5, 6.0], "target": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0], "vfov": 40.0, "aperture": 0.0}
    s.sky_bottom = s.sky_top = (1.0, 1.0, 1.0)
    out, _ = _render(oracle, s, 40, 30, spp=128, depth=64)
    c = out["color"]
    assert c[:, 3].mean() > 120
    assert np.array_equal(c[:, 0], c[:, 3]) and np.array_equal(c[:, 1], c[:, 3]) and np.array_equal(c[:, 2], c[:, 3])


def _next_floats(oracle, seed, n):
    import ctypes as C
    lib = oracle.load("strict")
    states = np.zeros(n, np.uint32)
    floats = np.zeros(n, np.float32)
    lib.oracle_kat_rng(seed, n, states.ctypes.data, floats.ctypes.data)
    return states, floats


@pytest.mark.parametrize("seed", [1, 700, 0x8C4CA03F ^ 0x7383ED49, 0xDEADBEEF])
def test_next_float_is_equidistributed(rt, oracle, seed):
    """2^20 NextFloat() draws of the restated Unity.Mathematics.Random: values are k / 2^23 in [0, 1), uniform over 256 bins (chi-squared, 255
    degrees of freedom), pairs of successive draws uniform over 64 x 64 cells (4095 degrees of freedom: NextFloat2, the jitter and both
    hemisphere samplers, consumes such pairs), mean and variance those of U(0, 1), and no draw repeats a state (period 2^32 - 1)."""
    n = 1 << 20
    states, f = _next_floats(oracle, seed, n)
    assert f.min() >= 0.0 and f.max() < 1.0
    assert np.array_equal(f * np.float32(1 << 23), np.floor(f * np.float32(1 << 23)))     # 23 random mantissa bits
    hist = np.bincount((f.astype(np.float64) * 256).astype(np.int64), minlength=256)
    chi2 = ((hist - n / 256.0) ** 2 / (n / 256.0)).sum()
    assert abs(chi2 - 255) < 5.0 * math.sqrt(2 * 255), chi2
    pairs = f.astype(np.float64).reshape(-1, 2)
    cell = (pairs[:, 0] * 64).astype(np.int64) * 64 + (pairs[:, 1] * 64).astype(np.int64)
    h2 = np.bincount(cell, minlength=4096)
    e = pairs.shape[0] / 4096.0
    chi2 = ((h2 - e) ** 2 / e).sum()
    assert abs(chi2 - 4095) < 5.0 * math.sqrt(2 * 4095), chi2
    fm = f.astype(np.float64)
    assert abs(fm.mean() - 0.5) < 5.0 * math.sqrt(1 / 12.0 / n) and abs(fm.var() - 1 / 12.0) < 5.0 * math.sqrt(1 / 180.0 / n)
    assert np.unique(states).size == n and states.min() != 0
    # lag-1 serial correlation of a uniform stream: 0 +- 1 / sqrt(n)
    r = np.corrcoef(fm[:-1], fm[1:])[0, 1]
    assert abs(r) < 5.0 / math.sqrt(n)


def test_cosine_hemisphere_and_uniform_direction_moments(rt, oracle):
    """The two direction samplers through Material.Scatter itself (oracle_kat_scatter: one Scatter call per draw on a lambert / a volume material):
    cosine-weighted directions have E[w] = 2/3 N and E[(w.N)^2] = 1/2 about ANY normal (the basis of UTIL/Tools.cs:19-37 must be orthonormal for
    that), isotropic directions E[w] = 0 and E[w w^T] = I / 3."""
    import ctypes as C
    lib = oracle.load("strict")
    fp = C.POINTER(C.c_float)
    n = 20000
    rng_state = C.c_uint32(12345)
    lam = S.lambertian((1.0, 1.0, 1.0))
    vol = S.volume((1.0, 1.0, 1.0), 1.0)
    for normal in ((0.0, 1.0, 0.0), (0.0, 0.0, -1.0), (0.6, -0.48, 0.64), (-0.70710678, 0.0, 0.70710678)):
        nn = np.array(normal, np.float64)
        nn /= np.linalg.norm(nn)
        dirs = np.zeros((n, 3))
        res = (C.c_float * 16)()
        for i in range(n):
            # a ray arriving along -N at the origin, hit normal N
            lib.oracle_kat_scatter(C.byref(lam), (C.c_float * 3)(*[float(x) for x in nn]), (C.c_float * 3)(*[float(-x) for x in nn]), 0.0,
                                   (C.c_float * 3)(0.0, 0.0, 0.0), (C.c_float * 3)(*[float(x) for x in nn]), 1.0, C.byref(rng_state), res)
            dirs[i] = (res[6], res[7], res[8])
        assert np.abs(np.linalg.norm(dirs, axis=1) - 1).max() < 1e-5
        cos = dirs @ nn
        assert cos.min() >= -1e-6
        # E[w] = 2/3 N: each component's standard error is <= 0.5 / sqrt(n)
        assert np.abs(dirs.mean(axis=0) - 2.0 / 3.0 * nn).max() < 5.0 * 0.5 / math.sqrt(n)
        assert abs((cos ** 2).mean() - 0.5) < 5.0 * 0.3 / math.sqrt(n)
    dirs = np.zeros((n, 3))
    res = (C.c_float * 16)()
    for i in range(n):
        lib.oracle_kat_scatter(C.byref(vol), (C.c_float * 3)(0.0, 1.0, 0.0), (C.c_float * 3)(0.0, -1.0, 0.0), 0.0, (C.c_float * 3)(0.0, 0.0, 0.0),
                               (C.c_float * 3)(0.0, 1.0, 0.0), 1.0, C.byref(rng_state), res)
        dirs[i] = (res[6], res[7], res[8])
    assert np.abs(np.linalg.norm(dirs, axis=1) - 1).max() < 1e-5
    assert np.abs(dirs.mean(axis=0)).max() < 5.0 * 0.58 / math.sqrt(n)
    m2 = dirs.T @ dirs / n
    assert np.abs(m2 - np.eye(3) / 3).max() < 5.0 * 0.3 / math.sqrt(n)


def _scatter(lib, mat, direction, normal, state):
    import ctypes as C
    res = (C.c_float * 16)()
    d = [float(x) for x in direction]
    n = [float(x) for x in normal]
    lib.oracle_kat_scatter(C.byref(mat), (C.c_float * 3)(*[-x for x in d]), (C.c_float * 3)(*d), 0.0, (C.c_float * 3)(0.0, 0.0, 0.0), (C.c_float * 3)(*n), 1.0,
                           C.byref(state), res)
    return np.array(res[:16], np.float64)


def test_smooth_glass_obeys_snell_and_the_mirror_law_with_schlick_frequencies(rt, oracle):
    """Optics, written down independently of the oracle: a smooth dielectric of index 1.5 sends a ray either along the mirror direction d - 2 (d.N) N or along Snell's
    refraction (coplanar with d and N, sin t = sin i / 1.5 entering, sin t = 1.5 sin i leaving, nothing beyond the critical angle asin(1 / 1.5) = 41.8 deg), and reflects with
    Schlick's probability r0 + (1 - r0)(1 - c)^5, r0 = ((1 - n) / (1 + n))^2 = 0.04, where c is the cosine of incidence entering and - the first book's convention, which
    the reference keeps (RT/Material.cs: cosine = IndexOfRefraction * dot(direction, normal)) - n times it leaving."""
    import ctypes as C
    lib = oracle.load("strict")
    glass = S.dielectric(1.5)
    n_ior = 1.5
    N = np.array((0.0, 1.0, 0.0))
    state = C.c_uint32(2024)
    draws = 6000
    for entering in (True, False):
        for deg in (0.0, 20.0, 35.0, 41.0, 43.0, 60.0, 80.0):
            th = math.radians(deg)
            # the ray travels in the x-y plane; entering: towards -N (from above), leaving: towards +N (from inside, the stored normal points outward)
            d = np.array((math.sin(th), -math.cos(th) if entering else math.cos(th), 0.0))
            mirror = d - 2.0 * (d @ N) * N
            sin_t = math.sin(th) / n_ior if entering else math.sin(th) * n_ior
            total_internal = sin_t >= 1.0
            if not total_internal:
                cos_t = math.sqrt(1.0 - sin_t * sin_t)
                refr = np.array((sin_t, -cos_t if entering else cos_t, 0.0))
            c = math.cos(th) if entering else n_ior * math.cos(th)
            r0 = ((1.0 - n_ior) / (1.0 + n_ior)) ** 2
            schlick = r0 + (1.0 - r0) * (1.0 - c) ** 5
            reflected = 0
            for _ in range(draws):
                out = _scatter(lib, glass, d, N, state)
                w = out[6:9]
                assert abs(np.linalg.norm(w) - 1.0) < 2e-6
                if np.abs(w - mirror).max() < 2e-6:
                    reflected += 1
                    assert tuple(out[0:3]) == (1.0, 1.0, 1.0)
                else:
                    assert not total_internal, (entering, deg)
                    assert np.abs(w - refr).max() < 3e-6, (entering, deg, w, refr)
            if total_internal:
                assert reflected == draws
            else:
                p = min(max(schlick, 0.0), 1.0)                    # (leaving near the critical angle the book's cosine exceeds 1 and the 'probability' with it: then every draw refracts... or none)
                if schlick >= 1.0:
                    assert reflected == draws
                elif schlick <= 0.0:
                    assert reflected == 0
                else:
                    assert abs(reflected / draws - p) < 5.0 * math.sqrt(p * (1.0 - p) / draws) + 1e-9, (entering, deg, reflected / draws, p)


def test_a_polished_metal_is_a_mirror_about_any_normal(rt, oracle):
    """Standard material, metallic 1, glossiness 1 (IsPerfectSpecular): every scattered ray leaves along d - 2 (d.N) N exactly, tinted by the albedo or - the glossy
    lobe - untinted; nothing diffuse."""
    import ctypes as C
    lib = oracle.load("strict")
    albedo = (0.8, 0.6, 0.2)
    mirror_mat = S.metal(albedo, 0.0)
    state = C.c_uint32(77)
    rng = np.random.default_rng(5)
    tinted = untinted = 0
    for _ in range(400):
        N = rng.normal(size=3)
        N /= np.linalg.norm(N)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if d @ N > -0.05:
            d = d - 2.0 * (d @ N) * N if d @ N > 0.05 else -N          # arrive from the normal's side
        N32 = np.array(N, np.float32).astype(np.float64)
        d32 = np.array(d, np.float32).astype(np.float64)
        out = _scatter(lib, mirror_mat, d32, N32, state)
        assert out[12] == 1.0                                           # IsPerfectSpecular
        want = d32 - 2.0 * (d32 @ N32) * N32
        assert np.abs(out[6:9] - want).max() < 3e-6
        refl = tuple(np.round(out[0:3], 6))
        if refl == (1.0, 1.0, 1.0):
            untinted += 1
        else:
            assert np.abs(out[0:3] - np.array(albedo, np.float32).astype(np.float64)).max() < 1e-7
            tinted += 1
    assert tinted > 0 and untinted > 0
