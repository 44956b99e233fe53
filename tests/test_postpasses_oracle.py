"""CPU tests of the oracle's post passes (CombineJob, FinalizeTexturesJob, ReduceMetricsJob) against numpy restatements."""
import numpy as np


def _inputs(seed=0, w=16, h=9):
    rng = np.random.default_rng(seed)
    n = w * h
    color = np.concatenate([rng.uniform(0, 40, (n, 3)), rng.integers(0, 12, (n, 1))], axis=1).astype(np.float32)
    color[rng.integers(0, n, 8), 3] = 0            # zero-sample pixels -> look-around
    color[5, 0] = np.nan                           # NaN handling
    normal = rng.normal(size=(n, 3)).astype(np.float32) * color[:, 3:4]
    albedo = rng.uniform(0, 2, (n, 3)).astype(np.float32) * np.maximum(color[:, 3:4], 1)
    return w, h, color, normal, albedo


def test_combine_matches_numpy(oracle):
    w, h, color, normal, albedo = _inputs()
    for debug in (False, True):
        for ldr in (False, True):
            oc, on, oa = oracle.combine(w, h, color, normal, albedo, debug, ldr)
            for i in range(w * h):
                c = color[i].copy()
                count = int(c[3])
                if not debug and count == 0:
                    j = i
                    while count == 0 and j - w >= 0:        # borrows the pixel BELOW (index -= Size.x, JOBS/CombineJob.cs:44)
                        j -= w
                        c = color[j].copy()
                        count = int(c[3])
                if count == 0:
                    want = (1, 0, 1) if debug else (0, 0, 0)
                elif np.isnan(c).any():
                    want = (0, 1, 1) if debug else (0, 0, 0)
                else:
                    want = c[:3] / np.float32(count)
                assert np.array_equal(oc[i], np.asarray(want, np.float32)), (i, debug)
                den = np.float32(max(count, 1))
                a = albedo[i] / den
                if ldr:
                    a = np.minimum(a, 1)
                assert np.array_equal(oa[i], a.astype(np.float32))
                nv = normal[i] / den
                ln = np.float32(nv[0] * nv[0] + nv[1] * nv[1]) + np.float32(nv[2] * nv[2])
                if ln > 1.175494351e-38:
                    assert np.allclose(on[i], nv / np.sqrt(ln), atol=1e-6)
                else:
                    assert np.all(on[i] == 0)


def test_finalize_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    n = 500
    color = rng.uniform(-0.2, 1.5, (n, 3)).astype(np.float32)
    normal = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    albedo = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    fc, fn, fa = oracle.finalize(color, normal, albedo)

    def gamma_byte(v):
        v = np.maximum(v.astype(np.float64), 0)
        g = np.maximum(1.055 * v ** float(np.float32(0.416666667)) - 0.055, 0)
        return np.clip(g, 0, 1) * 255

    for got, src in ((fc, color), (fn, normal * np.float32(0.5) + np.float32(0.5)), (fa, albedo)):
        want = gamma_byte(src)
        assert np.all(got[:, 3] == 255)
        diff = np.abs(got[:, :3].astype(np.float64) - np.floor(want))
        frac = want - np.floor(want)
        near_edge = (frac < 2e-3) | (frac > 1 - 2e-3)    # float32 pow vs float64 may straddle an integer boundary
        assert np.all((diff == 0) | (near_edge & (diff <= 1)))


def test_reduce_metrics_matches_numpy(rt, oracle):
    rng = np.random.default_rng(2)
    n = 1000
    diag = rng.integers(0, 50, (n, 1)).astype(np.float32)
    color = np.zeros((n, 4), np.float32)
    color[:, 3] = rng.integers(0, 9, n)
    scw = rng.uniform(0, 30, n).astype(np.float32)
    m = oracle.reduce_metrics(diag, color, scw)
    assert m.totalRayCount == int(diag.sum()) == m.totalRayCount64
    assert m.totalSamples == int(color[:, 3].sum()) == m.totalSamples64
    with np.errstate(divide="ignore", invalid="ignore"):
        w = scw / color[:, 3]
    assert m.sampleCountWeightExtrema.x == np.nanmin(w) and m.sampleCountWeightExtrema.y == np.nanmax(w)   # NaN (0/0) is skipped, inf is kept
    assert (m.sampleCountExtrema[0], m.sampleCountExtrema[1]) == (int(color[:, 3].min()), int(color[:, 3].max()))
    diag16 = np.zeros((n, 4), np.float32)
    diag16[:, 0] = diag[:, 0]
    assert oracle.reduce_metrics(diag16, color, scw).totalRayCount == m.totalRayCount
