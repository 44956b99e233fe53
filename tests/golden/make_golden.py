#!/usr/bin/env python3
"""Regenerates the committed golden vectors from the CPU oracle (strict build).

The reference ships no tests, golden images or known-answer vectors and cannot be executed here (Unity C# + Burst; no
dotnet/mono), so every fixture below is produced by the build's own oracle and justified by the line-by-line
correspondence documented in oracle/rtow_oracle.cpp.  "Parity unpinned" at the Unity.Mathematics / Burst boundary
(see oracle/README.md); what these fixtures pin is that the oracle - and therefore the HIP path checked against it -
does not drift between rounds.

  python tests/golden/make_golden.py          # rewrites tests/golden/*.json / *.npz
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracing-in-one-weekend_amd")
from oracle import binding as ob  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    lib = ob.load("strict")
    out = {}

    # (1) RNG known-answer vectors (assumed Unity.Mathematics.Random semantics)
    rng = {}
    for seed in (1, 700, 0x8C4CA03F, int(lib.oracle_kat_pixel_seed(1, 0)), int(lib.oracle_kat_pixel_seed(1, 12345))):
        st = np.zeros(16, np.uint32)
        fl = np.zeros(16, np.float32)
        lib.oracle_kat_rng(seed, 16, st.ctypes.data, fl.ctypes.data)
        rng[str(seed)] = {"states": [int(x) for x in st], "float_bits": [int(x) for x in fl.view(np.uint32)]}
    out["rng"] = rng

    # (5) scene fixtures
    cover = rt.scenes.cover_scene()
    with open(os.path.join(HERE, "cover_scene.json"), "w") as f:
        json.dump(cover.to_dict(), f, separators=(",", ":"))
    moving = rt.scenes.moving_scene()
    out["scenes"] = {
        "cover": {"entities": cover.entity_count, "sha256_positions": sha(np.stack(cover.positions)), "tentatives_used": cover.meta["tentatives_used"]},
        "moving": {"entities": moving.entity_count, "sha256_positions": sha(np.stack(moving.positions)), "moving": int(sum(moving.moving))},
    }

    # (6) image goldens: full AOV set of a 64x36 x 8 spp x depth 8 render + SHA-256 of config 1 (400x225)
    osc = ob.OracleScene(cover.desc())
    p = rt.scenes.make_params(cover, 64, 36, spp=8, trace_depth=8, diagnostics_stride=16)
    r = osc.sample_batch(p)
    np.savez_compressed(os.path.join(HERE, "cover_64x36_8spp_d8.npz"), color=r["color"], normal=r["normal"], albedo=r["albedo"], scw=r["scw"],
                        raycount=r["diag"][:, 0].copy())
    p1 = rt.scenes.make_params(cover, 400, 225, spp=8, trace_depth=8)
    r1 = osc.sample_batch(p1)
    out["config1_400x225_8spp_d8"] = {k: sha(r1[k]) for k in ("color", "normal", "albedo", "scw")}
    out["config1_400x225_8spp_d8"]["raycount"] = sha(r1["diag"][:, 0].copy())
    out["config1_400x225_8spp_d8"]["total_rays"] = float(r1["diag"][:, 0].sum())
    out["config1_400x225_8spp_d8"]["successful_samples"] = float(r1["color"][:, 3].sum())

    # (7) first-hit AOV golden: traceDepth 1, no jitter -> normal/albedo are the sample-0 fallbacks
    p2 = rt.scenes.make_params(cover, 96, 54, spp=1, trace_depth=1, jitter=False)
    r2 = osc.sample_batch(p2)
    np.savez_compressed(os.path.join(HERE, "cover_96x54_firsthit.npz"), normal=r2["normal"], albedo=r2["albedo"])
    osc.close()

    # moving scene + aperture (config 5 at reduced size)
    osm = ob.OracleScene(moving.desc())
    pm = rt.scenes.make_params(moving, 96, 54, spp=4, trace_depth=8)
    rm = osm.sample_batch(pm)
    out["moving_96x54_4spp_d8"] = {k: sha(rm[k]) for k in ("color", "normal", "albedo", "scw")}
    osm.close()

    # one small render per further feature (volumes, ties, textures, HDRI sky, texture-driven noise, adaptive sampling)
    sys.path.insert(0, HERE)
    from feature_cases import feature_cases, render_digests  # noqa: E402
    out["features"] = {name: render_digests(rt, ob, name, case) for name, case in feature_cases(rt).items()}

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
