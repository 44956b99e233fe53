"""Development helper: one soak case (GPU against the oracle, every pixel) with the differing pixels printed in full.
    python tests/soak_one.py <scene> <w> <h> <spp> <depth> [seed]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rt = importlib.import_module("raytracing-in-one-weekend_amd")
from oracle import binding as oracle  # noqa: E402


def main():
    name, w, h, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    seed = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    S = rt.scenes
    scene = {"twin_moving": lambda: S.twin_spheres_scene(True), "twin": S.twin_spheres_scene, "moving": S.moving_scene, "cover": S.cover_scene}[name]()
    desc = scene.desc()
    ctx = rt.Context(int(os.environ.get("SOAK_DEVICE", "0")), flags=int(os.environ.get("SOAK_FLAGS", "0")))
    ctx.upload_scene(desc)
    p = S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=seed)
    gpu = rt.sample_batch_host(ctx, p)
    osc = oracle.OracleScene(desc)
    ref = osc.sample_batch(p)
    bad = np.zeros(w * h, dtype=bool)
    for k in ("color", "normal", "albedo", "scw"):
        bad |= np.any(gpu[k].view(np.uint32).reshape(w * h, -1) != ref[k].view(np.uint32).reshape(w * h, -1), axis=1)
    bad |= gpu["diag"][:, 0] != ref["diag"][:, 0]
    idx = np.nonzero(bad)[0]
    print("lib", os.environ.get("RTOW_LIB_PATH", "product"), "differing pixels:", len(idx), idx[:10].tolist())
    for i in idx[:4]:
        for k in ("color", "normal", "albedo", "scw"):
            print("  pixel", int(i), k, "gpu", gpu[k].reshape(w * h, -1)[i].tolist(), "ref", ref[k].reshape(w * h, -1)[i].tolist())
        print("  pixel", int(i), "rays gpu", gpu["diag"][i, 0], "ref", ref["diag"][i, 0])
        # which sample count first goes wrong: re-render this pixel's frame at increasing spp (the stream is sequential per pixel)
        for s in range(1, spp + 1):
            q = S.make_params(scene, w, h, spp=s, trace_depth=depth, seed=seed)
            r1 = osc.sample_pixels(q, np.array([i], dtype=np.int32))
            g1 = rt.sample_batch_host(ctx, q)
            if any(not np.array_equal(g1[k].reshape(w * h, -1)[i].view(np.uint32), r1[k].reshape(1, -1)[0].view(np.uint32)) for k in ("color", "normal", "albedo", "scw")):
                print("  first differing sample count:", s, "gpu colour", g1["color"][i].tolist(), "ref", r1["color"][0].tolist(), "gpu rays", g1["diag"][i, 0], "ref", r1["diag"][0, 0])
                break
    osc.close()


if __name__ == "__main__":
    main()
