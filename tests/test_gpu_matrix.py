"""GPU: the rarely combined switches, combined.  Every defect of round 2 sat in a corner no test had walked into (a chain of the exact-tie
kernel with 16-byte diagnostics, a 25-entry hit list, a tie on a ray through a mesh edge), so this file walks the corners systematically:
scenes that live in the slow paths (spilled hit lists, the exact-tie procedure, ties on most rays) x diagnostics record size x tree in
LDS / in HBM x RNG policy x one batch / a chain of three through the host-buffer entry points, each against the oracle's batches one after
the other, on frames small enough for the oracle to finish all 240 in about a minute."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SCENES = {
    "volume_stack_48": (lambda S: S.volume_stack_scene(48, 0.125), 10),       # ~100 hits per camera ray: hit lists spill to HBM
    "decal_stack": (lambda S: S.decal_stack_scene(20), 6),                    # exact-tie kernels by scene size, a tie on most rays, 23 hits
    "twin_row": (lambda S: S.twin_row_scene(30, True), 6),                    # exact-tie kernels by duplicates, moving, up to 60 hits per tied ray
    # ... and one scene per remaining kernel family, so that the same corners are walked in the fast paths too
    "cover_60": (lambda S: S.cover_scene(60, 600), 8),                        # spheres
    "tiny_moving": (lambda S: S.tiny_scene(), 8),                             # moving spheres
    "mixed": (lambda S: S.mixed_scene(), 6),                                  # general entities, rank rule (14 entities)
    "mesh": (lambda S: S.mesh_scene(1), 6),                                   # triangles, exact-tie kernels by scene size
    "textured": (lambda S: S.textured_scene(), 6),                            # image textures
    "volumes": (lambda S: S.volume_scene(), 10),                              # fog, short hit lists
    "textured_volumes": (lambda S: S.textured_volume_scene(), 8),
}


@pytest.mark.parametrize("chain", [1, 3])
@pytest.mark.parametrize("policy", ["reference", "per_sample", "xoroshiro"])
@pytest.mark.parametrize("tree", ["lds", "hbm"])
@pytest.mark.parametrize("stride", [4, 16])
@pytest.mark.parametrize("name", sorted(SCENES))
def test_slow_path_scenes_under_every_switch(rt, oracle, name, stride, tree, policy, chain):
    S, a = rt.scenes, rt.abi
    make, depth = SCENES[name]
    scene = make(S)
    desc = scene.desc()
    w, h, spp = 48, 40, 3
    plist = [S.make_params(scene, w, h, spp=spp, trace_depth=depth, seed=31 + 5 * k, diagnostics_stride=stride,
                           rng_policy={"reference": a.RNG_REFERENCE, "per_sample": a.RNG_PER_SAMPLE, "xoroshiro": a.RNG_PER_SAMPLE_XOROSHIRO}[policy]) for k in range(chain)]
    osc = oracle.OracleScene(desc)
    ref, ref_diags = None, []
    for p in plist:
        ref = osc.sample_batch(p, None if ref is None else {k: ref[k] for k in ("color", "normal", "albedo", "scw")})
        ref_diags.append(ref["diag"])
    osc.close()
    with rt.Context(0, lds_scene_budget=64 if tree == "hbm" else 0) as ctx:
        ctx.upload_scene(desc)
        assert bool(ctx.scene_info().sceneInLds) == (tree == "lds")
        if chain == 1:
            gpu = rt.sample_batch_host(ctx, plist[0])
            gpu_diags = [gpu["diag"]]
        else:
            gpu = rt.sample_batch_chain_host(ctx, plist)
            gpu_diags = gpu["diag"]
    for k in ("color", "normal", "albedo", "scw"):
        assert np.array_equal(gpu[k].view(np.uint32), ref[k].view(np.uint32)), (k, int(np.any((gpu[k].view(np.uint32) != ref[k].view(np.uint32)).reshape(w * h, -1), axis=1).sum()))
    for b, (g, r) in enumerate(zip(gpu_diags, ref_diags)):
        assert np.array_equal(g[:, 0], r[:, 0]), ("ray counts of batch", b)
    assert ref["color"][:, 3].sum() > 0
