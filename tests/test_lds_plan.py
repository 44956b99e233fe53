"""CPU: the per-launch LDS plan of the sample kernel (csrc/rtow_kernels.h planLds / historyWords, round 6) held to its contract without a GPU.

The kernel's LDS is `[8 candidate rows][a traversal-stack row per inner level of the scene's tree][wide codes: 256 B][path-history rows][queues 384 B][scene image]`;
the host decides every size per launch and the launcher refuses a launch whose plan does not belong to its variant.  What must hold for every scene and trace depth:
the plan fits the CU's 160 KB, regions do not overlap, the deep-history variants get `traceDepth - 8` rows, and a scene that no longer fits whole degrades to a staged
tree top instead of failing."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim():
    csrc = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    src, so = os.path.join(ROOT, "tests", "native", "lds_plan_shim.cpp"), os.path.join(out_dir, "liblds_plan_shim.so")
    deps = [src] + [os.path.join(csrc, n) for n in ("rtow_kernels.h", "rtow_scene.h", "rtow_bvh.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-fPIC", "-shared", "--offload-host-only", "-x", "hip", src, "-o", so], check=True, capture_output=True)
    return C.CDLL(so)


def plan(shim, wide, depth, total, nodes, hist, budget=0):
    out = (C.c_uint * 8)()
    shim.shim_plan_lds(int(wide), depth, total, nodes, hist, budget, out)
    return dict(zip(("stackRows", "histOffset", "histRows", "frontBytes", "sceneBytes", "nodeCount", "allLds", "histSpillRows"), [int(x) for x in out]))


def test_cover_scene_plans(shim):
    k = (C.c_int * 5)()
    shim.shim_constants(k)
    lds_max, queue, cand, lanes, in_regs = [int(x) for x in k]
    assert (lds_max, queue, cand, lanes, in_regs) == (160 * 1024, 384, 8, 1024, 8)
    cover = dict(depth=11, total=73824, nodes=485)
    p = plan(shim, False, cover["depth"], cover["total"], cover["nodes"], 0)
    assert p["stackRows"] == 11 and p["frontBytes"] == (8 + 11) * 1024 * 2 and p["histOffset"] == 0 and p["allLds"] == 1 and p["sceneBytes"] == 73824
    # the reference host's committed trace depth: 24 history rows behind the stack, the scene still whole in LDS (the reason the rows are counted from the tree's own depth)
    p = plan(shim, False, cover["depth"], cover["total"], cover["nodes"], 32 - in_regs)
    assert p["histOffset"] == (8 + 11) * 2048 and p["frontBytes"] == p["histOffset"] + 24 * 2048 and p["allLds"] == 1
    assert p["frontBytes"] + queue + p["sceneBytes"] <= lds_max
    # round 5's fixed 24 stack rows would not have left room: 24 levels + 24 history rows + the scene exceed the CU
    assert (8 + 24) * 2048 + 24 * 2048 + queue + cover["total"] > lds_max
    assert p["histSpillRows"] == 0
    # the deepest paths the API allows: 56 rows do not fit next to the whole scene, and the scene-in-LDS kernels carry no code for rows elsewhere - the top of the tree is
    # staged instead (at least 256 nodes), the history gets the rows that fit, the rest of the rows live in HBM
    p = plan(shim, False, cover["depth"], cover["total"], cover["nodes"], 64 - in_regs)
    assert p["allLds"] == 0 and p["nodeCount"] >= 256 and p["histRows"] + p["histSpillRows"] == 56 and p["histSpillRows"] > 0
    assert p["frontBytes"] + queue + p["sceneBytes"] <= lds_max
    # the benchmark mesh (250 881 nodes, 21 levels, wide codes) at the reference host's committed trace depth 32: round 6's first plan had no room for 24 rows next to
    # 29 x 4 KB of stack and candidate rows and refused the launch - the plan now keeps the top 256 nodes and spills the rows that do not fit
    p = plan(shim, True, 21, 86 << 20, 250881, 32 - in_regs)
    assert p["allLds"] == 0 and p["nodeCount"] >= 256 and p["histRows"] + p["histSpillRows"] == 24 and p["histSpillRows"] > 0
    assert p["frontBytes"] + queue + p["sceneBytes"] <= lds_max


@pytest.mark.parametrize("wide", [False, True])
def test_every_plan_fits_and_is_ordered(shim, wide):
    lds_max, queue = 160 * 1024, 384
    for depth in (1, 2, 7, 11, 16, 21, 24):
        for hist in (0, 1, 9, 24, 32, 56):
            for total, nodes in ((64, 1), (73824, 485), (5 << 20, 9999), (86 << 20, 250881)):
                p = plan(shim, wide, depth, total, nodes, hist)
                code = 4 if wide else 2
                rows_end = (8 + depth) * 1024 * code + (256 if wide else 0)
                assert p["stackRows"] == depth
                assert p["histRows"] + p["histSpillRows"] == hist and p["histOffset"] == (rows_end if p["histRows"] else 0)
                assert p["frontBytes"] == rows_end + p["histRows"] * 2048
                assert p["frontBytes"] + queue + p["sceneBytes"] <= lds_max, (wide, depth, hist, total, p)
                assert p["nodeCount"] >= 1, "at least the root is staged: the walk starts in LDS"
                if p["allLds"]:
                    assert not wide and p["sceneBytes"] == total and p["nodeCount"] == nodes
                else:
                    assert p["sceneBytes"] == 64 * p["nodeCount"] and p["nodeCount"] <= nodes
                    assert p["nodeCount"] >= min(nodes, 256), "the history never takes the top of a tree that is beyond LDS"
                # a scene kept whole has every row in LDS (the scene-in-LDS kernels have no code for rows in HBM)
                if p["allLds"]:
                    assert p["histSpillRows"] == 0


def test_lds_budget_override_is_a_development_aid_only_downwards(shim):
    p = plan(shim, False, 11, 73824, 485, 0, budget=1024)
    assert p["allLds"] == 0 and p["nodeCount"] == 16
    q = plan(shim, False, 11, 73824, 485, 0, budget=10 << 20)
    assert q["allLds"] == 1


def test_history_width_rule(shim):
    hw = lambda **kw: shim.shim_history_words(kw.get("noise", 0), kw.get("per_sample", 0), kw.get("wide", 0), kw.get("ties", 0), kw.get("full_diag", 0), kw["depth"])
    assert [hw(depth=d) for d in (1, 8, 9, 16, 17, 32, 64)] == [4, 4, 8, 8, 32, 32, 32]
    assert hw(depth=5, full_diag=1) == 32 and hw(depth=5, noise=1) == 32 and hw(depth=5, noise=2) == 32
    assert hw(depth=8, per_sample=1) == 4 and hw(depth=9, per_sample=1) == 32 and hw(depth=8, per_sample=1, ties=1) == 32 and hw(depth=8, per_sample=1, wide=1) == 32
    assert hw(depth=8, wide=1) == 4 and hw(depth=12, wide=1) == 32 and hw(depth=12, wide=1, ties=1) == 8 and hw(depth=20, wide=1, ties=1) == 32
    # a launch under the tie watch runs the rank-rule kernel first and the exact-tie kernel on the marked pixels with ONE plan: the wider of the two never needs fewer rows
    for depth in range(1, 65):
        for wide in (0, 1):
            assert max(hw(depth=depth, wide=wide), hw(depth=depth, wide=wide, ties=1)) in (4, 8, 32)
