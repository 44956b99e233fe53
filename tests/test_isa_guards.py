"""CPU: properties of the COMPILED hot kernels, checked in the build container (hipcc cross-compiles gfx950 without a GPU).

The sample kernel lives at the edge of the register file (DESIGN.md 4.1 "Registers", 5.3): the headline variants are spill-free at 121-126
VGPRs, and the chained path's wide stores are inline assembly that the compiler's hazard recogniser cannot see into.  A ROCm bump, or an edit
that moves the allocation, can silently undo either - and the next GPU run would only show "slower" or, worse, a wrong albedo record in one
kernel under a chain.  So every CPU-suite run recompiles the two sphere translation units (~20 s each, in parallel) and asserts:

  * resource usage (-Rpass-analysis=kernel-resource-usage) of the reference-stream variants at trace depth <= 8 / <= 16, tree in LDS and in
    HBM: no VGPR spill, at most 128 VGPRs at four waves per SIMD, no scratch at all for the static-sphere kernels
    with the scene in LDS (the headline), at most the known 36 / 68 bytes elsewhere;
  * ISA (-save-temps): inside those kernels no scratch_ instruction at all; in EVERY kernel of the unit no write-through (`sc1`) store is left -
    round 2's chained path stored its accumulators with inline-asm `global_store_dwordx3/x4 ... sc1`, each needing an `s_nop 1` behind it (a VMEM
    store of more than 8 bytes still reads its data registers for up to two wait states after issue on gfx940+; LLVM pads the stores it emits
    itself - GCNHazardRecognizer::checkVALUHazardsHelper, VALUWaitStates = 2 - and cannot pad an asm statement); round 3 keeps a chunk's batches on
    one XCD and stores plainly, so the hazard family is gone and must stay gone - and each wide `sc1` LOAD (still inline asm) waits for its own
    data (`s_waitcnt vmcnt(0)`) before anything uses it.
"""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc")
FLAGS = ["-std=c++17", "-O3", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-x", "hip"]   # csrc/Makefile's


def _compile(unit, out_dir):
    src = os.path.join(CSRC, unit + ".hip")
    proc = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-c", "-save-temps=obj", "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(out_dir, unit + ".o")],
                          capture_output=True, text=True, cwd=CSRC)
    assert proc.returncode == 0, proc.stderr[-3000:]
    asm = [f for f in os.listdir(out_dir) if f.startswith(unit) and f.endswith(".s") and "amdgcn" in f]
    assert len(asm) == 1, os.listdir(out_dir)
    return proc.stderr, open(os.path.join(out_dir, asm[0])).read()


def _usage(remarks):
    """mangled kernel name -> {field: int} from the kernel-resource-usage remarks"""
    out = {}
    for block in re.split(r"remark: (?:[^\n]*?: )?Function Name: ", remarks)[1:]:      # "remark: Function Name:" or "remark: file:line:col: Function Name:" (-save-temps)
        name = block.split(" [")[0].strip()
        fields = {}
        for key, tag in (("vgprs", r"\bVGPRs"), ("agprs", "AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occupancy", r"Occupancy \[waves/SIMD\]"),
                         ("sgpr_spill", "SGPRs Spill"), ("vgpr_spill", "VGPRs Spill")):
            m = re.search(tag + r": (\d+)", block)
            assert m, (name, key)
            fields[key] = int(m.group(1))
        out[name] = fields
    return out


def _bodies(asm):
    """mangled kernel name -> list of instruction lines (comments and directives dropped)"""
    out = {}
    for m in re.finditer(r"^(_ZN4rtow[^\n:]*sample_batch_kernel[^\n:]*):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M):
        lines = []
        for line in m.group(2).splitlines():
            line = line.split(";")[0].strip()
            if line and not line.startswith(".") and not line.endswith(":"):
                lines.append(line)
        out[m.group(1)] = lines
    return out


def _variant(name):
    """template arguments of sample_batch_kernel<ALL_LDS, KIND, HW, DIAG, NOISE, PER_SAMPLE, GEO> out of the mangled name"""
    m = re.search(r"sample_batch_kernelILb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)EE", name)
    assert m, name
    return tuple(int(x) for x in m.groups())


@pytest.fixture(scope="module")
def compiled():
    with tempfile.TemporaryDirectory() as tmp:
        dirs = {u: os.path.join(tmp, u) for u in ("rtow_sample_spheres", "rtow_sample_spheres_motion")}
        for d in dirs.values():
            os.makedirs(d)
        with ThreadPoolExecutor(2) as pool:
            res = list(pool.map(lambda u: _compile(u, dirs[u]), dirs))
    return {u: r for u, r in zip(dirs, res)}


@pytest.mark.parametrize("unit,kind", [("rtow_sample_spheres", 0), ("rtow_sample_spheres_motion", 1)])
def test_hot_variants_stay_spill_free(compiled, unit, kind):
    remarks, asm = compiled[unit]
    usage = _usage(remarks)
    bodies = _bodies(asm)
    hot = deep = 0
    for name, u in usage.items():
        if "sample_batch_kernel" not in name:
            continue
        all_lds, k, hw, full_diag, noise, per_sample, geo = _variant(name)
        assert k == kind
        assert (geo & 3) == 0, name                                                         # one launch geometry since round 4: 1024 lanes per workgroup
        assert u["vgprs"] <= 128 and u["agprs"] == 0 and u["occupancy"] >= 4, (name, u)     # the workgroup must fit the CU in EVERY variant (128 VGPRs at four waves per SIMD)
        if hw in (4, 8) and not full_diag and noise == 0 and not per_sample and not (geo & 4):
            # the reference stream at depth <= 8 / <= 16: the benchmark's kernels
            hot += 1
            assert u["vgpr_spill"] == 0 and u["sgpr_spill"] <= 32, (name, u)      # SGPRs spill into VGPR lanes: 21 - 27 since round 5 read the cubemap's launch constants on use (29 - 33 before)
            # not one scratch instruction and no private segment in any of them (round 5: the cubemap's constants are read on use - until then 36 / 68 bytes were reserved)
            assert u["scratch"] == 0, (name, u)
            assert not [l for l in bodies[name] if l.startswith("scratch_")], name
        if hw == 32 and full_diag in (0, 1) and noise == 0 and not per_sample:
            # the generic reference-stream kernels: depth 17 .. 64, and - DIAG 1 - the 16-byte FULL_DIAGNOSTICS records = the reference host's COMMITTED configuration (traceDepth 32,
            # Assets/Prefabs/Raytracer.prefab:383-395; ProjectSettings.asset:590).  Round 6: their path history beyond depth 8 lives in LDS rows, the reference-tree counter walk in a
            # variant of its own (DIAG 2) - no private segment (rounds 1 - 5: 448 bytes per lane, 186 GB of HBM writes per 10-batch launch), no spilled VGPR
            deep += 1
            if geo & 16:
                # the twins with the lanes in a hurry (plain and chained launches of the same configuration; round 6): nothing spilled, not one scratch instruction; with the
                # scene in LDS they keep the register allocator's 36 bytes of dead slots (a 32-byte spill slot it does not use + the emergency slot) - they run those launches
                # 17 - 20 % faster than the variants without the code, which batch groups keep
                assert u["vgpr_spill"] == 0 and u["scratch"] <= (36 if all_lds else 0), (name, u)
                assert not [l for l in bodies[name] if l.startswith("scratch_")], name
            elif geo & 4:
                # wide codes (scenes beyond 65 535 entities): the code for history rows in HBM costs the moving-sphere kernel four spilled VGPRs (12 bytes); nothing more
                assert u["vgpr_spill"] <= 4 and u["scratch"] <= 16, (name, u)
            else:
                # (the moving-sphere kind with its scene in LDS keeps 36 bytes of dead spill slots - nine of the view's constants, never touched: not one scratch instruction)
                assert u["vgpr_spill"] == 0 and u["scratch"] <= (36 if kind == 1 and all_lds else 0), (name, u)
                assert not [l for l in bodies[name] if l.startswith("scratch_")], name
    assert hot == (6 if kind == 0 else 4), hot               # 2 history widths x (LDS | HBM), + the pinhole twins of the static-sphere kernels whose tree is beyond LDS (GEO bit 3)
    assert deep == (9 if kind == 0 else 5), deep                                   # DIAG 1: LDS, HBM, HBM with wide codes; DIAG 0: LDS, HBM (wide codes serve every deeper launch from DIAG 1: launchByDiagGeo); + the four
                                                             # twins with the lanes in a hurry of the static-sphere kind (GEO bit 4: LDS | HBM x DIAG 0 | 1)
    headline = [u for n, u in usage.items() if "sample_batch_kernel" in n and _variant(n) == (1, kind, 4, 0, 0, 0, 0)]
    assert len(headline) == 1 and headline[0]["vgprs"] <= 128, headline      # (the allocator takes all 128 since round 5 - no spill, no scratch; 124 / 127 before)


@pytest.mark.parametrize("unit", ["rtow_sample_spheres", "rtow_sample_spheres_motion"])
def test_coherent_accesses_of_the_chained_path(compiled, unit):
    _, asm = compiled[unit]
    bodies = _bodies(asm)
    assert len(bodies) >= 20
    stores = loads = 0
    for name, lines in bodies.items():
        for i, line in enumerate(lines):
            if re.match(r"global_store_\w+\b.*\bsc1\b", line) and not re.search(r"\bsc0\b", line):
                stores += 1                                                        # (sc0 sc1 = system scope: the host-visible overflow / cancel flags)
            if re.match(r"global_load_dwordx[34]\b.*\bsc1\b", line):
                loads += 1
                assert re.match(r"s_waitcnt vmcnt\(0\)", lines[i + 1]), (name, line, lines[i + 1])
    # every reference-stream variant has the chained path: three wide loads per pixel (colour x4, normal and albedo x3), no write-through store
    assert stores == 0 and loads >= 3 * 4, (stores, loads)
