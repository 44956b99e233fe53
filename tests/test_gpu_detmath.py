"""GPU: the device's deterministic sin/cos/log/pow (and IEEE / and sqrt) against the oracle's, bit for bit, over dense sweeps -
including EVERY value Unity's NextFloat() can return for log (ProbabilisticHit) and sincos(2*pi*u)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_detmath_matches_oracle_bit_for_bit():
    src = os.path.join(ROOT, "tests", "native", "detmath_parity.hip")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "detmath_parity")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-x", "hip", src, "-o", exe],
                       check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout


def test_exact_rcp_and_sqrt_equal_ieee_for_every_operand():
    """rtow::exact_rcp / exact_sqrt (csrc/rtow_exactmath.hip.h) are the path's `1 / x` and `sqrt(x)`: a hardware approximation plus one
    fused residual step where that is provably enough, the compiler's IEEE expansion elsewhere.  'Provably' = enumerated: all 2^32 float
    operands, on the device, against `1.0f / x`, `__builtin_sqrtf(x)` and their composition in normalize."""
    src = os.path.join(ROOT, "tests", "native", "exactmath_parity.hip")
    hdr = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc", "rtow_exactmath.hip.h")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "exactmath_parity")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-x", "hip", src, "-o", exe],
                       check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rcp 0 mismatches" in r.stdout and "sqrt 0 " in r.stdout and "rcp(sqrt) 0 " in r.stdout and "NaN -> inf 0 " in r.stdout, r.stdout


def test_exact_div3_equals_ieee_division():
    """rtow::exact_div3 (csrc/rtow_exactmath.hip.h): three quotients by one divisor from one correctly rounded reciprocal and one fused
    residual step each.  Binary division cannot be enumerated over operands, but whether that step gives the IEEE quotient depends on the
    mantissas only (inside the guarded exponent ranges everything scales by powers of two): all 2^23 x 2^23 mantissa pairs are checked on
    the device against `a / b`, then the function itself - guard and fallback included - on 2^32 arbitrary and 2^32 edge-exponent quadruples."""
    src = os.path.join(ROOT, "tests", "native", "exactdiv_parity.hip")
    hdr = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc", "rtow_exactmath.hip.h")
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "exactdiv_parity")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-x", "hip", src, "-o", exe],
                       check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=590)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mantissa pairs 70368744177664: 0 mismatches" in r.stdout and "random quadruples 4294967296: 0 mismatches" in r.stdout \
        and "edge-exponent quadruples 4294967296: 0 mismatches" in r.stdout, r.stdout


def test_finalize_byte_table_equals_the_pow_form_for_every_operand():
    """FinalizeTexturesJob's float -> byte conversion runs from a 255-step table (csrc/rtow_finalize.hip.h: hardware log2 / exp2 estimate, two
    comparisons) instead of nine deterministic pows per pixel.  Same byte for all 2^32 float operands - enumerated on the device against
    to_byte_exact, with the table built by the product's own kernel (the one stretch of floats where the polynomial pow is not monotone
    included: the table marks it and the kernel takes the exact form there)."""
    src = os.path.join(ROOT, "tests", "native", "finalize_parity.hip")
    csrc = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc")
    newest = max(os.path.getmtime(p) for p in (src, os.path.join(csrc, "rtow_finalize.hip.h"), os.path.join(csrc, "rtow_detmath.hip.h")))
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "finalize_parity")
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-x", "hip", src, "-o", exe],
                       check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "table vs exact 0 mismatches" in r.stdout, r.stdout
