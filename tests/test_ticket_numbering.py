"""The numbering of a launch's owned pixels (csrc/rtow_kernels.h: owned_pixel_xy - 8 x 8 tiles where the width is a multiple of 8, the rows behind
the last whole tile row and every other frame row by row) compiled for the host from the very same text: a bijection onto the owned pixels for
every frame shape tried, chunks of 64 tickets = one tile inside the tiled region.  The device side of the same function is exercised by
tests/test_gpu_parity.py::test_ticket_numbering_in_tiles_reaches_every_owned_pixel and by every full-frame test."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def numbering():
    hdr = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc", "rtow_kernels.h")
    text = open(hdr).read()
    block = text[text.index("// [ticket numbering: begin]"):text.index("// [ticket numbering: end]")]
    out_dir = os.path.join(ROOT, "tests", "build")
    os.makedirs(out_dir, exist_ok=True)
    inc = os.path.join(out_dir, "ticket_numbering_extracted.inc")
    if not os.path.exists(inc) or open(inc).read() != block:
        open(inc, "w").write(block)
    so = os.path.join(out_dir, "libticket_numbering_host.so")
    src = os.path.join(ROOT, "tests", "native", "ticket_numbering_host.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(inc)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-fPIC", "-shared", src, "-o", so], check=True, capture_output=True)
    lib = C.CDLL(so)
    lib.ticket_numbering_check.argtypes = [C.c_uint, C.c_uint, C.c_int]
    lib.ticket_numbering_check.restype = C.c_longlong
    return lib


@pytest.mark.parametrize("width,rows", [(1920, 1080), (1920, 135), (1920, 540), (3840, 2160), (3840, 270), (400, 225), (64, 24), (64, 27), (72, 13), (8, 64), (8, 7),
                                        (70, 16), (136, 5), (33, 19), (1, 1), (7, 300), (1280, 720), (1280, 90), (2560, 1440), (96, 54)])
def test_owned_pixel_numbering_is_a_bijection(numbering, width, rows):
    assert numbering.ticket_numbering_check(width, rows, 1) == 0
    assert numbering.ticket_numbering_check(width, rows, 0) == 0          # RTOW_TICKET_TILES = 0 builds: row by row
