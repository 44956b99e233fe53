// Test-only: owned_pixel_xy (raytracing-in-one-weekend_amd/csrc/rtow_kernels.h, the text between the "[ticket numbering: ...]" markers, extracted by
// tests/test_ticket_numbering.py) compiled for the host.  ticket_numbering_check returns 0 when the numbering of a width x rows frame visits every
// owned pixel exactly once and a chunk of 64 tickets inside the tiled region is one 8 x 8 tile; otherwise the number of the first offending ticket + 1.
#include <cstddef>
#include <cstdint>
#include <vector>
#define __host__
#define __device__
#include "../build/ticket_numbering_extracted.inc"

extern "C" long long ticket_numbering_check(unsigned width, unsigned rows, int tiles)
{
    const unsigned tilesPerRow = (tiles && width % kTileW == 0u) ? width / kTileW : 0u;
    const unsigned tiledPixels = tilesPerRow ? (rows / kTileH) * kTileH * width : 0u;
    std::vector<uint8_t> seen((std::size_t)width * rows, 0);
    for (unsigned n = 0; n < width * rows; n++) {
        int cx = -1, row = -1;
        owned_pixel_xy(n, width, tilesPerRow, tiledPixels, cx, row);
        if (cx < 0 || row < 0 || (unsigned)cx >= width || (unsigned)row >= rows) return (long long)n + 1;
        if (seen[(std::size_t)row * width + cx]++) return (long long)n + 1;
        if (n < tiledPixels) {
            int cx0, row0;
            owned_pixel_xy(n & ~63u, width, tilesPerRow, tiledPixels, cx0, row0);        // the chunk's first ticket: the tile's corner
            if (cx - cx0 < 0 || cx - cx0 >= (int)kTileW || row - row0 < 0 || row - row0 >= (int)kTileH) return (long long)n + 1;
        } else if ((unsigned)row * width + (unsigned)cx != n) {
            return (long long)n + 1;                                                     // behind the tiles: row-major numbers
        }
    }
    return 0;
}
