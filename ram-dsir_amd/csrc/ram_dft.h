// ram_dft.h -- the Random Amplitude Mixup passes as PRUNED DFTs ON THE MATRIX CORES (ram_dft.hip), called from rd_ram_mix (ram_fft.hip)
// when the caller has provided the coefficient tables (rd_ram_t.dft_tables, built once per geometry by rd_ram_dft_tables).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct RamDftGeom {
    int H, W, b;
    int ntile;          // 16-bin tiles of the kept row bins kx = 0..b
    int nks_w;          // 16-pixel k-steps of a row
};
inline RamDftGeom ram_dft_geom(int H, int W, int b) {
    RamDftGeom g;
    g.H = H; g.W = W; g.b = b;
    g.ntile = (b + 1 + 15) / 16;
    g.nks_w = W / 16;
    return g;
}
// the geometry can run on the matrix path at all
inline bool ram_dft_ok(int H, int W, int b) { return W % 16 == 0 && W <= 1024 && H <= 1024 && b >= 0; }

// byte offsets of the tables inside rd_ram_t.dft_tables
struct RamDftLayout {
    size_t row_fwd;     // [ntile][nks_w][3 terms][64 lanes] uint4
    size_t total;
};
inline RamDftLayout ram_dft_layout(const RamDftGeom& g) {
    RamDftLayout l;
    l.row_fwd = 0;
    l.total = (size_t)g.ntile * g.nks_w * 3 * 64 * 16;
    return l;
}

// pass A for uint8 images: rowspec[img][c][y][kx] (kx < KP) of `nimg` images (the B sources, then the B partners)
int ram_dft_row_fwd(const void* src, const void* trg, int B, int nimg, int H, int W, int b, int KP, float2* rowspec,
                    const void* tables, hipStream_t st);

