// glm_tile.h -- geometry of the per-run design table of k_glm_tile (glm_tile.hip), shared with the host code that builds it
#pragma once

template <int Q> struct TileCfg {
    static constexpr int K1 = Q + 1;                       // columns of the variant-independent design [1, z_1 .. z_q]
    static constexpr int KK = (K1 + 3) / 4;                // k-steps of the eta MFMA (16x16x4)
    static constexpr int NP = K1 * (K1 + 1) / 2;           // pairs (a >= b) of those columns: index a (a + 1) / 2 + b
    static constexpr int NCB = (NP + 15) / 16;             // 16-column blocks of the products table
    static constexpr int NC = NCB * 16;
    // one 16-sample tile, in doubles:  Zt [16 samples][16 cols] | ZtT [16 cols][16 samples] | ZZ32 [16 samples][NC] floats | ZZ64 [16][NC]
    static constexpr int OFF_ZT = 0, OFF_ZTT = 256, OFF_ZZ32 = 512, OFF_ZZ64 = 512 + 8 * NC, TILE_D = 512 + 8 * NC + 16 * NC;
    static constexpr int TBUF = 512 + 16 * NC;             // doubles of one LDS table buffer (head + the products in either precision)
    static constexpr int GST = 32 + NC + 5;                // doubles per variant row of the gather area (odd: conflict-free column reads)
};
