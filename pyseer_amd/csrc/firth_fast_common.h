// firth_fast_common.h -- what the one-pass Firth kernels share (firth_fast.hip: one wavefront per SIMD, 32 x 32 x 16 tiles; firth_fast_w.hip:
// two wavefronts per SIMD, 16 x 16 x 32 tiles): the monomial tables' row numbering and the contraction of the accumulated sums with V.
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include "glm_device.h"

typedef _Float16 ff_v8h __attribute__((ext_vector_type(8)));
typedef _Float16 ff_v2h __attribute__((ext_vector_type(2)));
typedef float ff_v16f __attribute__((ext_vector_type(16)));
typedef float ff_v2f __attribute__((ext_vector_type(2)));
typedef uint32_t ff_v4u __attribute__((ext_vector_type(4)));
typedef double ff_v2d __attribute__((ext_vector_type(2)));

#define FF_SCALE 4096.0            /* weights are scaled by 2^12 before the hi / lo split (their lo parts stay normal halves); exact */
#ifndef FF_TAU
#define FF_TAU 1e-7                /* an increase of the one-pass F beyond this leaves for the exact rounds, whose comparison is the reference's own.  = the evaluation noise of the one-pass F (the matrix-core part of log det: ~1e-6 rho, rho <= 0.05).  Round 6: 1e-5 -> 1e-7 at the same rate (profiles/r06/ab_ff_tau.txt); the reference itself never halves on a rise between F's rounding noise and 1e-5 (profiles/r06/firth_subtau_search.txt) */
#endif
#ifndef FF_FIN_TOL
#define FF_FIN_TOL 1e-7            /* a fit whose LAST step (any coordinate) is at most this is finished in the pass that found it (F to second order, I11 as it is) */
#endif
#ifndef FF_ABL
#define FF_ABL 0                   /* timing ablations (results meaningless): 1 = no MFMAs, 2 = no LDS-DMA / barriers, 4 = no sample arithmetic; 8 (results valid) = every fit finished by the exact kernel */
#endif
#ifndef FF_RHO
#define FF_RHO 0.05               /* largest relative change of I's diagonal against the null model for a fit to be finished from the one-pass F */
#endif
#ifndef FF_STAGGER
#define FF_STAGGER 0              /* s_sleep units (64 cycles) between the wavefronts of a block after every barrier */
#endif
#ifndef FF_SCHED
#define FF_SCHED 8                 /* vector instructions between two MFMAs of a pair's instruction stream (0: the compiler's own order) */
#endif
#define FF_MAXIT 8                 /* passes after which a fit that has not met the stop rule leaves for the exact rounds */

__host__ __device__ constexpr int ff_tri(int a) { return a * (a + 1) / 2; }
__host__ __device__ constexpr int ff_tet(int a) { return a * (a + 1) * (a + 2) / 6; }
struct FFMono { int a, b, c; };
// (z1 = (1, z), Z1 = Q + 1 entries.)  Degree-3 table: row r = tet(a) + tri(b) + c holds z1_a z1_b z1_c, a >= b >= c.  Degree-2 table: the Z1
// monomials with the constant come first -- row a holds z1_a (a = 0: 1) -- so that tile 0 alone serves the k-row of I (sum w k z1_a); rows
// Z1 + tri(a - 1) + (b - 1) hold z1_a z1_b for a >= b >= 1.
__host__ __device__ constexpr FFMono ff_dec_tri(int r) { int a = 0; while (ff_tri(a + 1) <= r) ++a; return FFMono{a, r - ff_tri(a), 0}; }
__host__ __device__ constexpr FFMono ff_dec3(int r) { int a = 0; while (ff_tet(a + 1) <= r) ++a; const FFMono m = ff_dec_tri(r - ff_tet(a)); return FFMono{a, m.a, m.b}; }
__host__ __device__ constexpr FFMono ff_dec2(int r, int z1) { if (r < z1) return FFMono{r, 0, 0}; const FFMono m = ff_dec_tri(r - z1); return FFMono{m.a + 1, m.b + 1, 0}; }
__host__ __device__ constexpr int ff_row2(int a, int b, int z1) { return b == 0 ? a : z1 + ff_tri(a - 1) + (b - 1); }
__host__ __device__ constexpr int ff_design(int z1) { return z1 == 0 ? 0 : z1 + 1; }      // column of x = (1, k, z) that holds z1's entry

template <int Q> struct FFC {
    static constexpr int Z1 = Q + 1, PC = Q + 2, NH = PC * (PC + 1) / 2;
    static constexpr int N2 = ff_tri(Z1), N3 = ff_tet(Z1);
    static constexpr int T2 = (N2 + 31) / 32, T3 = (N3 + 31) / 32, NTA = T2 + T3, NACC = 2 * T2 + T3 + 1, AWK = 2 * T2 + T3;   // (AWK: the k-row of I)
    static_assert(Z1 <= 32, "the k-row of I lives in tile 0 of the degree-2 table");
    static constexpr int RS = (Q + 3 + 1) & ~1;           // doubles per sample record: z_s[Q], s = 1 - 2 y, live (1 / 0), -2^12 w0 (the null model's weight, scaled like the operands), padding to 16 bytes
};

// g_x += sum over the distinct arrangements of the multiset {A, B, C} of t V_yz: for every distinct element x, the other two (y, z) give
// 2 t V_yz if y != z, t V_yy otherwise
template <int PC, int A, int B, int C>
__device__ __forceinline__ void ff_contrib(double t, const double (&V)[PC * (PC + 1) / 2], double (&g)[PC])
{
    auto Vs = [&](int i, int j) { return i >= j ? V[sidx(i, j)] : V[sidx(j, i)]; };
    g[A] = fma(t * (B == C ? 1.0 : 2.0), Vs(B, C), g[A]);
    if (B != A) g[B] = fma(t * (A == C ? 1.0 : 2.0), Vs(A, C), g[B]);
    if (C != A && C != B) g[C] = fma(t * (A == B ? 1.0 : 2.0), Vs(A, B), g[C]);
}
// row R of the degree-3 table: the entry T_{abc} without k
template <int Q, int R>
__device__ __forceinline__ void ff_row3(float tv, const double (&V)[FFC<Q>::NH], double (&g)[FFC<Q>::PC])
{
    if constexpr (R < FFC<Q>::N3) {
        constexpr FFMono m = ff_dec3(R);
        ff_contrib<FFC<Q>::PC, ff_design(m.a), ff_design(m.b), ff_design(m.c)>((double)tv, V, g);
    }
}
// row R of the degree-2 table against c k: the entries {k, a, b}; a monomial that holds the constant also stands for {k, k, a} (k^2 = k), and
// the constant alone for {k, k, k}
template <int Q, int R>
__device__ __forceinline__ void ff_row2k(float tv, const double (&V)[FFC<Q>::NH], double (&g)[FFC<Q>::PC])
{
    if constexpr (R < FFC<Q>::N2) {
        constexpr FFMono m = ff_dec2(R, FFC<Q>::Z1);
        ff_contrib<FFC<Q>::PC, 1, ff_design(m.a), ff_design(m.b)>((double)tv, V, g);
        if constexpr (m.b == 0) ff_contrib<FFC<Q>::PC, 1, 1, ff_design(m.a)>((double)tv, V, g);
        if constexpr (m.a == 0 && m.b == 0) ff_contrib<FFC<Q>::PC, 1, 1, 1>((double)tv, V, g);
    }
}
// row R of the degree-2 table against w: the entry I_{ab} of the covariate block (design columns other than k)
template <int Q, int R>
__device__ __forceinline__ void ff_row2i(float tv, const double *__restrict__ inull, double (&I)[FFC<Q>::NH], double &rho)
{
    if constexpr (R < FFC<Q>::N2) {
        constexpr FFMono m = ff_dec2(R, FFC<Q>::Z1);
        const double v0 = inull[R];
        I[sidx(ff_design(m.a), ff_design(m.b))] = v0 + (double)tv;
        if constexpr (m.a == m.b) rho = fmax(rho, fabs((double)tv) / v0);       // how far the weights are from the null model's, on the diagonal
    }
}
// ---- one 32-row tile of sums (every lane holding all of its rows) into I, or contracted with V into the penalty's gradient -----------------
template <int Q, int TILE, int... Rs>
__device__ __forceinline__ void ff_tile3(const float (&row)[32], const double (&V)[FFC<Q>::NH], double (&g)[FFC<Q>::PC], std::integer_sequence<int, Rs...>)
{
    (ff_row3<Q, TILE * 32 + Rs>(row[Rs], V, g), ...);
}
template <int Q, int TILE, int... Rs>
__device__ __forceinline__ void ff_tile2k(const float (&row)[32], const double (&V)[FFC<Q>::NH], double (&g)[FFC<Q>::PC], std::integer_sequence<int, Rs...>)
{
    (ff_row2k<Q, TILE * 32 + Rs>(row[Rs], V, g), ...);
}
template <int Q, int TILE, int... Rs>
__device__ __forceinline__ void ff_tile2i(const float (&row)[32], const double *__restrict__ inull, double (&I)[FFC<Q>::NH], double &rho,
                                          std::integer_sequence<int, Rs...>)
{
    (ff_row2i<Q, TILE * 32 + Rs>(row[Rs], inull, I, rho), ...);
}
