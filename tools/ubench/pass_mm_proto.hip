// Prototype + microbenchmark of "the single-precision Newton pass as GEMMs throughout" (DESIGN.md section 9): is ~1 cycle per (variant, sample)
// reachable when eta, the Hessian's blocks and the score all run on the f16 matrix pipe and the vector ALU keeps only the element-wise part?
//
// Layout (v_mfma_f32_32x32x16_f16; lane = (n = lane & 31, lh = lane >> 5); A: row n, k = 8 lh .. + 7; B: column n, k = 8 lh .. + 7;
// C: column n, register r = row (r & 3) + 8 (r >> 2) + 4 lh).  A wavefront owns 32 variants; per 32-sample tile:
//   1. eta[sample][variant] = (1, z)(sample) . beta(variant): A = the run's (1, z) tile (hi + lo), B = the lane's own beta as halves (hi + lo):
//      three MFMAs into a zero tile.  A lane then holds ITS variant's eta at the 16 samples rho(r) = (r & 3) + 8 (r >> 2) + 4 lh.
//   2. element-wise on those 16 registers: eta += beta_k x, mu, w = mu (1 - mu), w x, sum mu x.
//   3. the registers, converted to halves, ARE B operands [samples x variants] (the k-order of an operand is free): registers 0..7 of the two
//      lanes of a variant are the 16 k-slots of one MFMA, registers 8..15 of a second.  The run's tables (products z_j z_k, and (1, z)) are
//      stored in that sample order: products x w, (1, z) x w (intercept row), (1, z) x w x (variant row), (1, z) x mu (score: g = Z^T y - it).
// Tables come straight from L2 here (13 GB per pass at the bench's size: the product kernel would stage them through LDS for a block).
// Checks the sums against a plain fp64 kernel, then times 2^18 variants x N samples.
// Build: hipcc -O3 --offload-arch=gfx950 pass_mm_proto.hip -o pass_mm_proto ; run: ./pass_mm_proto [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int Q = 10, NPROD = Q * (Q + 1) / 2, NCB = 2;     // 55 products in two 32-column blocks
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__host__ __device__ constexpr int rho(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// tables, per 32-sample tile (one uint4 = 8 halves per lane each):
//   ze[tile][2 (hi, lo)][64]      A of eta: lane (m = sample in tile, kg): coefficients 8 kg .. + 7 of (1, z_1 .. z_Q, 0 ...)
//   zp[tile][NCB][2 (half)][64]   A of the products: lane (m = product column, kg) of MFMA h: samples 16 h + {0,1,2,3,8,9,10,11} + 4 kg
//   zl[tile][2 (hi, lo)][2 (half)][64]   A of the linear block: lane (m = column of (1, z), kg), same sample order
__global__ __launch_bounds__(64, 3) void k_pass_mm(const uint32_t *__restrict__ bits, int NT, const v4u *__restrict__ ze, const v4u *__restrict__ zp,
                                                   const v4u *__restrict__ zl, const float *__restrict__ beta, float *__restrict__ out, int tmask)
{
    const int lane = threadIdx.x, n = lane & 31, lh = lane >> 5;
    const int64_t v = (int64_t)blockIdx.x * 32 + n;
    // the lane's own beta as the B operand of eta: k = 8 lh .. + 7 of (b0, bz_1 .. bz_Q, 0 ...), hi + lo
    v4u bh, bl;
    float bk;
    {
        const float *b = beta + v * (Q + 2);
        bk = b[1];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k0 = 8 * lh + 2 * t, k1 = k0 + 1;
            const float x0 = k0 == 0 ? b[0] : (k0 <= Q ? b[1 + k0] : 0.0f), x1 = k1 <= Q ? b[1 + k1] : 0.0f;
            const v2h hh = __builtin_convertvector(v2f{x0, x1}, v2h);
            const v2f back = __builtin_convertvector(hh, v2f);
            const v2h ll = __builtin_convertvector(v2f{x0 - back.x, x1 - back.y}, v2h);
            bh[t] = __builtin_bit_cast(uint32_t, hh); bl[t] = __builtin_bit_cast(uint32_t, ll);
        }
    }
    v16f accp[NCB], accw, accx, accm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accp[0][r] = 0; accp[1][r] = 0; accw[r] = 0; accx[r] = 0; accm[r] = 0; }
    float smx = 0.0f;                                           // sum mu x (the variant's own score entry: t11 - it)
    const v8h Bh = __builtin_bit_cast(v8h, bh), Bl = __builtin_bit_cast(v8h, bl);
    v4u e0 = ze[(0 * 2 + 0) * 64 + lane], e1 = ze[(0 * 2 + 1) * 64 + lane];
    uint32_t w32 = bits[(int64_t)0 * gridDim.x * 32 + v];
    for (int t0 = 0; t0 < NT; ++t0) {
        const int t = t0 & tmask;                               // (tmask = 0: every tile reads tile 0's tables -- the pass without its table traffic)
        const int tn = min(t0 + 1, NT - 1) & tmask;
        const v4u e0n = ze[((int64_t)tn * 2 + 0) * 64 + lane], e1n = ze[((int64_t)tn * 2 + 1) * 64 + lane];
        const uint32_t wn = bits[(int64_t)min(t0 + 1, NT - 1) * gridDim.x * 32 + v];
        v4u pz[NCB][2], lz[2][2];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int h = 0; h < 2; ++h) pz[cb][h] = zp[(((int64_t)t * NCB + cb) * 2 + h) * 64 + lane];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h) lz[p][h] = zl[(((int64_t)t * 2 + p) * 2 + h) * 64 + lane];
        v16f eta;
#pragma unroll
        for (int r = 0; r < 16; ++r) eta[r] = 0.0f;
        const v8h Eh = __builtin_bit_cast(v8h, e0), El = __builtin_bit_cast(v8h, e1);
        eta = __builtin_amdgcn_mfma_f32_32x32x16_f16(Eh, Bh, eta, 0, 0, 0);
        eta = __builtin_amdgcn_mfma_f32_32x32x16_f16(El, Bh, eta, 0, 0, 0);
        eta = __builtin_amdgcn_mfma_f32_32x32x16_f16(Eh, Bl, eta, 0, 0, 0);
        const uint32_t wsh = w32 >> (4 * lh);
        uint32_t Bw[2][4], Bx[2][4], Bmh[2][4], Bml[2][4];
#pragma unroll
        for (int pr = 0; pr < 8; ++pr) {                        // registers (2 pr, 2 pr + 1): one half2 of the operands
            float w2[2], x2[2], m2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 2 * pr + q;
                const float xb = (float)((wsh >> ((r & 3) + 8 * (r >> 2))) & 1u);
                const float e = fmaf(bk, xb, eta[r]);
                const float mu = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e * -1.4426950408889634f));
                const float w = fmaf(-mu, mu, mu);
                smx = fmaf(mu, xb, smx);
                w2[q] = w; x2[q] = w * xb; m2[q] = mu;
            }
            const int h = pr >> 2, s = pr & 3;
            Bw[h][s] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{w2[0], w2[1]}, v2h));
            Bx[h][s] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{x2[0], x2[1]}, v2h));
            const v2h mh = __builtin_convertvector(v2f{m2[0], m2[1]}, v2h);
            const v2f back = __builtin_convertvector(mh, v2f);
            Bmh[h][s] = __builtin_bit_cast(uint32_t, mh);
            Bml[h][s] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2f{m2[0] - back.x, m2[1] - back.y}, v2h));
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v8h W = __builtin_bit_cast(v8h, v4u{Bw[h][0], Bw[h][1], Bw[h][2], Bw[h][3]}), X = __builtin_bit_cast(v8h, v4u{Bx[h][0], Bx[h][1], Bx[h][2], Bx[h][3]});
            const v8h Mh = __builtin_bit_cast(v8h, v4u{Bmh[h][0], Bmh[h][1], Bmh[h][2], Bmh[h][3]}), Ml = __builtin_bit_cast(v8h, v4u{Bml[h][0], Bml[h][1], Bml[h][2], Bml[h][3]});
            const v8h Lh = __builtin_bit_cast(v8h, lz[0][h]), Ll = __builtin_bit_cast(v8h, lz[1][h]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) accp[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, pz[cb][h]), W, accp[cb], 0, 0, 0);
            accw = __builtin_amdgcn_mfma_f32_32x32x16_f16(Lh, W, accw, 0, 0, 0);
            accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(Lh, X, accx, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(Lh, Mh, accm, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(Lh, Ml, accm, 0, 0, 0);
            accm = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ll, Mh, accm, 0, 0, 0);
        }
        e0 = e0n; e1 = e1n; w32 = wn;
    }
    smx += __shfl_xor(smx, 32);
    // out[v][0 .. 63] products (columns of the two blocks), [64 .. 95] (1, z) x w, [96 .. 127] (1, z) x w x, [128 .. 159] (1, z) x mu, [160] sum mu x
    float *o = out + v * 161;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rho(r, lh);
        o[row] = accp[0][r]; o[32 + row] = accp[1][r]; o[64 + row] = accw[r]; o[96 + row] = accx[r]; o[128 + row] = accm[r];
    }
    if (lh == 0) o[160] = smx;
}

// the same sums, plainly (fp64; one thread per variant)
__global__ void k_ref(const uint32_t *__restrict__ bits, int NT, int N, const double *__restrict__ Z, const float *__restrict__ beta, double *__restrict__ out, int64_t Vtot)
{
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= Vtot) return;
    const float *b = beta + v * (Q + 2);
    double acc[161];
    for (int i = 0; i < 161; ++i) acc[i] = 0.0;
    for (int i = 0; i < N; ++i) {
        const int t = i >> 5;
        const double x = (double)((bits[(int64_t)t * Vtot + v] >> (i & 31)) & 1u);
        double lin[Q + 1]; lin[0] = 1.0;
        for (int j = 0; j < Q; ++j) lin[1 + j] = Z[(int64_t)i * Q + j];
        double eta = b[0] + b[1] * x;
        for (int j = 0; j < Q; ++j) eta += (double)b[2 + j] * lin[1 + j];
        const double mu = 1.0 / (1.0 + exp(-eta)), w = mu * (1.0 - mu);
        int c = 0;
        for (int j = 0; j < Q; ++j) for (int k = 0; k <= j; ++k, ++c) acc[c] += w * lin[1 + j] * lin[1 + k];
        for (int j = 0; j <= Q; ++j) { acc[64 + j] += w * lin[j]; acc[96 + j] += w * x * lin[j]; acc[128 + j] += mu * lin[j]; }
        acc[160] += mu * x;
    }
    for (int i = 0; i < 161; ++i) out[v * 161 + i] = acc[i];
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 4992, NT = N / 32;
    const int64_t Vchk = 2048, Vbig = 1 << 18;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> Z((size_t)N * Q);
    for (auto &z : Z) z = nd(rng);
    auto lin = [&](int i, int c) -> float { return c == 0 ? 1.0f : (c <= Q ? (float)Z[(size_t)i * Q + (c - 1)] : 0.0f); };
    // sample order of the k-slots: MFMA h, lane half kg, slot t -> sample 16 h + {0,1,2,3,8,9,10,11}[t] + 4 kg
    static const int so[8] = {0, 1, 2, 3, 8, 9, 10, 11};
    std::vector<uint16_t> ze((size_t)NT * 2 * 64 * 8), zp((size_t)NT * NCB * 2 * 64 * 8, 0), zl((size_t)NT * 2 * 2 * 64 * 8, 0);
    for (int t = 0; t < NT; ++t)
        for (int ln = 0; ln < 64; ++ln) {
            const int m = ln & 31, kg = ln >> 5;
            for (int s = 0; s < 8; ++s) {
                const float x = lin(t * 32 + m, 8 * kg + s);
                const uint16_t hi = f2h(x), lo = f2h(x - h2f(hi));
                ze[(((size_t)t * 2 + 0) * 64 + ln) * 8 + s] = hi; ze[(((size_t)t * 2 + 1) * 64 + ln) * 8 + s] = lo;
            }
            for (int h = 0; h < 2; ++h)
                for (int s = 0; s < 8; ++s) {
                    const int i = t * 32 + 16 * h + so[s] + 4 * kg;
                    for (int cb = 0; cb < NCB; ++cb) {
                        const int col = cb * 32 + m;
                        if (col < NPROD) {
                            int j = 0; while ((j + 1) * (j + 2) / 2 <= col) ++j;
                            const int k = col - j * (j + 1) / 2;
                            zp[((((size_t)t * NCB + cb) * 2 + h) * 64 + ln) * 8 + s] = f2h((float)(Z[(size_t)i * Q + j] * Z[(size_t)i * Q + k]));
                        }
                    }
                    const float x = lin(i, m);
                    const uint16_t hi = f2h(x), lo = f2h(x - h2f(hi));
                    zl[((((size_t)t * 2 + 0) * 2 + h) * 64 + ln) * 8 + s] = hi; zl[((((size_t)t * 2 + 1) * 2 + h) * 64 + ln) * 8 + s] = lo;
                }
        }
    auto run = [&](int64_t V, bool check) {
        std::vector<uint32_t> bits((size_t)NT * V);
        for (auto &b : bits) b = (uint32_t)rng() & (uint32_t)rng();          // ~25 % carriers
        std::vector<float> beta((size_t)V * (Q + 2));
        for (int64_t v = 0; v < V; ++v) { beta[v * (Q + 2)] = -0.5f + 0.1f * (float)nd(rng); beta[v * (Q + 2) + 1] = 0.5f * (float)nd(rng); for (int j = 0; j < Q; ++j) beta[v * (Q + 2) + 2 + j] = 0.3f * (float)nd(rng); }
        uint32_t *d_bits; v4u *d_ze, *d_zp, *d_zl; float *d_beta, *d_out; double *d_Z, *d_ref;
        HIPCHK(hipMalloc(&d_bits, bits.size() * 4)); HIPCHK(hipMalloc(&d_ze, ze.size() * 2)); HIPCHK(hipMalloc(&d_zp, zp.size() * 2)); HIPCHK(hipMalloc(&d_zl, zl.size() * 2));
        HIPCHK(hipMalloc(&d_beta, beta.size() * 4)); HIPCHK(hipMalloc(&d_out, (size_t)V * 161 * 4)); HIPCHK(hipMalloc(&d_Z, Z.size() * 8));
        HIPCHK(hipMemcpy(d_bits, bits.data(), bits.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_ze, ze.data(), ze.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_zp, zp.data(), zp.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_zl, zl.data(), zl.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_beta, beta.data(), beta.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d_Z, Z.data(), Z.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_pass_mm, dim3((unsigned)(V / 32)), dim3(64), 0, 0, d_bits, NT, d_ze, d_zp, d_zl, d_beta, d_out, -1);
        HIPCHK(hipDeviceSynchronize());
        if (check) {
            HIPCHK(hipMalloc(&d_ref, (size_t)V * 161 * 8));
            hipLaunchKernelGGL(k_ref, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, 0, d_bits, NT, N, d_Z, d_beta, d_ref, V);
            HIPCHK(hipDeviceSynchronize());
            std::vector<float> o((size_t)V * 161); std::vector<double> r((size_t)V * 161);
            HIPCHK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(r.data(), d_ref, r.size() * 8, hipMemcpyDeviceToHost));
            double worst[5] = {0, 0, 0, 0, 0}, scale[5] = {0, 0, 0, 0, 0};
            for (int64_t v = 0; v < V; ++v)
                for (int i = 0; i < 161; ++i) {
                    const int grp = i < 64 ? 0 : i < 96 ? 1 : i < 128 ? 2 : i < 160 ? 3 : 4;
                    const int col = i < 64 ? i : (i - 64) % 32;
                    if ((grp == 0 && col >= NPROD) || (grp >= 1 && grp <= 3 && col > Q)) continue;
                    worst[grp] = fmax(worst[grp], fabs((double)o[v * 161 + i] - r[v * 161 + i])); scale[grp] = fmax(scale[grp], fabs(r[v * 161 + i]));
                }
            const char *nm[5] = {"products x w", "(1, z) x w", "(1, z) x w x", "(1, z) x mu", "sum mu x"};
            for (int g = 0; g < 5; ++g) printf("check %-14s max |err| %.3e (largest entry %.3e: %.1e relative)\n", nm[g], worst[g], scale[g], worst[g] / scale[g]);
            hipFree(d_ref);
        } else {
            for (int tmask = -1; tmask <= 0; ++tmask) {
                hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
                HIPCHK(hipEventRecord(e0));
                for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k_pass_mm, dim3((unsigned)(V / 32)), dim3(64), 0, 0, d_bits, NT, d_ze, d_zp, d_zl, d_beta, d_out, tmask);
                HIPCHK(hipEventRecord(e1)); HIPCHK(hipEventSynchronize(e1));
                float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
                printf("k_pass_mm%s: %lld variants x %d samples: %.3f ms  (%.2f SIMD-cycles per (variant, sample) at 2.4 GHz; tables from L2: %.1f GB per pass)\n",
                       tmask ? "" : " (every tile = tile 0: no table traffic)", (long long)V, N, ms, ms * 1e-3 * 2.4e9 * 1024 / ((double)V * N), tmask ? (double)(V / 32) * NT * 64 * 16 * 10 / 1e9 : 0.0);
            }
        }
        hipFree(d_bits); hipFree(d_ze); hipFree(d_zp); hipFree(d_zl); hipFree(d_beta); hipFree(d_out); hipFree(d_Z);
    };
    run(Vchk, true);
    run(Vbig, false);
    return 0;
}
