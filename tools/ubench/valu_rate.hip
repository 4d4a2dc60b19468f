// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction of the instructions the fixed-effects kernels are made of.
// Every kernel runs ITERS x 16 independent instructions of one kind per wave (8 accumulator chains, so dependent latency is hidden),
// at W waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITERS 4096
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, float *outf)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 1e-9;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f f0 = {1.f, 2.f}, f1 = {3.f, 4.f}, f2 = {5.f, 6.f}, f3 = {7.f, 8.f}, f4 = f0, f5 = f1, f6 = f2, f7 = f3, fb = {1.0000001f, 0.999999f}, fc = {1e-9f, 1e-9f};
    float s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8, sb = 1.0000001f, sc = 1e-9f;
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) { asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                      "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
        if (KIND == 1) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                      "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                      : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(sb), "v"(sc)); }
        if (KIND == 2) { asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                                      "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                                      : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb), "v"(fc)); }
        if (KIND == 3) { asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,1]\n"
                                      "v_pk_fma_f32 %0, %0, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %4, %4, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %9 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel_hi:[1,0,1]\n"
                                      : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb), "v"(fc)); }
        if (KIND == 4) { asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                                      "v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %9\n v_add_f64 %3, %3, %9\n v_add_f64 %4, %4, %9\n v_add_f64 %5, %5, %9\n v_add_f64 %6, %6, %9\n v_add_f64 %7, %7, %9\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); }
        if (KIND == 5) { asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                                      "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (KIND == 6) { asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                      "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(c)); }      // one SGPR-pair operand, as the covariate rows arrive
        if (KIND == 7) { asm volatile("v_ldexp_f64 %0, %0, 1\n v_rndne_f64 %1, %1\n v_ldexp_f64 %2, %2, 1\n v_rndne_f64 %3, %3\n v_ldexp_f64 %4, %4, 1\n v_rndne_f64 %5, %5\n v_ldexp_f64 %6, %6, 1\n v_rndne_f64 %7, %7\n"
                                      "v_ldexp_f64 %0, %0, -1\n v_rndne_f64 %1, %1\n v_ldexp_f64 %2, %2, -1\n v_rndne_f64 %3, %3\n v_ldexp_f64 %4, %4, -1\n v_rndne_f64 %5, %5\n v_ldexp_f64 %6, %6, -1\n v_rndne_f64 %7, %7\n"
                                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    outf[blockIdx.x * blockDim.x + threadIdx.x] = f0.x + f1.y + f2.x + f3.y + f4.x + f5.y + f6.x + f7.y + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
}
template <int KIND> static void run(const char *name, int waves_per_simd, double *o, float *of, double ghz)
{
    const int blocks = 256 * waves_per_simd;                       // 256 CUs x (4 waves of a 256-thread block = one per SIMD) x W
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, o, of); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, o, of); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)ITERS * 16 * waves_per_simd;
    printf("%-34s W=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD at %.1f GHz\n", name, waves_per_simd, ms, ms * 1e-3 * ghz * 1e9 / inst_per_simd, ghz);
}

// dependent-issue latency: NCH independent chains of v_fma_f64 (16 instructions per iteration, round-robin over the chains)
template <int NCH, int PK>
__global__ __launch_bounds__(256) void kc(double *out)
{
    double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    double b = 1.0000001, c = 1e-9;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[u % NCH]) : "v"(b), "v"(c));
            else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[u % NCH]) : "v"(b), "v"(c));
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH, int PK> static void runc(int waves_per_simd, double *o, double ghz)
{
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kc<NCH, PK>), dim3(blocks), dim3(256), 0, 0, o); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((kc<NCH, PK>), dim3(blocks), dim3(256), 0, 0, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s, %d independent chain(s) per wave, W=%d: %.2f cycles per wave-instruction per SIMD at %.1f GHz\n", PK ? "v_pk_fma_f32" : "v_fma_f64", NCH, waves_per_simd,
           ms * 1e-3 * ghz * 1e9 / ((double)ITERS * 16 * waves_per_simd), ghz);
}

// conversions and selects (one kind per kernel, 16 independent instructions per iteration)
template <int KIND>
__global__ __launch_bounds__(256) void kv(double *out)
{
    double a[8]; unsigned u[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; f[i] = threadIdx.x + 0.5f * i; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (KIND == 1) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            if (KIND == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            if (KIND == 3) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i]));
            if (KIND == 4) asm volatile("v_bfe_u32 %0, %1, 3, 1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (KIND == 5) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            if (KIND == 6) asm volatile("v_lshrrev_b64 %0, 3, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (KIND == 7) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
            if (KIND == 8) asm volatile("v_max_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if (KIND == 9) asm volatile("v_cmp_eq_u64 vcc, %0, %1" :: "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + u[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> static void runv(const char *name, int w, double *o)
{
    const int blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kv<KIND>, dim3(blocks), dim3(256), 0, 0, o); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kv<KIND>, dim3(blocks), dim3(256), 0, 0, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-18s W=%d: %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, w, ms * 1e-3 * 2.4e9 / ((double)ITERS * 16 * w));
}

// MFMA beside the vector ALU within ONE wavefront: per iteration NM v_mfma_f32_32x32x16_f16 (independent accumulators) interleaved with NV
// independent v_fma_f64 (8 chains).  If the two units overlap inside a wavefront, time = max of the two; if not, the sum.
typedef _Float16 ub_v8h __attribute__((ext_vector_type(8)));
typedef float ub_v16f __attribute__((ext_vector_type(16)));
template <int NM, int NV>
__global__ __launch_bounds__(256) void km(double *out)
{
    double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    double b = 1.0000001, c = 1e-9;
    ub_v16f acc[4]; for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    ub_v8h A, B; for (int r = 0; r < 8; ++r) { A[r] = (_Float16)(threadIdx.x * 0.001f + r); B[r] = (_Float16)(r * 0.5f); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < NM) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[u]) : "v"(A), "v"(B));
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(u * (NV / 4) + q) & 7]) : "v"(b), "v"(c));
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV> static void runm(int w, double *o)
{
    const int blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((km<NM, NV>), dim3(blocks), dim3(256), 0, 0, o); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((km<NM, NV>), dim3(blocks), dim3(256), 0, 0, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("per iteration %d MFMA 32x32x16 f16 + %2d v_fma_f64, W=%d: %.1f cycles per iteration per SIMD at 2.4 GHz\n", NM, NV, w, ms * 1e-3 * 2.4e9 / ((double)ITERS * w));
}
int main()
{
    double *o; float *of; hipMalloc(&o, 8 << 20); hipMalloc(&of, 4 << 20);
    for (int w = 1; w <= 2; ++w) { runm<4, 0>(w, o); runm<0, 32>(w, o); runm<4, 32>(w, o); runm<4, 16>(w, o); runm<4, 64>(w, o); runm<1, 32>(w, o); runm<2, 32>(w, o); }
    if (getenv("UB_MFMA_ONLY")) return 0;
    for (int w = 1; w <= 8; w *= 2) {
        run<0>("v_fma_f64", w, o, of, 2.4); run<6>("v_fma_f64 (SGPR operand)", w, o, of, 2.4); run<4>("v_mul_f64 / v_add_f64", w, o, of, 2.4);
        run<1>("v_fma_f32", w, o, of, 2.4); run<2>("v_pk_fma_f32", w, o, of, 2.4); run<3>("v_pk_fma_f32 op_sel_hi (broadcast)", w, o, of, 2.4);
        run<5>("v_rcp_f64", w, o, of, 2.4); run<7>("v_ldexp_f64 / v_rndne_f64", w, o, of, 2.4);
    }
    for (int w = 1; w <= 8; w *= 2) { runc<1, 0>(w, o, 2.4); runc<2, 0>(w, o, 2.4); runc<4, 0>(w, o, 2.4); runc<8, 0>(w, o, 2.4); runc<1, 1>(w, o, 2.4); runc<2, 1>(w, o, 2.4); runc<4, 1>(w, o, 2.4); }
    for (int w = 1; w <= 2; ++w) { runv<0>("v_cvt_f64_u32", w, o); runv<1>("v_cvt_f32_f64", w, o); runv<2>("v_cvt_f64_f32", w, o); runv<3>("v_cvt_i32_f64", w, o);
        runv<4>("v_bfe_u32", w, o); runv<5>("v_cndmask_b32", w, o); runv<6>("v_lshrrev_b64", w, o); runv<7>("v_cvt_f32_u32", w, o); runv<8>("v_max_f64", w, o); runv<9>("v_cmp_eq_u64", w, o); }
    return 0;
}
