// Does the fp64 matrix instruction run BESIDE the fp64 vector ALU on gfx950?  Per iteration NM v_mfma_f64_16x16x4_f64 (independent
// accumulators) interleaved with NV independent v_fma_f64 (8 chains), at W wavefronts per SIMD.  If the two units overlap, time = max of the
// two; if they share a data path, the sum.  Second part: ds_read_b128 throughput when the 64 lanes read 64 / 16 / 4 / 1 distinct addresses
// (the sample records of the Firth kernels are read by 16 lanes each).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64.hip -o mfma_f64 ; run: ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ __launch_bounds__(256) void km(double *out)
{
    double a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    double b = 1.0000001, c = 1e-9;
    v4d acc[4]; for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
    double A = threadIdx.x * 1e-3, B = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u < NM) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(A), "v"(B));
#pragma unroll
            for (int q = 0; q < NV / 4; ++q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[(u * (NV / 4) + q) & 7]) : "v"(b), "v"(c));
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV> static void runm(int w, double *o)
{
    const int blocks = 256 * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((km<NM, NV>), dim3(blocks), dim3(256), 0, 0, o); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((km<NM, NV>), dim3(blocks), dim3(256), 0, 0, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("per iteration %d MFMA f64 16x16x4 + %2d v_fma_f64, W=%d: %.1f cycles per iteration per SIMD at 2.4 GHz\n", NM, NV, w, ms * 1e-3 * 2.4e9 / ((double)ITERS * w));
}

// ds_read_b128, 16 per iteration, DIST distinct addresses per wavefront (lanes l and l' read the same 16 bytes iff l % DIST == l' % DIST ...
// here: group = lane / (64 / DIST)), all waves of the CU at once
template <int DIST>
__global__ __launch_bounds__(512) void kl(double *out)
{
    __shared__ __attribute__((aligned(16))) double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / (64 / DIST);
    const double *p = lds + wave * 1024 + grp * 14;                 // (record stride of 14 doubles, as the kernels')
    v4d s = {0, 0, 0, 0};
    typedef double v2d __attribute__((ext_vector_type(2)));
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            v2d x;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const double *)p), "n"(16 * (u & 7)));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            s[0] += 0.0 * x.x;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}
template <int DIST> static void runl(double *o)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kl<DIST>, dim3(256), dim3(512), 0, 0, o); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kl<DIST>, dim3(256), dim3(512), 0, 0, o); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ds_read_b128, %2d distinct addresses per wavefront, 8 wavefronts per CU: %.1f cycles per wave-instruction per CU at 2.4 GHz\n", DIST, ms * 1e-3 * 2.4e9 / ((double)ITERS * 16 * 8));
}
int main()
{
    double *o; hipMalloc(&o, 8 << 20);
    for (int w = 1; w <= 4; w *= 2) { runm<4, 0>(w, o); runm<0, 32>(w, o); runm<4, 32>(w, o); runm<4, 64>(w, o); runm<2, 32>(w, o); runm<1, 32>(w, o); runm<4, 16>(w, o); }
    runl<64>(o); runl<16>(o); runl<4>(o); runl<1>(o);
    return 0;
}
