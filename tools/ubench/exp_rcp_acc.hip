// Accuracy of the pieces a leaner k_glm_ll would be made of (gfx950; measured and NOT adopted -- DESIGN.md section 9): v_rcp_f64 raw / after one / after two Newton steps on u in (1, 2];
// e^-x by the degree-13 polynomial (glm_device.h: exp_neg) against a 64-entry table + degree-5 polynomial.  Reference: long double on the host.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off exp_rcp_acc.hip -o exp_rcp_acc ; run: ./exp_rcp_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__device__ __forceinline__ double exp_neg13(double x)
{
    const double u = -fmin(x, 800.0);
    const double kf = rint(u * 1.4426950408889634074);
    double r = fma(kf, -6.93147180369123816490e-01, u);
    r = fma(kf, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09); p = fma(p, r, 2.505210838544172e-08); p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06); p = fma(p, r, 2.48015873015873e-05); p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.3888888888888889e-03); p = fma(p, r, 8.333333333333333e-03); p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
__device__ __forceinline__ double exp_neg_tab(double x, const double *etab)
{
    const double u = -fmin(x, 800.0);
    const double mf = rint(u * 92.332482616893656758);                       // 64 / ln 2
    double r = fma(mf, -6.93147180369123816490e-01 / 64.0, u);
    r = fma(mf, -1.90821492927058770002e-10 / 64.0, r);
    const int m = (int)mf;
    const double T = etab[m & 63];
    double c = fma(r, 1.0 / 120.0, 1.0 / 24.0); c = fma(c, r, 1.0 / 6.0); c = fma(c, r, 0.5);
    const double q = fma(c, r * r, r);
    return ldexp(fma(T, q, T), m >> 6);
}
__global__ void k(const double *x, int n, double *o)
{
    __shared__ double etab[64];
    if (threadIdx.x < 64) etab[threadIdx.x] = exp2((double)threadIdx.x / 64.0);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double xi = x[i];
    o[i] = exp_neg13(xi); o[n + i] = exp_neg_tab(xi, etab);
    const double u = 1.0 + o[i];                                             // u in (1, 2]
    double inv = __builtin_amdgcn_rcp(u);
    o[2 * n + i] = inv;
    inv = fma(fma(-u, inv, 1.0), inv, inv); o[3 * n + i] = inv;
    inv = fma(fma(-u, inv, 1.0), inv, inv); o[4 * n + i] = inv;
    o[5 * n + i] = u;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), o(6 * (size_t)n);
    std::mt19937_64 g(1); std::uniform_real_distribution<double> U(0.0, 1.0);
    for (int i = 0; i < n; ++i) x[i] = (i & 3) == 0 ? U(g) * 0.01 : (i & 3) == 1 ? U(g) * 2.0 : (i & 3) == 2 ? U(g) * 40.0 : U(g) * 700.0;
    double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, dout);
    hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    double e13 = 0, etb = 0, r0 = 0, r1 = 0, r2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double ex = expl(-(long double)x[i]);
        e13 = fmax(e13, (double)fabsl((o[i] - ex) / ex)); etb = fmax(etb, (double)fabsl((o[n + i] - ex) / ex));
        const long double iv = 1.0L / (long double)o[5 * (size_t)n + i];
        r0 = fmax(r0, (double)fabsl((o[2 * (size_t)n + i] - iv) / iv)); r1 = fmax(r1, (double)fabsl((o[3 * (size_t)n + i] - iv) / iv)); r2 = fmax(r2, (double)fabsl((o[4 * (size_t)n + i] - iv) / iv));
    }
    printf("{\"exp_neg_degree13_max_rel\": %.3e, \"exp_neg_table64_degree5_max_rel\": %.3e, \"rcp_f64_raw_max_rel\": %.3e, \"rcp_one_newton_max_rel\": %.3e, \"rcp_two_newton_max_rel\": %.3e, \"eps\": %.3e}\n", e13, etb, r0, r1, r2, 2.220446049250313e-16);
    return 0;
}
