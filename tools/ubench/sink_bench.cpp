#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <random>
#include <vector>
#include "../../include/seerhip.h"
int main() {
    const int64_t n = 1 << 20;
    std::mt19937_64 g(1); std::uniform_real_distribution<double> u(0, 1);
    std::vector<std::vector<double>> c(6, std::vector<double>(n));
    for (int k = 0; k < 6; ++k) for (int64_t i = 0; i < n; ++i) c[k][i] = (k == 1 || k == 2) ? std::pow(10.0, -30 * u(g)) : u(g) - (k == 3 ? 0.5 : 0);
    const double *cols[6]; for (int k = 0; k < 6; ++k) cols[k] = c[k].data();
    std::vector<char> names(n * 31, 'A'); std::vector<int64_t> off(n + 1), sel(n);
    for (int64_t i = 0; i <= n; ++i) off[i] = i * 31;
    for (int64_t i = 0; i < n; ++i) sel[i] = i;
    std::vector<uint32_t> flags(n, 0); std::vector<int32_t> lin(n, 1);
    const char *labels[3] = {"a", "bb", "c"};
    std::vector<char> out(200 << 20);
    double best = 1e9;
    for (int r = 0; r < 10; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        int64_t k = sh_format_rows(names.data(), off.data(), sel.data(), n, cols, 6, nullptr, 0, nullptr, lin.data(), labels, 3, flags.data(), out.data(), out.size());
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
        if (r == 0) printf("%lld bytes\n", (long long)k);
    }
    printf("min %.1f ms = %.0f ns/row\n", best * 1e3, best * 1e9 / n);
}
