#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdint>
#include "../../safepy_amd/csrc/draws.cpp"
template <class F> double timeit(F f) { auto t0 = std::chrono::steady_clock::now(); f(); return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
int main() {
    const int64_t k = 3789;
    std::vector<uint32_t> steps(k + 64);
    DrawStream ds(0);
    ds.shuffle_targets(k, steps.data());
    double tv4 = 0, tv2 = 0, tv1 = 0, tsc = 0; size_t w4 = 0, w2 = 0, w1 = 0, wsc = 0;
    for (int q = 0; q < 1000; ++q) {
        int64_t i = k - 1;
        while (i > 0) {
            const uint32_t mask = mask_for((uint32_t)i); const int64_t lo = mask >> 1;
            size_t r0 = ds.rp; (void)r0;
            auto words = [&](size_t before) { return (size_t)0; };
            (void)words;
            if (mask >= 2047) { size_t a = ds.rp; int64_t i0 = i; tv4 += timeit([&] { i = ds.vector_run<4>(i, lo, mask, k, steps.data()); }); w4 += (i0 - i); (void)a; }
            if (mask >= 511) { int64_t i0 = i; tv2 += timeit([&] { i = ds.vector_run<2>(i, lo, mask, k, steps.data()); }); w2 += (i0 - i); }
            { int64_t i0 = i; tv1 += timeit([&] { i = ds.vector_run<1>(i, lo, mask, k, steps.data()); }); w1 += (i0 - i); }
            { int64_t i0 = i; tsc += timeit([&] { i = ds.scalar_run(i, lo, mask, k, steps.data()); }); wsc += (i0 - i); }
        }
    }
    printf("per 1000 perms: v4 %.2f ms (%zu draws)  v2 %.2f ms (%zu)  v1 %.2f ms (%zu)  scalar %.2f ms (%zu)\n", tv4 / 1e3, w4, tv2 / 1e3, w2, tv1 / 1e3, w1, tsc / 1e3, wsc);
    return 0;
}
