#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdint>
#include "../../safepy_amd/csrc/draws.cpp"
int main() {
    const int64_t k = 3789, P = 1000;
    std::vector<uint32_t> steps(k * 128);
    for (int rep = 0; rep < 3; ++rep) {
        DrawStream ds(0);
        auto t0 = std::chrono::steady_clock::now();
        for (int64_t q = 0; q < P; ++q) ds.shuffle_targets(k, steps.data() + (q % 128) * k);
        auto t1 = std::chrono::steady_clock::now();
        printf("draws: %.3f ms per 1000 perms (avx512 %d)\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), (int)ds.use_avx512);
    }
    // breakdown: how many words go through the vector path vs scalar
    return 0;
}
