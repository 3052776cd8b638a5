// Where the time of the host draw loop (safepy_amd/csrc/draws.cpp, vector_run) goes: the same batch-resolved rejection
// loop over PRE-GENERATED raw words (no helper thread), with parts switched off one at a time (results are then wrong;
// only the time matters).  Build: g++ -O3 -std=c++17 -march=native draw_breakdown.cpp -o draw_breakdown -lpthread
#include <immintrin.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../safepy_amd/csrc/draws.cpp"

static inline uint32_t mask_of(uint32_t i) { uint32_t m = i; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; return m; }

// MODE bits: 1 = no compress store, 2 = no ambiguity resolution, 4 = register compress + full store, 8 = branch-free first two
template <int NV, int MODE>
__attribute__((target("avx512f,popcnt,bmi,bmi2"))) static int64_t run(const uint32_t *raw, size_t &rp, int64_t i, int64_t lo, uint32_t mask,
                                                                       int64_t k, uint32_t *steps) {
    constexpr int W = 16 * NV;
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    alignas(64) uint32_t lanes[W];
    while (i - W > lo) {
        const __m512i sure_thr = _mm512_set1_epi32((int)(i - W)), maybe_thr = _mm512_set1_epi32((int)i);
        __m512i v[NV];
        uint64_t acc = 0, maybe = 0;
        for (int j = 0; j < NV; ++j) {
            v[j] = _mm512_and_si512(_mm512_loadu_si512(raw + rp + 16 * j), vmask);
            acc |= (uint64_t)_mm512_cmple_epu32_mask(v[j], sure_thr) << (16 * j);
            maybe |= (uint64_t)_mm512_cmple_epu32_mask(v[j], maybe_thr) << (16 * j);
        }
        uint64_t amb = maybe & ~acc;
        if (!(MODE & 2)) {
            if (MODE & 8) {
                for (int j = 0; j < NV; ++j) _mm512_store_si512(lanes + 16 * j, v[j]);
                for (int rep = 0; rep < 2; ++rep) {
                    const unsigned t = (unsigned)__builtin_ctzll(amb | (1ull << (W - 1)));
                    const uint64_t bit = 1ull << t;
                    const unsigned before = (unsigned)__builtin_popcountll(acc & (bit - 1));
                    acc |= (lanes[t] <= (uint32_t)i - before) ? bit : 0;
                    amb &= amb - 1;
                }
                while (amb) {
                    const unsigned t = (unsigned)__builtin_ctzll(amb);
                    amb &= amb - 1;
                    const unsigned before = (unsigned)__builtin_popcountll(acc & ((1ull << t) - 1));
                    if (lanes[t] <= (uint32_t)i - before) acc |= 1ull << t;
                }
            } else if (amb) {
                for (int j = 0; j < NV; ++j) _mm512_store_si512(lanes + 16 * j, v[j]);
                do {
                    const unsigned t = (unsigned)__builtin_ctzll(amb);
                    amb &= amb - 1;
                    const unsigned before = (unsigned)__builtin_popcountll(acc & ((1ull << t) - 1));
                    if (lanes[t] <= (uint32_t)i - before) acc |= 1ull << t;
                } while (amb);
            }
        }
        if (!(MODE & 1)) {
            uint32_t *dst = steps + (k - 1 - i);
            for (int j = 0; j < NV; ++j) {
                const __mmask16 a = (__mmask16)(acc >> (16 * j));
                if (MODE & 4) _mm512_storeu_si512(dst, _mm512_maskz_compress_epi32(a, v[j]));
                else _mm512_mask_compressstoreu_epi32(dst, a, v[j]);
                dst += __builtin_popcount((unsigned)a);
            }
        }
        i -= __builtin_popcountll(acc);
        rp += W;
    }
    return i;
}

// per-lane "sure" thresholds (lane t has at most t accepted lanes before it), level boundary by truncation at the
// r-th accepted lane (pdep), so the loop runs a whole level without a scalar tail
template <int NV>
__attribute__((target("avx512f,popcnt,bmi,bmi2"))) static int64_t run_lane(const uint32_t *raw, size_t &rp, int64_t i, int64_t lo, uint32_t mask,
                                                                            int64_t k, uint32_t *steps) {
    constexpr int W = 16 * NV;
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    const __m512i lane_id = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    alignas(64) uint32_t lanes[W];
    while (i > lo) {
        const __m512i vi = _mm512_set1_epi32((int)i);
        __m512i v[NV];
        uint64_t acc = 0, maybe = 0;
        for (int j = 0; j < NV; ++j) {
            v[j] = _mm512_and_si512(_mm512_loadu_si512(raw + rp + 16 * j), vmask);
            const __m512i sure = _mm512_sub_epi32(vi, _mm512_add_epi32(lane_id, _mm512_set1_epi32(16 * j)));
            acc |= (uint64_t)_mm512_cmple_epu32_mask(v[j], sure) << (16 * j);
            maybe |= (uint64_t)_mm512_cmple_epu32_mask(v[j], vi) << (16 * j);
        }
        uint64_t amb = maybe & ~acc;
        if (__builtin_expect(amb != 0, 0)) {
            for (int j = 0; j < NV; ++j) _mm512_store_si512(lanes + 16 * j, v[j]);
            do {
                const unsigned t = (unsigned)__builtin_ctzll(amb);
                amb &= amb - 1;
                const unsigned before = (unsigned)__builtin_popcountll(acc & ((1ull << t) - 1));
                if (lanes[t] <= (uint32_t)i - before) acc |= 1ull << t;
            } while (amb);
        }
        unsigned n_acc = (unsigned)__builtin_popcountll(acc);
        unsigned consumed = W;
        const uint64_t room = (uint64_t)(i - lo);                 // accepts left on this level (mask)
        if (__builtin_expect(n_acc >= room, 0)) {         // (== too: the lanes after the last accept of a level belong to the next mask)
            const unsigned pos = (unsigned)__builtin_ctzll(_pdep_u64(1ull << (room - 1), acc));
            acc &= (2ull << pos) - 1;
            n_acc = (unsigned)room;
            consumed = pos + 1;
        }
        uint32_t *dst = steps + (k - 1 - i);
        for (int j = 0; j < NV; ++j) {
            const __mmask16 a = (__mmask16)(acc >> (16 * j));
            _mm512_mask_compressstoreu_epi32(dst, a, v[j]);
            dst += __builtin_popcount((unsigned)a);
        }
        i -= n_acc;
        rp += consumed;
    }
    return i;
}

// variant: ambiguous lanes looked up in the raw words themselves (no vector store -> scalar load), FIRST resolution step
// unconditional (a batch without ambiguous lanes re-checks its last lane, which is harmless), a loop only for more
template <int NV, int VAR>
__attribute__((target("avx512f,popcnt,bmi,bmi2"))) static int64_t run_lane2(const uint32_t *raw, size_t &rp, int64_t i, int64_t lo, uint32_t mask,
                                                                             int64_t k, uint32_t *steps) {
    constexpr int W = 16 * NV;
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    const __m512i lane_id = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    while (i > lo) {
        const __m512i vi = _mm512_set1_epi32((int)i);
        __m512i v[NV];
        uint64_t acc = 0, maybe = 0;
        for (int j = 0; j < NV; ++j) {
            v[j] = _mm512_and_si512(_mm512_loadu_si512(raw + rp + 16 * j), vmask);
            const __m512i sure = _mm512_sub_epi32(vi, _mm512_add_epi32(lane_id, _mm512_set1_epi32(16 * j)));
            acc |= (uint64_t)_mm512_cmple_epu32_mask(v[j], sure) << (16 * j);
            maybe |= (uint64_t)_mm512_cmple_epu32_mask(v[j], vi) << (16 * j);
        }
        uint64_t amb = maybe & ~acc;
        const uint32_t *words = raw + rp;
        if (VAR & 1) {                               // one unconditional step
            const unsigned t = (unsigned)__builtin_ctzll(amb | (1ull << (W - 1)));
            const uint64_t bit = 1ull << t;
            const unsigned before = (unsigned)__builtin_popcountll(acc & (bit - 1));
            acc |= ((words[t] & mask) <= (uint32_t)i - before) ? bit : 0;
            amb &= amb - 1;
        }
        if (VAR & 2) {                               // a second one
            const unsigned t = (unsigned)__builtin_ctzll(amb | (1ull << (W - 1)));
            const uint64_t bit = 1ull << t;
            const unsigned before = (unsigned)__builtin_popcountll(acc & (bit - 1));
            acc |= ((words[t] & mask) <= (uint32_t)i - before) ? bit : 0;
            amb &= amb - 1;
        }
        if (!(VAR & 4))
        while (__builtin_expect(amb != 0, 0)) {
            const unsigned t = (unsigned)__builtin_ctzll(amb);
            amb &= amb - 1;
            const unsigned before = (unsigned)__builtin_popcountll(acc & ((1ull << t) - 1));
            acc |= ((words[t] & mask) <= (uint32_t)i - before) ? (1ull << t) : 0;
        }
        unsigned n_acc = (unsigned)__builtin_popcountll(acc);
        unsigned consumed = W;
        const uint64_t room = (uint64_t)(i - lo);
        if (__builtin_expect(n_acc >= room, 0)) {
            const unsigned pos = (unsigned)__builtin_ctzll(_pdep_u64(1ull << (room - 1), acc));
            acc &= (2ull << pos) - 1;
            n_acc = (unsigned)room;
            consumed = pos + 1;
        }
        uint32_t *dst = steps + (k - 1 - i);
        for (int j = 0; j < NV; ++j) {
            const __mmask16 a = (__mmask16)(acc >> (16 * j));
            if (VAR & 8) _mm512_storeu_si512(dst, _mm512_maskz_compress_epi32(a, v[j]));
            else _mm512_mask_compressstoreu_epi32(dst, a, v[j]);
            dst += __builtin_popcount((unsigned)a);
        }
        i -= n_acc;
        rp += consumed;
    }
    return i;
}

template <int TOPNV, int MIDNV, int VAR>
static double shuffle_lane2(const uint32_t *raw, size_t n_raw, int64_t k, int perms, uint32_t *steps, size_t *used) {
    size_t rp = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < perms; ++q) {
        int64_t i = k - 1;
        if (rp + 3 * k > n_raw) rp = 0;
        while (i > 0) {
            const uint32_t mask = mask_of((uint32_t)i);
            const int64_t lo = mask >> 1;
            if (mask >= 2047) i = TOPNV == 4 ? run_lane2<4, VAR>(raw, rp, i, lo, mask, k, steps) : TOPNV == 2 ? run_lane2<2, VAR>(raw, rp, i, lo, mask, k, steps) : run_lane2<1, VAR>(raw, rp, i, lo, mask, k, steps);
            else if (mask >= 127) i = MIDNV == 4 ? run_lane2<4, VAR>(raw, rp, i, lo, mask, k, steps) : MIDNV == 2 ? run_lane2<2, VAR>(raw, rp, i, lo, mask, k, steps) : run_lane2<1, VAR>(raw, rp, i, lo, mask, k, steps);
            else if (mask >= 31) i = run_lane2<1, VAR>(raw, rp, i, lo, mask, k, steps);
            while (i > lo) {
                const uint32_t v = raw[rp++] & mask;
                steps[k - 1 - i] = v;
                i -= (v <= (uint32_t)i);
            }
        }
    }
    *used = rp;
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template <int TOPNV, int MIDNV>
static double shuffle_lane(const uint32_t *raw, size_t n_raw, int64_t k, int perms, uint32_t *steps, size_t *used) {
    size_t rp = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < perms; ++q) {
        int64_t i = k - 1;
        if (rp + 3 * k > n_raw) rp = 0;
        while (i > 0) {
            const uint32_t mask = mask_of((uint32_t)i);
            const int64_t lo = mask >> 1;
            if (mask >= 2047) i = TOPNV == 4 ? run_lane<4>(raw, rp, i, lo, mask, k, steps) : TOPNV == 2 ? run_lane<2>(raw, rp, i, lo, mask, k, steps) : run_lane<1>(raw, rp, i, lo, mask, k, steps);
            else if (mask >= 127) i = MIDNV == 4 ? run_lane<4>(raw, rp, i, lo, mask, k, steps) : MIDNV == 2 ? run_lane<2>(raw, rp, i, lo, mask, k, steps) : run_lane<1>(raw, rp, i, lo, mask, k, steps);
            else if (mask >= 31) i = run_lane<1>(raw, rp, i, lo, mask, k, steps);
            while (i > lo) {
                const uint32_t v = raw[rp++] & mask;
                steps[k - 1 - i] = v;
                i -= (v <= (uint32_t)i);
            }
        }
    }
    *used = rp;
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template <int MODE, int MAXNV>
static double shuffle_all(const uint32_t *raw, size_t n_raw, int64_t k, int perms, uint32_t *steps, size_t *used) {
    size_t rp = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int q = 0; q < perms; ++q) {
        int64_t i = k - 1;
        if (rp + 3 * k > n_raw) rp = 0;
        while (i > 0) {
            const uint32_t mask = mask_of((uint32_t)i);
            const int64_t lo = mask >> 1;
            if (MAXNV >= 4 && mask >= 2047) i = run<4, MODE>(raw, rp, i, lo, mask, k, steps);
            if (MAXNV >= 2 && mask >= 511) i = run<2, MODE>(raw, rp, i, lo, mask, k, steps);
            i = run<1, MODE>(raw, rp, i, lo, mask, k, steps);
            while (i > lo) {
                const uint32_t v = raw[rp++] & mask;
                steps[k - 1 - i] = v;
                i -= (v <= (uint32_t)i);
            }
        }
    }
    *used = rp;
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
    const int64_t k = 3789;
    const int perms = 1000;
    const size_t n_raw = (size_t)1 << 23;
    std::vector<uint32_t> raw(n_raw + 256), steps(k + 128);
    MT19937 rng(0);
    auto t0 = std::chrono::steady_clock::now();
    rng.bulk(raw.data(), n_raw);
    const double gen_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("MT19937 bulk: %.3f ms per %zu words = %.3f ns/word (one perm of %lld needs ~%.0f words)\n", gen_ms, n_raw, 1e6 * gen_ms / n_raw,
           (long long)k, 5200.0);
    size_t used = 0;
    for (int rep = 0; rep < 3; ++rep) {
        printf("full (mem compress)        : %.3f ms / 1000 perms\n", shuffle_all<0, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("full, reg compress + store : %.3f\n", shuffle_all<4, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("branch-free first two      : %.3f\n", shuffle_all<8, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("branch-free + reg compress : %.3f\n", shuffle_all<12, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("no store                   : %.3f\n", shuffle_all<1, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("no ambiguity resolution    : %.3f\n", shuffle_all<2, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("neither                    : %.3f\n", shuffle_all<3, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("W <= 32 (full)             : %.3f\n", shuffle_all<0, 2>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("W <= 32 reg compress       : %.3f\n", shuffle_all<4, 2>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("W = 16 only (full)         : %.3f\n", shuffle_all<0, 1>(raw.data(), n_raw, k, perms, steps.data(), &used));
    }
    {   // correctness of the new loop against the plain scalar rule, a few permutations
        std::vector<uint32_t> ref(k + 128), got(k + 128);
        size_t rp_ref = 0, rp_got = 0;
        bool ok = true;
        for (int q = 0; q < 200 && ok; ++q) {
            int64_t i = k - 1;
            while (i > 0) { const uint32_t mask = mask_of((uint32_t)i); const uint32_t v = raw[rp_ref++] & mask; ref[k - 1 - i] = v; i -= (v <= (uint32_t)i); }
            i = k - 1;
            while (i > 0) {
                const uint32_t mask = mask_of((uint32_t)i); const int64_t lo = mask >> 1;
                if (mask >= 2047) i = run_lane<4>(raw.data(), rp_got, i, lo, mask, k, got.data());
                else if (mask >= 127) i = run_lane<2>(raw.data(), rp_got, i, lo, mask, k, got.data());
                else if (mask >= 31) i = run_lane<1>(raw.data(), rp_got, i, lo, mask, k, got.data());
                while (i > lo) { const uint32_t v = raw[rp_got++] & mask; got[k - 1 - i] = v; i -= (v <= (uint32_t)i); }
            }
            ok = rp_ref == rp_got && !memcmp(ref.data(), got.data(), (k - 1) * 4);
            if (!ok) { int64_t f = 0; while (f < k - 1 && ref[f] == got[f]) ++f; printf("perm %d: rp %zu vs %zu, first difference at step %lld (i = %lld): %u vs %u\n", q, rp_ref, rp_got, (long long)f, (long long)(k - 1 - f), ref[f], got[f]); }
        }
        printf("new loop equals the scalar rule on 200 permutations: %s\n", ok ? "yes" : "NO");
    }
    for (int rep = 0; rep < 3; ++rep) {
        printf("per-lane thresholds 4/4 : %.3f\n", shuffle_lane<4, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("per-lane thresholds 4/2 : %.3f\n", shuffle_lane<4, 2>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("per-lane thresholds 2/2 : %.3f\n", shuffle_lane<2, 2>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("per-lane thresholds 2/1 : %.3f\n", shuffle_lane<2, 1>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("per-lane thresholds 1/1 : %.3f\n", shuffle_lane<1, 1>(raw.data(), n_raw, k, perms, steps.data(), &used));
    }
    {
        std::vector<uint32_t> ref(k + 128), got(k + 128);
        size_t rp_ref = 0, rp_a = 0, rp_b = 0;
        bool ok = true;
        for (int q = 0; q < 300 && ok; ++q) {
            int64_t i = k - 1;
            while (i > 0) { const uint32_t mask = mask_of((uint32_t)i); const uint32_t v = raw[rp_ref++] & mask; ref[k - 1 - i] = v; i -= (v <= (uint32_t)i); }
            for (int var = 0; var < 2; ++var) {
                size_t &rp = var ? rp_b : rp_a;
                i = k - 1;
                while (i > 0) {
                    const uint32_t mask = mask_of((uint32_t)i); const int64_t lo = mask >> 1;
                    if (mask >= 2047) i = var ? run_lane2<4, 3>(raw.data(), rp, i, lo, mask, k, got.data()) : run_lane2<2, 1>(raw.data(), rp, i, lo, mask, k, got.data());
                    else if (mask >= 127) i = var ? run_lane2<2, 3>(raw.data(), rp, i, lo, mask, k, got.data()) : run_lane2<1, 0>(raw.data(), rp, i, lo, mask, k, got.data());
                    else if (mask >= 31) i = var ? run_lane2<1, 9>(raw.data(), rp, i, lo, mask, k, got.data()) : run_lane2<1, 1>(raw.data(), rp, i, lo, mask, k, got.data());
                    while (i > lo) { const uint32_t v = raw[rp++] & mask; got[k - 1 - i] = v; i -= (v <= (uint32_t)i); }
                }
                ok = ok && rp_ref == rp && !memcmp(ref.data(), got.data(), (k - 1) * 4);
            }
        }
        printf("lane2 variants equal the scalar rule on 300 permutations: %s\n", ok ? "yes" : "NO");
    }
    {   // MT19937 into a small, cache-resident buffer (what an in-thread generator would cost)
        MT19937 g2(0);
        std::vector<uint32_t> small(4992);
        auto a0 = std::chrono::steady_clock::now();
        uint32_t sink = 0;
        for (int r = 0; r < 2000; ++r) { g2.bulk(small.data(), small.size()); sink ^= small[r % 4992]; }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();
        printf("MT19937 into a 20 KB buffer: %.3f ns/word (sink %u)\n", 1e6 * ms / (2000.0 * 4992), sink);
    }
    for (int rep = 0; rep < 3; ++rep) {
        printf("lane2 2/1 raw lookup, loop only     : %.3f\n", shuffle_lane2<2, 1, 0>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 2/1 one unconditional step    : %.3f\n", shuffle_lane2<2, 1, 1>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 2/1 two unconditional steps   : %.3f\n", shuffle_lane2<2, 1, 3>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 2/2 one unconditional step    : %.3f\n", shuffle_lane2<2, 2, 1>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 4/2 one unconditional step    : %.3f\n", shuffle_lane2<4, 2, 1>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 2/1 one step + reg compress   : %.3f\n", shuffle_lane2<2, 1, 9>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 2/1 NO resolution (floor)     : %.3f\n", shuffle_lane2<2, 1, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
        printf("lane2 4/2 NO resolution (floor)     : %.3f\n", shuffle_lane2<4, 2, 4>(raw.data(), n_raw, k, perms, steps.data(), &used));
    }
    // the product's threaded stream, for reference
    for (int rep = 0; rep < 2; ++rep) {
        DrawStream ds(0);
        std::vector<uint32_t> st(k * 128);
        auto a = std::chrono::steady_clock::now();
        for (int64_t q = 0; q < perms; ++q) ds.shuffle_targets(k, st.data() + (q % 128) * k);
        printf("product stream (draws.cpp): %.3f ms / 1000 perms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count());
    }
    return 0;
}
