// The draw stream of the legacy NumPy permutation (pure host C++, compiled by the host
// compiler -- not a HIP translation unit -- so that it can use x86 vector intrinsics).
//
// np.random.seed(s) = MT19937 init_genrand(s); np.random.permutation(x) = Fisher-Yates from the
// top, j = random_interval(i): mask = smallest 2^b - 1 >= i, draw (next_u32 & mask) until <= i
// (the stream is continuous across permutations).  This file produces,
// for one shuffle of k items, the accepted swap targets in STEP order:
//     steps[s] = j drawn for i = k-1-s,   s = 0 .. k-2.
//
// Rejection sampling is sequential in principle (whether a draw is accepted depends on how
// many were accepted before it), but inside a batch of 16 draws the threshold moves by at most
// 16: a draw v <= i-16 is accepted whatever happened before it, a draw v > i is rejected
// whatever happened before it, and only i-16 < v <= i is ambiguous -- about 16/mask of the
// draws.  So batches without an ambiguous draw are resolved with one vector compare and one
// compress-store (AVX-512), and the rare ambiguous batch falls back to the scalar loop.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "draws.h"

namespace {

struct MT19937 {
    uint32_t mt[624];
    int pos;

    explicit MT19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + static_cast<uint32_t>(i);
        pos = 624;
    }

#if defined(__x86_64__)
    __attribute__((target_clones("default", "avx2", "avx512f")))
#endif
    void refill() {
        int i = 0;
        for (; i < 624 - 397; ++i) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
            mt[i] = mt[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        for (; i < 623; ++i) {
            const uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
            mt[i] = mt[i - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        }
        const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
        pos = 0;
    }

#if defined(__x86_64__)
    __attribute__((target_clones("default", "avx2", "avx512f")))
#endif
    static void temper(uint32_t *out, const uint32_t *in, size_t n) {
        for (size_t i = 0; i < n; ++i) {
            uint32_t y = in[i];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[i] = y;
        }
    }

    void bulk(uint32_t *out, size_t count) {
        size_t done = 0;
        while (done < count) {
            if (pos == 624) refill();
            const size_t take = std::min<size_t>(624 - pos, count - done);
            temper(out + done, mt + pos, take);
            pos += static_cast<int>(take);
            done += take;
        }
    }
};

inline uint32_t mask_for(uint32_t i) {
    uint32_t m = i;
    m |= m >> 1;
    m |= m >> 2;
    m |= m >> 4;
    m |= m >> 8;
    m |= m >> 16;
    return m;
}

}  // namespace

struct DrawStream {
    MT19937 rng;
    std::vector<uint32_t> raw;
    size_t rp = 0, avail = 0;
    bool use_avx512 = false;

    explicit DrawStream(uint32_t seed) : rng(seed), raw(1 << 15) {
#if defined(__x86_64__)
        use_avx512 = __builtin_cpu_supports("avx512f");
#endif
    }

    inline void ensure(size_t want) {
        // keep at least `want` unread outputs contiguous (leftovers are moved to the front)
        if (avail - rp >= want) return;
        const size_t left = avail - rp;
        memmove(raw.data(), raw.data() + rp, left * sizeof(uint32_t));
        rng.bulk(raw.data() + left, raw.size() - left);
        rp = 0;
        avail = raw.size();
    }

    // scalar: until i leaves (stop, i0]; returns the new i
    inline int64_t scalar_run(int64_t i, int64_t stop, uint32_t mask, int64_t k, uint32_t *steps) {
        while (i > stop) {
            ensure(1);
            const uint32_t v = raw[rp++] & mask;
            steps[k - 1 - i] = v;                 // a rejected draw is overwritten by the next one
            i -= (v <= static_cast<uint32_t>(i));
        }
        return i;
    }

#if defined(__x86_64__)
    __attribute__((target("avx512f")))
    int64_t vector_run(int64_t i, int64_t lo, uint32_t mask, int64_t k, uint32_t *steps) {
        const __m512i vmask = _mm512_set1_epi32(static_cast<int>(mask));
        while (i - 16 > lo) {
            ensure(16);
            const __m512i v = _mm512_and_si512(_mm512_loadu_si512(raw.data() + rp), vmask);
            const __mmask16 sure = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32(static_cast<int>(i - 16)));
            const __mmask16 maybe = _mm512_cmple_epu32_mask(v, _mm512_set1_epi32(static_cast<int>(i)));
            if (sure != maybe) {                  // an ambiguous draw: resolve this batch one by one
                const size_t end = rp + 16;
                while (rp < end) {
                    const uint32_t x = raw[rp++] & mask;
                    steps[k - 1 - i] = x;
                    i -= (x <= static_cast<uint32_t>(i));
                }
                continue;
            }
            _mm512_mask_compressstoreu_epi32(steps + (k - 1 - i), sure, v);
            i -= __builtin_popcount(static_cast<unsigned>(sure));
            rp += 16;
        }
        return i;
    }
#endif

    void shuffle_targets(int64_t k, uint32_t *steps) {
        int64_t i = k - 1;
        while (i > 0) {
            const uint32_t mask = mask_for(static_cast<uint32_t>(i));
            const int64_t lo = mask >> 1;         // the mask is unchanged while i is in (lo, mask]
#if defined(__x86_64__)
            if (use_avx512) i = vector_run(i, lo, mask, k, steps);
#endif
            i = scalar_run(i, lo, mask, k, steps);
        }
    }
};

// Streaming copy (no read-for-ownership of the destination lines): the draw thread hands a
// chunk of targets to the swap workers on other cores; writing the shared buffer with ordinary
// stores would first have to pull every line back from the workers' caches.
void draws_nt_copy(void *dst, const void *src, size_t bytes) {
#if defined(__x86_64__)
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const __m128i *s = static_cast<const __m128i *>(src);
        __m128i *d = static_cast<__m128i *>(dst);
        for (size_t i = 0; i < n16; ++i) _mm_stream_si128(d + i, _mm_load_si128(s + i));
        _mm_sfence();
        memcpy(static_cast<char *>(dst) + n16 * 16, static_cast<const char *>(src) + n16 * 16, bytes - n16 * 16);
        return;
    }
#endif
    memcpy(dst, src, bytes);
}

DrawStream *draw_stream_new(uint32_t seed) { return new DrawStream(seed); }
void draw_stream_free(DrawStream *s) { delete s; }
void draw_stream_targets(DrawStream *s, int64_t k, uint32_t *steps) { s->shuffle_targets(k, steps); }
