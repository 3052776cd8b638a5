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
// draws.  So a batch is resolved with two vector compares and one compress-store (AVX-512);
// the few ambiguous lanes are settled one by one in lane order, each against the number of
// accepted lanes before it (a popcount of the mask built so far), which is exactly the
// sequential rule.  The raw MT19937 output is produced by a helper thread, one block ahead.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>
#include <cstdio>
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

// Raw output blocks come from a helper thread (MT19937 state update + tempering, about a
// quarter of the stream's cost), up to kSlots - 1 blocks ahead of the consumer; single producer,
// single consumer.  A producer that finds the ring full sleeps (it is 3-4x faster than the
// consumer, and a spinning helper per rank would eat into a container's CPU quota).
struct DrawStream {
    static constexpr size_t kBlock = 1 << 15;        // words per block
    static constexpr size_t kCarry = 128;             // unread words carried in front of a fresh block
    static constexpr int kSlots = 8;
    MT19937 rng;
    std::vector<uint32_t> slot[kSlots];              // [kCarry + kBlock]
    std::atomic<int> ready[kSlots];                  // 1 = filled by the producer, 0 = free
    std::atomic<bool> stop{false};
    std::thread producer;
    bool started = false;
    int cur = 0;                                     // slot the consumer reads
    const uint32_t *raw_ptr = nullptr;               // words of the current slot (carry included)
    size_t rp = 0, avail = 0;
    bool use_avx512 = false;
    bool reg_compress = false;

    explicit DrawStream(uint32_t seed) : rng(seed) {
        for (int b = 0; b < kSlots; ++b) {
            slot[b].resize(kCarry + kBlock);
            ready[b].store(0, std::memory_order_relaxed);
        }
#if defined(__x86_64__)
        use_avx512 = __builtin_cpu_supports("avx512f");
        // memory-form vpcompressd measured as fast as register compress + store on Zen 5 (EPYC 9575F) and
        // faster on the Xeon of the build container; "reg" remains selectable for other cores
        if (const char *e = getenv("SAFE_HIP_DRAW_COMPRESS")) reg_compress = e[0] == 'r';   // "reg" / "mem"
#endif
    }

    ~DrawStream() {
        if (started) {
            stop.store(true, std::memory_order_release);
            producer.join();
        }
    }

    // SAFE_HIP_DRAW_PAIR=1: the consumer (draw thread) stays on the core it runs on and this producer moves to that
    // core's SMT sibling -- the raw words then travel through the shared L1/L2 instead of between cores
    // (tools/ubench/draw_stream: 2.4 vs 3.0-3.3 ms per 1000 x 3789 draws)
    int pair_cpu = -1;
    void pair_with_sibling() {
        const char *e = getenv("SAFE_HIP_DRAW_PAIR");
        if (!(e && e[0] == '1')) return;
        const int cpu = sched_getcpu();
        if (cpu < 0) return;
        char path[128];
        snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        FILE *f = fopen(path, "r");
        if (!f) return;
        int a = -1, b = -1;
        char sep = 0;
        const int got = fscanf(f, "%d%c%d", &a, &sep, &b);
        fclose(f);
        if (got != 3 || (sep != ',' && sep != '-') || a < 0 || b < 0) return;
        const int sib = a == cpu ? b : a;
        cpu_set_t allowed, one;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0 || !CPU_ISSET(sib, &allowed) || !CPU_ISSET(cpu, &allowed)) return;
        CPU_ZERO(&one);
        CPU_SET(cpu, &one);
        if (pthread_setaffinity_np(pthread_self(), sizeof(one), &one) == 0) pair_cpu = sib;
    }

    void produce(int b) {                            // owns `rng` once started
        if (pair_cpu >= 0) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(pair_cpu, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
        }
        for (;;) {
            while (ready[b].load(std::memory_order_acquire) != 0) {
                if (stop.load(std::memory_order_acquire)) return;
                std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
            if (stop.load(std::memory_order_acquire)) return;
            rng.bulk(slot[b].data() + kCarry, kBlock);
            ready[b].store(1, std::memory_order_release);
            b = (b + 1) % kSlots;
        }
    }

    // keep at least `want` (<= kCarry) unread outputs contiguous
    inline void ensure(size_t want) {
        if (avail - rp >= want) return;
        next_block();
    }

    void next_block() {
        if (!started) {                              // first use: block 0 inline, then the helper runs ahead
            rng.bulk(slot[0].data() + kCarry, kBlock);
            ready[0].store(1, std::memory_order_relaxed);
            started = true;
            cur = 0;
            raw_ptr = slot[0].data();
            rp = kCarry;
            avail = kCarry + kBlock;
            pair_with_sibling();
            producer = std::thread([this] { produce(1); });
            return;
        }
        const int nxt = (cur + 1) % kSlots;
        while (ready[nxt].load(std::memory_order_acquire) == 0) std::this_thread::yield();
        // carry the unread tail of the current slot in front of the next block (the producer
        // only ever writes behind kCarry)
        const size_t left = avail - rp;              // < want <= kCarry
        memcpy(slot[nxt].data() + (kCarry - left), raw_ptr + rp, left * sizeof(uint32_t));
        ready[cur].store(0, std::memory_order_release);      // the producer may refill the old slot
        cur = nxt;
        raw_ptr = slot[nxt].data();
        rp = kCarry - left;
        avail = kCarry + kBlock;
    }

    // scalar: until i leaves (floor_i, i0]; returns the new i
    inline int64_t scalar_run(int64_t i, int64_t floor_i, uint32_t mask, int64_t k, uint32_t *steps) {
        while (i > floor_i) {
            ensure(1);
            const uint32_t v = raw_ptr[rp++] & mask;
            steps[k - 1 - i] = v;                 // a rejected draw is overwritten by the next one
            i -= (v <= static_cast<uint32_t>(i));
        }
        return i;
    }

#if defined(__x86_64__)
    // NV vectors of 16 draws per batch.  The acceptance threshold moves by at most W = 16 * NV inside
    // a batch, so v <= i - W is accepted and v > i rejected whatever came before; the lanes in
    // between (about W / mask of them) are settled in lane order against the exact number of accepted
    // lanes before them.  Wider batches amortise the loop-carried chain i -> compare -> mask ->
    // popcount -> i, which is what bounds this loop.
    template <int NV>
    __attribute__((target("avx512f,popcnt,bmi,bmi2")))
    int64_t vector_run(int64_t i, int64_t lo, uint32_t mask, int64_t k, uint32_t *steps) {
        constexpr int W = 16 * NV;
        const __m512i vmask = _mm512_set1_epi32(static_cast<int>(mask));
        alignas(64) uint32_t lanes[W];
        while (i - W > lo) {
            ensure(W);
            const __m512i sure_thr = _mm512_set1_epi32(static_cast<int>(i - W));
            const __m512i maybe_thr = _mm512_set1_epi32(static_cast<int>(i));
            __m512i v[NV];
            uint64_t acc = 0, maybe = 0;
            for (int j = 0; j < NV; ++j) {
                v[j] = _mm512_and_si512(_mm512_loadu_si512(raw_ptr + rp + 16 * j), vmask);
                acc |= static_cast<uint64_t>(_mm512_cmple_epu32_mask(v[j], sure_thr)) << (16 * j);
                maybe |= static_cast<uint64_t>(_mm512_cmple_epu32_mask(v[j], maybe_thr)) << (16 * j);
            }
            uint64_t amb = maybe & ~acc;
            if (amb) {
                for (int j = 0; j < NV; ++j) _mm512_store_si512(lanes + 16 * j, v[j]);
                do {
                    const unsigned t = static_cast<unsigned>(__builtin_ctzll(amb));
                    amb &= amb - 1;
                    const unsigned before = static_cast<unsigned>(__builtin_popcountll(acc & ((1ull << t) - 1ull)));
                    if (lanes[t] <= static_cast<uint32_t>(i) - before) acc |= 1ull << t;
                } while (amb);
            }
            uint32_t *dst = steps + (k - 1 - i);
            for (int j = 0; j < NV; ++j) {
                const __mmask16 a = static_cast<__mmask16>(acc >> (16 * j));
                if (reg_compress) _mm512_storeu_si512(dst, _mm512_maskz_compress_epi32(a, v[j]));   // tail lanes are overwritten later
                else _mm512_mask_compressstoreu_epi32(dst, a, v[j]);
                dst += __builtin_popcount(static_cast<unsigned>(a));
            }
            i -= __builtin_popcountll(acc);
            rp += W;
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
            if (use_avx512) {
                if (mask >= 2047) i = vector_run<4>(i, lo, mask, k, steps);
                if (mask >= 511) i = vector_run<2>(i, lo, mask, k, steps);
                i = vector_run<1>(i, lo, mask, k, steps);
            }
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
