// Context, error reporting and raw device-memory helpers of libsafe_hip.so.
#include "common.h"
#include "ring.h"

#include <atomic>
#include <chrono>
#include <map>
#include <thread>

static thread_local char g_error[1024] = "";

void safe_trace(const char *what) {
    static const bool on = getenv("SAFE_HIP_TRACE") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[safe_hip %9.3f ms] %s\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}

void safe_warn_diagnostic(const char *name) {
    static std::mutex mu;
    static std::vector<std::string> seen;
    std::lock_guard<std::mutex> lk(mu);
    for (const std::string &n : seen)
        if (n == name) return;
    seen.emplace_back(name);
    fprintf(stderr, "[safe_hip] WARNING: diagnostic switch %s is set: kernels skip work, the results of this process are INVALID\n", name);
}

void safe_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

// Host waits.  By default hipStreamSynchronize (the runtime spins: lowest latency, one busy core per waiting thread).  With
// blocking waits switched on (safe_set_blocking_sync / SAFE_HIP_BLOCKING_SYNC=1: several ranks sharing few host cores) the
// thread sleeps on an interrupt-backed event instead -- a few tens of microseconds later, no CPU meanwhile.
static std::atomic<int> g_blocking_sync{-1};
std::atomic<long long> g_alloc_calls{0};

static bool blocking_sync_on() {
    int v = g_blocking_sync.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = getenv("SAFE_HIP_BLOCKING_SYNC");
        v = e && atoi(e) != 0 ? 1 : 0;
        g_blocking_sync.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

bool safe_blocking_sync_selected() { return blocking_sync_on(); }

unsigned safe_event_flags(unsigned base) { return blocking_sync_on() ? (base | hipEventBlockingSync) : base; }

hipError_t safe_stream_sync(hipStream_t s) {
    if (!blocking_sync_on()) return hipStreamSynchronize(s);
    thread_local std::map<int, hipEvent_t> events;               // one blocking event per (thread, device)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipEvent_t &ev = events[dev];
    if (!ev && (e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(ev, s)) != hipSuccess) return e;
    return hipEventSynchronize(ev);
}

int ctx_scratch(safe_ctx *ctx, int slot, size_t bytes, void **out) {
    if (bytes == 0) bytes = 1;
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
            SAFE_HIP_CHECK(safe_stream_sync(ctx->side_stream));
            SAFE_HIP_CHECK(hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 8;
        g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipMalloc(&ctx->scratch[slot], want);
        if (e != hipSuccess) {
            safe_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            return SAFE_E_NOMEM;
        }
        ctx->scratch_bytes[slot] = want;
    }
    *out = ctx->scratch[slot];
    return SAFE_OK;
}

static size_t block_class(size_t bytes) { return (std::max<size_t>(bytes, 1) + 255) / 256 * 256; }

int ctx_block_alloc(safe_ctx *ctx, size_t bytes, void **out) {
    const size_t want = block_class(bytes);
    for (size_t i = 0; i < ctx->block_cache.size(); ++i)
        if (ctx->block_cache[i].first == want) {
            *out = ctx->block_cache[i].second;
            ctx->block_cache.erase(ctx->block_cache.begin() + static_cast<std::ptrdiff_t>(i));
            return SAFE_OK;
        }
    g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess) {
        safe_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        return SAFE_E_NOMEM;
    }
    return SAFE_OK;
}

// the caller has synchronised every stream that may still touch the block
void ctx_block_free(safe_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    const size_t have = block_class(bytes);
    if (have > (8u << 20) || ctx->block_cache.size() >= 32) {
        (void)hipFree(p);
        return;
    }
    ctx->block_cache.emplace_back(have, p);
}

int ctx_events(safe_ctx *ctx, bool timing, size_t count, hipEvent_t **out) {
    std::vector<hipEvent_t> &pool = timing ? ctx->ev_timing : ctx->ev_plain;
    while (pool.size() < count) {
        hipEvent_t e = nullptr;
        SAFE_HIP_CHECK(timing ? hipEventCreateWithFlags(&e, safe_event_flags(hipEventDefault)) : hipEventCreateWithFlags(&e, safe_event_flags(hipEventDisableTiming)));
        pool.push_back(e);
    }
    *out = pool.data();
    return SAFE_OK;
}

int ctx_pinned(safe_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_bytes < bytes) {
        if (ctx->pinned) {
            SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
            SAFE_HIP_CHECK(safe_stream_sync(ctx->side_stream));
            SAFE_HIP_CHECK(hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            safe_set_error("hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            return SAFE_E_NOMEM;
        }
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return SAFE_OK;
}

extern "C" {

int safe_abi_version(void) { return SAFE_HIP_ABI_VERSION; }

const char *safe_last_error(void) { return g_error; }

int safe_device_count(int *count) {
    SAFE_REQUIRE(count != nullptr, "safe_device_count: count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *count = c;
    return SAFE_OK;
}

int safe_device_pci_bus_id(int device, char *buf, size_t buf_len) {
    SAFE_REQUIRE(buf != nullptr && buf_len >= 16, "safe_device_pci_bus_id: buffer of at least 16 bytes needed");
    SAFE_HIP_CHECK(hipDeviceGetPCIBusId(buf, static_cast<int>(buf_len), device));
    return SAFE_OK;
}

// The runtime opens a queue of its own for host <-> device copies the first time it needs one -- as far as can be told from
// outside, the first time a copy finds no free DMA engine -- and opening a queue on this platform means populating a 173 MiB
// context-save area plus 16 MiB in host memory on the spot: 7-12 ms, taken by one of the runtime's threads in the middle of
// whatever call is running.  Left alone that happened in one run of `bench.py --steps 20 --warmup 5` out of three (busy boxes) to
// ten, inside the first timed steps: one step of 10-15 ms among 3 ms ones, 4105-4108 minor faults, resident set +189 MiB
// (bench.py's step probe).  It never happens with HSA_ENABLE_SDMA=0 (every copy a shader copy from the start -- but the step is
// 0.8 ms slower), and it stops happening when the context starts with a burst of copies in both directions on a few streams at
// once: 0 of 90 driver-style runs with the burst against 7 of 60 without, interleaved on the same boxes, medians unchanged
// (round 5, a switch then).  Cost: ~35 ms of context creation (199 against 165 ms), 64 MiB of
// pinned and device memory for its duration.  Four streams (16 measured the same).  Best effort: errors are
// ignored.
static void ctx_open_copy_queue(safe_ctx *ctx) {
    const int n_streams = 4;
    const size_t bytes = size_t(8) << 20;
    void *host = nullptr, *dev = nullptr;
    if (hipHostMalloc(&host, 2 * n_streams * bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&dev, 2 * n_streams * bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (host) (void)hipHostFree(host);
        return;
    }
    memset(host, 0, 2 * n_streams * bytes);
    std::vector<hipStream_t> streams(n_streams, nullptr);
    for (hipStream_t &s : streams)
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
    for (int round = 0; round < 3; ++round)
        for (int i = 0; i < n_streams; ++i) {
            if (!streams[i]) continue;
            char *h = static_cast<char *>(host) + 2 * i * bytes, *d = static_cast<char *>(dev) + 2 * i * bytes;
            (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, streams[i]);
            (void)hipMemcpyAsync(h + bytes, d + bytes, bytes, hipMemcpyDeviceToHost, streams[i]);
        }
    for (hipStream_t s : streams)
        if (s) (void)hipStreamSynchronize(s);
    for (hipStream_t s : streams)
        if (s) (void)hipStreamDestroy(s);
    (void)hipFree(dev);
    (void)hipHostFree(host);
    (void)hipGetLastError();
    (void)ctx;
}

int safe_ctx_create(int device, safe_ctx **out) {
    SAFE_REQUIRE(out != nullptr, "safe_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        (void)hipGetLastError();
        safe_set_error("safe_ctx_create: no HIP device available (%s); libsafe_hip has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return SAFE_E_HIP;
    }
    SAFE_REQUIRE(device >= 0 && device < count, "safe_ctx_create: device %d out of range [0,%d)", device, count);
    SAFE_HIP_CHECK(hipSetDevice(device));
    if (blocking_sync_on()) {
        // every host wait of the runtime on this device (synchronous copies, torch's own synchronize) sleeps too; refused by
        // some runtimes once the device is active: the library's own waits (safe_stream_sync, blocking events) do not depend on it
        if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();
    }
    safe_ctx *ctx = new safe_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    SAFE_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    ctx->num_cu = prop.multiProcessorCount;
    ctx->hbm_bytes = static_cast<int64_t>(prop.totalGlobalMem);
    snprintf(ctx->arch, sizeof(ctx->arch), "%s", prop.gcnArchName);
    SAFE_HIP_CHECK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    {
        int lo = 0, hi = 0;       // numerically lowest value = highest priority
        SAFE_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        SAFE_HIP_CHECK(hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, hi));
    }
    SAFE_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    for (hipStream_t &ms : ctx->more_streams) SAFE_HIP_CHECK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
    SAFE_HIP_CHECK(hipEventCreateWithFlags(&ctx->t0, safe_event_flags(hipEventDefault)));
    SAFE_HIP_CHECK(hipEventCreateWithFlags(&ctx->t1, safe_event_flags(hipEventDefault)));
    SAFE_HIP_CHECK(hipEventCreateWithFlags(&ctx->k0, safe_event_flags(hipEventDefault)));
    SAFE_HIP_CHECK(hipEventCreateWithFlags(&ctx->k1, safe_event_flags(hipEventDefault)));
    ctx_open_copy_queue(ctx);
    *out = ctx;
    return SAFE_OK;
}

int safe_ctx_destroy(safe_ctx *ctx) {
    if (!ctx) return SAFE_OK;
    (void)hipSetDevice(ctx->device);
    (void)safe_stream_sync(ctx->stream);
    if (ctx->t0) (void)hipEventDestroy(ctx->t0);
    if (ctx->t1) (void)hipEventDestroy(ctx->t1);
    if (ctx->k0) (void)hipEventDestroy(ctx->k0);
    if (ctx->k1) (void)hipEventDestroy(ctx->k1);
    perms_cache_drop(ctx);
    draw_worker_shutdown(ctx);
    if (ctx->ring) ring_close(ctx->ring);
    ctx->ring = nullptr;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->d2h_ring) (void)hipHostFree(ctx->d2h_ring);
    for (hipEvent_t e : ctx->d2h_events)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->xc_events)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_timing) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_plain) (void)hipEventDestroy(e);
    for (int i = 0; i < safe_ctx::N_SCRATCH; ++i)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    for (auto &b : ctx->block_cache) (void)hipFree(b.second);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    for (hipStream_t ms : ctx->more_streams)
        if (ms) (void)hipStreamDestroy(ms);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SAFE_OK;
}

int safe_set_blocking_sync(int on) {
    g_blocking_sync.store(on ? 1 : 0, std::memory_order_relaxed);
    return SAFE_OK;
}

int safe_ctx_set_stream(safe_ctx *ctx, void *hip_stream) {
    SAFE_REQUIRE(ctx != nullptr, "safe_ctx_set_stream: ctx is NULL");
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return SAFE_OK;
}

int safe_ctx_sync(safe_ctx *ctx) {
    SAFE_REQUIRE(ctx != nullptr, "safe_ctx_sync: ctx is NULL");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

int safe_ctx_info(safe_ctx *ctx, int *num_cu, int64_t *hbm_bytes, char *arch, size_t arch_len) {
    SAFE_REQUIRE(ctx != nullptr, "safe_ctx_info: ctx is NULL");
    if (num_cu) *num_cu = ctx->num_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    if (arch && arch_len) snprintf(arch, arch_len, "%s", ctx->arch);
    return SAFE_OK;
}

int safe_dev_alloc(safe_ctx *ctx, size_t bytes, void **out_dev) {
    SAFE_REQUIRE(ctx != nullptr && out_dev != nullptr, "safe_dev_alloc: NULL argument");
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    uint8_t *p = nullptr;
    SAFE_TRY(dev_alloc(&p, bytes));
    *out_dev = p;
    return SAFE_OK;
}

int safe_dev_free(safe_ctx *ctx, void *dev) {
    SAFE_REQUIRE(ctx != nullptr, "safe_dev_free: ctx is NULL");
    if (!dev) return SAFE_OK;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    SAFE_HIP_CHECK(hipFree(dev));
    return SAFE_OK;
}

int safe_dev_memset(safe_ctx *ctx, void *dev, int value, size_t bytes) {
    SAFE_REQUIRE(ctx != nullptr && dev != nullptr, "safe_dev_memset: NULL argument");
    SAFE_HIP_CHECK(hipMemsetAsync(dev, value, bytes, ctx->stream));
    return SAFE_OK;
}

int safe_memcpy_h2d(safe_ctx *ctx, void *dev, const void *host, size_t bytes) {
    SAFE_REQUIRE(ctx != nullptr && (bytes == 0 || (dev && host)), "safe_memcpy_h2d: NULL argument");
    if (bytes == 0) return SAFE_OK;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

// A result matrix read back into a fresh NumPy array (SAFE.nes and its siblings: 139 MB each at configs[1], 1.6 GB at configs[3])
// is bound by the host side of a pageable copy: one thread takes every first-touch page fault and does the copy out of the
// runtime's staging buffer (8.2 ms per 139 MB = 17 GB/s measured).  Here the DMA engine fills a ring of pinned 4 MB slots
// (~50 GB/s) and a few host threads copy finished slots into the destination -- faults and copies in parallel.
static int d2h_pipelined(safe_ctx *ctx, char *host, const char *dev, size_t bytes, int workers, bool *fallback) {
    constexpr int R = safe_ctx::D2H_SLOTS;
    constexpr size_t S = safe_ctx::D2H_SLOT_BYTES;
    // one read-back at a time per context (ctypes releases the GIL: two Python threads may call safe_memcpy_d2h on one context)
    std::lock_guard<std::mutex> ring_guard(ctx->d2h_mu);
    if (!ctx->d2h_ring) {
        // published only when complete: a failed allocation or event leaves no half-made ring behind, and the caller takes the
        // plain copy (*fallback)
        void *ring_mem = nullptr;
        hipEvent_t evs[R] = {};
        g_alloc_calls.fetch_add(1, std::memory_order_relaxed);
        bool ok = hipHostMalloc(&ring_mem, R * S, hipHostMallocDefault) == hipSuccess;
        for (int i = 0; ok && i < R; ++i) ok = hipEventCreateWithFlags(&evs[i], safe_event_flags(hipEventDisableTiming)) == hipSuccess;
        if (!ok) {
            for (hipEvent_t e : evs)
                if (e) (void)hipEventDestroy(e);
            if (ring_mem) (void)hipHostFree(ring_mem);
            (void)hipGetLastError();
            *fallback = true;
            return SAFE_OK;
        }
        for (int i = 0; i < R; ++i) ctx->d2h_events[i] = evs[i];
        ctx->d2h_ring = ring_mem;
    }
    const bool sleepy = safe_blocking_sync_selected();     // few cores per rank: the waits below yield instead of spinning
    const int64_t n_chunks = static_cast<int64_t>((bytes + S - 1) / S);
    std::atomic<int64_t> issued{0};                       // chunks whose copy into their slot has been queued (event recorded)
    std::atomic<int64_t> slot_done[R];                    // per slot: the last chunk copied out of it, + 1
    for (auto &d : slot_done) d.store(0, std::memory_order_relaxed);
    std::atomic<int> failed{0};
    char *ring = static_cast<char *>(ctx->d2h_ring);
    const int device = ctx->device;
    auto work = [&](int w) {
        (void)hipSetDevice(device);
        for (int64_t c = w; c < n_chunks; c += workers) {
            while (issued.load(std::memory_order_acquire) <= c) {
                if (failed.load(std::memory_order_relaxed)) return;
                if (sleepy) std::this_thread::yield();
#if defined(__x86_64__)
                else __builtin_ia32_pause();
#endif
            }
            const int s = static_cast<int>(c % R);
            if (hipEventSynchronize(ctx->d2h_events[s]) != hipSuccess) {
                failed.store(1);
                return;
            }
            const size_t off = static_cast<size_t>(c) * S, len = std::min(S, bytes - off);
            memcpy(host + off, ring + static_cast<size_t>(s) * S, len);
            slot_done[s].store(c + 1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int w = 0; w < workers; ++w) pool.emplace_back(work, w);
    hipError_t e = hipSuccess;
    for (int64_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
        const int s = static_cast<int>(c % R);
        while (c >= R && slot_done[s].load(std::memory_order_acquire) < c - R + 1) {      // the slot's previous chunk has been copied out
            if (failed.load(std::memory_order_relaxed)) break;
            if (sleepy) std::this_thread::yield();
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
        }
        if (failed.load(std::memory_order_relaxed)) break;
        const size_t off = static_cast<size_t>(c) * S, len = std::min(S, bytes - off);
        e = hipMemcpyAsync(ring + static_cast<size_t>(s) * S, dev + off, len, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->d2h_events[s], ctx->stream);
        if (e == hipSuccess) issued.store(c + 1, std::memory_order_release);
    }
    if (e != hipSuccess) failed.store(1);
    for (std::thread &t : pool) t.join();
    if (e == hipSuccess) e = safe_stream_sync(ctx->stream);
    if (e != hipSuccess || failed.load()) {
        safe_set_error("safe_memcpy_d2h: %s", e != hipSuccess ? hipGetErrorString(e) : "a copy thread failed");
        return SAFE_E_HIP;
    }
    return SAFE_OK;
}

int safe_memcpy_d2h(safe_ctx *ctx, void *host, const void *dev, size_t bytes) {
    SAFE_REQUIRE(ctx != nullptr && (bytes == 0 || (dev && host)), "safe_memcpy_d2h: NULL argument");
    if (bytes == 0) return SAFE_OK;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    // SAFE_HIP_D2H_THREADS=0: the plain copy; default: 4 copy threads for 64 MB and more (2 when blocking waits are selected: a rank
    // that has few cores to itself).  tools/probe/d2h_threads.py: 139 MB 7.7 -> 3.6 ms, 1.6 GB 93 -> 36 ms (4 threads; more do not
    // help); a 17 MB copy into recycled memory runs at 55 GB/s as it is and stays on the plain path.
    int workers = safe_blocking_sync_selected() ? 2 : 4;
    if (const char *env = getenv("SAFE_HIP_D2H_THREADS")) workers = atoi(env);
    if (workers > 0 && bytes >= (size_t(64) << 20)) {
        bool fallback = false;
        const int rc = d2h_pipelined(ctx, static_cast<char *>(host), static_cast<const char *>(dev), bytes, std::min(workers, 16), &fallback);
        if (rc != SAFE_OK || !fallback) return rc;          // (fallback: the pinned ring could not be made -- the plain copy below)
    }
    SAFE_HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

int safe_memcpy_d2h_resident(safe_ctx *ctx, void *host, const void *dev, size_t bytes) {
    SAFE_REQUIRE(ctx != nullptr && (bytes == 0 || (dev && host)), "safe_memcpy_d2h_resident: NULL argument");
    if (bytes == 0) return SAFE_OK;
    SAFE_HIP_CHECK(hipSetDevice(ctx->device));
    SAFE_HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    return SAFE_OK;
}

int safe_timer_start(safe_ctx *ctx) {
    SAFE_REQUIRE(ctx != nullptr, "safe_timer_start: ctx is NULL");
    SAFE_HIP_CHECK(hipEventRecord(ctx->t0, ctx->stream));
    return SAFE_OK;
}

int safe_timer_stop_ms(safe_ctx *ctx, double *elapsed_ms) {
    SAFE_REQUIRE(ctx != nullptr && elapsed_ms != nullptr, "safe_timer_stop_ms: NULL argument");
    SAFE_HIP_CHECK(hipEventRecord(ctx->t1, ctx->stream));
    SAFE_HIP_CHECK(hipEventSynchronize(ctx->t1));
    float ms = 0.f;
    SAFE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->t0, ctx->t1));
    *elapsed_ms = ms;
    return SAFE_OK;
}

int safe_alloc_count(int64_t *calls) {
    SAFE_REQUIRE(calls, "safe_alloc_count: NULL argument");
    *calls = g_alloc_calls.load(std::memory_order_relaxed);
    return SAFE_OK;
}

int safe_last_mfma_slices(safe_ctx *ctx, int *slices) {
    SAFE_REQUIRE(ctx && slices, "safe_last_mfma_slices: NULL argument");
    *slices = ctx->last_slices;
    return SAFE_OK;
}

int safe_last_mfma_filter(safe_ctx *ctx, int *core_slices, int64_t *undecided) {
    SAFE_REQUIRE(ctx && core_slices && undecided, "safe_last_mfma_filter: NULL argument");
    *core_slices = ctx->last_core_slices;
    *undecided = ctx->last_undecided;
    return SAFE_OK;
}

int safe_last_kernel_stats(safe_ctx *ctx, char *name, size_t name_len, double *avg_ms, int64_t *launches) {
    SAFE_REQUIRE(ctx != nullptr, "safe_last_kernel_stats: ctx is NULL");
    if (name && name_len) snprintf(name, name_len, "%s", ctx->last_kernel.name.c_str());
    if (avg_ms) *avg_ms = ctx->last_kernel.launches ? ctx->last_kernel.total_ms / ctx->last_kernel.launches : 0.0;
    if (launches) *launches = ctx->last_kernel.launches;
    return SAFE_OK;
}

int safe_last_kernel_busy_ms(safe_ctx *ctx, double *busy_ms) {
    SAFE_REQUIRE(ctx && busy_ms, "safe_last_kernel_busy_ms: NULL argument");
    *busy_ms = ctx->last_kernel.busy_ms;
    return SAFE_OK;
}

}  // extern "C"
