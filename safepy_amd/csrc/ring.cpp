// Node-shared permutation stream (see ring.h): shared-memory ring + futex doorbells.  Host code only.
#include "ring.h"

#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <climits>

#include "common.h"
#include "draws.h"

namespace {

constexpr uint32_t kMagic = 0x53414645u;      // "SAFE"
constexpr uint32_t kVersion = 1;
constexpr int kMaxPeers = 64;
constexpr int kMaxSlots = 16;
constexpr uint64_t kLeft = 0xFFFFFFFFull;     // "consumed" value of a consumer that has left the call

struct alignas(64) PeerState {
    std::atomic<uint64_t> consumed;           // (generation << 32) | chunks of that generation this consumer is done with
    std::atomic<uint32_t> attached;
};

struct RingHeader {
    std::atomic<uint32_t> magic;              // written last by the producer
    uint32_t version;
    int32_t local_world;
    int32_t reserved;
    int64_t capacity;                         // bytes of slot space behind the header
    alignas(64) std::atomic<uint64_t> ready;  // (generation << 32) | chunks published in that generation
    alignas(64) std::atomic<uint32_t> bell;   // futex word: bumped by the producer after every change of `ready`
    alignas(64) std::atomic<uint32_t> progress;   // futex word: bumped by consumers after every change of their state
    alignas(64) RingCall call;                // the current generation's call (written before `ready` announces it)
    int64_t slot_bytes;
    int32_t n_slots;
    alignas(64) PeerState peer[kMaxPeers];
};

constexpr size_t kHeaderBytes = (sizeof(RingHeader) + 4095) & ~size_t(4095);

int futex_wait(std::atomic<uint32_t> *word, uint32_t seen, double seconds) {
    timespec ts;
    ts.tv_sec = static_cast<time_t>(seconds);
    ts.tv_nsec = static_cast<long>((seconds - static_cast<double>(ts.tv_sec)) * 1e9);
    return static_cast<int>(syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAIT, seen, &ts, nullptr, 0));
}

void futex_wake_all(std::atomic<uint32_t> *word) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE, INT_MAX, nullptr, nullptr, 0);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double ring_timeout_s() {
    static const double t = [] {
        const char *e = getenv("SAFE_HIP_RING_TIMEOUT_S");
        const double v = e ? atof(e) : 120.0;
        return v > 0.0 ? v : 120.0;
    }();
    return t;
}

}  // namespace

struct PermRing {
    char name[96] = {0};
    int fd = -1;
    uint8_t *base = nullptr;
    size_t bytes = 0;
    RingHeader *hdr = nullptr;
    int local_rank = 0, local_world = 1;
    bool producer = false;
    bool unlinked = false;
    uint64_t generation = 0;          // calls made through this ring (every rank of the node counts the same calls)
    bool in_call = false;
    int n_slots = 0;
    int64_t slot_bytes = 0;
    int64_t fetched = 0;              // consumer: chunks of the current call already copied out
};

static uint8_t *slot_ptr(PermRing *r, int64_t ci) { return r->base + kHeaderBytes + static_cast<size_t>(ci % r->n_slots) * r->slot_bytes; }

static void producer_unlink_if_all_attached(PermRing *r) {
    if (!r->producer || r->unlinked) return;
    for (int p = 1; p < r->local_world; ++p)
        if (r->hdr->peer[p].attached.load(std::memory_order_acquire) == 0) return;
    shm_unlink(r->name);             // every rank holds a mapping: the name is no longer needed (nothing is left behind on a crash)
    r->unlinked = true;
}

int ring_open(const char *name, int local_rank, int local_world, int64_t capacity_bytes, double timeout_s, PermRing **out) {
    SAFE_REQUIRE(name && out, "ring_open: NULL argument");
    SAFE_REQUIRE(local_world >= 1 && local_world <= kMaxPeers && local_rank >= 0 && local_rank < local_world,
                 "ring_open: local rank %d of %d out of range (at most %d ranks per node)", local_rank, local_world, kMaxPeers);
    SAFE_REQUIRE(capacity_bytes >= 4096 && strlen(name) < 80 && name[0] != '\0' && !strchr(name, '/'),
                 "ring_open: bad name or capacity");
    *out = nullptr;
    PermRing *r = new PermRing();
    snprintf(r->name, sizeof(r->name), "/safe_hip.%s", name);
    r->local_rank = local_rank;
    r->local_world = local_world;
    r->producer = local_rank == 0;
    const size_t cap = (static_cast<size_t>(capacity_bytes) + 4095) & ~size_t(4095);
    r->bytes = kHeaderBytes + cap;
    if (r->producer) {
        r->fd = shm_open(r->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (r->fd < 0 && errno == EEXIST) {                       // a leftover of a crashed job under the same name
            shm_unlink(r->name);
            r->fd = shm_open(r->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        }
        if (r->fd < 0 || ftruncate(r->fd, static_cast<off_t>(r->bytes)) != 0) {
            safe_set_error("ring_open: cannot create %s (%zu bytes): %s", r->name, r->bytes, strerror(errno));
            if (r->fd >= 0) {
                close(r->fd);
                shm_unlink(r->name);
            }
            delete r;
            return SAFE_E_NOMEM;
        }
    } else {
        const double t_end = now_s() + timeout_s;
        for (;;) {
            r->fd = shm_open(r->name, O_RDWR, 0600);
            if (r->fd >= 0) {
                struct stat st;
                if (fstat(r->fd, &st) == 0 && static_cast<size_t>(st.st_size) >= r->bytes) break;    // created AND sized
                close(r->fd);
                r->fd = -1;
            }
            if (now_s() > t_end) {
                safe_set_error("ring_open: %s did not appear within %.0f s (is local rank 0 of this node running the same call?)",
                               r->name, timeout_s);
                delete r;
                return SAFE_E_VALUE;
            }
            usleep(200);
        }
    }
    void *m = mmap(nullptr, r->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, r->fd, 0);
    if (m == MAP_FAILED) {
        safe_set_error("ring_open: mmap of %s failed: %s", r->name, strerror(errno));
        close(r->fd);
        if (r->producer) shm_unlink(r->name);
        delete r;
        return SAFE_E_NOMEM;
    }
    r->base = static_cast<uint8_t *>(m);
    r->hdr = reinterpret_cast<RingHeader *>(m);
    if (r->producer) {                                            // (fresh shm pages are zero: every counter starts at 0)
        r->hdr->version = kVersion;
        r->hdr->local_world = local_world;
        r->hdr->capacity = static_cast<int64_t>(cap);
        r->hdr->magic.store(kMagic, std::memory_order_release);
    } else {
        const double t_end = now_s() + timeout_s;
        while (r->hdr->magic.load(std::memory_order_acquire) != kMagic) {
            if (now_s() > t_end) {
                safe_set_error("ring_open: %s was never initialised", r->name);
                ring_close(r);
                return SAFE_E_VALUE;
            }
            usleep(100);
        }
        if (r->hdr->version != kVersion || r->hdr->local_world != local_world || r->hdr->capacity != static_cast<int64_t>(cap)) {
            safe_set_error("ring_open: %s was created for %d ranks / %lld bytes, this rank expects %d / %zu", r->name,
                           r->hdr->local_world, (long long)r->hdr->capacity, local_world, cap);
            ring_close(r);
            return SAFE_E_VALUE;
        }
        r->hdr->peer[local_rank].attached.store(1, std::memory_order_release);
        r->hdr->progress.fetch_add(1, std::memory_order_release);
        futex_wake_all(&r->hdr->progress);
    }
    *out = r;
    return SAFE_OK;
}

void ring_close(PermRing *r) {
    if (!r) return;
    if (r->in_call) ring_end_call(r);
    if (r->producer && !r->unlinked) shm_unlink(r->name);
    if (r->base) munmap(r->base, r->bytes);
    if (r->fd >= 0) close(r->fd);
    delete r;
}

bool ring_is_producer(const PermRing *r) { return r->producer; }
int64_t ring_capacity(const PermRing *r) { return r->hdr->capacity; }

int ring_slots_for(const PermRing *r, int64_t slot_bytes) {
    if (slot_bytes <= 0) return 0;
    const int64_t padded = (slot_bytes + 4095) & ~int64_t(4095);
    const int64_t s = r->hdr->capacity / padded;
    return s < 2 ? 0 : static_cast<int>(std::min<int64_t>(s, kMaxSlots));
}

int ring_begin_call(PermRing *r, const RingCall &call, int64_t slot_bytes) {
    SAFE_REQUIRE(r && r->producer && !r->in_call, "ring_begin_call: not the producer, or a call is still open");
    const int slots = ring_slots_for(r, slot_bytes);
    SAFE_REQUIRE(slots >= 2, "ring_begin_call: a chunk of %lld bytes does not fit the ring twice", (long long)slot_bytes);
    RingHeader *h = r->hdr;
    const uint64_t g = r->generation + 1;
    // every consumer must have joined AND left generation g - 1: its slots and the `ready` word are about to be reused (a
    // consumer that were still to join g - 1 would find generation g announced and could not tell what it missed)
    const double t_end = now_s() + ring_timeout_s();
    for (;;) {
        const uint32_t seen = h->progress.load(std::memory_order_acquire);
        bool all = true;
        for (int p = 1; p < r->local_world && all && g > 1; ++p)
            all = h->peer[p].consumed.load(std::memory_order_acquire) == (((g - 1) << 32) | kLeft);
        if (all) break;
        if (now_s() > t_end) {
            safe_set_error("shared permutation stream: a rank of this node has not finished the previous call after %.0f s "
                           "(the ranks of a node must make the same calls in the same order)", ring_timeout_s());
            return SAFE_E_VALUE;
        }
        futex_wait(&h->progress, seen, 0.5);
    }
    h->call = call;
    h->slot_bytes = (slot_bytes + 4095) & ~int64_t(4095);
    h->n_slots = slots;
    r->generation = g;
    r->n_slots = slots;
    r->slot_bytes = h->slot_bytes;
    r->in_call = true;
    h->ready.store(g << 32, std::memory_order_release);          // announces generation g with 0 chunks
    h->bell.fetch_add(1, std::memory_order_release);
    futex_wake_all(&h->bell);
    producer_unlink_if_all_attached(r);
    return SAFE_OK;
}

int ring_publish(PermRing *r, int64_t ci, const void *src, size_t bytes) {
    SAFE_REQUIRE(r && r->producer && r->in_call, "ring_publish: no open call");
    SAFE_REQUIRE(static_cast<int64_t>(bytes) <= r->slot_bytes, "ring_publish: chunk larger than a slot");
    RingHeader *h = r->hdr;
    const uint64_t g = r->generation;
    if (ci >= r->n_slots) {                                      // the slot still holds chunk ci - n_slots: everyone must be past it
        const uint64_t need = static_cast<uint64_t>(ci - r->n_slots + 1);
        const double t_end = now_s() + ring_timeout_s();
        for (;;) {
            const uint32_t seen = h->progress.load(std::memory_order_acquire);
            bool all = true;
            for (int p = 1; p < r->local_world && all; ++p) {
                const uint64_t c = h->peer[p].consumed.load(std::memory_order_acquire);
                all = (c >> 32) == g && (c & 0xFFFFFFFFull) >= need;
            }
            if (all) break;
            if (now_s() > t_end) {
                safe_set_error("shared permutation stream: a rank of this node has not consumed chunk %lld after %.0f s",
                               (long long)(ci - r->n_slots), ring_timeout_s());
                return SAFE_E_VALUE;
            }
            futex_wait(&h->progress, seen, 0.5);
        }
    }
    draws_nt_copy(slot_ptr(r, ci), src, bytes);
    std::atomic_thread_fence(std::memory_order_release);         // (non-temporal stores: fence before the flag)
    h->ready.store((g << 32) | static_cast<uint64_t>(ci + 1), std::memory_order_release);
    h->bell.fetch_add(1, std::memory_order_release);
    futex_wake_all(&h->bell);
    producer_unlink_if_all_attached(r);
    return SAFE_OK;
}

int ring_join_call(PermRing *r, const RingCall &call, int64_t slot_bytes) {
    SAFE_REQUIRE(r && !r->producer && !r->in_call, "ring_join_call: not a consumer, or a call is still open");
    RingHeader *h = r->hdr;
    const uint64_t g = r->generation + 1;
    const double t_end = now_s() + ring_timeout_s();
    for (;;) {
        const uint32_t seen = h->bell.load(std::memory_order_acquire);
        const uint64_t rd = h->ready.load(std::memory_order_acquire);
        if ((rd >> 32) == g) break;
        if ((rd >> 32) > g) {
            safe_set_error("shared permutation stream: this rank joins call %llu but the node's producer is at call %llu "
                           "(the ranks of a node must make the same calls in the same order)", (unsigned long long)g,
                           (unsigned long long)(rd >> 32));
            return SAFE_E_VALUE;
        }
        if (now_s() > t_end) {
            safe_set_error("shared permutation stream: local rank 0 did not start call %llu within %.0f s", (unsigned long long)g,
                           ring_timeout_s());
            return SAFE_E_VALUE;
        }
        futex_wait(&h->bell, seen, 0.5);
    }
    r->generation = g;
    const RingCall &c = h->call;
    const int64_t padded = (slot_bytes + 4095) & ~int64_t(4095);
    if (c.n != call.n || c.k != call.k || c.count != call.count || c.chunk_rows != call.chunk_rows ||
        c.movable_hash != call.movable_hash || h->slot_bytes != padded) {
        // leave at once so that the producer is not held up by a rank that cannot take part
        h->peer[r->local_rank].consumed.store((g << 32) | kLeft, std::memory_order_release);
        h->progress.fetch_add(1, std::memory_order_release);
        futex_wake_all(&h->progress);
        safe_set_error("shared permutation stream: this rank's call (n=%lld, movable rows=%lld, permutations=%lld) differs from "
                       "local rank 0's (n=%lld, movable rows=%lld, permutations=%lld) or marks other rows",
                       (long long)call.n, (long long)call.k, (long long)call.count, (long long)c.n, (long long)c.k, (long long)c.count);
        return SAFE_E_VALUE;
    }
    r->n_slots = h->n_slots;
    r->slot_bytes = h->slot_bytes;
    r->fetched = 0;
    r->in_call = true;
    h->peer[r->local_rank].consumed.store(g << 32, std::memory_order_release);
    h->progress.fetch_add(1, std::memory_order_release);
    futex_wake_all(&h->progress);
    return SAFE_OK;
}

int ring_fetch(PermRing *r, int64_t ci, void *dst, size_t bytes, double *waited_ms) {
    SAFE_REQUIRE(r && !r->producer && r->in_call, "ring_fetch: no open call");
    SAFE_REQUIRE(ci == r->fetched, "ring_fetch: chunks must be fetched in order (asked %lld, next is %lld)", (long long)ci,
                 (long long)r->fetched);
    SAFE_REQUIRE(static_cast<int64_t>(bytes) <= r->slot_bytes, "ring_fetch: chunk larger than a slot");
    RingHeader *h = r->hdr;
    const uint64_t g = r->generation;
    const double t0 = now_s(), t_end = t0 + ring_timeout_s();
    for (;;) {
        const uint32_t seen = h->bell.load(std::memory_order_acquire);
        const uint64_t rd = h->ready.load(std::memory_order_acquire);
        if ((rd >> 32) != g) {
            safe_set_error("shared permutation stream: the producer moved on to call %llu while this rank was reading call %llu",
                           (unsigned long long)(rd >> 32), (unsigned long long)g);
            return SAFE_E_VALUE;
        }
        if ((rd & 0xFFFFFFFFull) > static_cast<uint64_t>(ci)) break;
        if (now_s() > t_end) {
            safe_set_error("shared permutation stream: chunk %lld was not published within %.0f s", (long long)ci, ring_timeout_s());
            return SAFE_E_VALUE;
        }
        futex_wait(&h->bell, seen, 0.5);                         // blocks: no CPU while the producer draws
    }
    if (waited_ms) *waited_ms += 1e3 * (now_s() - t0);
    memcpy(dst, slot_ptr(r, ci), bytes);
    r->fetched = ci + 1;
    h->peer[r->local_rank].consumed.store((g << 32) | static_cast<uint64_t>(ci + 1), std::memory_order_release);
    h->progress.fetch_add(1, std::memory_order_release);
    futex_wake_all(&h->progress);
    return SAFE_OK;
}

void ring_end_call(PermRing *r) {
    if (!r || !r->in_call) return;
    r->in_call = false;
    if (r->producer) return;                                     // (the producer's next ring_begin_call waits for the consumers)
    RingHeader *h = r->hdr;
    h->peer[r->local_rank].consumed.store((r->generation << 32) | kLeft, std::memory_order_release);
    h->progress.fetch_add(1, std::memory_order_release);
    futex_wake_all(&h->progress);
}

extern "C" {

int safe_ring_open(const char *name, int local_rank, int local_world, int64_t capacity_bytes, safe_ring **out) {
    return ring_open(name, local_rank, local_world, capacity_bytes, ring_timeout_s(), out);
}

int safe_ring_close(safe_ring *ring) {
    ring_close(ring);
    return SAFE_OK;
}

int safe_ring_begin(safe_ring *ring, int64_t n, int64_t k, int64_t count, uint64_t movable_hash, int64_t slot_bytes) {
    SAFE_REQUIRE(ring, "safe_ring_begin: NULL ring");
    RingCall call;
    call.n = n;
    call.k = k;
    call.count = count;
    call.chunk_rows = 128;
    call.movable_hash = movable_hash;
    return ring_is_producer(ring) ? ring_begin_call(ring, call, slot_bytes) : ring_join_call(ring, call, slot_bytes);
}

int safe_ring_publish(safe_ring *ring, int64_t chunk, const void *src_host, size_t bytes) {
    SAFE_REQUIRE(ring && src_host, "safe_ring_publish: NULL argument");
    return ring_publish(ring, chunk, src_host, bytes);
}

int safe_ring_fetch(safe_ring *ring, int64_t chunk, void *dst_host, size_t bytes) {
    SAFE_REQUIRE(ring && dst_host, "safe_ring_fetch: NULL argument");
    return ring_fetch(ring, chunk, dst_host, bytes, nullptr);
}

int safe_ring_end(safe_ring *ring) {
    SAFE_REQUIRE(ring, "safe_ring_end: NULL ring");
    ring_end_call(ring);
    return SAFE_OK;
}

}  // extern "C"
