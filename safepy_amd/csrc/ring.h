// Node-shared permutation stream: a POSIX shared-memory ring through which ONE process of a node hands the row maps of
// every pipeline chunk to the other ranks of that node (ring.cpp; plain host C++, no HIP).
//
// Why: the legacy NumPy stream (safepy/safe_extras.py:46-58) is sequential -- how many MT19937 words a shuffle consumes
// depends on its rejections -- so a rank cannot draw "its part"; in round 2 EVERY rank of a node drew the whole stream
// (one draw thread + swap workers each: 8 ranks = 40 busy host threads, and the serial 1.7-2.2 ms per 1000 permutations
// on every rank).  Here the node's rank 0 draws and replays once and publishes each chunk's row maps; the other ranks
// block on a futex until a chunk is there, copy it to their own pinned staging buffer and upload it.
#pragma once
#include <cstddef>
#include <cstdint>

struct PermRing;

// what a call must agree on across the ranks of a node (checked by every consumer)
struct RingCall {
    int64_t n = 0, k = 0, count = 0, chunk_rows = 0;
    uint64_t movable_hash = 0;
};

// local_rank 0 creates /dev/shm/<name> (capacity_bytes of chunk slots + a header) and waits for nobody; the others attach,
// waiting up to timeout_s for the segment to appear.  Returns 0 or a negative SAFE_E_* code (message via safe_set_error).
int ring_open(const char *name, int local_rank, int local_world, int64_t capacity_bytes, double timeout_s, PermRing **out);
void ring_close(PermRing *r);
bool ring_is_producer(const PermRing *r);
int64_t ring_capacity(const PermRing *r);
// slots a call with chunks of `slot_bytes` gets (0: does not fit -- every rank then draws for itself, decided identically everywhere)
int ring_slots_for(const PermRing *r, int64_t slot_bytes);

// producer: announce the next call (waits until every consumer has left the previous one)
int ring_begin_call(PermRing *r, const RingCall &call, int64_t slot_bytes);
// producer: chunk `ci` of the current call = `bytes` at `src`; waits for the slot to be free (all consumers past ci - slots)
int ring_publish(PermRing *r, int64_t ci, const void *src, size_t bytes);
// consumer: join the next call (waits for the producer's announcement, verifies `call`)
int ring_join_call(PermRing *r, const RingCall &call, int64_t slot_bytes);
// consumer: chunk `ci` copied to `dst`; *waited_ms (optional) = time blocked waiting for the producer
int ring_fetch(PermRing *r, int64_t ci, void *dst, size_t bytes, double *waited_ms);
// either side: done with the current call (a consumer that leaves early never holds the producer up)
void ring_end_call(PermRing *r);
