// Host draw stream of the legacy NumPy permutation (draws.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

struct DrawStream;
DrawStream *draw_stream_new(uint32_t seed);
void draw_stream_free(DrawStream *s);
// accepted swap targets of one shuffle of k items in step order: steps[s] = j for i = k-1-s
void draw_stream_targets(DrawStream *s, int64_t k, uint32_t *steps);
// which form of the masked-rejection loop this host runs: "avx512" (batches of 16-64 words) or "scalar"
const char *draws_path_name();
// streaming (non-temporal) copy for hand-offs to other cores
void draws_nt_copy(void *dst, const void *src, size_t bytes);
// 32-bit swap targets (all < 65536) packed to 16 bits
void draws_pack_u16(uint16_t *dst, const uint32_t *src, size_t count);
