// K5, matrix-core form: the permutation test of run_permutations (safepy/safe_extras.py:36-70)
// for quantitative attributes as an exact fixed-point block-sparse GEMM on the i8 MFMA pipe.
//
// The reference evaluates S_p = A . B0[perm_p] with dgemm.  Here
//   * every attribute column is scaled by a power of two and rounded to a 47-bit integer,
//     which is split into six balanced base-256 digits (i8 "slices", Ozaki-style splitting);
//     A is 0/1, so each slice product accumulates EXACTLY in the i32 MFMA accumulators and
//     the six partial sums recombine into the exact 64-bit integer sum.  Scores of different
//     permutations are compared as integers: deterministic, independent of summation order,
//     and at least as accurate as the reference's f64 dot products (quantisation 2^-46 of
//     the column maximum, far below dgemm's own rounding of a ~k-term sum);
//   * nodes are renumbered along a space-filling curve of the layout (the neighborhoods are
//     balls of the layout, safe.py:389-417), which makes A block-sparse: a 256-row group only
//     touches the 32-column blocks near it (fill 30-50 % at config-5 density), and only those
//     blocks are multiplied;
//   * the permuted operand B0[cur_p] is never materialised: the rows a block needs are
//     gathered from the resident slice matrix straight into LDS (transposed on the way with
//     v_perm so that every lane finds its 16 k-values contiguous), one 128-row super-step
//     ahead of the MFMAs;
//   * <= / >= counters, observed scores and accumulators live in registers for the whole
//     (row group, 32-column tile, permutation span) task; tasks are queued per XCD by column
//     tile so the slice rows a tile needs stay in that XCD's L2.
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>

#include "common.h"
#include <atomic>
#include <thread>

namespace {

constexpr int MF_R = 256;             // rows per group = 8 waves x 32
constexpr int MF_NS = 6;              // i8 slices per value
constexpr int MF_CN = 6;              // counts form: 32-column tiles per task (one i8 plane each)
constexpr int MF_SS = 1024 + 16;      // LDS bytes per (k-step, slice): [2 halves][32 lanes][16 B] + 16 B skew
constexpr int MF_KS = MF_NS * MF_SS;  // LDS bytes per k-step (32 attribute rows)
// The general kernel's LDS form of a gathered k-step.  mf_trg(slices): rows as they come ([32 rows][RS bytes]; RS = 32 bytes per
// slice, padded to an ODD multiple of 32 so that the eight rows a transposing read touches fall into eight different groups of
// eight banks) and the B operand read with ds_read_b64_tr_b8; else transposed in registers by the gather threads (v_perm) into
// [slice][2 halves][32 lanes][16 B] and read with ds_read_b128.  Same box, configs[4] rank share (tools/probe/trg_ab.sh): six
// slices 1.384 -> 1.347 s with the transposing reads, but three slices 0.929 -> 0.970 s and the filtered z-scores (four slices)
// 2.513 -> 2.657 s -- eight waves read every tile, and two 8-byte reads per slice instead of one 16-byte read cost them more LDS
// issue slots than the gather threads save on few slices -- so: six slices and more (-DSAFE_MFMA_TRG=0: never; =2: always).
#ifndef SAFE_MFMA_TRG
#define SAFE_MFMA_TRG 1
#endif
constexpr bool mf_trg(int ns) { return SAFE_MFMA_TRG == 2 || (SAFE_MFMA_TRG == 1 && ns >= 6); }
constexpr int mf_rs(int ns) { return 32 * (ns | 1); }                                  // row bytes in LDS (transposing-read form)
constexpr int mf_ks(int ns) { return mf_trg(ns) ? 32 * mf_rs(ns) : ns * MF_SS; }       // LDS bytes per k-step
constexpr int MF_BUF = 4 * MF_KS;     // one buffer = one super-step = 4 k-steps
constexpr int MF_MAXBLK = 4096;       // column blocks per row group the kernel can index from LDS
constexpr int MF_SHIFT_BITS = 45;     // |q| < 2^46 after scaling: six balanced base-256 digits hold +-(2^47 - ...)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 2048 - (exponent of the least significant set bit of x), x finite and non-zero: with the exponent of the column
// maximum it tells how many bits an EXACT fixed-point image of the column needs (k_mfma_colfinish)
__device__ __forceinline__ unsigned int neg_lowbit_of(double x) {
    const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(x));
    const int e = static_cast<int>((b >> 52) & 0x7FFull);
    unsigned long long m = b & ((1ull << 52) - 1ull);
    int low;
    if (e == 0) low = -1074 + __ffsll(static_cast<unsigned long long>(m)) - 1;
    else {
        m |= 1ull << 52;
        low = e - 1023 - 52 + __ffsll(static_cast<unsigned long long>(m)) - 1;
    }
    return static_cast<unsigned int>(2048 - low);
}

// ---------------------------------------------------------------------------------------
// column statistics: max |b|, sum b^2 and count over the non-NaN non-zero entries
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_mfma_colstats(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                       int64_t col0, int64_t mloc, int rows_per_block,
                                                       unsigned long long *__restrict__ maxbits, double *__restrict__ sumsq,
                                                       unsigned int *__restrict__ cnt, unsigned int *__restrict__ neg_lowbit) {
    // 32 columns x 8 row lanes; the lane index runs along whichever axis is contiguous
    __shared__ double s_max[8][33], s_sq[8][33];
    __shared__ unsigned int s_cnt[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c0 = static_cast<int64_t>(blockIdx.x) * 32, r0 = static_cast<int64_t>(blockIdx.y) * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const bool col_major = rs == 1;            // Fortran order: consecutive rows are adjacent
    double mx = 0.0, sq = 0.0;
    unsigned int ct = 0, nl = 0;
    __shared__ unsigned int s_nl[8][33];
    if (!col_major) {
        const int64_t j = c0 + tx;
        if (j < mloc)
            for (int64_t r = r0 + ty; r < r1; r += 8) {
                const double x = static_cast<double>(reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs]);
                if (x == x && x != 0.0) {
                    mx = fmax(mx, fabs(x));
                    sq += x * x;
                    ++ct;
                    if (fabs(x) < __longlong_as_double(0x7FF0000000000000ll)) nl = max(nl, neg_lowbit_of(x));
                }
            }
        s_max[ty][tx] = mx;
        s_sq[ty][tx] = sq;
        s_cnt[ty][tx] = ct;
        s_nl[ty][tx] = nl;
    } else {
        // lanes along rows; column = c0 + ty + 8*i
        for (int i = 0; i < 4; ++i) {
            const int64_t j = c0 + ty + 8 * i;
            mx = 0.0;
            sq = 0.0;
            ct = 0;
            nl = 0;
            if (j < mloc)
                for (int64_t r = r0 + tx; r < r1; r += 32) {
                    const double x = static_cast<double>(reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs]);
                    if (x == x && x != 0.0) {
                        mx = fmax(mx, fabs(x));
                        sq += x * x;
                        ++ct;
                        if (fabs(x) < __longlong_as_double(0x7FF0000000000000ll)) nl = max(nl, neg_lowbit_of(x));
                    }
                }
            for (int o = 16; o > 0; o >>= 1) {
                mx = fmax(mx, __shfl_xor(mx, o, 32));
                sq += __shfl_xor(sq, o, 32);
                ct += __shfl_xor(ct, o, 32);
                nl = max(nl, static_cast<unsigned int>(__shfl_xor(static_cast<int>(nl), o, 32)));
            }
            if (tx == 0 && j < mloc && ct) {
                atomicMax(&maxbits[j], static_cast<unsigned long long>(__double_as_longlong(mx)));
                atomicAdd(&sumsq[j], sq);
                atomicAdd(&cnt[j], ct);
                atomicMax(&neg_lowbit[j], nl);
            }
        }
        return;
    }
    __syncthreads();
    if (ty == 0) {
        for (int k = 1; k < 8; ++k) {
            mx = fmax(mx, s_max[k][tx]);
            sq += s_sq[k][tx];
            ct += s_cnt[k][tx];
            nl = max(nl, s_nl[k][tx]);
        }
        const int64_t j = c0 + tx;
        if (j < mloc && ct) {
            atomicMax(&maxbits[j], static_cast<unsigned long long>(__double_as_longlong(mx)));
            atomicAdd(&sumsq[j], sq);
            atomicAdd(&cnt[j], ct);
            atomicMax(&neg_lowbit[j], nl);
        }
    }
}

// per column: the power-of-two scale of its fixed-point image, and how many bits the image needs.
//   exact   the column's values all sit on one binary grid that spans <= 46 bits from its lowest set bit to the top bit of
//           its maximum (integers, f32-valued data of moderate range, ...): scale = 2^-lowbit, q = b * scale EXACTLY;
//   rounded otherwise (f64 data in general): the maximum is mapped just below 2^46 and q = rint(b * scale), grid step
//           2^-46 of the column maximum (k_mfma_slice counts what that does to small values, k_mfma_colcheck judges).
// need[0] = max over columns of the bits needed (47 = rounded): the call runs with 2 / 4 / 6 i8 slices (<= 14 / 30 / 46 bits).
__global__ void k_mfma_colfinish(const unsigned long long *__restrict__ maxbits, const unsigned int *__restrict__ neg_lowbit,
                                 int64_t mloc, int *__restrict__ shift, double *__restrict__ scale, int *__restrict__ bad,
                                 int *__restrict__ need) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= mloc) return;
    const double mx = __longlong_as_double(static_cast<long long>(maxbits[j]));
    int sh = 0, bits = 1;
    if (mx > 0.0 && mx < __longlong_as_double(0x7FF0000000000000ll)) {
        const int top = ilogb(mx), low = 2048 - static_cast<int>(neg_lowbit[j]);
        const int span = top - low + 1;
        if (span <= MF_SHIFT_BITS + 1) {
            sh = -low;                                   // exact: the lowest set bit becomes the unit
            bits = span;
        } else {
            sh = MF_SHIFT_BITS - top;
            bits = MF_SHIFT_BITS + 2;
        }
    } else if (mx != 0.0) {
        atomicOr(bad, 1);                               // infinities are not representable at all
    }
    shift[j] = sh;
    scale[j] = ldexp(1.0, -sh);
    atomicMax(need, bits);
}

__host__ __device__ __forceinline__ int mfma_slices_for(int need_bits) { return need_bits <= 14 ? 2 : need_bits <= 30 ? 4 : MF_NS; }

// ---------------------------------------------------------------------------------------
// slices: bs[column tile][row][slice][32 columns] i8 (tile-major: the rows one XCD gathers for its tile are packed, not
// interleaved with the neighbouring tiles' bytes in the same cache lines), row n = zeros.  q = rint(b * 2^shift)
// as a 47-bit integer, digits d_t in [-128, 127] with q = sum d_t 256^t.
// Also counts, per column, the values far below the column maximum (< 2^-20 max) that had to be rounded:
// k_mfma_colcheck declines the path when they are more than a thousandth of a column (the grid, set by the
// maximum, would be too coarse for sums made of such values).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_mfma_slice(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                    int64_t col0, int64_t mloc, int64_t n_ct, const int *__restrict__ shift,
                                                    const unsigned long long *__restrict__ maxbits, const int *__restrict__ need,
                                                    unsigned char *__restrict__ bs, unsigned int *__restrict__ n_small,
                                                    unsigned int *__restrict__ n_small_rounded, int64_t split_off, long long *__restrict__ q64) {
    __shared__ double tile[32][33];
    __shared__ unsigned int s_small[32], s_rounded[32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t ct = blockIdx.x, r0 = static_cast<int64_t>(blockIdx.y) * 32;
    const int64_t c0 = ct * 32;
    const bool col_major = rs == 1;
    if (threadIdx.x < 32) s_small[threadIdx.x] = s_rounded[threadIdx.x] = 0;
    for (int i = 0; i < 4; ++i) {
        const int a = ty + 8 * i;                                  // the slow index of the read
        const int64_t r = col_major ? r0 + tx : r0 + a, j = col_major ? c0 + a : c0 + tx;
        double x = 0.0;
        if (r < n && j < mloc) x = static_cast<double>(reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs]);
        if (col_major) tile[tx][a] = x;
        else tile[a][tx] = x;
    }
    __syncthreads();
    const int64_t j = c0 + tx;
    const int sh = j < mloc ? shift[j] : 0;
    // "small": more than 2^20 below the column maximum -- held to fewer than 26 significant bits on a rounding grid
    const double small_below = j < mloc ? ldexp(__longlong_as_double(static_cast<long long>(maxbits[j])), -20) : 0.0;
    const int ns = mfma_slices_for(*need);
    // split form (split_off != 0 and six slices: the filtered permutation test): digits 0-2 and digits 3-5 are two matrices of
    // three slices each, the high one split_off bytes behind the low one
    const bool split = split_off != 0 && ns == MF_NS;
    const int64_t row_bytes = (split ? MF_NS / 2 : ns) * 32, tile_bytes = (n + 1) * row_bytes;   // tile-major: bs[column tile][row][slice][32]
    const long long bias = ns == 2 ? 0x8080ll : ns == 4 ? 0x80808080ll : 0x808080808080ll;
    unsigned int k_small = 0, k_rounded = 0;
    for (int i = 0; i < 4; ++i) {
        const int rr = ty + 8 * i;
        const int64_t r = r0 + rr;
        if (r > n) continue;                                       // row n is the zero row
        double x = tile[rr][tx];
        if (!(x == x) || r == n) x = 0.0;                          // NaN -> 0 (safe_extras.py:10)
        if (!(fabs(x) < __longlong_as_double(0x7FF0000000000000ll))) x = 0.0;   // +-inf: the path is declined anyway
        const double scaled = ldexp(x, sh);
        const double q = rint(scaled);
        if (x != 0.0 && fabs(x) < small_below) {
            ++k_small;
            k_rounded += q != scaled;
        }
        const unsigned long long u = static_cast<unsigned long long>(static_cast<long long>(q) + bias);
        // (filtered form) the whole fixed-point value once more as one 64-bit word, [row][column]: what k_mfma_resolve sums
        if (q64 && split && j < mloc) q64[r * mloc + j] = static_cast<long long>(q);
        unsigned char *dst = bs + ct * tile_bytes + r * row_bytes + tx;
#pragma unroll
        for (int t = 0; t < MF_NS; ++t)
            if (t < ns)
                dst[(split && t >= MF_NS / 2 ? split_off + (t - MF_NS / 2) * 32 : t * 32)] =
                    static_cast<unsigned char>(((u >> (8 * t)) & 0xFFu) ^ 0x80u);
    }
    if (k_small) atomicAdd(&s_small[tx], k_small);
    if (k_rounded) atomicAdd(&s_rounded[tx], k_rounded);
    __syncthreads();
    if (threadIdx.x < 32 && c0 + threadIdx.x < mloc) {
        if (s_small[threadIdx.x]) atomicAdd(&n_small[c0 + threadIdx.x], s_small[threadIdx.x]);
        if (s_rounded[threadIdx.x]) atomicAdd(&n_small_rounded[c0 + threadIdx.x], s_rounded[threadIdx.x]);
    }
}

// z-scores: the scale of the SQUARES of a column (always the rounding regime: b*b is itself a rounded f64 product,
// as in the reference's np.power(B0, 2)): max^2 sits just below 2^46.  scale2[j] = 2^-shift2[j].
// inexact[j] = 1 unless both the values and their squares are held exactly (values on one binary grid spanning <= 22 bits).
__global__ void k_mfma_colfinish_sq(const unsigned long long *__restrict__ maxbits, const unsigned int *__restrict__ neg_lowbit,
                                    int64_t mloc, int *__restrict__ shift2, double *__restrict__ scale2,
                                    unsigned int *__restrict__ inexact, int *__restrict__ bad) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= mloc) return;
    const double mx = __longlong_as_double(static_cast<long long>(maxbits[j]));
    const double m2 = mx * mx;
    int sh = 0;
    unsigned int rounded = 0;
    if (mx != 0.0) {
        if (m2 < __longlong_as_double(0x7FF0000000000000ll) && m2 >= __longlong_as_double(0x0010000000000000ll)) {
            sh = MF_SHIFT_BITS - ilogb(m2);
            const int span = ilogb(mx) - (2048 - static_cast<int>(neg_lowbit[j])) + 1;
            rounded = 2 * span > MF_SHIFT_BITS;
        } else {
            atomicOr(bad, 1);                           // squares overflow or leave the normal range
        }
    }
    shift2[j] = sh;
    scale2[j] = ldexp(1.0, -sh);
    inexact[j] = rounded;
}

// z-score slices: bs[16-column tile][row][7 slices][32 bytes]; bytes 0-15 of slice t = digit t of q1 = fixed-point B0 of
// the tile's columns, bytes 16-31 = digit t of q2 = rint(B0*B0 * 2^shift2); slice 6 = the not-NaN flags (bytes 0-15).
// Row n = zeros (and not counted).  Counts the small rounded values of B0 like k_mfma_slice.
template <typename T>
__global__ __launch_bounds__(256) void k_mfma_slice_z(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                      int64_t col0, int64_t mloc, int64_t n_ct, const int *__restrict__ shift,
                                                      const int *__restrict__ shift2, const unsigned long long *__restrict__ maxbits,
                                                      unsigned char *__restrict__ bs, unsigned int *__restrict__ n_small,
                                                      unsigned int *__restrict__ n_small_rounded, unsigned int *__restrict__ n_zero,
                                                      int64_t zf_hi_off, int64_t zf_lo_off, longlong2 *__restrict__ z64) {
    __shared__ double tile[32][17];
    __shared__ unsigned int s_small[16], s_rounded[16], s_zero[16];
    const int64_t ct = blockIdx.x, r0 = static_cast<int64_t>(blockIdx.y) * 32, c0 = ct * 16;
    const bool col_major = rs == 1;
    if (threadIdx.x < 16) s_small[threadIdx.x] = s_rounded[threadIdx.x] = s_zero[threadIdx.x] = 0;
    for (int i = 0; i < 2; ++i) {
        // the fast thread index runs along whichever axis is contiguous in memory
        const int rr = col_major ? (threadIdx.x & 31) : (threadIdx.x >> 4) + 16 * i;
        const int cc = col_major ? (threadIdx.x >> 5) + 8 * i : (threadIdx.x & 15);
        const int64_t r = r0 + rr, j = c0 + cc;
        double x = __longlong_as_double(0x7FF8000000000000ll);
        if (r < n && j < mloc) x = static_cast<double>(reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs]);
        tile[rr][cc] = x;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = c0 + tx;
    const int sh = j < mloc ? shift[j] : 0, sh2 = j < mloc ? shift2[j] : 0;
    const double small_below = j < mloc ? ldexp(__longlong_as_double(static_cast<long long>(maxbits[j])), -20) : 0.0;
    const int64_t row_bytes = (MF_NS + 1) * 32, tile_bytes = (n + 1) * row_bytes; // tile-major: bs[16-column tile][row][slice][32]
    const long long bias = 0x808080808080ll;
    unsigned int k_small = 0, k_rounded = 0, k_zero = 0;
    for (int i = 0; i < 2; ++i) {
        const int rr = ty + 16 * i;
        const int64_t r = r0 + rr;
        if (r > n) continue;
        double x = tile[rr][tx];
        const bool present = (x == x) && r < n && j < mloc;            // safe_extras.py:19 (~isnan)
        k_zero += present && x == 0.0;
        if (!present) x = 0.0;                                          // NaN -> 0 (safe_extras.py:10)
        if (!(fabs(x) < __longlong_as_double(0x7FF0000000000000ll))) x = 0.0;   // +-inf: the path is declined anyway
        const double scaled = ldexp(x, sh);
        const double q1 = rint(scaled);
        const T xt = static_cast<T>(x);
        const double q2 = rint(ldexp(static_cast<double>(static_cast<T>(xt * xt)), sh2));   // np.power(B0, 2) rounds in B's own type
        if (x != 0.0 && fabs(x) < small_below) {
            ++k_small;
            k_rounded += q1 != scaled;
        }
        const unsigned long long u1 = static_cast<unsigned long long>(static_cast<long long>(q1) + bias);
        const unsigned long long u2 = static_cast<unsigned long long>(static_cast<long long>(q2) + bias);
        unsigned char *dst = bs + ct * tile_bytes + r * row_bytes + tx;
#pragma unroll
        for (int t = 0; t < MF_NS; ++t) {
            dst[t * 32] = static_cast<unsigned char>(((u1 >> (8 * t)) & 0xFFu) ^ 0x80u);
            dst[t * 32 + 16] = static_cast<unsigned char>(((u2 >> (8 * t)) & 0xFFu) ^ 0x80u);
        }
        dst[MF_NS * 32] = present ? 1 : 0;
        dst[MF_NS * 32 + 16] = 0;
        if (zf_hi_off) {
            // the filtered form's layouts: high digits (3-5) of values | squares and the not-NaN slice, 128-byte rows; low digits
            // (0-2), 96-byte rows (read by the resolve kernel only)
            unsigned char *hi = bs + zf_hi_off + ct * ((n + 1) * 128) + r * 128 + tx;
            unsigned char *lo = bs + zf_lo_off + ct * ((n + 1) * 96) + r * 96 + tx;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                lo[t * 32] = static_cast<unsigned char>(((u1 >> (8 * t)) & 0xFFu) ^ 0x80u);
                lo[t * 32 + 16] = static_cast<unsigned char>(((u2 >> (8 * t)) & 0xFFu) ^ 0x80u);
                hi[t * 32] = static_cast<unsigned char>(((u1 >> (8 * (t + 3))) & 0xFFu) ^ 0x80u);
                hi[t * 32 + 16] = static_cast<unsigned char>(((u2 >> (8 * (t + 3))) & 0xFFu) ^ 0x80u);
            }
            hi[3 * 32] = present ? 1 : 0;
            hi[3 * 32 + 16] = present ? 1 : 0;      // (also under the squares' columns: their lanes count the members of the rows THEY test)
            // ... and both fixed-point values as one 16-byte word, [row][column] (the square is never negative: bit 62 carries the
            // not-NaN flag): what k_mfma_resolve_z sums
            if (z64 && j < mloc)
                z64[r * mloc + j] = make_longlong2(static_cast<long long>(q1), static_cast<long long>(q2) | (present ? (1ll << 62) : 0ll));
        }
    }
    if (k_small) atomicAdd(&s_small[tx], k_small);
    if (k_rounded) atomicAdd(&s_rounded[tx], k_rounded);
    if (k_zero) atomicAdd(&s_zero[tx], k_zero);
    __syncthreads();
    if (threadIdx.x < 16 && c0 + threadIdx.x < mloc) {
        if (s_small[threadIdx.x]) atomicAdd(&n_small[c0 + threadIdx.x], s_small[threadIdx.x]);
        if (s_rounded[threadIdx.x]) atomicAdd(&n_small_rounded[c0 + threadIdx.x], s_rounded[threadIdx.x]);
        if (s_zero[threadIdx.x]) atomicAdd(&n_zero[c0 + threadIdx.x], s_zero[threadIdx.x]);
    }
}

// z-scores (n_zero != NULL) also decline a column that is BOTH held inexactly and has exact zeros: a z-score does not change
// when its values are scaled, so neighborhoods whose non-zero members are one value each (sparse columns) give mathematically
// EQUAL scores from different values, the reference decides such comparisons by the rounding of its f64 operations, and only
// the very same operands reproduce that (the f64 kernels do; a rounded fixed-point image cannot).
__global__ void k_mfma_colcheck(const unsigned int *__restrict__ cnt, const unsigned int *__restrict__ n_small,
                                const unsigned int *__restrict__ n_small_rounded, const unsigned int *__restrict__ n_zero,
                                const unsigned int *__restrict__ inexact, int64_t mloc, int *__restrict__ bad) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= mloc) return;
    if (n_zero && n_zero[j] && inexact[j]) atomicOr(bad, 1);
    // values that the grid holds to fewer than 26 bits must be a negligible part of the column (< 1 in 1024): sums made of
    // such values only would be compared at the grid's resolution, not at f64's
    if (1024ull * n_small_rounded[j] > cnt[j]) atomicOr(bad, 1);
}

// source-row maps of a span: row 0 = identity (observed score), row 1 + q = permutation p_base + q
__global__ __launch_bounds__(256) void k_mfma_src(const int32_t *__restrict__ order, int64_t n_src, int64_t n,
                                                  const int32_t *__restrict__ table, int64_t p_base,
                                                  int32_t *__restrict__ out, int dbg_window = 0) {
    const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (u >= n_src) return;
    const int64_t q = blockIdx.y;
    const int32_t node = order[u];
    int32_t src = q == 0 ? node : table[(p_base + q - 1) * (n + 1) + node];
    if (dbg_window > 0) src %= dbg_window;                  // diagnostic (SAFE_HIP_MFMA_DBG_WINDOW): every gather inside a small L2-resident window
    out[q * n_src + u] = src;
}

// ---------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t expand4(uint32_t word, uint32_t bit) {   // bits [bit, bit+4) -> 4 bytes of 0/1
    const uint32_t nib = __builtin_amdgcn_ubfe(word, bit, 4u);
    return static_cast<uint32_t>(__umul24(nib, 0x204081u)) & 0x01010101u;             // copies at bit 0, 7, 14, 21: no overlap
}

// 4 x 4 byte transpose: in[r] = 4 columns of row r  ->  out[c] = 4 rows of column c
__device__ __forceinline__ void transpose4(const uint32_t (&in)[4], uint32_t (&out)[4]) {
    const uint32_t lo01 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u);   // r0c0 r1c0 r0c1 r1c1
    const uint32_t hi01 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);   // r0c2 r1c2 r0c3 r1c3
    const uint32_t lo23 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u);
    const uint32_t hi23 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);
    out[0] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
}

// (diagnostic switches of the general kernel -- SAFE_HIP_MFMA_DBG bits 1 / 4 / 8: no transposed stores / no barrier / no score
// completion, WRONG results -- exist only in a library built with make DIAG=1)
#ifdef SAFE_HIP_DIAG
#define MF_DIAG(x) (x)
#else
#define MF_DIAG(x) false
#endif
// COUNTS = false: the permutation test (six i8 slices of ONE 32-column tile per task).
// COUNTS = true : observed counts only, for 0/1 attributes (hypergeometric path, 'sum' scores): the six
//                 planes of a task are six adjacent 32-column TILES with one plane each, n_q = 1, and
//                 the epilogue writes through `hl` (table lookup or plain counts).
// Z = true   : z-scores (safe_extras.py:19-31).  A task is a tile of SIXTEEN attribute columns: bytes 0-15 of a slice row
//                are the digits of B0, bytes 16-31 the digits of B0^2 (both six slices), and a seventh slice carries the
//                not-NaN flags -- so the same 32-column MFMA tile accumulates sum, sum of squares and count at once.  The
//                lane that owns column a of a row and the lane that owns column 16 + a differ in lane bit 2: when a score
//                completes the square sums cross over with one shuffle and the owner of column a evaluates
//                mean / sqrt(EXX - mean^2) in f64, in the reference's order of operations, on the EXACT integer sums.
//                Counters then hold (#>= << 16 | #<=) like the f64 kernels' (NaN scores compare false).
// EPI (counts form only): which epilogue this instantiation carries -- 0 = packed u16 counts for k_hyp_emit (the default split
// form), 1 = plain counts, 2 = table lookup fused into the epilogue.  One kernel with all three kept the fused form's pipelined
// table values (16 x double2 + nodes + slab offsets) in the register budget of the main loop: 60 VGPRs spilled.
// FM (filtered permutation test of six-slice columns, the exact counts with half the matrix work):
// every value is q = hi * 2^24 + lo with hi = digits 3-5 and lo = digits 0-2 (|lo| <= MF_LO_MAX), so a score is
// v = 2^24 V_hi + V_lo with |V_lo| <= members * MF_LO_MAX.  The matrix cores only form V_hi; against the EXACT observed
// score O (two observe passes, FM = 1: one over the low and one over the high digits) a permuted score is
//   certainly smaller   when V_hi <  T0          T0 = floor((O - B) / 2^24),  B = members * MF_LO_MAX
//   certainly greater   when V_hi >  T0 + W      W  = floor(2 B / 2^24) + 1
// and otherwise (a window of ~members units of 2^24 against a spread of ~2^25 sqrt(members): ~1e-5 of the compares on
// N(0,1) data) it is appended to a list and decided EXACTLY by k_mfma_resolve, which sums the low digits of the
// neighborhood's members.  Counters therefore equal the six-slice kernel's bit for bit.
//   FM = 0  classic (all slices on the matrix cores, exact compare in the kernel)
//   FM = 1  observe: n_q = 1, the completed score is stored (shift 0) or added (<< shift) to obs64[column][row]
//   FM = 2  filter: every q is a permutation; thresholds from obs64 at the start of the task
constexpr long long MF_LO_MAX = 128ll * (1ll + 256ll + 65536ll);          // |digits 0-2 of a balanced base-256 number|
struct MfmaFilt {
    long long *obs64 = nullptr;               // [column][padded row] exact observed scores (fixed point)
    const int32_t *rowcnt = nullptr;          // [padded row] members of the row's neighborhood (0: padding)
    ulonglong2 *amb = nullptr;                // undecided compares: {row | column << 32, V_hi << 16 | permutation}
    unsigned int *amb_count = nullptr;
    unsigned int amb_cap = 0;
    int obs_shift = 0;
    int p_base = 0;                           // the launch's first permutation
    unsigned long long *prof = nullptr;       // (diagnostic build 512) cycles per phase, summed over waves: [wave 0-3][8]
    const double *zobs = nullptr;             // z-scores: the observed scores ns[node][column] (NaN = no test)
};

template <bool COUNTS, int NS, bool Z = false, bool SKIP = true, int EPI = 0, bool PREF = true, int FM = 0>
__global__ __launch_bounds__(512) void k_permtest_mfma(
    const unsigned char *__restrict__ bs, int64_t row_bytes, int64_t tile_bytes, const int32_t *__restrict__ srcp, int64_t n_src, int n_q,
    const int32_t *__restrict__ blk_ptr, const int32_t *__restrict__ blk_kb, const uint32_t *__restrict__ blk_bits,
    const int2 *__restrict__ tasks, const int32_t *__restrict__ q_off, unsigned int *__restrict__ q_ctr, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_padr, const int32_t *__restrict__ rowmap,
    const double *__restrict__ col_scale, double *__restrict__ ns_out, HypLookup hl, MfmaFilt fa) {
    static_assert(!Z || !COUNTS, "z-scores are a permutation test");
    static_assert(!Z || (FM == 0 && NS == MF_NS + 1) || (FM == 2 && NS == MF_NS / 2 + 1), "z-scores: six (filtered: three high) value | square slices + the not-NaN slice");
    static_assert(FM == 0 || Z || (NS == MF_NS / 2 && !COUNTS), "filtered form: three slices of 'sum' scores");
    constexpr bool MF_TRG = mf_trg(NS);
    constexpr int KS = mf_ks(NS), BUF = 4 * KS, RS = mf_rs(NS);             // LDS bytes per k-step / per super-step buffer / per row (MF_TRG)
    constexpr int MF_SQ_LANE = MF_TRG ? 16 : 4;   // z-scores: the lane holding column c + 16 (the squares) of the lane that holds column c
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [2][BUF] + kb list
    __shared__ int slot_box;
    __shared__ unsigned int wg_xmax;
    unsigned int published = 0;
    if (threadIdx.x == 0) wg_xmax = 0;                                        // (ordered before its first use by the task-fetch barriers)
    int32_t *kb_list = reinterpret_cast<int32_t *>(lds + 2 * BUF);
    // observed scores of this thread's 16 outputs (exact 64-bit integers), [r][thread]: read once per permutation
    long long *obs = reinterpret_cast<long long *>(lds + 2 * BUF + MF_MAXBLK * sizeof(int32_t)) + threadIdx.x;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lam = lane & 31, h = lane >> 5;
    // this wave's 32 rows of the group: waves w and w + 4 share a SIMD (a workgroup's waves go round the four SIMDs) and own the
    // ADJACENT pieces 2 (w & 3) and 2 (w & 3) + 1 -- the pairing build_blocks balances, and the one k_permtest_mfma_f's
    // 64-row waves have
    const int wrow = (2 * (wave & 3) + (wave >> 2)) * 32;
    // gather role: thread -> (k-step of the super-step, row quad, 16-byte chunk of the row segment)
    constexpr int CH = 2 * NS, GT = 4 * 8 * CH;
    const bool gth = tid < GT;
    // EVERY thread issues the gather's loads (the 512 - GT threads without a gather role repeat another thread's: same cache
    // lines, results dropped): with the loads inside `if (gth)` the compiler's wait-count pass had to assume the fewest
    // outstanding loads at every wait, and a gather wave then waited for the membership words it had requested a moment ago --
    // a full L2 round trip at the top of every super-step
    const int gt = tid % GT;
    const int chunk = gt % CH, rq = (gt / CH) % 8, ks_g = gt / (8 * CH);
    const int s_g = chunk >> 1, half_g = chunk & 1;
    // this thread's LDS write base inside a buffer; column i of its 16 goes to lane slot
    // (i & 3) + 4 * half + 8 * (i >> 2)
    const uint32_t w_base = MF_TRG ? static_cast<uint32_t>(ks_g * KS + (4 * rq) * RS + chunk * 16)
                                   : static_cast<uint32_t>(ks_g * KS + s_g * MF_SS + (rq >> 2) * 512 + (4 * half_g) * 16 + (rq & 3) * 4);
    // MFMA B operand of this lane.  MF_TRG: within 16 lanes, lane l reads the 8-byte piece (row l >> 1, half l & 1) of the 16-byte
    // column group (lane >> 4) & 1 and receives column l of the 8 x 16 tile (tools/ubench/tr8_probe.hip)
    const uint32_t r_base = MF_TRG ? static_cast<uint32_t>((16 * h + ((lane & 15) >> 1)) * RS + 16 * ((lane >> 4) & 1) + 8 * (lane & 1))
                                   : static_cast<uint32_t>(h * 512 + lam * 16);
    const int col_in_tile = MF_TRG ? lam : 16 * ((lam >> 2) & 1) + 4 * (lam >> 3) + (lam & 3);
    (void)s_g, (void)half_g;

    const int home = blockIdx.x & 7;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const int qx = (home + attempt) & 7;
        const int q_begin = q_off[qx], q_len = q_off[qx + 1] - q_begin;
        for (;;) {
            if (tid == 0) slot_box = static_cast<int>(atomicAdd(&q_ctr[qx], 1u));
            __syncthreads();
            const int slot = slot_box;
            __syncthreads();
            if (slot >= q_len) break;
            const int2 task = tasks[q_begin + slot];
            const int g = task.x, ct = task.y;
            const int b0 = blk_ptr[g], nb = blk_ptr[g + 1] - b0, S = nb >> 2;
            if (S == 0) continue;
            for (int i = tid; i < nb; i += 512) kb_list[i] = blk_kb[b0 + i];
            __syncthreads();

            const unsigned char *bs_ct = bs + static_cast<int64_t>(ct) * tile_bytes + chunk * 16;
            const uint32_t *bits_w = blk_bits + static_cast<int64_t>(b0) * MF_R + wrow + lam;
            const int total = n_q * S;

            // z-scores: this lane's attribute column and the power-of-two scales of its sum / sum of squares
            const int64_t colz = static_cast<int64_t>(ct) * 16 + (col_in_tile & 15);
            double sc1 = 1.0, sc2 = 1.0;
            if (Z && colz < mloc) {
                sc1 = col_scale[colz];
                sc2 = col_scale[mloc + colz];
            }

            v16i acc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][r] = 0;
            uint32_t cnt[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cnt[r] = 0;
            // filter: this lane's thresholds -- T0 in the LDS slot of the observed score, the window width in a register
            // (0xFFFFFFFF: padding row / column or an empty neighborhood: nothing is counted, nothing recorded)
            uint32_t win[(FM == 2 && !Z) ? 16 : 1];
            double sc1sq = 1.0;
            if constexpr (FM == 2 && Z) {
                // z-scores, filtered: the observed scores of this lane's 16 outputs (formed by the seven-slice kernel in a pass of
                // its own) go to the LDS slots; NaN (padding, squares' lanes, fewer than three values, no spread) = no test
                // the lanes of the VALUE columns test rows r = 0..7 of their 16 accumulator rows, the lanes of the SQUARES' columns
                // (idle in the seven-slice form: one lane-bit away, same attribute) rows 8..15 -- slot rr of either holds its row
                sc1sq = sc1 * sc1;
                const int r_off = (col_in_tile & 16) ? 8 : 0;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rr + r_off;
                    const int64_t u = static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int32_t node = rowmap[u];
                    const bool live = node >= 0 && colz < mloc;
                    const double o = live ? fa.zobs[static_cast<int64_t>(node) * mloc + colz] : __longlong_as_double(0x7FF8000000000000ll);
                    obs[rr * 512] = __double_as_longlong(o);
                }
            }
            if constexpr (FM == 2 && !Z) {
                const int64_t colf = static_cast<int64_t>(ct) * 32 + col_in_tile;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t u = static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int32_t members = fa.rowcnt[u];
                    const bool live = colf < mloc && members > 0;
                    const long long bound = static_cast<long long>(members) * MF_LO_MAX;
                    const long long o = live ? fa.obs64[colf * n_padr + u] : 0ll;
                    obs[r * 512] = live ? (o - bound) >> 24 : 0ll;
                    win[r] = live ? static_cast<uint32_t>((2 * bound) >> 24) + 1u : 0xFFFFFFFFu;
                }
            }

            auto load_src = [&](int q, int t) -> int4 {
                q = q < n_q ? q : n_q - 1;                           // (look-ahead past the task's end: a valid row of indices, never used)
                const int kb = kb_list[4 * t + ks_g];
                return *reinterpret_cast<const int4 *>(srcp + static_cast<int64_t>(q) * n_src + static_cast<int64_t>(kb) * 32 + 4 * rq);
            };
            auto load_rows = [&](const int4 &src, uint4 (&L)[4]) {
                L[0] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.x) * row_bytes);
                L[1] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.y) * row_bytes);
                L[2] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.z) * row_bytes);
                L[3] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.w) * row_bytes);
            };
            // column word cw (columns 4cw .. 4cw+3 of this thread's chunk) of the four rows in L:
            // transpose to k-contiguous bytes and write the four lane slots b + 8cw (+ 4 half)
            auto store_quarter = [&](const uint4 (&L)[4], int cw, int buf) {
                unsigned char *dst = lds + buf * BUF + w_base;
                if constexpr (MF_TRG) {
                    uint4 v;                                                     // row cw of the thread's four, as it came
                    v.x = cw == 0 ? L[0].x : cw == 1 ? L[1].x : cw == 2 ? L[2].x : L[3].x;   // (component-wise: a select between
                    v.y = cw == 0 ? L[0].y : cw == 1 ? L[1].y : cw == 2 ? L[2].y : L[3].y;   // whole uint4 lvalues keeps L in
                    v.z = cw == 0 ? L[0].z : cw == 1 ? L[1].z : cw == 2 ? L[2].z : L[3].z;   // scratch memory)
                    v.w = cw == 0 ? L[0].w : cw == 1 ? L[1].w : cw == 2 ? L[2].w : L[3].w;
                    *reinterpret_cast<uint4 *>(dst + cw * RS) = v;
                    return;
                }
                const uint32_t w[4] = {cw == 0 ? L[0].x : cw == 1 ? L[0].y : cw == 2 ? L[0].z : L[0].w,
                                       cw == 0 ? L[1].x : cw == 1 ? L[1].y : cw == 2 ? L[1].z : L[1].w,
                                       cw == 0 ? L[2].x : cw == 1 ? L[2].y : cw == 2 ? L[2].z : L[2].w,
                                       cw == 0 ? L[3].x : cw == 1 ? L[3].y : cw == 2 ? L[3].z : L[3].w};
                uint32_t o[4];
                transpose4(w, o);
#pragma unroll
                for (int b = 0; b < 4; ++b) *reinterpret_cast<uint32_t *>(dst + (b + 8 * cw) * 16) = o[b];
            };
            auto advance = [&](int &qq, int &tt) {
                if (++tt == S) {
                    tt = 0;
                    ++qq;
                }
            };

            // Pipeline (one super-step = 4 k-steps = 128 gathered attribute rows per iteration):
            //   rows of super-step it+2 are requested at the top of iteration it, land while it and
            //   it+1 compute, and are transposed into the other LDS buffer during it+1;
            //   their source-row indices were requested one iteration earlier still.
            // vmcnt retires in order, so inside an iteration the loads needed soonest (membership
            // words, source indices) are issued before the row gathers, and the two row register
            // sets / index registers swap roles between iterations instead of being copied
            // (a copy would wait for the gather that was just issued).
            uint4 L_a[4], L_b[4];
            int4 src_a = make_int4(0, 0, 0, 0), src_b = make_int4(0, 0, 0, 0);
            int q1 = 0, t1 = 0, q2, t2, q3, t3;                      // (q, t) of iterations it + 1, + 2, + 3
            advance(q1, t1);
            q2 = q1, t2 = t1;
            advance(q2, t2);
            {
                // (all look-ahead loads are unconditional -- past the task's end they fetch valid rows nobody uses: a load inside
                // `if (more)` made the wait-count pass assume the path without it, and every super-step then began with a wait
                // for the membership words requested a moment earlier)
                const int4 s0 = load_src(0, 0);
                load_rows(s0, L_b);
                const int4 s1 = load_src(q1, t1);
                load_rows(s1, L_a);                                  // stored during iteration 0
                src_a = load_src(q2, t2);                            // gathered at the top of iteration 0
                if (gth) {
#pragma unroll
                    for (int cw = 0; cw < 4; ++cw) store_quarter(L_b, cw, 0);
                }
            }
            uint32_t aw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) aw[k] = bits_w[static_cast<int64_t>(k) * MF_R];
            __syncthreads();

            int q = 0, t = 0;
            auto body = [&](int it, uint4 (&L_store)[4], uint4 (&L_load)[4], const int4 &src_use, int4 &src_load)
                            __attribute__((always_inline)) {
                const int buf = it & 1;
                q3 = q2, t3 = t2;
                advance(q3, t3);
                const bool more1 = it + 1 < total;
                uint32_t aw_next[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) aw_next[k] = bits_w[static_cast<int64_t>(4 * t1 + k) * MF_R];
                src_load = load_src(q3, t3);
                load_rows(src_use, L_load);

                const unsigned char *bbuf = lds + buf * BUF + r_base;
                // this wave's 32 x 32 piece of a block may hold no member at all (42 % of the pieces at configs[4]: a 256-row
                // group spans more of the layout than one neighborhood radius): multiplying zeros is skipped, wave-uniformly
                // -- the SIMD's matrix pipe goes to its other wave.  (The LDS operand reads stay unconditional: skipping them too
                // measured 1.44 s against 1.37 s at the configs[4] rank share -- the branches break the one-k-step-ahead read
                // schedule and cost four spilled registers.)
                bool nz[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) nz[k] = !SKIP || __builtin_amdgcn_ballot_w64(aw[k] != 0u) != 0ull;
                // PF: the B operands of k-step k+1 are read from LDS before the MFMAs of k-step k are issued (two operand sets).
                // !PF: one operand set, read right before its MFMAs and only for the pieces that hold members -- the SIMD's other
                // wave covers the LDS latency; the z-score form (seven slices: no room for a second set) always runs this way
                constexpr bool PF = PREF && (!Z || NS <= 4);
                v4i b_cur[NS], b_nxt[PF ? NS : 1];
                auto read_operand = [&](int k, int s) -> v4i {
                    if constexpr (MF_TRG) {
                        typedef int v2i __attribute__((ext_vector_type(2)));
                        typedef __attribute__((address_space(3))) v2i lds_v2i;
                        const unsigned char *at = bbuf + k * KS + s * 32;
                        const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at));            // k = 16 h + 0..7
                        const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at + 8 * RS));   // k = 16 h + 8..15
                        v4i r;
                        r[0] = lo[0], r[1] = lo[1], r[2] = hi[0], r[3] = hi[1];
                        return r;
                    } else {
                        return *reinterpret_cast<const v4i *>(bbuf + k * KS + s * MF_SS);
                    }
                };
                if constexpr (PF) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) b_cur[s] = read_operand(0, s);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (PF) {
                        if (k < 3) {
#pragma unroll
                            for (int s = 0; s < NS; ++s)
                                b_nxt[PF ? s : 0] = read_operand(k + 1, s);
                        }
                        __builtin_amdgcn_sched_barrier(0);           // keep the LDS reads ahead of this k-step's MFMAs
                    }
                    if (nz[k]) {
                        if constexpr (!PF) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) b_cur[s] = read_operand(k, s);
                        }
                        v4i a;
                        a[0] = static_cast<int>(expand4(aw[k], 16 * h));
                        a[1] = static_cast<int>(expand4(aw[k], 16 * h + 4));
                        a[2] = static_cast<int>(expand4(aw[k], 16 * h + 8));
                        a[3] = static_cast<int>(expand4(aw[k], 16 * h + 12));
#pragma unroll
                        for (int s = 0; s < NS; ++s) acc[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_cur[s], acc[s], 0, 0, 0);
                    }
                    // a quarter of the next super-step's tile goes to the other buffer while the
                    // matrix pipe works through this k-step
                    if (gth && more1 && !MF_DIAG(hl.dbg & 1)) store_quarter(L_store, k, buf ^ 1);
                    if constexpr (PF) {
#pragma unroll
                        for (int s = 0; s < NS; ++s) b_cur[s] = b_nxt[PF ? s : 0];
                    }
                }

                if constexpr (COUNTS) {
                    // (the counts are written once, after the loop)
                } else if (t == S - 1 && !MF_DIAG(hl.dbg & 8)) {     // a score is complete
                    uint32_t undecided = 0;                                 // (filter) outputs the high digits leave open
                    if constexpr (Z && FM == 2) {
                        // The filtered z-score test.  Only the HIGH digits were multiplied: S1 = a + S1_low, S2 = b + S2_low with
                        // |low| <= e = count * MF_LO_MAX.  z >= o is a statement about the sign of S1 and of
                        //     T = S1^2 sc1^2 (1 + o^2) - o^2 S2 sc2 count          (mu / sigma >= o, squared and cleared of count^2)
                        // so it is decided here whenever the signs are certain despite the low parts (and the f64 evaluation: a
                        // 2^-40 margin); whatever is not -- including every score whose variance is not clearly positive --
                        // goes to k_mfma_resolve_z, which evaluates the reference's formula on the exact sums.
                        // The work is split between the value lane and the squares' lane of an attribute: the value lane hands over
                        // its sums of rows 8..15 and receives the squares of rows 0..7 (one exchange per row pair), then BOTH evaluate
                        // eight outputs (all 64 lanes busy; with the value lanes doing all sixteen this was 27 % of the kernel)
                        const bool sq = (col_in_tile & 16) != 0;
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            long long x = static_cast<long long>(acc[2][rr]), y = static_cast<long long>(acc[2][rr + 8]);
#pragma unroll
                            for (int s = 1; s >= 0; --s) {
                                x = (x << 8) + static_cast<long long>(acc[s][rr]);
                                y = (y << 8) + static_cast<long long>(acc[s][rr + 8]);
                            }
                            const long long give = sq ? x : y;
                            const int g_lo = __shfl_xor(static_cast<int>(give), MF_SQ_LANE), g_hi = __shfl_xor(static_cast<int>(give >> 32), MF_SQ_LANE);
                            const long long got = (static_cast<long long>(g_hi) << 32) | static_cast<long long>(static_cast<uint32_t>(g_lo));
                            const long long v = sq ? got : x, w = sq ? y : got;        // sums / sums of squares (high digits) of MY row
                            const double o = __longlong_as_double(obs[rr * 512]);
                            const double members = static_cast<double>(sq ? acc[3][rr + 8] : acc[3][rr]);   // (the not-NaN slice is in both halves)
                            if (o == o && members >= 3.0) {                  // (observed NaN: no test; fewer than 3 values: the score is NaN -- safe_extras.py:30)
                                const double a = static_cast<double>(v) * 16777216.0, b = static_cast<double>(w) * 16777216.0;
                                const double e = members * static_cast<double>(MF_LO_MAX);
                                const double o2 = o * o, k1 = sc1sq * (1.0 + o2), k2 = o2 * sc2;
                                const double slack1 = 2.0 * fabs(a) * e + e * e;                         // |S1^2 - a^2| <=
                                const double p1 = a * a * k1, p2 = b * members * k2;
                                const double T = p1 - p2;
                                const double E = k1 * slack1 + k2 * members * e + (p1 + fabs(p2)) * 0x1p-40;
                                const double V = b * members * sc2 - a * a * sc1sq;                      // count^2 * variance, high parts
                                const bool var_ok = V - (sc2 * members * e + sc1sq * slack1) > fabs(b) * members * sc2 * 0x1p-20;
                                const bool t_pos = T > E, t_neg = T < -E, s_pos = a > e, s_neg = a < -e;
                                const bool greater = var_ok && (o >= 0.0 ? (s_pos && t_pos) : (s_pos || t_neg));
                                const bool smaller = var_ok && (o >= 0.0 ? (s_neg || t_neg) : (s_neg && t_pos));
                                cnt[rr] += greater ? (1u << 16) : (smaller ? 1u : 0u);
                                undecided |= (!greater && !smaller) ? (1u << rr) : 0u;
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < ((Z && FM == 2) ? 0 : 16); ++r) {
                        constexpr int NV = Z ? (FM == 2 ? MF_NS / 2 : MF_NS) : NS;   // value slices
                        long long v = static_cast<long long>(acc[NV - 1][r]);
#pragma unroll
                        for (int s = NV - 2; s >= 0; --s) v = (v << 8) + static_cast<long long>(acc[s][r]);
                        if constexpr (Z && FM == 2) {
                            // (handled above: eight outputs per lane)
                        } else if constexpr (Z) {
                            // columns 0-15 of the tile: sum (and count, slice 6); columns 16-31: sum of squares, one lane-bit away (MF_SQ_LANE)
                            const int lo_sq = __shfl_xor(static_cast<int>(v), MF_SQ_LANE), hi_sq = __shfl_xor(static_cast<int>(v >> 32), MF_SQ_LANE);
                            const long long w = (static_cast<long long>(hi_sq) << 32) | static_cast<long long>(static_cast<uint32_t>(lo_sq));
                            const double members = static_cast<double>(acc[NS - 1][r]);
                            const double mean = (static_cast<double>(v) * sc1) / members;          // safe_extras.py:21-23
                            const double exx = (static_cast<double>(w) * sc2) / members;           // safe_extras.py:25-26
                            const double sd = sqrt(exx - mean * mean);                            // safe_extras.py:27
                            double zs = mean / sd;                                                // safe_extras.py:28
                            if (sd == 0.0) zs = __longlong_as_double(0x7FF8000000000000ll);       // safe_extras.py:29
                            if (members < 3.0) zs = __longlong_as_double(0x7FF8000000000000ll);   // safe_extras.py:30
                            if (q == 0) {
                                obs[r * 512] = __double_as_longlong(zs);
                            } else {
                                const double o = __longlong_as_double(obs[r * 512]);
                                cnt[r] += (static_cast<uint32_t>(zs >= o) << 16) | static_cast<uint32_t>(zs <= o);
                            }
                        } else if constexpr (FM == 1 && !Z) {
                            const int64_t colf = static_cast<int64_t>(ct) * 32 + col_in_tile;
                            if (colf < mloc) {
                                long long *dst = fa.obs64 + colf * n_padr + static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h;
                                *dst = fa.obs_shift ? *dst + (v << fa.obs_shift) : v;
                            }
                        } else if constexpr (FM == 2 && !Z) {
                            const long long x = v - obs[r * 512];
                            const bool below = x < 0;
                            const bool above = static_cast<unsigned long long>(x) > static_cast<unsigned long long>(win[FM == 2 ? r : 0]);
                            cnt[r] += below ? (1u << 16) : (above ? 1u : 0u);         // certainly smaller / certainly greater
                            undecided |= (!below && !above && win[FM == 2 ? r : 0] != 0xFFFFFFFFu) ? (1u << r) : 0u;
                        } else if (q == 0) {
                            obs[r * 512] = v;
                        } else {
                            const long long o = obs[r * 512];
                            cnt[r] += (static_cast<uint32_t>(v < o) << 16) | static_cast<uint32_t>(v > o);
                        }
                        if constexpr (FM != 2) {
#pragma unroll
                            for (int s = 0; s < NS; ++s) acc[s][r] = 0;
                        }
                    }
                    if constexpr (FM == 2 && Z) {
                        if (__builtin_expect(undecided != 0u, 0)) {
                            const unsigned long long u0 = static_cast<unsigned long long>(static_cast<int64_t>(g) * MF_R + wrow + 4 * h);
                            for (uint32_t left = undecided; left;) {
                                const int r = __builtin_ctz(left) + ((col_in_tile & 16) ? 8 : 0);       // (slot -> accumulator row of this lane)
                                left &= left - 1u;
                                const unsigned int at = atomicAdd(fa.amb_count, 1u);
                                if (at >= fa.amb_cap) continue;
                                fa.amb[at] = make_ulonglong2((u0 + static_cast<unsigned long long>((r & 3) + 8 * (r >> 2))) |
                                                                 (static_cast<unsigned long long>(colz) << 32),
                                                             (3ull << 62) | static_cast<unsigned long long>(fa.p_base + q));
                            }
                        }
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[s][r] = 0;
                    }
                    if constexpr (FM == 2 && !Z) {
                        // rare (~1e-5 of the compares): the low digits decide (k_mfma_resolve)
                        if (__builtin_expect(undecided != 0u, 0)) {
                            // a per-lane loop over the set bits with a DYNAMIC element index (a select chain over the sixteen
                            // accumulator elements): unrolled over r the sixteen copies of this block cost the main loop 100
                            // spilled registers although it almost never runs
                            const unsigned long long colf = static_cast<unsigned long long>(static_cast<int64_t>(ct) * 32 + col_in_tile);
                            const unsigned long long u0 = static_cast<unsigned long long>(static_cast<int64_t>(g) * MF_R + wrow + 4 * h);
                            for (uint32_t left = undecided; left;) {
                                const int r = __builtin_ctz(left);
                                left &= left - 1u;
                                const unsigned int at = atomicAdd(fa.amb_count, 1u);
                                if (at >= fa.amb_cap) continue;
                                long long v = static_cast<long long>(acc[NS - 1][r]);
                                for (int s = NS - 2; s >= 0; --s) v = (v << 8) + static_cast<long long>(acc[s][r]);
                                const unsigned long long u = u0 + static_cast<unsigned long long>((r & 3) + 8 * (r >> 2));
                                fa.amb[at] = make_ulonglong2(u | (colf << 32), (static_cast<unsigned long long>(v) << 16) |
                                                                                   static_cast<unsigned long long>(fa.p_base + q));
                            }
                        }
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[s][r] = 0;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) aw[k] = aw_next[k];
                if (!MF_DIAG(hl.dbg & 4)) __syncthreads();
                q = q1, t = t1;
                q1 = q2, t1 = t2;
                q2 = q3, t2 = t3;
            };
            for (int it = 0; it < total; it += 2) {
                body(it, L_a, L_b, src_a, src_b);
                if (it + 1 < total) body(it + 1, L_b, L_a, src_b, src_a);
            }

            if constexpr (COUNTS) {
                // ---- the counts of six column tiles are complete (n_q = 1): write them through `hl`.
                //      Loads first, in batches the hardware can overlap: row -> node -> table slab -> value
                int32_t node[EPI == 0 ? 1 : 16];
                uint32_t slab[EPI == 0 ? 1 : 16];
                uint32_t kofs[NS];
                bool col_ok[NS];
                const uint32_t n_kid_u = static_cast<uint32_t>(hl.n_kid);     // table layout [size id][x][count id]
                if constexpr (EPI != 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        node[EPI == 0 ? 0 : r] = rowmap[static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        slab[EPI == 0 ? 0 : r] = (EPI == 2 && node[EPI == 0 ? 0 : r] >= 0)
                                                     ? static_cast<uint32_t>(hl.nid[node[EPI == 0 ? 0 : r]]) * static_cast<uint32_t>(hl.n_kid * hl.xs) : 0u;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int64_t col = (static_cast<int64_t>(ct) * NS + s) * 32 + col_in_tile;
                        col_ok[s] = col < mloc;
                        kofs[s] = (EPI == 2 && col_ok[s]) ? static_cast<uint32_t>(hl.kid[col]) : 0u;
                    }
                }
                if constexpr (EPI == 0) {
                    // split form: the counts leave as u16, six tiles of one (row, column-in-tile) packed into
                    // 12 bytes, 384 contiguous bytes per row and half-wave; k_hyp_emit streams the results out
                    static_assert(NS == 6, "packed counts hold six tiles");
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t u = static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h;
                        unsigned int *dst = hl.cnt16 + ((static_cast<int64_t>(ct) * n_padr + u) * 32 + col_in_tile) * 3;
                        uint3 w;
                        w.x = static_cast<uint32_t>(acc[0][r]) | (static_cast<uint32_t>(acc[1][r]) << 16);
                        w.y = static_cast<uint32_t>(acc[2][r]) | (static_cast<uint32_t>(acc[3][r]) << 16);
                        w.z = static_cast<uint32_t>(acc[4][r]) | (static_cast<uint32_t>(acc[5][r]) << 16);
                        *reinterpret_cast<uint3 *>(dst) = w;             // stays in L2 / MALL for the emit kernel
                    }
                    // largest count of the call: the emit kernel stages table columns [0, max] in LDS
                    int mx = 0;
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = max(mx, acc[s][r]);
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) mx = max(mx, __shfl_xor(mx, d));
                    if (lane == 0) atomicMax(&wg_xmax, static_cast<unsigned int>(mx));   // (LDS; published below, once per task at most)
                } else if constexpr (EPI == 1) {                      // plain counts ('sum' scores of 0/1 attributes)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int64_t col = (static_cast<int64_t>(ct) * NS + s) * 32 + col_in_tile;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (col_ok[s] && node[r] >= 0)
                                hl.pvalues_pos[static_cast<int64_t>(node[r]) * mloc + col] = static_cast<double>(acc[s][r]);
                    }
                } else {
                    // Software pipeline over the 12 (column tile, row half) rounds: the table values of round k+1
                    // are requested BEFORE the stores of round k are issued.  vmcnt retires in order, so a round
                    // that gathers only after storing waits for its predecessor's 24 write acknowledgements on
                    // top of its own gather latency, twelve times per task; in this order the wait for round
                    // k+1's values lets the stores of round k stay in flight.  The rounds are branch-free
                    // (padding rows / columns store to a per-lane dummy slot): with branches around the stores
                    // the compiler's wait-count pass cannot count them and serialises the rounds again.
                    double *const dummy = hl.dummy + lane;
                    auto fetch = [&](int s, int half, double2 (&val)[8]) __attribute__((always_inline)) {
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const int r = half * 8 + rr;
                            val[rr] = hl.tab[slab[r] + kofs[s] + ((col_ok[s] && node[r] >= 0) ? static_cast<uint32_t>(acc[s][r]) * n_kid_u : 0u)];
                        }
                    };
                    unsigned int hits[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) hits[s] = 0;
                    auto emit = [&](int s, int half, const double2 (&val)[8]) __attribute__((always_inline)) {
                        const int64_t col = (static_cast<int64_t>(ct) * NS + s) * 32 + col_in_tile;
#pragma unroll
                        for (int rr = 0; rr < 8; ++rr) {
                            const int r = half * 8 + rr;
                            const bool ok = col_ok[s] && node[r] >= 0;
                            const int64_t o = static_cast<int64_t>(node[r]) * mloc + col;
                            const bool hit = ok && val[rr].x < hl.p_cut;                    // safe.py:468-470 (nes_p_cut)
                            *(ok ? hl.pvalues_pos + o : dummy) = val[rr].x;
                            *(ok ? hl.nes + o : dummy) = val[rr].y;                         // -log10 p from the table (safe.py:608)
                            *(ok ? hl.nes_binary + o : dummy) = hit ? 1.0 : 0.0;
                            hits[s] += hit;
                        }
                    };
                    double2 va[8], vb[8];
                    fetch(0, 0, va);
#pragma unroll
                    for (int k = 0; k < 2 * NS; k += 2) {
                        fetch(k >> 1, 1, vb);
                        emit(k >> 1, 0, va);
                        if (k + 2 < 2 * NS) fetch((k >> 1) + 1, 0, va);
                        emit(k >> 1, 1, vb);
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int64_t col = (static_cast<int64_t>(ct) * NS + s) * 32 + col_in_tile;
                        if (hits[s]) atomicAdd(&hl.enriched[col], hits[s]);
                    }
                }
            }
            // ---- task epilogue: observed scores (first span only) and the counters
            const int64_t col = Z ? colz : static_cast<int64_t>(ct) * 32 + col_in_tile;
            if constexpr (Z && FM == 2) {                                 // eight outputs per lane: value lanes rows 0..7, squares' lanes 8..15
                if (col < mloc) {
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = rr + ((col_in_tile & 16) ? 8 : 0);
                        const int64_t u = static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (cnt[rr]) atomicAdd(&gl_counts[col * n_padr + u], cnt[rr]);
                    }
                }
            } else if (!COUNTS && FM != 1 && col < mloc && !(Z && (col_in_tile & 16))) {
                const double sc = (ns_out && !Z) ? col_scale[col] : 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t u = static_cast<int64_t>(g) * MF_R + wrow + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (cnt[r]) atomicAdd(&gl_counts[col * n_padr + u], cnt[r]);
                    if (FM == 0 && ns_out) {
                        const int32_t node = rowmap[u];
                        if (node >= 0)
                            ns_out[static_cast<int64_t>(node) * mloc + col] =
                                Z ? __longlong_as_double(obs[r * 512]) : static_cast<double>(obs[r * 512]) * sc;
                    }
                }
            }
            __syncthreads();                                         // kb_list / buffers are reused by the next task
            if constexpr (COUNTS) {
                // the call's largest count: every wave of every task on ONE global address cost 8 % of the kernel; the
                // workgroup keeps its own maximum in LDS and publishes it only when it grew
                if (EPI == 0 && tid == 0 && wg_xmax > published) {
                    published = wg_xmax;
                    atomicMax(hl.xmax, published);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// the filtered form's own kernel (FM = 2 above is the same arithmetic in the general kernel's shape)
// ---------------------------------------------------------------------------------------
// What the general kernel's shape costs once only three slices are multiplied: every wave reads the whole operand tile from
// LDS for its 32 rows (1 KB per MFMA), 320 of its 512 threads repeat gather loads they do not need, the membership words of a
// k-step arrive as four scalar-width loads, and all eight waves meet at one barrier per 128 gathered rows.  Here
//   * a workgroup is FOUR waves of 64 rows (two 32 x 32 pieces per wave: every LDS operand read feeds two MFMAs, 96
//     accumulator registers), and TWO workgroups share a CU -- the other workgroup's matrix work covers this one's barrier,
//     gather and score completion;
//   * the membership words of a super-step are one 16-byte load per piece (blk_bits4: [super-step][row][4 k-steps]);
//   * threads without a gather role take the source maps' padding block (every index = the zero row: one cache line per load
//     instruction, no select) instead of repeating a neighbour's rows;
//   * the thresholds are 32-bit: y = floor(V_hi / 16) against Y0 = floor((O - B') / 2^28), B' = B + 15 * 2^24, with one window
//     width per task (from the group's largest neighborhood: neighborhoods below 2048 members); what the test leaves open is
//     appended for k_mfma_resolve, which forms that score from all six digits (the record carries no partial sum);
//   * <= / >= counters are 8-bit fields (two outputs per register), flushed to memory every 255 permutations.
// (diagnostic builds) keeps a value -- and the loads / MFMAs that produce it -- alive without using it
__device__ __forceinline__ void mf_keep(uint32_t x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void mf_keep(const uint4 &x) { asm volatile("" ::"v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)); }
__device__ __forceinline__ void mf_keep(const v4i &x) { asm volatile("" ::"v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3])); }
constexpr int MF_F_MAXBLK = 2048;         // column blocks per row group this kernel can index from LDS (else the general kernel)
// DBG: diagnostic builds that skip work (WRONG results; only instantiated with -DSAFE_HIP_DIAG, selected by SAFE_HIP_MFMA_DBG):
// 1 no transposes / LDS stores, 2 no MFMAs, 4 no barrier per super-step, 8 no score completion, 16 no membership-word loads,
// 32 no source-index loads, 64 no row gathers, 128 always the same LDS buffer, 256 no LDS operand reads.  A template parameter:
// as a kernel argument the tests cost the main loop 16 spilled registers and made it three times slower.
// TR: the gathered rows go to LDS AS THEY ARE ([k][three slices x 32 bytes], one ds_write_b128 per row and thread) and the MFMA
// operand is read with the transposing LDS read (two ds_read_b64_tr_b8 per slice and k-step: within a 16-lane group lane l
// receives byte l & 7 of the 8-byte pieces 2 j + (l >> 3), j = 0..7 -- pieces laid over eight consecutive rows, that is column l
// of an 8 x 16 byte tile, eight consecutive k: tools/ubench/tr8_probe.hip) -- no v_perm transposes, no 4-byte scatter stores.
// !TR: 4 x 4 byte transposes in registers, k-contiguous LDS image, ds_read_b128 (the general kernel's operand path).
template <int DBG, bool TR = true>
__global__ __launch_bounds__(256, 2) void k_permtest_mfma_f(
    const unsigned char *__restrict__ bs, int64_t tile_bytes, const int32_t *__restrict__ srcp, int64_t n_src, int n_q,
    const int32_t *__restrict__ blk_ptr, const int32_t *__restrict__ blk_kb, const uint4 *__restrict__ blk_bits4,
    const int32_t *__restrict__ grp_maxcnt, const int2 *__restrict__ tasks, const int32_t *__restrict__ q_off,
    unsigned int *__restrict__ q_ctr, int64_t mloc, unsigned int *__restrict__ gl_counts, int64_t n_padr, MfmaFilt fa) {
    constexpr int dbg = DBG;
    constexpr int NS = MF_NS / 2, KS = TR ? 32 * NS * 32 : NS * MF_SS, BUF = 4 * KS;
    constexpr int64_t row_bytes = NS * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // [2][BUF] | kb list | Y0 [32][256] | counters [8][256]
    __shared__ int slot_box;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lam = lane & 31, h = lane >> 5;
    int32_t *kb_list = reinterpret_cast<int32_t *>(lds + 2 * BUF);
    int32_t *y0s = reinterpret_cast<int32_t *>(lds + 2 * BUF + MF_F_MAXBLK * sizeof(int32_t)) + tid;
    // counters: ONE 8-bit field per output in LDS ([8][256] words; output o = 16 p + r in word r & 7, byte 2 p + (r >> 3)) that counts
    // the permutations NOT certainly greater than the observed score; flushed every 255 permutations as
    // (#smaller << 16 | #greater) = (field << 16 | permutations - field).  An undecided compare is counted "smaller" and taken back
    // at once (a rare global atomic of -(1 << 16)); k_mfma_resolve then adds what it really was
    uint32_t *cnts = reinterpret_cast<uint32_t *>(lds + 2 * BUF + MF_F_MAXBLK * sizeof(int32_t) + 32 * 256 * sizeof(int32_t)) + tid;
    constexpr int CH = 2 * NS, GT = 4 * 8 * CH;                               // 192 gather threads = waves 0-2
    const bool gth = tid < GT;
    const int gt = gth ? tid : 0;
    const int chunk = gt % CH, rq = (gt / CH) % 8, ks_g = gt / (8 * CH);
    const int s_g = chunk >> 1, half_g = chunk & 1;
    // TR: a thread's row i goes to [k-step][row 4 rq + i][chunk]; a lane reads the 8-byte piece (row (l & 15) >> 1 of its eight,
    // half l & 1) of column half (lane >> 4) & 1 and k half lane >> 5 -- lane l then owns column l & 31 of the tile
    const uint32_t w_base = TR ? static_cast<uint32_t>(ks_g * KS + (4 * rq) * 96 + chunk * 16)
                               : static_cast<uint32_t>(ks_g * KS + s_g * MF_SS + (rq >> 2) * 512 + (4 * half_g) * 16 + (rq & 3) * 4);
    const uint32_t r_base = TR ? static_cast<uint32_t>((16 * h + ((lane & 15) >> 1)) * 96 + 16 * ((lane >> 4) & 1) + 8 * (lane & 1))
                               : static_cast<uint32_t>(h * 512 + lam * 16);
    const int col_in_tile = TR ? lam : 16 * ((lam >> 2) & 1) + 4 * (lam >> 3) + (lam & 3);
    // threads without a gather role (wave 3) take their source indices from the PADDING block of the source maps (index n_kb:
    // every entry is the zero row n), so their four row loads hit one cache line and need no select
    const int kb_pad = static_cast<int>(n_src / 32) - 1;

    const int home = blockIdx.x & 7;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const int qx = (home + attempt) & 7;
        const int q_begin = q_off[qx], q_len = q_off[qx + 1] - q_begin;
        for (;;) {
            if (tid == 0) slot_box = static_cast<int>(atomicAdd(&q_ctr[qx], 1u));
            __syncthreads();
            const int slot = slot_box;
            __syncthreads();
            if (slot >= q_len) break;
            const int2 task = tasks[q_begin + slot];
            const int g = task.x, ct = task.y;
            const int b0 = blk_ptr[g], nb = blk_ptr[g + 1] - b0, S = nb >> 2;
            if (S == 0) continue;
            for (int i = tid; i < nb; i += 256) kb_list[i] = blk_kb[b0 + i];

            const unsigned char *bs_ct = bs + static_cast<int64_t>(ct) * tile_bytes + (gth ? chunk * 16 : 0);
            const uint4 *bits_w = blk_bits4 + static_cast<int64_t>(b0 >> 2) * MF_R + wave * 64 + lane;   // lane l: row l of the wave's 64
            const int total = n_q * S;
            const int64_t colf = static_cast<int64_t>(ct) * 32 + col_in_tile;
            const int64_t u_lane = static_cast<int64_t>(g) * MF_R + wave * 64 + 4 * h;     // + 32 p + (r & 3) + 8 (r >> 2)

            // thresholds of this lane's 32 outputs (32-bit, in LDS); the window width is one number per task
            const long long b_max = static_cast<long long>(grp_maxcnt[g]) * MF_LO_MAX;
            const uint32_t wc = static_cast<uint32_t>((2 * b_max + (15ll << 24)) >> 28) + 2u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                long long o64[16];
                int32_t members[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t u = u_lane + 32 * p + (r & 3) + 8 * (r >> 2);
                    members[r] = fa.rowcnt[u];
                    o64[r] = colf < mloc ? fa.obs64[colf * n_padr + u] : 0ll;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long bp = static_cast<long long>(members[r]) * MF_LO_MAX + (15ll << 24);
                    // padding rows (-1 members) and padding columns only ever form y = 0: Y0 = 1 counts them "smaller" without any
                    // further test (their counters are never read); an EMPTY neighborhood of a real row takes the general rule
                    // (every compare a tie: undecided, settled as "equal" by the resolve kernel)
                    y0s[(16 * p + r) * 256] = (colf < mloc && members[r] >= 0) ? static_cast<int32_t>((o64[r] - bp) >> 28) : 1;
                }
            }
            __syncthreads();                                         // kb_list

            v16i acc[2][NS];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][s][r] = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) cnts[j * 256] = 0;
            int since_flush = 0;
            unsigned long long prof_acc[5] = {0, 0, 0, 0, 0};

            auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t w = cnts[j * 256];
                    cnts[j * 256] = 0;
                    if (colf < mloc) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) {                    // byte f: piece f >> 1, accumulator element j + 8 (f & 1)
                            const uint32_t smaller = (w >> (8 * f)) & 0xFFu;
                            const int r = j + 8 * (f & 1);
                            atomicAdd(&gl_counts[colf * n_padr + u_lane + 32 * (f >> 1) + (r & 3) + 8 * (r >> 2)],
                                      (smaller << 16) | (static_cast<uint32_t>(since_flush) - smaller));
                        }
                    }
                }
            };
            auto load_src_kb = [&](int q, int kb) -> int4 {
                q = q < n_q ? q : n_q - 1;
                return *reinterpret_cast<const int4 *>(srcp + static_cast<int64_t>(q) * n_src + static_cast<int64_t>(kb) * 32 + 4 * rq);
            };
            auto load_src = [&](int q, int t) -> int4 { return load_src_kb(q, gth ? kb_list[4 * t + ks_g] : kb_pad); };
            auto load_rows = [&](const int4 &src, uint4 (&L)[4]) {
                L[0] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.x) * row_bytes);
                L[1] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.y) * row_bytes);
                L[2] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.z) * row_bytes);
                L[3] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(src.w) * row_bytes);
            };
            auto store_quarter = [&](const uint4 (&L)[4], int cw, int buf) {
                unsigned char *dst = lds + buf * BUF + w_base;
                if constexpr (TR) {
                    uint4 v;                                                     // row cw of the thread's four, as it came
                    v.x = cw == 0 ? L[0].x : cw == 1 ? L[1].x : cw == 2 ? L[2].x : L[3].x;
                    v.y = cw == 0 ? L[0].y : cw == 1 ? L[1].y : cw == 2 ? L[2].y : L[3].y;
                    v.z = cw == 0 ? L[0].z : cw == 1 ? L[1].z : cw == 2 ? L[2].z : L[3].z;
                    v.w = cw == 0 ? L[0].w : cw == 1 ? L[1].w : cw == 2 ? L[2].w : L[3].w;
                    *reinterpret_cast<uint4 *>(dst + cw * 96) = v;
                    return;
                }
                const uint32_t w[4] = {cw == 0 ? L[0].x : cw == 1 ? L[0].y : cw == 2 ? L[0].z : L[0].w,
                                       cw == 0 ? L[1].x : cw == 1 ? L[1].y : cw == 2 ? L[1].z : L[1].w,
                                       cw == 0 ? L[2].x : cw == 1 ? L[2].y : cw == 2 ? L[2].z : L[2].w,
                                       cw == 0 ? L[3].x : cw == 1 ? L[3].y : cw == 2 ? L[3].z : L[3].w};
                uint32_t o[4];
                transpose4(w, o);
#pragma unroll
                for (int b = 0; b < 4; ++b) *reinterpret_cast<uint32_t *>(dst + (b + 8 * cw) * 16) = o[b];
            };
            auto advance = [&](int &qq, int &tt) {
                if (++tt == S) {
                    tt = 0;
                    ++qq;
                }
            };

            // the pipeline of k_permtest_mfma: rows of super-step it + 2 requested at the top of iteration it, transposed into
            // the other buffer during it + 1; their source indices one iteration earlier still
            uint4 L_a[4], L_b[4];
            int4 src_a = make_int4(0, 0, 0, 0), src_b = make_int4(0, 0, 0, 0);
            int q1 = 0, t1 = 0, q2, t2, q3, t3;
            advance(q1, t1);
            q2 = q1, t2 = t1;
            advance(q2, t2);
            {
                const int4 s0 = load_src(0, 0);
                load_rows(s0, L_b);
                const int4 s1 = load_src(q1, t1);
                load_rows(s1, L_a);
                src_a = load_src(q2, t2);
                if (gth) {
#pragma unroll
                    for (int cw = 0; cw < 4; ++cw) store_quarter(L_b, cw, 0);
                }
            }
            // membership words of a super-step: lane l loads the four words of row l; v_permlane32_swap then gives every lane the
            // words of row (l & 31) of piece 0 and of piece 1 (one 16-byte load per lane instead of two)
            auto split_rows = [&](const uint4 &raw, uint4 (&w)[2]) __attribute__((always_inline)) {
                typedef unsigned int v2u __attribute__((ext_vector_type(2)));
                const v2u x = __builtin_amdgcn_permlane32_swap(raw.x, raw.x, false, false);
                const v2u y = __builtin_amdgcn_permlane32_swap(raw.y, raw.y, false, false);
                const v2u z = __builtin_amdgcn_permlane32_swap(raw.z, raw.z, false, false);
                const v2u ww = __builtin_amdgcn_permlane32_swap(raw.w, raw.w, false, false);
                w[0] = make_uint4(x[0], y[0], z[0], ww[0]);
                w[1] = make_uint4(x[1], y[1], z[1], ww[1]);
            };
            uint4 aw[2];
            split_rows(bits_w[0], aw);
            int kb_next;                                             // column block of the source indices requested next (read one iteration early)
            {
                int q3p = q2, t3p = t2;
                advance(q3p, t3p);
                kb_next = gth ? kb_list[4 * t3p + ks_g] : kb_pad;
            }
            __syncthreads();

            int q = 0, t = 0;
            auto body = [&](int it, uint4 (&L_store)[4], uint4 (&L_load)[4], const int4 &src_use, int4 &src_load)
                            __attribute__((always_inline)) {
                const int buf = it & 1;
                q3 = q2, t3 = t2;
                advance(q3, t3);
                const bool more1 = it + 1 < total;
                unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                if (dbg & 512) c0 = __builtin_amdgcn_s_memtime();
                uint4 aw_raw = aw[0];
                if (!(dbg & 16)) aw_raw = bits_w[static_cast<int64_t>(t1) * MF_R];
                if (!(dbg & 32)) src_load = load_src_kb(q3, kb_next);
                {
                    int q4 = q3, t4 = t3;
                    advance(q4, t4);
                    kb_next = gth ? kb_list[4 * t4 + ks_g] : kb_pad;      // (consumed at the top of the next iteration: no wait here)
                }
                // the four row loads of the gather are issued one per k-step below: right after the barrier all four waves of the
                // workgroup (and often the CU's other workgroup) would queue 24 of them at once -- a wave spent ~450 cycles per
                // super-step getting its six loads accepted
                const int32_t row_of[4] = {src_use.x, src_use.y, src_use.z, src_use.w};

                const unsigned char *bbuf = lds + ((dbg & 128) ? 0 : buf * BUF) + r_base;
                if (dbg & 512) c1 = __builtin_amdgcn_s_memtime();
                // two operand sets: the slices of k-step k + 1 are read before the MFMAs of k-step k are issued
                v4i b_cur[NS], b_nxt[NS];
                auto read_operand = [&](int k, int s) -> v4i {
                    if constexpr (TR) {
                        typedef int v2i __attribute__((ext_vector_type(2)));
                        typedef __attribute__((address_space(3))) v2i lds_v2i;
                        const unsigned char *at = bbuf + k * KS + s * 32;
                        const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at));             // k = 16 h + 0..7
                        const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at + 8 * 96));    // k = 16 h + 8..15
                        v4i r;
                        r[0] = lo[0], r[1] = lo[1], r[2] = hi[0], r[3] = hi[1];
                        return r;
                    } else {
                        return *reinterpret_cast<const v4i *>(bbuf + k * KS + s * MF_SS);
                    }
                };
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    if (!(dbg & 256)) b_cur[s] = read_operand(0, s);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < 3) {
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            if (!(dbg & 256)) b_nxt[s] = read_operand(k + 1, s);
                    }
                    if (!(dbg & 64)) L_load[k] = *reinterpret_cast<const uint4 *>(bs_ct + static_cast<int64_t>(row_of[k]) * row_bytes);
                    __builtin_amdgcn_sched_barrier(0);               // the reads and the row load stay ahead of this k-step's MFMAs
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const uint32_t word = k == 0 ? aw[p].x : k == 1 ? aw[p].y : k == 2 ? aw[p].z : aw[p].w;
                        if (dbg & 2) {
                            mf_keep(word);
#pragma unroll
                            for (int s = 0; s < NS; ++s) mf_keep(b_cur[s]);
                        }
                        if (__builtin_amdgcn_ballot_w64(word != 0u) != 0ull && !(dbg & 2)) {   // (a piece without members is skipped)
                            v4i a;
                            if (dbg & 1024) {                        // (diagnostic: what a FREE bit -> i8 expansion would give)
                                a[0] = a[1] = a[2] = a[3] = static_cast<int>(word);
                            } else {
                                a[0] = static_cast<int>(expand4(word, 16 * h));
                                a[1] = static_cast<int>(expand4(word, 16 * h + 4));
                                a[2] = static_cast<int>(expand4(word, 16 * h + 8));
                                a[3] = static_cast<int>(expand4(word, 16 * h + 12));
                            }
#pragma unroll
                            for (int s = 0; s < NS; ++s) acc[p][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_cur[s], acc[p][s], 0, 0, 0);
                        }
                    }
                    if (gth && more1 && !(dbg & 1)) store_quarter(L_store, k, buf ^ 1);
                    if ((dbg & 1) && k == 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) mf_keep(L_store[i]);
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) b_cur[s] = b_nxt[s];
                }
                uint4 aw_next[2];
                split_rows(aw_raw, aw_next);

                if (dbg & 512) c2 = __builtin_amdgcn_s_memtime();
                if (t == S - 1 && !(dbg & 8)) {                      // the scores of permutation q are complete
                    uint32_t open = 0;                                // outputs the test leaves undecided
                    const int32_t wcs = static_cast<int32_t>(wc);
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {                  // sixteen outputs (four counter words) at a time: their thresholds are read together
                        int32_t y0v[4][4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int f = 0; f < 4; ++f) y0v[jj][f] = y0s[(16 * (f >> 1) + 4 * jb + jj + 8 * (f & 1)) * 256];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const int j = 4 * jb + jj;
                            uint32_t inc = 0;
#pragma unroll
                            for (int f = 0; f < 4; ++f) {
                                const int p = f >> 1, r = j + 8 * (f & 1);
                                const int32_t y = acc[p][2][r] * 4096 + acc[p][1][r] * 16 + (acc[p][0][r] >> 4);   // floor(V_hi / 16)
                                const int32_t d = y - y0v[jj][f];
                                inc |= (d < wcs) ? (1u << (8 * f)) : 0u;               // not certainly greater (d < 0: certainly smaller)
                                open |= (static_cast<uint32_t>(d) < wc) ? (1u << (16 * p + r)) : 0u;
                            }
                            if (inc) __hip_atomic_fetch_add(&cnts[j * 256], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (__builtin_expect(open != 0u, 0)) {
                        // rare (~1e-5 of the compares): the resolve kernel forms the exact score from all six digits.  (Nothing here
                        // touches the accumulators: a per-lane loop that indexed them dynamically moved all 96 of them to scratch
                        // memory, unrolled copies cost the main loop its registers.)
                        for (uint32_t left = open; left;) {
                            const int o = __builtin_ctz(left);
                            left &= left - 1u;
                            const int64_t u = u_lane + 32 * (o >> 4) + (o & 3) + 8 * ((o & 15) >> 2);
                            atomicAdd(&gl_counts[colf * n_padr + u], 0xFFFF0000u);       // it was counted "smaller" above: taken back
                            const unsigned int at = atomicAdd(fa.amb_count, 1u);
                            if (at < fa.amb_cap)
                                fa.amb[at] = make_ulonglong2(static_cast<unsigned long long>(u) | (static_cast<unsigned long long>(colf) << 32),
                                                             (1ull << 63) | static_cast<unsigned long long>(fa.p_base + q));
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[p][s][r] = 0;
                    if (++since_flush == 255) {
                        flush();
                        since_flush = 0;
                    }
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) aw[p] = aw_next[p];
                if (dbg & 512) {
                    mf_keep(aw[0]);                                  // (the wait for the membership words belongs to this phase)
                    mf_keep(aw[1]);
                    c3 = __builtin_amdgcn_s_memtime();
                }
                if (!(dbg & 4)) __syncthreads();
                if (dbg & 512) {
                    const unsigned long long c4 = __builtin_amdgcn_s_memtime();
                    prof_acc[0] += c1 - c0;                          // issue of the look-ahead loads
                    prof_acc[1] += c2 - c1;                          // operand reads, MFMAs, transposed stores
                    prof_acc[2] += c3 - c2;                          // score completion (+ the wait for the next membership words)
                    prof_acc[3] += c4 - c3;                          // barrier
                    prof_acc[4] += 1;
                }
                q = q1, t = t1;
                q1 = q2, t1 = t2;
                q2 = q3, t2 = t3;
            };
            for (int it = 0; it < total; it += 2) {
                body(it, L_a, L_b, src_a, src_b);
                if (it + 1 < total) body(it + 1, L_b, L_a, src_b, src_a);
            }
            if ((dbg & 512) && lane == 0 && fa.prof) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    atomicAdd(&fa.prof[wave * 8 + i], prof_acc[i]);
                    prof_acc[i] = 0;
                }
            }
            if (since_flush) flush();
            if (dbg & 8) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mf_keep(static_cast<uint32_t>(acc[p][s][r]));
            }
            __syncthreads();                                         // kb_list / buffers / thresholds are reused by the next task
        }
    }
}


// ---------------------------------------------------------------------------------------
// k_permtest_mfma_g (round 6): k_permtest_mfma_f's shape -- four waves of 64 rows, two workgroups per CU, thresholds and 8-bit
// counters in LDS -- with three changes that take instructions and waits out of the super-step:
//  * the gather is LDS-DMA (global_load_lds_dwordx4).  A super-step's gathered rows lie in LDS as they are in memory,
//    [k-step][32 rows][96 bytes]; the DMA's destination is wave-uniform base + lane x 16, so the 192 16-byte chunks of a
//    k-step are three wave-instructions: wave w stages k-step w, lane l of instruction j the chunk 64 j + l = row (64 j + l) / 6,
//    bytes 16 ((64 j + l) % 6) .. -- six adjacent lanes read one 96-byte row (1-2 cache lines; a first form with one row per
//    lane and [plane][row][16 B] in LDS touched 64 lines per instruction and spent 530 cycles per super-step issuing them).
//    No staging registers (k_permtest_mfma_f: 32), no ds_write, no wait for gathered rows inside the k-loop, all four waves
//    take part.  The DMAs of super-step it + 1 are issued at the top of super-step it (their buffer was last read in it - 1:
//    the barrier in between orders that) and have the whole super-step to land; the one vmcnt(0) is in front of the barrier.
//  * the membership words come bit-permuted (bs_bits4p): operand register j of lane half h = (word >> (4 h + j)) & 0x01010101 --
//    8 VALU per 32 x 32 piece instead of 12 (bit-field extract, multiply, mask per four bits).
//  * score completion: thresholds are stored as Y0 + W, so "not certainly greater" is the SIGN of d' = y - (Y0 + W); the four
//    signs of a counter word are collected with one v_alignbit each (top bytes side by side, one AND + shift per word), and
//    "undecided" (-W <= d' < 0) is detected for the lane's 32 outputs at once from the unsigned maximum of d' (v_max3_u32);
//    the per-output mask is only formed in the rare wave that has one (2.5 % of the wave-permutations at configs[4]).
// DBG (diagnostic builds only): 2 no MFMAs (nor expansions), 4 no barrier, 8 no score completion, 64 no DMA, 512 per-phase cycle
// counters, 1024 the operand registers without their expansion (the word itself: what a FREE bit -> i8 expansion would give).
template <int DBG>
__global__ __launch_bounds__(256, 2) void k_permtest_mfma_g(
    const unsigned char *__restrict__ bs, int64_t tile_bytes, const int32_t *__restrict__ srcp, int64_t n_src, int n_q,
    const int32_t *__restrict__ blk_ptr, const int32_t *__restrict__ blk_kb, const uint4 *__restrict__ blk_bits4p,
    const int32_t *__restrict__ grp_maxcnt, const int2 *__restrict__ tasks, const int32_t *__restrict__ q_off,
    unsigned int *__restrict__ q_ctr, int64_t mloc, unsigned int *__restrict__ gl_counts, int64_t n_padr, MfmaFilt fa) {
    constexpr int dbg = DBG;
    constexpr int NS = MF_NS / 2, KS = 32 * NS * 32, BUF = 4 * KS;            // k-step = 32 rows x 96 B, buffer = one super-step = 12 KB
    constexpr int64_t row_bytes = NS * 32;
    // ONE shared array (a second __shared__ object makes the compiler drain the DMA queue before every LDS read):
    // [2][BUF] | kb list | Y0 + W [32][256] | counters [8][256] | task slot
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lam = lane & 31, h = lane >> 5;
    int32_t *kb_list = reinterpret_cast<int32_t *>(lds + 2 * BUF);
    int32_t *y0s = reinterpret_cast<int32_t *>(lds + 2 * BUF + MF_F_MAXBLK * sizeof(int32_t)) + tid;
    uint32_t *cnts = reinterpret_cast<uint32_t *>(lds + 2 * BUF + MF_F_MAXBLK * sizeof(int32_t) + 32 * 256 * sizeof(int32_t)) + tid;
    int *slot_box = reinterpret_cast<int *>(lds + 2 * BUF + MF_F_MAXBLK * sizeof(int32_t) + 32 * 256 * sizeof(int32_t) + 8 * 256 * sizeof(uint32_t));
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte *)lds));      // LDS byte address of the array
    // staging role: k-step `wave` of the super-step; instruction j: chunk 64 j + lane = row r_g[j] of the block, bytes 16 c_g[j] ..
    const int r_g[3] = {lane / 6, (64 + lane) / 6, (128 + lane) / 6};
    const int c16_g[3] = {16 * (lane % 6), 16 * ((64 + lane) % 6), 16 * ((128 + lane) % 6)};
    const int dma_base = __builtin_amdgcn_readfirstlane(wave * KS);
    // operand read (ds_read_b64_tr_b8): lane l of a 16-lane group supplies piece l = row (l >> 1) of eight, bytes 8 (l & 1) .. of a
    // 16-byte column half; the group (lane >> 4) & 1 is the column half, the lane half h the k rows 16 h ..; lane l then owns
    // column l & 31 of the tile
    const uint32_t r_base = static_cast<uint32_t>((16 * h + ((lane & 15) >> 1)) * 96 + 16 * ((lane >> 4) & 1) + 8 * (lane & 1));
    const uint32_t sh0 = 4u * h, sh1 = sh0 + 1u, sh2 = sh0 + 2u, sh3 = sh0 + 3u;

    const int home = blockIdx.x & 7;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const int qx = (home + attempt) & 7;
        const int q_begin = q_off[qx], q_len = q_off[qx + 1] - q_begin;
        for (;;) {
            if (tid == 0) *slot_box = static_cast<int>(atomicAdd(&q_ctr[qx], 1u));
            __syncthreads();
            const int slot = *slot_box;
            __syncthreads();
            if (slot >= q_len) break;
            const int2 task = tasks[q_begin + slot];
            const int g = task.x, ct = task.y;
            const int b0 = blk_ptr[g], nb = blk_ptr[g + 1] - b0, S = nb >> 2;
            if (S == 0) continue;
            for (int i = tid; i < nb; i += 256) kb_list[i] = blk_kb[b0 + i];

            const unsigned char *bs_ct = bs + static_cast<int64_t>(ct) * tile_bytes;
            const uint4 *bits_w = blk_bits4p + static_cast<int64_t>(b0 >> 2) * MF_R + wave * 64 + lane;   // lane l: row l of the wave's 64
            const int total = n_q * S;
            const int64_t colf = static_cast<int64_t>(ct) * 32 + lam;
            const int64_t u_lane = static_cast<int64_t>(g) * MF_R + wave * 64 + 4 * h;     // + 32 p + (r & 3) + 8 (r >> 2)

            // thresholds of this lane's 32 outputs, stored as Y0 + W (32-bit, in LDS); the window width is one number per task
            const long long b_max = static_cast<long long>(grp_maxcnt[g]) * MF_LO_MAX;
            const uint32_t wc = static_cast<uint32_t>((2 * b_max + (15ll << 24)) >> 28) + 2u;
            const int32_t wcs = static_cast<int32_t>(wc);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                long long o64[16];
                int32_t members[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t u = u_lane + 32 * p + (r & 3) + 8 * (r >> 2);
                    members[r] = fa.rowcnt[u];
                    o64[r] = colf < mloc ? fa.obs64[colf * n_padr + u] : 0ll;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long bp = static_cast<long long>(members[r]) * MF_LO_MAX + (15ll << 24);
                    // padding rows (-1 members) and padding columns only ever form y = 0: Y0 = 1 counts them "smaller" without any
                    // further test (their counters are never read); an EMPTY neighborhood of a real row takes the general rule
                    y0s[(16 * p + r) * 256] = ((colf < mloc && members[r] >= 0) ? static_cast<int32_t>((o64[r] - bp) >> 28) : 1) + wcs;
                }
            }
            __syncthreads();                                         // kb_list

            v16i acc[2][NS];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][s][r] = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) cnts[j * 256] = 0;
            int since_flush = 0;
            unsigned long long prof_acc[5] = {0, 0, 0, 0, 0};

            auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t w = cnts[j * 256];
                    cnts[j * 256] = 0;
                    if (colf < mloc) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) {                    // byte f: piece f >> 1, accumulator element j + 8 (f & 1)
                            const uint32_t smaller = (w >> (8 * f)) & 0xFFu;
                            const int r = j + 8 * (f & 1);
                            atomicAdd(&gl_counts[colf * n_padr + u_lane + 32 * (f >> 1) + (r & 3) + 8 * (r >> 2)],
                                      (smaller << 16) | (static_cast<uint32_t>(since_flush) - smaller));
                        }
                    }
                }
            };
            struct Src3 {
                int32_t a, b, c;
            };
            auto src_of = [&](int q, int kb) -> Src3 {
                q = q < n_q ? q : n_q - 1;
                const int32_t *at = srcp + static_cast<int64_t>(q) * n_src + static_cast<int64_t>(kb) * 32;
                return Src3{at[r_g[0]], at[r_g[1]], at[r_g[2]]};
            };
            // The three DMA instructions of this thread: chunks 64 j + lane (j = 0..2) of k-step `wave` of the super-step in buffer `buf`.
            // Written as asm: through the builtin the compiler drains the DMA queue (vmcnt(0)) before the next LDS read it cannot
            // prove disjoint from the destination -- right after the issue.  The DMAs are therefore not in the compiler's vmcnt
            // bookkeeping: every counted load whose wait follows them is younger (so that wait covers them), and the explicit
            // vmcnt(0) in front of the super-step's barrier is what publishes the rows.  ONE statement: three statements (each
            // saving and restoring m0, each a scheduling fence) measured 10.5 against 9.6 ms per launch.
            auto stage = [&](const Src3 &src, int buf) __attribute__((always_inline)) {
                if (dbg & 64) return;
                const unsigned char *from = bs_ct + static_cast<int64_t>(src.a) * row_bytes + c16_g[0];
                const unsigned char *from1 = bs_ct + static_cast<int64_t>(src.b) * row_bytes + c16_g[1];
                const unsigned char *from2 = bs_ct + static_cast<int64_t>(src.c) * row_bytes + c16_g[2];
                const uint32_t to = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lds_base + buf * BUF + dma_base)));
                uint32_t keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %4\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\t"
                    "s_add_u32 m0, %4, 0x400\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %2, off\n\t"
                    "s_add_u32 m0, %4, 0x800\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %3, off\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(from), "v"(from1), "v"(from2), "s"(to)
                    : "memory", "scc");
            };
            auto advance = [&](int &qq, int &tt) {
                if (++tt == S) {
                    tt = 0;
                    ++qq;
                }
            };
            // membership words of a super-step: lane l loads the four words of row l; v_permlane32_swap then gives every lane the
            // words of row (l & 31) of piece 0 and of piece 1 (one 16-byte load per lane instead of two)
            auto split_rows = [&](const uint4 &raw, uint4 (&w)[2]) __attribute__((always_inline)) {
                typedef unsigned int v2u __attribute__((ext_vector_type(2)));
                const v2u x = __builtin_amdgcn_permlane32_swap(raw.x, raw.x, false, false);
                const v2u y = __builtin_amdgcn_permlane32_swap(raw.y, raw.y, false, false);
                const v2u z = __builtin_amdgcn_permlane32_swap(raw.z, raw.z, false, false);
                const v2u ww = __builtin_amdgcn_permlane32_swap(raw.w, raw.w, false, false);
                w[0] = make_uint4(x[0], y[0], z[0], ww[0]);
                w[1] = make_uint4(x[1], y[1], z[1], ww[1]);
            };

            // pipeline: super-step it computes from buffer it & 1; at its top the rows of it + 1 are handed to the DMA (source
            // index loaded during it - 1), the source index of it + 2 and the membership words of it + 1 are requested
            int q1 = 0, t1 = 0, q2, t2;
            advance(q1, t1);
            q2 = q1, t2 = t1;
            advance(q2, t2);
            stage(src_of(0, kb_list[wave]), 0);
            Src3 src_nx = src_of(q1, kb_list[4 * t1 + wave]);
            int kb_next = kb_list[4 * t2 + wave];
            int32_t y0r[32];                                         // the thresholds, read at the top of a permutation's last super-step
#pragma unroll
            for (int o = 0; o < 32; ++o) y0r[o] = 0;
            uint4 aw[2];
            split_rows(bits_w[0], aw);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            int q = 0, t = 0;
            for (int it = 0; it < total; ++it) {
                const int buf = it & 1;
                int q3 = q2, t3 = t2;
                advance(q3, t3);
                unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                if (dbg & 512) c0 = __builtin_amdgcn_s_memtime();
                // The three DMAs of super-step it + 1.  A wave spends ~570 cycles of the super-step's ~2100 getting them accepted (the
                // address unit takes a 64-lane gather at about a lane per cycle, and the CU's eight waves issue 24 of them per
                // super-step); issued one per k-step instead they cost the k-loop what they save here (measured: 9.75 against
                // 9.60 ms per launch).  Unconditional: behind the task's last super-step the clamped index stages rows nobody
                // reads -- a branch would put a vmcnt(0) where the paths meet again.
                stage(src_nx, buf ^ 1);
                src_nx = src_of(q2, kb_next);
                kb_next = kb_list[4 * t3 + wave];                        // (consumed at the top of the next iteration)
                if (t == S - 1 && !(dbg & 8)) {                          // (their LDS latency passes under the k-loop)
#pragma unroll
                    for (int o = 0; o < 32; ++o) y0r[o] = y0s[o * 256];
                }
                const uint4 aw_raw = bits_w[static_cast<int64_t>(t1) * MF_R];

                const unsigned char *bbuf = lds + buf * BUF + r_base;
                if (dbg & 512) c1 = __builtin_amdgcn_s_memtime();
                // two operand sets: the slices of k-step k + 1 are read before the MFMAs of k-step k are issued
                v4i b_cur[NS], b_nxt[NS];
                auto read_operand = [&](int k, int s) -> v4i {
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(3))) v2i lds_v2i;
                    const unsigned char *at = bbuf + k * KS + s * 32;
                    const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at));            // k rows 16 h + 0..7
                    const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at + 8 * 96));   // k rows 16 h + 8..15
                    v4i r;
                    r[0] = lo[0], r[1] = lo[1], r[2] = hi[0], r[3] = hi[1];
                    return r;
                };
#pragma unroll
                for (int s = 0; s < NS; ++s) b_cur[s] = read_operand(0, s);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < 3) {
#pragma unroll
                        for (int s = 0; s < NS; ++s) b_nxt[s] = read_operand(k + 1, s);
                    }
                    // (skipping the reads of a k-step whose two pieces hold no member measured slower, 9.87 against 9.61 ms per
                    // launch: two more wave-level branches per k-step cost more than the LDS reads they save)
                    __builtin_amdgcn_sched_barrier(0);               // the reads stay ahead of this k-step's MFMAs
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const uint32_t word = k == 0 ? aw[p].x : k == 1 ? aw[p].y : k == 2 ? aw[p].z : aw[p].w;
                        if (dbg & 2) {
                            mf_keep(word);
#pragma unroll
                            for (int s = 0; s < NS; ++s) mf_keep(b_cur[s]);
                        }
                        if (__builtin_amdgcn_ballot_w64(word != 0u) != 0ull && !(dbg & 2)) {   // (a piece without members is skipped)
                            v4i a;
                            if (dbg & 1024) {
                                a[0] = a[1] = a[2] = a[3] = static_cast<int>(word);
                            } else {
                                a[0] = static_cast<int>((word >> sh0) & 0x01010101u);
                                a[1] = static_cast<int>((word >> sh1) & 0x01010101u);
                                a[2] = static_cast<int>((word >> sh2) & 0x01010101u);
                                a[3] = static_cast<int>((word >> sh3) & 0x01010101u);
                            }
#pragma unroll
                            for (int s = 0; s < NS; ++s) acc[p][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_cur[s], acc[p][s], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) b_cur[s] = b_nxt[s];
                }

                if (dbg & 512) c2 = __builtin_amdgcn_s_memtime();
                if (t == S - 1 && !(dbg & 8)) {                      // the scores of permutation q are complete
                    uint32_t mx = 0u;                                 // unsigned maximum of d' = y - (Y0 + W) over the lane's outputs
                    {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            uint32_t tops = 0u;                       // the top bytes of the four d' side by side: their sign bits are the increments
#pragma unroll
                            for (int f = 3; f >= 0; --f) {
                                const int p = f >> 1, r = j + 8 * (f & 1);
                                const int32_t y = acc[p][2][r] * 4096 + acc[p][1][r] * 16 + (acc[p][0][r] >> 4);   // floor(V_hi / 16)
                                const uint32_t d = static_cast<uint32_t>(y - y0r[16 * p + r]);
                                tops = __builtin_amdgcn_alignbit(tops, d, 24);
                                mx = d > mx ? d : mx;
                            }
                            // d' < 0: not certainly greater (counted "smaller"; an undecided one is taken back below)
                            __hip_atomic_fetch_add(&cnts[j * 256], (tops >> 7) & 0x01010101u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (__builtin_expect(~mx < wc, 0)) {
                        // rare (~1e-5 of the compares): -W <= d' < 0 for one of this lane's outputs -- find them (the thresholds again)
                        // and hand them to the resolve kernel, which forms the exact score from all six digits
                        uint32_t open = 0;
#pragma unroll
                        for (int o = 0; o < 32; ++o) {
                            const int p = o >> 4, r = o & 15;
                            const int32_t y = acc[p][2][r] * 4096 + acc[p][1][r] * 16 + (acc[p][0][r] >> 4);
                            const uint32_t d = static_cast<uint32_t>(y - y0r[o]);
                            open |= (~d < wc) ? (1u << o) : 0u;
                        }
                        for (uint32_t left = open; left;) {
                            const int o = __builtin_ctz(left);
                            left &= left - 1u;
                            const int64_t u = u_lane + 32 * (o >> 4) + (o & 3) + 8 * ((o & 15) >> 2);
                            atomicAdd(&gl_counts[colf * n_padr + u], 0xFFFF0000u);       // it was counted "smaller" above: taken back
                            const unsigned int at = atomicAdd(fa.amb_count, 1u);
                            if (at < fa.amb_cap)
                                fa.amb[at] = make_ulonglong2(static_cast<unsigned long long>(u) | (static_cast<unsigned long long>(colf) << 32),
                                                             (1ull << 63) | static_cast<unsigned long long>(fa.p_base + q));
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[p][s][r] = 0;
                    if (++since_flush == 255) {
                        flush();
                        since_flush = 0;
                    }
                }
                split_rows(aw_raw, aw);
                if (dbg & 512) {
                    mf_keep(aw[0]);                                  // (the wait for the membership words belongs to this phase)
                    mf_keep(aw[1]);
                    c3 = __builtin_amdgcn_s_memtime();
                }
                if (!(dbg & 4)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next super-step's rows has landed
                    __syncthreads();
                }
                if (dbg & 512) {
                    const unsigned long long c4 = __builtin_amdgcn_s_memtime();
                    prof_acc[0] += c1 - c0;                          // DMA issue, look-ahead loads
                    prof_acc[1] += c2 - c1;                          // operand reads, expansions, MFMAs
                    prof_acc[2] += c3 - c2;                          // score completion (+ the wait for the next membership words)
                    prof_acc[3] += c4 - c3;                          // DMA landed + barrier
                    prof_acc[4] += 1;
                }
                q = q1, t = t1;
                q1 = q2, t1 = t2;
                q2 = q3, t2 = t3;
            }
            if ((dbg & 512) && lane == 0 && fa.prof) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    atomicAdd(&fa.prof[wave * 8 + i], prof_acc[i]);
                    prof_acc[i] = 0;
                }
            }
            if (since_flush) flush();
            if (dbg & 8) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mf_keep(static_cast<uint32_t>(acc[p][s][r]));
            }
            __syncthreads();                                         // kb_list / buffers / thresholds are reused by the next task
        }
    }
}


// ---------------------------------------------------------------------------------------
// k_permtest_mfma_gz (round 6): the FILTERED Z-SCORE test (k_permtest_mfma's FM = 2, Z) in k_permtest_mfma_g's shape -- four waves
// of 64 rows (two 32 x 32 pieces each: an operand read feeds two MFMAs), two workgroups per CU, LDS-DMA gather, bit-permuted
// membership words.  A task is a tile of SIXTEEN attributes: bytes 0-15 of a slice row are value digits, bytes 16-31 the digits of
// the squares; slices 0-2 = the three HIGH digits, slice 3 = the not-NaN flags (under both halves): 128-byte rows, eight adjacent
// lanes of a DMA instruction read one row, four instructions per wave and super-step (wave w stages k-step w).
// The test is the general kernel's, operation for operation (the decision rules and their margins are the ones the parity tests
// pin): S1 = a + low, S2 = b + low with |low| <= count * MF_LO_MAX; z >= o is a statement about the signs of S1 and of
// T = S1^2 sc1^2 (1 + o^2) - o^2 S2 sc2 count; what the high parts leave open goes to k_mfma_resolve_z.  The value lane of an
// attribute (tile column c) and its squares' lane (column c + 16 = lane ^ 16) split the work: the value lane tests accumulator rows
// 0..7 of a piece, the squares' lane rows 8..15, after one exchange per row pair.  Observed scores (f64) sit in LDS, [16 outputs]
// [thread]; counters are 8-bit LDS fields (#certainly greater | #certainly smaller, two outputs per word), flushed every 255
// permutations as (#>= << 16 | #<=) atomics -- the f64 kernels' counter form, which k_counts_finalize<true> reads with the
// observed scores.
// DBG (diagnostic builds only): 2 no MFMAs, 4 no barrier, 8 no score completion, 64 no DMA, 512 per-phase cycle counters.
constexpr int MF_GZ_MAXBLK = 1024;        // column blocks per row group this kernel can index from LDS (else the general kernel)
// F32 (default): the same decision in single precision.  In units of 2^24 x the value grid, u = the high value sum, w = the high
// sum of squares, c = count, d = c x MF_LO_MAX / 2^24 >= |low parts|:
//     T' = u^2 - rho gamma w c,   gamma = sc2 / (sc1^2 2^24) (a power of two),   rho = o^2 / (1 + o^2)   (T = T' x a positive factor)
//     |T' - its true value| <= d (2 |u| + d) + rho gamma c d     (the low parts)   +   2^-21 (u^2 + rho gamma |w| c)   (f32 roundings)
// decided when |T'| exceeds (1 + 2^-10) x the first bound + 2^-19 x the second -- margins a superset of the f64 form's (which
// already include the reference's own roundings), so every compare this form decides the f64 form decides the same way, and the
// counters stay equal to the seven-slice kernel's; per output the workgroup keeps copysign(rho, o) as f32 in LDS (NaN = no test).
// 16 outputs cost ~800 VALU instructions instead of ~2850 (f64 products, 64-bit sums and two exchanges per output).
template <int DBG, bool F32 = true>
__global__ __launch_bounds__(256, 2) void k_permtest_mfma_gz(
    const unsigned char *__restrict__ bs, int64_t tile_bytes, const int32_t *__restrict__ srcp, int64_t n_src, int n_q,
    const int32_t *__restrict__ blk_ptr, const int32_t *__restrict__ blk_kb, const uint4 *__restrict__ blk_bits4p,
    const int2 *__restrict__ tasks, const int32_t *__restrict__ q_off, unsigned int *__restrict__ q_ctr, int64_t mloc,
    unsigned int *__restrict__ gl_counts, int64_t n_padr, const int32_t *__restrict__ rowmap, const double *__restrict__ col_scale,
    MfmaFilt fa) {
    constexpr int dbg = DBG;
    constexpr int NS = MF_NS / 2 + 1, KS = 32 * NS * 32, BUF = 4 * KS;        // k-step = 32 rows x 128 B, buffer = one super-step = 16 KB
    constexpr int64_t row_bytes = NS * 32;
    // ONE shared array: [2][BUF] | kb list | observed scores f64 [16][256] | counters [8][256] | task slot
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int lam = lane & 31, h = lane >> 5;
    int32_t *kb_list = reinterpret_cast<int32_t *>(lds + 2 * BUF);
    long long *obs = reinterpret_cast<long long *>(lds + 2 * BUF + MF_GZ_MAXBLK * sizeof(int32_t)) + tid;
    float *rhos = reinterpret_cast<float *>(lds + 2 * BUF + MF_GZ_MAXBLK * sizeof(int32_t)) + tid;      // (F32: the same space, [16][256] floats)
    uint32_t *cnts = reinterpret_cast<uint32_t *>(lds + 2 * BUF + MF_GZ_MAXBLK * sizeof(int32_t) + 16 * 256 * sizeof(long long)) + tid;
    int *slot_box = reinterpret_cast<int *>(lds + 2 * BUF + MF_GZ_MAXBLK * sizeof(int32_t) + 16 * 256 * sizeof(long long) + 8 * 256 * sizeof(uint32_t));
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte *)lds));      // LDS byte address of the array
    // staging role: k-step `wave` of the super-step; instruction j: chunk 64 j + lane = row 8 j + (lane >> 3), bytes 16 (lane & 7) ..
    const int r_g0 = lane >> 3, c16_g = 16 * (lane & 7);
    const int dma_base = __builtin_amdgcn_readfirstlane(wave * KS);
    const uint32_t r_base = static_cast<uint32_t>((16 * h + ((lane & 15) >> 1)) * 128 + 16 * ((lane >> 4) & 1) + 8 * (lane & 1));
    const uint32_t sh0 = 4u * h, sh1 = sh0 + 1u, sh2 = sh0 + 2u, sh3 = sh0 + 3u;
    const bool sq = (lam & 16) != 0;                                          // this lane owns a column of squares
    const int r_off = sq ? 8 : 0;
    const uint32_t sq_mask = sq ? 0xFFFFFFFFu : 0u;

    const int home = blockIdx.x & 7;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const int qx = (home + attempt) & 7;
        const int q_begin = q_off[qx], q_len = q_off[qx + 1] - q_begin;
        for (;;) {
            if (tid == 0) *slot_box = static_cast<int>(atomicAdd(&q_ctr[qx], 1u));
            __syncthreads();
            const int slot = *slot_box;
            __syncthreads();
            if (slot >= q_len) break;
            const int2 task = tasks[q_begin + slot];
            const int g = task.x, ct = task.y;
            const int b0 = blk_ptr[g], nb = blk_ptr[g + 1] - b0, S = nb >> 2;
            if (S == 0) continue;
            for (int i = tid; i < nb; i += 256) kb_list[i] = blk_kb[b0 + i];

            const unsigned char *bs_ct = bs + static_cast<int64_t>(ct) * tile_bytes + c16_g;
            const uint4 *bits_w = blk_bits4p + static_cast<int64_t>(b0 >> 2) * MF_R + wave * 64 + lane;   // lane l: row l of the wave's 64
            const int total = n_q * S;
            const int64_t colz = static_cast<int64_t>(ct) * 16 + (lam & 15);
            const int64_t u_lane = static_cast<int64_t>(g) * MF_R + wave * 64 + 4 * h;     // + 32 p + (r & 3) + 8 (r >> 2)
            double sc1 = 1.0, sc2 = 1.0;
            if (colz < mloc) {
                sc1 = col_scale[colz];
                sc2 = col_scale[mloc + colz];
            }
            const double sc1sq = sc1 * sc1;
            // the observed scores of this lane's 16 outputs (8 per piece: value lanes accumulator rows 0..7, squares' lanes 8..15);
            // NaN (padding, fewer than three values, no spread) = no test
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = rr + r_off;
                    const int32_t node = rowmap[u_lane + 32 * p + (r & 3) + 8 * (r >> 2)];
                    const bool live = node >= 0 && colz < mloc;
                    const double o = live ? fa.zobs[static_cast<int64_t>(node) * mloc + colz] : __longlong_as_double(0x7FF8000000000000ll);
                    if constexpr (F32) {
                        const double o2 = o * o;
                        const double rho = o2 > 0x1p1000 ? 1.0 : o2 / (1.0 + o2);            // (o = +-inf: the limit)
                        rhos[(8 * p + rr) * 256] = o == o ? copysignf(static_cast<float>(rho), o < 0.0 ? -1.0f : 1.0f) : __int_as_float(0x7FC00000);
                    } else {
                        obs[(8 * p + rr) * 256] = __double_as_longlong(o);
                    }
                }
            const float gamma = static_cast<float>(sc2 / (sc1sq * 16777216.0));            // (powers of two: exact)
            __syncthreads();                                         // kb_list

            v16i acc[2][NS];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][s][r] = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) cnts[j * 256] = 0;
            int since_flush = 0;
            unsigned long long prof_acc[5] = {0, 0, 0, 0, 0};

            // counter word j: outputs 2 j and 2 j + 1 (output o = 8 p + rr), bytes {#greater, #smaller} of each
            auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t w = cnts[j * 256];
                    cnts[j * 256] = 0;
                    if (colz < mloc && w) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int o = 2 * j + e, r = (o & 7) + r_off;
                            const uint32_t add = (((w >> (16 * e)) & 0xFFu) << 16) | ((w >> (16 * e + 8)) & 0xFFu);
                            if (add) atomicAdd(&gl_counts[colz * n_padr + u_lane + 32 * (o >> 3) + (r & 3) + 8 * (r >> 2)], add);
                        }
                    }
                }
            };
            struct Src4 {
                int32_t a, b, c, d;
            };
            auto src_of = [&](int q, int kb) -> Src4 {
                q = q < n_q ? q : n_q - 1;
                const int32_t *at = srcp + static_cast<int64_t>(q) * n_src + static_cast<int64_t>(kb) * 32 + r_g0;
                return Src4{at[0], at[8], at[16], at[24]};
            };
            // (one asm statement: see k_permtest_mfma_g)
            auto stage = [&](const Src4 &src, int buf) __attribute__((always_inline)) {
                if (dbg & 64) return;
                const unsigned char *from = bs_ct + static_cast<int64_t>(src.a) * row_bytes;
                const unsigned char *from1 = bs_ct + static_cast<int64_t>(src.b) * row_bytes;
                const unsigned char *from2 = bs_ct + static_cast<int64_t>(src.c) * row_bytes;
                const unsigned char *from3 = bs_ct + static_cast<int64_t>(src.d) * row_bytes;
                const uint32_t to = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lds_base + buf * BUF + dma_base)));
                uint32_t keep;
                asm volatile(
                    "s_mov_b32 %0, m0\n\t"
                    "s_mov_b32 m0, %5\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %1, off\n\t"
                    "s_add_u32 m0, %5, 0x400\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %2, off\n\t"
                    "s_add_u32 m0, %5, 0x800\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %3, off\n\t"
                    "s_add_u32 m0, %5, 0xc00\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %4, off\n\t"
                    "s_mov_b32 m0, %0"
                    : "=&s"(keep)
                    : "v"(from), "v"(from1), "v"(from2), "v"(from3), "s"(to)
                    : "memory", "scc");
            };
            auto advance = [&](int &qq, int &tt) {
                if (++tt == S) {
                    tt = 0;
                    ++qq;
                }
            };
            auto split_rows = [&](const uint4 &raw, uint4 (&w)[2]) __attribute__((always_inline)) {
                typedef unsigned int v2u __attribute__((ext_vector_type(2)));
                const v2u x = __builtin_amdgcn_permlane32_swap(raw.x, raw.x, false, false);
                const v2u y = __builtin_amdgcn_permlane32_swap(raw.y, raw.y, false, false);
                const v2u z = __builtin_amdgcn_permlane32_swap(raw.z, raw.z, false, false);
                const v2u ww = __builtin_amdgcn_permlane32_swap(raw.w, raw.w, false, false);
                w[0] = make_uint4(x[0], y[0], z[0], ww[0]);
                w[1] = make_uint4(x[1], y[1], z[1], ww[1]);
            };

            int q1 = 0, t1 = 0, q2, t2;
            advance(q1, t1);
            q2 = q1, t2 = t1;
            advance(q2, t2);
            stage(src_of(0, kb_list[wave]), 0);
            Src4 src_nx = src_of(q1, kb_list[4 * t1 + wave]);
            int kb_next = kb_list[4 * t2 + wave];
            uint4 aw[2];
            split_rows(bits_w[0], aw);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            int q = 0, t = 0;
            for (int it = 0; it < total; ++it) {
                const int buf = it & 1;
                int q3 = q2, t3 = t2;
                advance(q3, t3);
                unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                if (dbg & 512) c0 = __builtin_amdgcn_s_memtime();
                stage(src_nx, buf ^ 1);                                  // (unconditional: see k_permtest_mfma_g)
                src_nx = src_of(q2, kb_next);
                kb_next = kb_list[4 * t3 + wave];
                const uint4 aw_raw = bits_w[static_cast<int64_t>(t1) * MF_R];

                const unsigned char *bbuf = lds + buf * BUF + r_base;
                if (dbg & 512) c1 = __builtin_amdgcn_s_memtime();
                v4i b_cur[NS], b_nxt[NS];
                auto read_operand = [&](int k, int s) -> v4i {
                    typedef int v2i __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(3))) v2i lds_v2i;
                    const unsigned char *at = bbuf + k * KS + s * 32;
                    const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at));             // k rows 16 h + 0..7
                    const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i *)(at + 8 * 128));   // k rows 16 h + 8..15
                    v4i r;
                    r[0] = lo[0], r[1] = lo[1], r[2] = hi[0], r[3] = hi[1];
                    return r;
                };
#pragma unroll
                for (int s = 0; s < NS; ++s) b_cur[s] = read_operand(0, s);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < 3) {
#pragma unroll
                        for (int s = 0; s < NS; ++s) b_nxt[s] = read_operand(k + 1, s);
                    }
                    __builtin_amdgcn_sched_barrier(0);               // the reads stay ahead of this k-step's MFMAs
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const uint32_t word = k == 0 ? aw[p].x : k == 1 ? aw[p].y : k == 2 ? aw[p].z : aw[p].w;
                        if (dbg & 2) {
                            mf_keep(word);
#pragma unroll
                            for (int s = 0; s < NS; ++s) mf_keep(b_cur[s]);
                        }
                        if (__builtin_amdgcn_ballot_w64(word != 0u) != 0ull && !(dbg & 2)) {   // (a piece without members is skipped)
                            v4i a;
                            a[0] = static_cast<int>((word >> sh0) & 0x01010101u);
                            a[1] = static_cast<int>((word >> sh1) & 0x01010101u);
                            a[2] = static_cast<int>((word >> sh2) & 0x01010101u);
                            a[3] = static_cast<int>((word >> sh3) & 0x01010101u);
#pragma unroll
                            for (int s = 0; s < NS; ++s) acc[p][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b_cur[s], acc[p][s], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) b_cur[s] = b_nxt[s];
                }

                if (dbg & 512) c2 = __builtin_amdgcn_s_memtime();
                if (t == S - 1 && !(dbg & 8)) {                      // the scores of permutation q are complete
                    // The filtered z-score test of k_permtest_mfma (FM = 2, Z), operation for operation.
                    uint32_t undecided = 0;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
#pragma unroll
                        for (int rp = 0; rp < 4; ++rp) {
                            uint32_t inc = 0;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int rr = 2 * rp + e;
                                if constexpr (F32) {
                                    auto hi_sum = [&](int r) -> float {           // acc2 x 65536 + acc1 x 256 + acc0 (each below 2^24: exact conversions)
                                        return fmaf(static_cast<float>(acc[p][2][r]), 65536.0f,
                                                    fmaf(static_cast<float>(acc[p][1][r]), 256.0f, static_cast<float>(acc[p][0][r])));
                                    };
                                    // (selects between the value lane's and the squares' lane's roles are bit selects with a per-lane mask,
                                    // the decision is mask logic without short-circuit branches: as written with ?: and && the compiler
                                    // re-derived the lane predicate for every select and branched around every clause -- 2700 instructions)
                                    const uint32_t m_lo = __float_as_uint(hi_sum(rr)), m_hi = __float_as_uint(hi_sum(rr + 8));
                                    const uint32_t got = static_cast<uint32_t>(__shfl_xor(static_cast<int>((m_lo & sq_mask) | (m_hi & ~sq_mask)), 16));
                                    const float u = __uint_as_float((got & sq_mask) | (m_lo & ~sq_mask));      // high sum of values of MY row
                                    const float w = __uint_as_float((m_hi & sq_mask) | (got & ~sq_mask));      // high sum of squares
                                    const float cf = static_cast<float>(static_cast<int>((static_cast<uint32_t>(acc[p][3][rr + 8]) & sq_mask) |
                                                                                         (static_cast<uint32_t>(acc[p][3][rr]) & ~sq_mask)));
                                    const float rs = rhos[(8 * p + rr) * 256], rho = fabsf(rs);
                                    const float d = cf * (static_cast<float>(MF_LO_MAX) / 16777216.0f * 1.000244140625f);   // >= the low parts, in units of 2^24
                                    const float au = fabsf(u), uu = u * u, gwc = (gamma * w) * cf, kwc = rho * gwc;
                                    const float slack_u = d * fmaf(2.0f, au, d), gcd = gamma * (cf * d);
                                    const float T = uu - kwc, V = gwc - uu;
                                    const float e_t = fmaf(0x1p-19f, uu + fabsf(kwc), fmaf(rho, gcd, slack_u) * 1.0009765625f);
                                    const float e_v = fmaf(0x1p-18f, uu + fabsf(gwc), (slack_u + gcd) * 1.0009765625f);
                                    const float m_s = fmaf(au, 0x1p-20f, d);
                                    // (observed NaN: no test; fewer than 3 values: the permuted score is NaN -- safe_extras.py:30)
                                    const bool live = (rs == rs) & (cf >= 3.0f), ok = live & (V > e_v);
                                    const bool t_pos = T > e_t, t_neg = -T > e_t, s_pos = u > m_s, s_neg = -u > m_s;
                                    const bool nn = __float_as_int(rs) >= 0;                                   // o >= 0
                                    const bool greater = ok & ((nn & s_pos & t_pos) | (!nn & (s_pos | t_neg)));
                                    const bool smaller = ok & ((nn & (s_neg | t_neg)) | (!nn & s_neg & t_pos));
                                    inc |= (greater ? (1u << (16 * e)) : 0u) | (smaller ? (1u << (16 * e + 8)) : 0u);
                                    undecided |= (live & !greater & !smaller) ? (1u << (8 * p + rr)) : 0u;
                                } else {
                                long long x = static_cast<long long>(acc[p][2][rr]), y = static_cast<long long>(acc[p][2][rr + 8]);
#pragma unroll
                                for (int s = 1; s >= 0; --s) {
                                    x = (x << 8) + static_cast<long long>(acc[p][s][rr]);
                                    y = (y << 8) + static_cast<long long>(acc[p][s][rr + 8]);
                                }
                                const long long give = sq ? x : y;
                                const int g_lo = __shfl_xor(static_cast<int>(give), 16), g_hi = __shfl_xor(static_cast<int>(give >> 32), 16);
                                const long long got = (static_cast<long long>(g_hi) << 32) | static_cast<long long>(static_cast<uint32_t>(g_lo));
                                const long long v = sq ? got : x, w = sq ? y : got;        // sums / sums of squares (high digits) of MY row
                                const double o = __longlong_as_double(obs[(8 * p + rr) * 256]);
                                const double members = static_cast<double>(sq ? acc[p][3][rr + 8] : acc[p][3][rr]);   // (the not-NaN slice is in both halves)
                                if (o == o && members >= 3.0) {                  // (observed NaN: no test; fewer than 3 values: the score is NaN -- safe_extras.py:30)
                                    const double a = static_cast<double>(v) * 16777216.0, b = static_cast<double>(w) * 16777216.0;
                                    const double eb = members * static_cast<double>(MF_LO_MAX);
                                    const double o2 = o * o, k1 = sc1sq * (1.0 + o2), k2 = o2 * sc2;
                                    const double slack1 = 2.0 * fabs(a) * eb + eb * eb;                      // |S1^2 - a^2| <=
                                    const double p1 = a * a * k1, p2 = b * members * k2;
                                    const double T = p1 - p2;
                                    const double E = k1 * slack1 + k2 * members * eb + (p1 + fabs(p2)) * 0x1p-40;
                                    const double V = b * members * sc2 - a * a * sc1sq;                      // count^2 * variance, high parts
                                    const bool var_ok = V - (sc2 * members * eb + sc1sq * slack1) > fabs(b) * members * sc2 * 0x1p-20;
                                    const bool t_pos = T > E, t_neg = T < -E, s_pos = a > eb, s_neg = a < -eb;
                                    const bool greater = var_ok && (o >= 0.0 ? (s_pos && t_pos) : (s_pos || t_neg));
                                    const bool smaller = var_ok && (o >= 0.0 ? (s_neg || t_neg) : (s_neg && t_pos));
                                    inc |= greater ? (1u << (16 * e)) : (smaller ? (1u << (16 * e + 8)) : 0u);
                                    undecided |= (!greater && !smaller) ? (1u << (8 * p + rr)) : 0u;
                                }
                            }
                            }
                            if (inc) __hip_atomic_fetch_add(&cnts[(4 * p + rp) * 256], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (__builtin_expect(undecided != 0u, 0)) {
                        for (uint32_t left = undecided; left;) {
                            const int o = __builtin_ctz(left), r = (o & 7) + r_off;
                            left &= left - 1u;
                            const unsigned int at = atomicAdd(fa.amb_count, 1u);
                            if (at >= fa.amb_cap) continue;
                            fa.amb[at] = make_ulonglong2(static_cast<unsigned long long>(u_lane + 32 * (o >> 3) + (r & 3) + 8 * (r >> 2)) |
                                                             (static_cast<unsigned long long>(colz) << 32),
                                                         (3ull << 62) | static_cast<unsigned long long>(fa.p_base + q));
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[p][s][r] = 0;
                    if (++since_flush == 255) {
                        flush();
                        since_flush = 0;
                    }
                }
                split_rows(aw_raw, aw);
                if (dbg & 512) {
                    mf_keep(aw[0]);
                    mf_keep(aw[1]);
                    c3 = __builtin_amdgcn_s_memtime();
                }
                if (!(dbg & 4)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next super-step's rows has landed
                    __syncthreads();
                }
                if (dbg & 512) {
                    const unsigned long long c4 = __builtin_amdgcn_s_memtime();
                    prof_acc[0] += c1 - c0;
                    prof_acc[1] += c2 - c1;
                    prof_acc[2] += c3 - c2;
                    prof_acc[3] += c4 - c3;
                    prof_acc[4] += 1;
                }
                q = q1, t = t1;
                q1 = q2, t1 = t2;
                q2 = q3, t2 = t3;
            }
            if ((dbg & 512) && lane == 0 && fa.prof) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    atomicAdd(&fa.prof[wave * 8 + i], prof_acc[i]);
                    prof_acc[i] = 0;
                }
            }
            if (since_flush) flush();
            if (dbg & 8) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mf_keep(static_cast<uint32_t>(acc[p][s][r]));
            }
            __syncthreads();                                         // kb_list / buffers / observed scores are reused by the next task
        }
    }
}

// ---------------------------------------------------------------------------------------
// host: node order, block structure
// ---------------------------------------------------------------------------------------
uint64_t hilbert_index(uint32_t x, uint32_t y, int bits) {
    uint64_t d = 0;
    for (uint32_t s = 1u << (bits - 1); s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += static_cast<uint64_t>(s) * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) {
                x = s - 1 - (x & (s - 1));
                y = s - 1 - (y & (s - 1));
            }
            std::swap(x, y);
        }
        x &= s - 1;
        y &= s - 1;
    }
    return d;
}

void order_by_layout(const std::vector<double> &xy, int64_t n, std::vector<int32_t> *order) {
    double x0 = xy[0], x1 = xy[0], y0 = xy[1], y1 = xy[1];
    for (int64_t i = 0; i < n; ++i) {
        x0 = std::min(x0, xy[2 * i]);
        x1 = std::max(x1, xy[2 * i]);
        y0 = std::min(y0, xy[2 * i + 1]);
        y1 = std::max(y1, xy[2 * i + 1]);
    }
    const double span = std::max(std::max(x1 - x0, y1 - y0), 1e-300);
    std::vector<uint64_t> key(n);
    for (int64_t i = 0; i < n; ++i) {
        const double fx = (xy[2 * i] - x0) / span, fy = (xy[2 * i + 1] - y0) / span;
        const uint32_t ix = static_cast<uint32_t>(std::min(65535.0, std::max(0.0, fx * 65535.0)));
        const uint32_t iy = static_cast<uint32_t>(std::min(65535.0, std::max(0.0, fy * 65535.0)));
        key[i] = (fx == fx && fy == fy) ? hilbert_index(ix, iy, 16) : ~0ull;
    }
    order->resize(n);
    std::iota(order->begin(), order->end(), 0);
    std::stable_sort(order->begin(), order->end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
}

// Cuthill-McKee (breadth first, neighbours by ascending degree) over the membership graph
void order_by_graph(const std::vector<int32_t> &row_ptr, const std::vector<int32_t> &col, int64_t n,
                    std::vector<int32_t> *order) {
    order->clear();
    order->reserve(n);
    std::vector<char> seen(n, 0);
    std::vector<int32_t> by_degree(n);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    auto deg = [&](int32_t v) { return row_ptr[v + 1] - row_ptr[v]; };
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int32_t a, int32_t b) { return deg(a) < deg(b); });
    std::vector<int32_t> nb;
    for (int32_t start : by_degree) {
        if (seen[start]) continue;
        seen[start] = 1;
        size_t head = order->size();
        order->push_back(start);
        while (head < order->size()) {
            const int32_t v = (*order)[head++];
            nb.clear();
            for (int32_t e = row_ptr[v]; e < row_ptr[v + 1]; ++e)
                if (!seen[col[e]]) {
                    seen[col[e]] = 1;
                    nb.push_back(col[e]);
                }
            std::stable_sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) { return deg(a) < deg(b); });
            order->insert(order->end(), nb.begin(), nb.end());
        }
    }
}

int build_blocks(safe_nbr *nbr) {
    if (nbr->blocks_ready) return SAFE_OK;
    safe_ctx *ctx = nbr->ctx;
    const int64_t n = nbr->n, nnz = nbr->nnz;
    std::vector<int32_t> row_ptr(n + 1), col(std::max<int64_t>(nnz, 1));
    SAFE_HIP_CHECK(hipMemcpy(row_ptr.data(), nbr->row_ptr, (n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (nnz) SAFE_HIP_CHECK(hipMemcpy(col.data(), nbr->col, nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> order;
    if (static_cast<int64_t>(nbr->h_xy.size()) == 2 * n) order_by_layout(nbr->h_xy, n, &order);
    else order_by_graph(row_ptr, col, n, &order);
    std::vector<int32_t> pos(n);
    for (int64_t u = 0; u < n; ++u) pos[order[u]] = static_cast<int32_t>(u);

    const int64_t n_groups = ceil_div(n, MF_R), n_kb = ceil_div(n, 32), n_src = (n_kb + 1) * 32;
    std::vector<int32_t> h_order(n_src, static_cast<int32_t>(n)), h_rowmap(n_groups * MF_R, -1), h_rowcnt(n_groups * MF_R, -1);
    for (int64_t u = 0; u < n; ++u) {
        h_order[u] = order[u];
        h_rowmap[u] = order[u];
        h_rowcnt[u] = row_ptr[order[u] + 1] - row_ptr[order[u]];
    }
    // row groups are independent: built by a few host threads into per-group lists, then laid end to end
    std::vector<int32_t> ptr(n_groups + 1, 0), kbs;
    std::vector<uint32_t> bits;
    std::vector<std::vector<int32_t>> g_kbs(n_groups);
    std::vector<std::vector<uint32_t>> g_bits(n_groups);
    const int n_thr = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({8, n_groups, static_cast<int64_t>(std::thread::hardware_concurrency())})));
    std::atomic<int64_t> next_group{0};
    const bool deal_blocks = true;
    auto worker = [&]() {
        std::vector<int32_t> slot(n_kb, -1), touched;
        for (;;) {
            const int64_t g = next_group.fetch_add(1);
            if (g >= n_groups) break;
            touched.clear();
            const int64_t u0 = g * MF_R, u1 = std::min<int64_t>(n, u0 + MF_R);
            for (int64_t u = u0; u < u1; ++u) {
                const int32_t node = order[u];
                for (int32_t e = row_ptr[node]; e < row_ptr[node + 1]; ++e) {
                    const int32_t kb = pos[col[e]] >> 5;
                    if (slot[kb] < 0) {
                        slot[kb] = 0;
                        touched.push_back(kb);
                    }
                }
            }
            std::sort(touched.begin(), touched.end());
            const int64_t count = ceil_div(static_cast<int64_t>(touched.size()), 4) * 4;
            for (size_t i = 0; i < touched.size(); ++i) slot[touched[i]] = static_cast<int32_t>(i);
            std::vector<int32_t> &kb_out = g_kbs[g];
            std::vector<uint32_t> &bit_out = g_bits[g];
            kb_out.assign(touched.begin(), touched.end());
            kb_out.resize(count, static_cast<int32_t>(n_kb));          // padding blocks: all-zero rows, no members
            bit_out.assign(count * MF_R, 0u);
            for (int64_t u = u0; u < u1; ++u) {
                const int32_t node = order[u];
                for (int32_t e = row_ptr[node]; e < row_ptr[node + 1]; ++e) {
                    const int32_t p = pos[col[e]];
                    bit_out[static_cast<int64_t>(slot[p >> 5]) * MF_R + (u - u0)] |= 1u << (p & 31);
                }
            }
            for (int32_t kb : touched) slot[kb] = -1;
            // The kernel works through a group four blocks (one super-step) at a time, every wave on its own 32 (or 64) rows, and skips
            // the pieces that hold no member; the pieces 2k and 2k + 1 share a SIMD's matrix pipe and all waves meet at a barrier
            // after every super-step.  Deal the blocks into super-steps so that the busiest SIMD of each has as little to do as
            // possible (greedy, heaviest blocks first): at configs[4] 4.98 instead of 5.83 multiply steps per super-step on the
            // critical SIMD (8 without the skip; tools/notes/mfma_subtile_stats.py).  The sum over a group's blocks does not
            // depend on their order.
            const int64_t real = static_cast<int64_t>(touched.size()), n_ss = count / 4;
            if (deal_blocks && n_ss > 1) {
                std::vector<std::array<int, 4>> simd(real);
                std::vector<int> weight(real, 0);
                for (int64_t i = 0; i < real; ++i) {
                    simd[i] = {0, 0, 0, 0};
                    for (int w = 0; w < 8; ++w) {
                        uint32_t any = 0;
                        for (int r = 0; r < 32; ++r) any |= bit_out[i * MF_R + w * 32 + r];
                        if (any) {
                            ++simd[i][w >> 1];                                   // pieces 2k and 2k + 1 share a SIMD (both kernels)
                            ++weight[i];
                        }
                    }
                }
                std::vector<int64_t> by_weight(real);
                std::iota(by_weight.begin(), by_weight.end(), 0);
                std::stable_sort(by_weight.begin(), by_weight.end(), [&](int64_t a, int64_t b) { return weight[a] > weight[b]; });
                std::vector<std::array<int, 4>> load(n_ss, std::array<int, 4>{0, 0, 0, 0});
                std::vector<int> filled(n_ss, 0);
                std::vector<std::vector<int64_t>> members(n_ss);
                for (int64_t i : by_weight) {
                    int64_t best = -1;
                    int best_max = 0, best_fill = 0;
                    for (int64_t ss = 0; ss < n_ss; ++ss) {
                        if (filled[ss] >= 4) continue;
                        int mx = 0;
                        for (int q = 0; q < 4; ++q) mx = std::max(mx, load[ss][q] + simd[i][q]);
                        if (best < 0 || mx < best_max || (mx == best_max && filled[ss] < best_fill)) {
                            best = ss;
                            best_max = mx;
                            best_fill = filled[ss];
                        }
                    }
                    for (int q = 0; q < 4; ++q) load[best][q] += simd[i][q];
                    ++filled[best];
                    members[best].push_back(i);
                }
                std::vector<int32_t> kb_new(count, static_cast<int32_t>(n_kb));
                std::vector<uint32_t> bit_new(count * MF_R, 0u);
                int64_t at = 0;
                for (int64_t ss = 0; ss < n_ss; ++ss) {
                    std::sort(members[ss].begin(), members[ss].end());           // ascending column blocks inside a super-step
                    for (size_t j = 0; j < 4; ++j, ++at) {
                        if (j >= members[ss].size()) continue;                    // (padding block: no members, source row n)
                        const int64_t i = members[ss][j];
                        kb_new[at] = kb_out[i];
                        std::copy(bit_out.begin() + i * MF_R, bit_out.begin() + (i + 1) * MF_R, bit_new.begin() + at * MF_R);
                    }
                }
                kb_out.swap(kb_new);
                bit_out.swap(bit_new);
            }
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_thr; ++t) pool.emplace_back(worker);
        worker();
        for (auto &th : pool) th.join();
    }
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t count = static_cast<int64_t>(g_kbs[g].size());
        SAFE_REQUIRE(count <= MF_MAXBLK, "membership row group touches %lld column blocks (limit %d)", (long long)count, MF_MAXBLK);
        ptr[g + 1] = ptr[g] + static_cast<int32_t>(count);
    }
    kbs.resize(ptr[n_groups]);
    bits.resize(static_cast<size_t>(ptr[n_groups]) * MF_R);
    for (int64_t g = 0; g < n_groups; ++g) {
        std::copy(g_kbs[g].begin(), g_kbs[g].end(), kbs.begin() + ptr[g]);
        std::copy(g_bits[g].begin(), g_bits[g].end(), bits.begin() + static_cast<size_t>(ptr[g]) * MF_R);
    }
    // the same bits with the four blocks of a super-step side by side, and the largest neighborhood of every group
    std::vector<uint32_t> bits4(bits.size());
    for (size_t b = 0; b < kbs.size(); ++b)
        for (int r = 0; r < MF_R; ++r) bits4[((b >> 2) * MF_R + r) * 4 + (b & 3)] = bits[b * MF_R + r];
    // ... and in the operand order of k_permtest_mfma_g: the MFMA's A operand of lane half h holds member 16 h + 4 j + b of the
    // block in byte b of register j -- stored at bit 8 b + 4 h + j, register j is one shift and one AND of the word
    std::vector<uint32_t> bits4p(bits4.size());
    for (size_t i = 0; i < bits4.size(); ++i) {
        const uint32_t w = bits4[i];
        uint32_t o = 0;
        for (int h = 0; h < 2; ++h)
            for (int j = 0; j < 4; ++j)
                for (int b = 0; b < 4; ++b) o |= ((w >> (16 * h + 4 * j + b)) & 1u) << (8 * b + 4 * h + j);
        bits4p[i] = o;
    }
    std::vector<int32_t> h_grpmax(n_groups, 0);
    nbr->bs_max_group_blocks = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        for (int r = 0; r < MF_R; ++r) h_grpmax[g] = std::max(h_grpmax[g], h_rowcnt[g * MF_R + r]);
        nbr->bs_max_group_blocks = std::max<int64_t>(nbr->bs_max_group_blocks, ptr[g + 1] - ptr[g]);
    }
    nbr->bs_groups = n_groups;
    nbr->bs_blocks = static_cast<int64_t>(kbs.size());
    nbr->bs_pieces = 0;                                                   // (a wave skips the MFMAs of a 32 x 32 piece without members)
    for (size_t piece = 0; piece + 32 <= bits.size(); piece += 32) {
        uint32_t any = 0;
        for (int r = 0; r < 32; ++r) any |= bits[piece + r];
        nbr->bs_pieces += any != 0;
    }
    nbr->bs_src = n_src;
    nbr->h_bs_ptr = ptr;
    nbr->h_bs_rowmap = h_rowmap;
    SAFE_TRY(dev_alloc(&nbr->bs_order, n_src));
    SAFE_TRY(dev_alloc(&nbr->bs_rowmap, n_groups * MF_R));
    SAFE_TRY(dev_alloc(&nbr->bs_rowcnt, n_groups * MF_R));
    SAFE_TRY(dev_alloc(&nbr->bs_ptr, n_groups + 1));
    SAFE_TRY(dev_alloc(&nbr->bs_grpmax, n_groups));
    SAFE_TRY(dev_alloc(&nbr->bs_bits4, bits4.size() / 4));
    SAFE_HIP_CHECK(hipMemcpy(nbr->bs_grpmax, h_grpmax.data(), h_grpmax.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!bits4.empty()) SAFE_HIP_CHECK(hipMemcpy(nbr->bs_bits4, bits4.data(), bits4.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    SAFE_TRY(dev_alloc(&nbr->bs_bits4p, bits4p.size() / 4));
    if (!bits4p.empty()) SAFE_HIP_CHECK(hipMemcpy(nbr->bs_bits4p, bits4p.data(), bits4p.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    SAFE_TRY(dev_alloc(&nbr->bs_kb, kbs.size()));
    SAFE_TRY(dev_alloc(&nbr->bs_bits, bits.size()));
    SAFE_HIP_CHECK(hipMemcpy(nbr->bs_order, h_order.data(), n_src * sizeof(int32_t), hipMemcpyHostToDevice));
    SAFE_HIP_CHECK(hipMemcpy(nbr->bs_rowmap, h_rowmap.data(), h_rowmap.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    SAFE_HIP_CHECK(hipMemcpy(nbr->bs_rowcnt, h_rowcnt.data(), h_rowcnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    SAFE_HIP_CHECK(hipMemcpy(nbr->bs_ptr, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!kbs.empty()) SAFE_HIP_CHECK(hipMemcpy(nbr->bs_kb, kbs.data(), kbs.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!bits.empty()) SAFE_HIP_CHECK(hipMemcpy(nbr->bs_bits, bits.data(), bits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
#ifdef SAFE_HIP_DIAG
    if (getenv("SAFE_HIP_MFMA_DBG_NOMEMBERS")) safe_warn_diagnostic("SAFE_HIP_MFMA_DBG_NOMEMBERS");
    if (getenv("SAFE_HIP_MFMA_DBG_NOMEMBERS") && !bits.empty())          // diagnostic: every piece empty -> no MFMA is issued (wrong results)
    {
        SAFE_HIP_CHECK(hipMemset(nbr->bs_bits, 0, bits.size() * sizeof(uint32_t)));
        SAFE_HIP_CHECK(hipMemset(nbr->bs_bits4, 0, bits.size() * sizeof(uint32_t)));
        SAFE_HIP_CHECK(hipMemset(nbr->bs_bits4p, 0, bits.size() * sizeof(uint32_t)));
    }
#endif
    nbr->blocks_ready = true;
    (void)ctx;
    return SAFE_OK;
}

}  // namespace

namespace {

// planes of 0/1 attributes: bs[row][group of six 32-column tiles][tile][32 columns] = (x == 1), row n = zeros
template <typename T>
__global__ __launch_bounds__(256) void k_mfma_planes01(const void *__restrict__ raw, int64_t n, int64_t rs, int64_t cs,
                                                       int64_t col0, int64_t mloc, int64_t n_grp, unsigned char *__restrict__ bs) {
    __shared__ unsigned char tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t ct = blockIdx.x, r0 = static_cast<int64_t>(blockIdx.y) * 32, c0 = ct * 32;     // ct: 32-column tile
    const bool col_major = rs == 1;
    for (int i = 0; i < 4; ++i) {
        const int a = ty + 8 * i;
        const int64_t r = col_major ? r0 + tx : r0 + a, j = col_major ? c0 + a : c0 + tx;
        unsigned char v = 0;
        if (r < n && j < mloc) v = reinterpret_cast<const T *>(raw)[r * rs + (col0 + j) * cs] == static_cast<T>(1) ? 1 : 0;
        if (col_major) tile[tx][a] = v;
        else tile[a][tx] = v;
    }
    __syncthreads();
    const int64_t row_bytes = n_grp * MF_CN * 32;
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + ty + 8 * i;
        if (r > n) continue;
        bs[r * row_bytes + ct * 32 + tx] = r == n ? 0 : tile[ty + 8 * i][tx];   // tile ct = plane (ct % 6) of group ct / 6
    }
}

// the same planes from a C-order matrix (column stride 1): no transpose, a thread turns four consecutive
// columns into four bytes (16- / 32-byte loads when the row pitch and the shard offset allow it), a wave
// reads 1 KiB (f32) of one row
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_mfma_planes01_rows(const T *__restrict__ raw, int64_t n, int64_t rs, int64_t col0, int64_t mloc,
                                                            int64_t row_bytes, unsigned char *__restrict__ bs) {
    const int64_t j = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
    if (j >= row_bytes) return;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 8;
    if (VEC && j + 4 <= mloc && r0 + 8 <= n) {
        // the common case: eight rows, all eight loads in flight before the first compare (the general loop below compiles to
        // load - wait - store per row: one load in flight per thread)
        T v[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const T *src = raw + (r0 + i) * rs + col0 + j;
            if constexpr (sizeof(T) == 4) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                v[i][0] = q.x, v[i][1] = q.y, v[i][2] = q.z, v[i][3] = q.w;
            } else {
                const double2 q0 = *reinterpret_cast<const double2 *>(src), q1 = *reinterpret_cast<const double2 *>(src + 2);
                v[i][0] = q0.x, v[i][1] = q0.y, v[i][2] = q1.x, v[i][3] = q1.y;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) w |= (v[i][k] == static_cast<T>(1) ? 1u : 0u) << (8 * k);
            *reinterpret_cast<uint32_t *>(bs + (r0 + i) * row_bytes + j) = w;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t r = r0 + i;
        if (r > n) break;
        uint32_t w = 0;
        if (r < n) {
            const T *src = raw + r * rs + col0 + j;
            if (VEC && j + 4 <= mloc) {
                T v[4];
                if constexpr (sizeof(T) == 4) {
                    const float4 q = *reinterpret_cast<const float4 *>(src);
                    v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
                } else {
                    const double2 q0 = *reinterpret_cast<const double2 *>(src), q1 = *reinterpret_cast<const double2 *>(src + 2);
                    v[0] = q0.x, v[1] = q0.y, v[2] = q1.x, v[3] = q1.y;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) w |= (v[k] == static_cast<T>(1) ? 1u : 0u) << (8 * k);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (j + k < mloc) w |= (src[k] == static_cast<T>(1) ? 1u : 0u) << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t *>(bs + r * row_bytes + j) = w;
    }
}

// Second half of the split hypergeometric form.  A workgroup = (one neighborhood size, <= 64 of the rows
// that have it, 8 groups of six 32-column tiles); wave w owns one group, lane (c, hh) the columns
// (6 grp + hh + 2j) * 32 + c, j = 0..2, so every store instruction of a wave covers 64 consecutive
// columns = 512 contiguous bytes of one row of p / nes / nes_binary.
// The table slab of the size -- [0 .. largest count of the call][annotation count id], one contiguous run -- is staged in
// LDS once per workgroup, so a lookup is one ds_read_b128 instead of a 64-line global gather (which
// costs the texture path 64 cycles and was a third of the kernel).  Rows are software-pipelined and the
// loop is branch-free: vmcnt counts stores too and retires in order, so the counts of batch i+1 are
// requested BEFORE the stores of batch i are issued (waiting for them then leaves those stores in
// flight); padding rows / columns store to a per-lane dummy slot.
// rows[i] = {row position, node, neighborhood-size id, -}; tasks[t] = [first, last) into rows.
template <int UN, bool STAGED>
__device__ __forceinline__ void hyp_emit_rows(const unsigned int *__restrict__ src, const int4 *__restrict__ rows, int2 task,
                                              const double2 *__restrict__ lut, uint32_t n_kid, const uint32_t (&kofs)[3], const bool (&ok)[3],
                                              const int64_t (&col)[3], int hh, int lane, int64_t mloc, const HypLookup &hl) {
    double *const dummy = hl.dummy + lane;
    unsigned int hits[3] = {0u, 0u, 0u};
    int4 r[UN];
    uint32_t w[UN][3];
    auto load_batch = [&](int i0, int4 (&rr)[UN], uint32_t (&ww)[UN][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            rr[i] = rows[min(i0 + i, task.y - 1)];                           // wave-uniform: scalar loads
            rr[i].w = i0 + i < task.y;
        }
#pragma unroll
        for (int i = 0; i < UN; ++i)                                         // three dword loads: a 96-bit tuple carried round
#pragma unroll
            for (int q = 0; q < 3; ++q) ww[i][q] = src[static_cast<int64_t>(rr[i].x) * 96 + q];   // the loop gets copied (= waited for) at once
    };
    // two batches per trip, the register sets swapping roles (a copy would wait for the loads just issued);
    // rows past the end of the task are dead (dummy stores), so the trip needs no branch
    auto batch = [&](int i0, const int4 (&rc)[UN], const uint32_t (&wc)[UN][3], int4 (&rn)[UN], uint32_t (&wn)[UN][3]) __attribute__((always_inline)) {
        load_batch(i0 + UN, rn, wn);
        __builtin_amdgcn_sched_barrier(0);                                   // the loads stay ahead of this batch's stores
        double2 val[UN][3];
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            const uint32_t x[3] = {hh ? wc[i][0] >> 16 : wc[i][0] & 0xffffu, hh ? wc[i][1] >> 16 : wc[i][1] & 0xffffu,
                                   hh ? wc[i][2] >> 16 : wc[i][2] & 0xffffu};
#pragma unroll
            for (int j = 0; j < 3; ++j) val[i][j] = lut[kofs[j] + (ok[j] ? x[j] * n_kid : 0u)];   // [x][count id]
        }
#pragma unroll
        for (int i = 0; i < UN; ++i) {
            const int64_t o = static_cast<int64_t>(rc[i].y) * mloc;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                // a row past the end of the task is a second copy of the task's last row: it stores the same
                // values to the same addresses again (a shared dummy slot would be one hot L2 line) and counts no hits
                const bool live = ok[j];
                const bool hit = live && val[i][j].x < hl.p_cut;                   // safe.py:468-470 (nes_p_cut)
                __builtin_nontemporal_store(val[i][j].x, live ? hl.pvalues_pos + o + col[j] : dummy);
                __builtin_nontemporal_store(val[i][j].y, live ? hl.nes + o + col[j] : dummy);   // -log10 p from the table (safe.py:608)
                __builtin_nontemporal_store(hit ? 1.0 : 0.0, live ? hl.nes_binary + o + col[j] : dummy);
                hits[j] += hit && rc[i].w != 0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int4 r2[UN];
    uint32_t w2[UN][3];
    load_batch(task.x, r, w);
    for (int i0 = task.x; i0 < task.y; i0 += 2 * UN) {
        batch(i0, r, w, r2, w2);
        batch(i0 + UN, r2, w2, r, w);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (hits[j]) atomicAdd(&hl.enriched[col[j]], hits[j]);
}



template <int UN, bool TAIL>
__global__ __launch_bounds__(512) void k_hyp_emit(const unsigned int *__restrict__ cnt16, int64_t n_padr, int64_t n_grp,
                                                  const int4 *__restrict__ rows, const int2 *__restrict__ tasks, int64_t mloc,
                                                  HypLookup hl, int lds_entries, int order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char emit_lds[];
    double2 *slab = reinterpret_cast<double2 *>(emit_lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, hh = lane >> 5;
    // order 0: the column blocks of a task are neighbours in dispatch order; 1 / 2: all tasks of one column block, then the
    // next block (1: last block first -- the counts the matrix-core kernel wrote last are read first, out of the memory-side cache)
    const unsigned int cblk = order == 0 ? blockIdx.x : order == 1 ? gridDim.y - 1 - blockIdx.y : blockIdx.y;
    const int2 task = tasks[order == 0 ? blockIdx.y : blockIdx.x];
    const int nid = rows[task.x].z;
    const uint32_t xc = min(static_cast<uint32_t>(*hl.xmax) + 1u, static_cast<uint32_t>(hl.xs));
    const uint32_t n_kid = static_cast<uint32_t>(hl.n_kid), xs = static_cast<uint32_t>(hl.xs);
    const bool staged = n_kid * xc <= static_cast<uint32_t>(lds_entries);      // uniform over the whole launch
    const double2 *tab_n = hl.tab + static_cast<int64_t>(nid) * n_kid * xs;
    if (staged) {
        const uint32_t total = n_kid * xc;                                   // [x < xc][count id]: one contiguous run of the table
        for (uint32_t e0 = threadIdx.x; e0 < total; e0 += 4 * 512) {         // four loads in flight per thread
            double2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = tab_n[min(e0 + q * 512u, total - 1u)];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (e0 + q * 512u < total) slab[e0 + q * 512u] = v[q];
        }
        __syncthreads();
    }
    const int64_t grp = static_cast<int64_t>(cblk) * 8 + wave;
    if (grp >= n_grp) return;
    int64_t col[3];
    uint32_t kofs[3];
    bool ok[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        col[j] = (grp * 6 + hh + 2 * j) * 32 + c;
        ok[j] = col[j] < mloc;
        kofs[j] = ok[j] ? static_cast<uint32_t>(hl.kid[col[j]]) : 0u;
    }
    const unsigned int *src = cnt16 + (grp * n_padr * 32 + c) * 3;
    // whole trips of 2 UN rows first, the last 1 .. 2 UN - 1 rows in trips of two: at most one repeated row per task
    // instead of up to 2 UN - 1 (tasks are short -- 26 rows on average at 20 000 nodes -- and repeated rows are real stores)
    const int bulk = TAIL ? task.x + (task.y - task.x) / (2 * UN) * (2 * UN) : task.y;
    if (staged) {
        if (bulk > task.x) hyp_emit_rows<UN, true>(src, rows, make_int2(task.x, bulk), slab, n_kid, kofs, ok, col, hh, lane, mloc, hl);
        if (bulk < task.y) hyp_emit_rows<1, true>(src, rows, make_int2(bulk, task.y), slab, n_kid, kofs, ok, col, hh, lane, mloc, hl);
    } else {
        if (bulk > task.x) hyp_emit_rows<UN, false>(src, rows, make_int2(task.x, bulk), tab_n, n_kid, kofs, ok, col, hh, lane, mloc, hl);
        if (bulk < task.y) hyp_emit_rows<1, false>(src, rows, make_int2(bulk, task.y), tab_n, n_kid, kofs, ok, col, hh, lane, mloc, hl);
    }
}

// planes, task queues and source map of the counts form, enqueued on ctx->stream
struct CountsSetup {
    const std::vector<int2> *tasks = nullptr;   // the membership handle's cached list (host memory behind an asynchronous copy)
    int32_t q_off[9] = {0};
    unsigned char *d_bs = nullptr;
    int2 *d_tasks = nullptr;
    int32_t *d_qoff = nullptr, *d_src = nullptr;
    unsigned int *d_qctr = nullptr;       // [0..7] queue counters, [12] largest count (split form)
    int64_t n_grp = 0, row_bytes = 0, n_src = 0, mloc = 0;
};

int counts_setup(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, CountsSetup *cs) {
    SAFE_TRY(build_blocks(nbr));
    const int64_t n = nbr->n, mloc = col1 - col0;
    const int64_t n_ct = ceil_div(mloc, 32), n_grp = ceil_div(n_ct, MF_CN), row_bytes = n_grp * MF_CN * 32, n_src = nbr->bs_src;
    cs->n_grp = n_grp, cs->row_bytes = row_bytes, cs->n_src = n_src, cs->mloc = mloc;
    // task list, queue words and source map first: they do not depend on the planes, and behind the planes kernel in stream
    // order they were 45 us of copies, fills and a tiny kernel between it and the count kernel
    if (nbr->counts_tasks_grp != n_grp) {
        std::vector<int32_t> g_order(nbr->bs_groups);
        std::iota(g_order.begin(), g_order.end(), 0);
        const std::vector<int32_t> &bp = nbr->h_bs_ptr;
        std::stable_sort(g_order.begin(), g_order.end(), [&](int32_t a, int32_t b) { return bp[a + 1] - bp[a] > bp[b + 1] - bp[b]; });
        nbr->counts_tasks.clear();
        nbr->counts_qoff[0] = 0;
        for (int qx = 0; qx < 8; ++qx) {
            for (int64_t ct = qx; ct < n_grp; ct += 8)
                for (int32_t g : g_order) nbr->counts_tasks.push_back(make_int2(g, static_cast<int>(ct)));
            nbr->counts_qoff[qx + 1] = static_cast<int32_t>(nbr->counts_tasks.size());
        }
        nbr->counts_tasks_grp = n_grp;
    }
    const std::vector<int2> &tasks = nbr->counts_tasks;
    cs->tasks = &nbr->counts_tasks;
    memcpy(cs->q_off, nbr->counts_qoff, sizeof(cs->q_off));
    void *ws = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 3, tasks.size() * sizeof(int2) + 16 * sizeof(int32_t) + 16 * sizeof(unsigned int), &ws));
    cs->d_tasks = static_cast<int2 *>(ws);
    cs->d_qoff = reinterpret_cast<int32_t *>(cs->d_tasks + tasks.size());
    cs->d_qctr = reinterpret_cast<unsigned int *>(cs->d_qoff + 16);
    SAFE_TRY(ctx_scratch(ctx, 4, static_cast<size_t>(n_src) * sizeof(int32_t), reinterpret_cast<void **>(&cs->d_src)));
    SAFE_HIP_CHECK(hipMemcpyAsync(cs->d_tasks, tasks.data(), tasks.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(cs->d_qoff, nbr->counts_qoff, sizeof(cs->q_off), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(cs->d_qctr, 0, 16 * sizeof(unsigned int), ctx->stream));
    hipLaunchKernelGGL(k_mfma_src, dim3(ceil_div(n_src, 256), 1), dim3(256), 0, ctx->stream, nbr->bs_order, n_src, n,
                       static_cast<const int32_t *>(nullptr), 0, cs->d_src);
    SAFE_TRY(ctx_scratch(ctx, 1, static_cast<size_t>(n + 1) * row_bytes, reinterpret_cast<void **>(&cs->d_bs)));
    if (attr->col_stride == 1) {                                          // C order: straight through
        const dim3 grid(ceil_div(row_bytes / 4, 256), ceil_div(n + 1, 8));
        const bool f32 = attr->dtype == SAFE_DTYPE_F32;
        const int64_t per16 = f32 ? 4 : 2;                                   // elements per 16 bytes
        const bool vec = attr->row_stride % per16 == 0 && col0 % per16 == 0 && reinterpret_cast<uintptr_t>(attr->raw) % 16 == 0;
#define PLANES_ROWS(T, V)                                                                                                      \
    hipLaunchKernelGGL((k_mfma_planes01_rows<T, V>), grid, dim3(256), 0, ctx->stream, static_cast<const T *>(attr->raw), n,   \
                       attr->row_stride, col0, mloc, row_bytes, cs->d_bs)
        if (f32) {
            if (vec) PLANES_ROWS(float, true);
            else PLANES_ROWS(float, false);
        } else {
            if (vec) PLANES_ROWS(double, true);
            else PLANES_ROWS(double, false);
        }
#undef PLANES_ROWS
    } else {
        const dim3 grid(n_grp * MF_CN, ceil_div(n + 1, 32));
        if (attr->dtype == SAFE_DTYPE_F32)
            hipLaunchKernelGGL(k_mfma_planes01<float>, grid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                               attr->col_stride, col0, mloc, n_grp, cs->d_bs);
        else
            hipLaunchKernelGGL(k_mfma_planes01<double>, grid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                               attr->col_stride, col0, mloc, n_grp, cs->d_bs);
    }
    const size_t lds_bytes = 2 * (4 * mf_ks(MF_CN)) + MF_MAXBLK * sizeof(int32_t);
    for (const void *fn : {reinterpret_cast<const void *>(k_permtest_mfma<true, MF_CN, false, true, 0>),
                           reinterpret_cast<const void *>(k_permtest_mfma<true, MF_CN, false, true, 1>),
                           reinterpret_cast<const void *>(k_permtest_mfma<true, MF_CN, false, true, 2>)})
        SAFE_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    return SAFE_OK;
}

void counts_launch(safe_ctx *ctx, safe_nbr *nbr, const CountsSetup &cs, const HypLookup &hl, int spare_cus = 0) {
    const size_t lds_bytes = 2 * (4 * mf_ks(MF_CN)) + MF_MAXBLK * sizeof(int32_t);
    const int64_t blocks = std::min<int64_t>(static_cast<int64_t>(cs.tasks->size()), std::max(1, ctx->num_cu - spare_cus));
#define COUNTS_LAUNCH(EPI)                                                                                                              \
    hipLaunchKernelGGL((k_permtest_mfma<true, MF_CN, false, true, EPI>), dim3(blocks), dim3(512), lds_bytes, ctx->stream, cs.d_bs,      \
                       cs.row_bytes, static_cast<int64_t>(MF_CN * 32), cs.d_src, cs.n_src, 1, nbr->bs_ptr, nbr->bs_kb, nbr->bs_bits,    \
                       cs.d_tasks, cs.d_qoff, cs.d_qctr, cs.mloc, static_cast<unsigned int *>(nullptr), nbr->bs_groups * MF_R,           \
                       nbr->bs_rowmap, static_cast<const double *>(nullptr), static_cast<double *>(nullptr), hl, MfmaFilt{})
    if (hl.cnt16) COUNTS_LAUNCH(0);                 // packed counts for k_hyp_emit
    else if (!hl.tab) COUNTS_LAUNCH(1);             // plain counts
    else COUNTS_LAUNCH(2);                          // table lookup in the epilogue
#undef COUNTS_LAUNCH
}

}  // namespace

// Fused form: counts + epilogue (table lookup when hl.tab, plain counts otherwise) in one kernel.
// (tried: three tiles per task at 128 VGPRs so that two workgroups share a CU and one's store epilogue
// overlaps the other's matrix phase -- the main loop spills and the kernel is 1.5x slower)
int launch_mfma_counts(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, const HypLookup &hl) {
    CountsSetup cs;
    SAFE_TRY(counts_setup(ctx, nbr, attr, col0, col1, &cs));
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    counts_launch(ctx, nbr, cs, hl);
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    ctx->last_kernel.name = "k_permtest_mfma<counts>";
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));                 // the task vector is host memory
    return SAFE_OK;
}

// Split form of the table lookup (default): the matrix-core kernel leaves packed u16 counts (8 % of the
// output bytes) and k_hyp_emit streams p / nes / nes_binary out, instead of a fused epilogue whose
// stores cannot overlap the next task's matrix phase (one workgroup per CU).  The first half needs
// nothing from the hypergeometric table, so the caller builds the table on the side stream meanwhile.
struct MfmaCountsSplit {
    CountsSetup cs;
    unsigned int *cnt16 = nullptr;
    std::vector<int4> rows;
    std::vector<int2> tasks;
    int4 *d_rows = nullptr;
    int2 *d_tasks = nullptr;
};

bool mfma_counts_split_applicable(const safe_nbr *nbr) {
    const char *split_env = getenv("SAFE_HIP_HYP_SPLIT");
    return nbr->max_count < 65536 && !(split_env && !strcmp(split_env, "0"));
}

int mfma_counts_split_begin(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, int64_t col0, int64_t col1, MfmaCountsSplit **out) {
    MfmaCountsSplit *st = new MfmaCountsSplit;
    *out = st;
    SAFE_TRY(counts_setup(ctx, nbr, attr, col0, col1, &st->cs));
    const int64_t n_padr = nbr->bs_groups * MF_R;
    SAFE_TRY(ctx_scratch(ctx, 5, static_cast<size_t>(st->cs.n_grp) * n_padr * 32 * 3 * sizeof(unsigned int), reinterpret_cast<void **>(&st->cnt16)));
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    // The count kernel is persistent and takes a CU whole (256 VGPRs x 2 waves per SIMD): with one workgroup
    // per CU nothing else runs until it ends -- not even the copy kernels of the side stream.
    // A few CUs are left to the side stream (neighborhood sizes and the copy kernels of the id vectors).
    HypLookup hl{};
    hl.cnt16 = st->cnt16;
    hl.xmax = st->cs.d_qctr + 12;
    counts_launch(ctx, nbr, st->cs, hl, 8);
    SAFE_HIP_CHECK(hipGetLastError());
    return SAFE_OK;
}

void mfma_counts_split_free(MfmaCountsSplit *st) { delete st; }

const unsigned int *mfma_counts_split_xmax(const MfmaCountsSplit *st) { return st->cs.d_qctr + 12; }

// rows grouped by neighborhood size for the second half, uploaded on `hs` (the side stream, idle while the count kernel runs on
// ctx->stream; the caller makes ctx->stream wait for it): h_nid[node] = neighborhood-size id (host)
// pinned_stage: pinned host memory for n int4 + n int2 (the uploads must not go through the runtime's pageable staging)
int mfma_counts_split_rows(safe_ctx *ctx, safe_nbr *nbr, MfmaCountsSplit *st, const int32_t *h_nid, hipStream_t hs, void *pinned_stage) {
    const int64_t n_padr = nbr->bs_groups * MF_R;
    constexpr int EMIT_CHUNK = 64;
    // every group cut into chunks of <= 64 rows, long chunks first (counting sort: the ids are dense)
    int32_t n_ids = 0;
    for (int64_t i = 0; i < nbr->n; ++i) n_ids = std::max(n_ids, h_nid[i] + 1);
    std::vector<int32_t> first(n_ids + 1, 0);
    for (int64_t i = 0; i < nbr->n; ++i) ++first[h_nid[i] + 1];
    for (int32_t k = 0; k < n_ids; ++k) first[k + 1] += first[k];
    st->rows.resize(nbr->n);
    for (int64_t u = 0; u < n_padr; ++u) {
        const int32_t node = nbr->h_bs_rowmap[u];
        if (node >= 0) st->rows[first[h_nid[node]]++] = make_int4(static_cast<int>(u), node, h_nid[node], 1);
    }
    for (size_t a = 0; a < st->rows.size();) {
        size_t b = a;
        while (b < st->rows.size() && st->rows[b].z == st->rows[a].z) ++b;
        for (size_t c = a; c < b; c += EMIT_CHUNK) st->tasks.push_back(make_int2(static_cast<int>(c), static_cast<int>(std::min(b, c + EMIT_CHUNK))));
        a = b;
    }
    std::stable_sort(st->tasks.begin(), st->tasks.end(), [](const int2 &a, const int2 &b) { return a.y - a.x > b.y - b.x; });
    void *ws = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 8, st->rows.size() * sizeof(int4) + st->tasks.size() * sizeof(int2), &ws));
    st->d_rows = static_cast<int4 *>(ws);
    st->d_tasks = reinterpret_cast<int2 *>(st->d_rows + st->rows.size());
    const size_t bytes = st->rows.size() * sizeof(int4) + st->tasks.size() * sizeof(int2);
    memcpy(pinned_stage, st->rows.data(), st->rows.size() * sizeof(int4));
    memcpy(static_cast<char *>(pinned_stage) + st->rows.size() * sizeof(int4), st->tasks.data(), st->tasks.size() * sizeof(int2));
    SAFE_HIP_CHECK(hipMemcpyAsync(ws, pinned_stage, bytes, hipMemcpyHostToDevice, hs));
    return SAFE_OK;
}

// second half: hl carries the table, the ids and the outputs.  Nothing here waits for the stream: the row / task vectors live in
// `st` until the caller has synchronised (mfma_counts_split_free)
int mfma_counts_split_emit(safe_ctx *ctx, safe_nbr *nbr, MfmaCountsSplit *st, const HypLookup &hl_in) {
    HypLookup hl = hl_in;
    hl.cnt16 = st->cnt16;
    hl.xmax = st->cs.d_qctr + 12;
    const int64_t n_padr = nbr->bs_groups * MF_R, n_grp = st->cs.n_grp;
    constexpr int EMIT_UN = 2;                              // rows per batch: 2 measured best of 1, 2, 3, 4, 8 (1.21 / 1.30 / 1.53 ms at 4 / 8)
    const int4 *d_rows = st->d_rows;
    const int2 *d_tasks = st->d_tasks;
    const int64_t want = hl.n_kid * hl.xs * static_cast<int64_t>(sizeof(double2));
    const char *lds_env = getenv("SAFE_HIP_EMIT_LDS_KB");                // tests: 0 forces the global-gather loop
    const int64_t lds_cap = lds_env ? std::max(0, atoi(lds_env)) * 1024ll : (64ll << 10);
    const int lds_bytes = static_cast<int>(std::min<int64_t>(want, std::min<int64_t>(lds_cap, 64 << 10)));
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));                 // the timed kernel of this form is the HBM-bound one
    constexpr int order = 0;                                // (blockIdx.x walks the column groups: the other order measured slower)
    const dim3 egrid(ceil_div(n_grp, 8), st->tasks.size());
    const int lds_entries = lds_bytes / static_cast<int>(sizeof(double2));
#define EMIT_LAUNCH(U, T)                                                                                                              \
    do {                                                                                                                            \
        SAFE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_hyp_emit<U, T>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                           lds_bytes));                                                                             \
        hipLaunchKernelGGL((k_hyp_emit<U, T>), egrid, dim3(512), lds_bytes, ctx->stream, st->cnt16, n_padr, n_grp, d_rows, d_tasks,    \
                           st->cs.mloc, hl, lds_entries, order);                                                                      \
    } while (0)
    // (16-byte pair stores -- a lane owning two adjacent columns of two rows -- measured flat against this form at configs[3] in
    // round 5, 1.205 vs 1.190 ms: the emit kernel is not bound by the width of its stores; that kernel is gone)
    static_assert(EMIT_UN == 2, "k_hyp_emit<2, true> is the measured best");
    EMIT_LAUNCH(2, true);
#undef EMIT_LAUNCH
    SAFE_HIP_CHECK(hipGetLastError());
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    ctx->last_kernel.name = "k_hyp_emit";                               // (after k_permtest_mfma<counts> and k_hyp_table)
    return SAFE_OK;
}

namespace {
// ---------------------------------------------------------------------------------------
// filtered form: the compares the high digits could not decide, and the observed scores
// ---------------------------------------------------------------------------------------
// One wave per undecided compare {row u, column, permutation, V_hi}: V_lo = sum over the members of the row's neighborhood of
// the low digits (0-2) of the permuted attribute row, then the EXACT sign of v - o = (V_hi << 24) + V_lo - O.
// Undecided compares of one task share a column tile, so the slice rows they gather are the ones the task kept in its L2.
__global__ __launch_bounds__(256) void k_mfma_resolve(const ulonglong2 *__restrict__ amb, const unsigned int *__restrict__ amb_count,
                                                      unsigned int amb_cap, const long long *__restrict__ obs64, int64_t n_padr,
                                                      const int32_t *__restrict__ rowmap, const int32_t *__restrict__ row_ptr,
                                                      const int32_t *__restrict__ col_idx, const int32_t *__restrict__ table, int64_t n,
                                                      const unsigned char *__restrict__ bs_lo, int64_t tile_bytes, int64_t hi_off,
                                                      const long long *__restrict__ q64, int64_t mloc, unsigned int *__restrict__ gl_counts) {
    const unsigned int count = min(*amb_count, amb_cap);
    const int lane = threadIdx.x & 63;
    const unsigned int wave0 = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    constexpr int64_t row_bytes = (MF_NS / 2) * 32;
    for (unsigned int rec = wave0; rec < count; rec += n_waves) {
        const ulonglong2 w = amb[rec];
        const int64_t u = static_cast<int64_t>(w.x & 0xFFFFFFFFull), col = static_cast<int64_t>(w.x >> 32);
        const int64_t perm = static_cast<int64_t>(w.y & 0xFFFFull);
        const bool full = (w.y >> 63) != 0ull;                                       // no partial sum: all six digits are summed here
        const long long v_hi = full ? 0ll : static_cast<long long>(w.y) >> 16;
        const int32_t node = rowmap[u];
        const int32_t e0 = row_ptr[node], e1 = row_ptr[node + 1];
        const unsigned char *base = bs_lo + (col >> 5) * tile_bytes + (col & 31);
        const int32_t *cur = table + perm * (n + 1);
        long long s = 0;
        if (full && q64) {                                                       // one 8-byte load per member instead of six bytes from two tiles
            for (int32_t e = e0 + lane; e < e1; e += 64) s += q64[static_cast<int64_t>(cur[col_idx[e]]) * mloc + col];
        } else
        for (int32_t e = e0 + lane; e < e1; e += 64) {
            const signed char *d = reinterpret_cast<const signed char *>(base + static_cast<int64_t>(cur[col_idx[e]]) * row_bytes);
            s += static_cast<long long>(static_cast<int>(d[0]) + 256 * static_cast<int>(d[32]) + 65536 * static_cast<int>(d[64]));
            if (full) {
                const signed char *dh = d + hi_off;
                s += static_cast<long long>(static_cast<int>(dh[0]) + 256 * static_cast<int>(dh[32]) + 65536 * static_cast<int>(dh[64])) << 24;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const long long diff = (v_hi << 24) - obs64[col * n_padr + u] + s;          // v - o, exactly
            const unsigned int add = diff < 0 ? (1u << 16) : diff > 0 ? 1u : 0u;
            if (add) atomicAdd(&gl_counts[col * n_padr + u], add);
        }
    }
}

// z-scores: one wave per undecided compare {row u, column, permutation}.  The exact sums of the permuted neighborhood -- values,
// squares (all six digits of each) and the number of non-NaN members -- then the reference's formula (safe_extras.py:19-31) in
// its order of operations, exactly as the seven-slice kernel evaluates it, against the observed score.
__global__ __launch_bounds__(256) void k_mfma_resolve_z(const ulonglong2 *__restrict__ amb, const unsigned int *__restrict__ amb_count,
                                                        unsigned int amb_cap, const double *__restrict__ zobs, int64_t mloc, int64_t n_padr,
                                                        const int32_t *__restrict__ rowmap, const int32_t *__restrict__ row_ptr,
                                                        const int32_t *__restrict__ col_idx, const int32_t *__restrict__ table, int64_t n,
                                                        const unsigned char *__restrict__ bs_lo, const unsigned char *__restrict__ bs_hi,
                                                        const longlong2 *__restrict__ z64, const double *__restrict__ col_scale,
                                                        unsigned int *__restrict__ gl_counts) {
    const unsigned int count = min(*amb_count, amb_cap);
    const int lane = threadIdx.x & 63;
    const unsigned int wave0 = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    for (unsigned int rec = wave0; rec < count; rec += n_waves) {
        const ulonglong2 w = amb[rec];
        const int64_t u = static_cast<int64_t>(w.x & 0xFFFFFFFFull), col = static_cast<int64_t>(w.x >> 32);
        const int64_t perm = static_cast<int64_t>(w.y & 0xFFFFull);
        const int32_t node = rowmap[u];
        const int32_t e0 = row_ptr[node], e1 = row_ptr[node + 1];
        const unsigned char *lo = bs_lo + (col >> 4) * ((n + 1) * 96) + (col & 15);
        const unsigned char *hi = bs_hi + (col >> 4) * ((n + 1) * 128) + (col & 15);
        const int32_t *cur = table + perm * (n + 1);
        long long s1 = 0, s2 = 0;
        int present = 0;
        if (z64) {                                                               // one 16-byte load per member
            for (int32_t e = e0 + lane; e < e1; e += 64) {
                const longlong2 v = z64[static_cast<int64_t>(cur[col_idx[e]]) * mloc + col];
                s1 += v.x;
                s2 += v.y & ((1ll << 62) - 1ll);
                present += static_cast<int>(v.y >> 62);
            }
        } else
        for (int32_t e = e0 + lane; e < e1; e += 64) {
            const int64_t src = cur[col_idx[e]];
            const signed char *l = reinterpret_cast<const signed char *>(lo + src * 96), *hh = reinterpret_cast<const signed char *>(hi + src * 128);
            s1 += static_cast<long long>(static_cast<int>(l[0]) + 256 * static_cast<int>(l[32]) + 65536 * static_cast<int>(l[64])) +
                  (static_cast<long long>(static_cast<int>(hh[0]) + 256 * static_cast<int>(hh[32]) + 65536 * static_cast<int>(hh[64])) << 24);
            s2 += static_cast<long long>(static_cast<int>(l[16]) + 256 * static_cast<int>(l[48]) + 65536 * static_cast<int>(l[80])) +
                  (static_cast<long long>(static_cast<int>(hh[16]) + 256 * static_cast<int>(hh[48]) + 65536 * static_cast<int>(hh[80])) << 24);
            present += hh[96];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
            present += __shfl_xor(present, o);
        }
        if (lane == 0) {
            const double o = zobs[static_cast<int64_t>(node) * mloc + col];
            const double sc1 = col_scale[col], sc2 = col_scale[mloc + col];
            const double members = static_cast<double>(present);
            const double mean = (static_cast<double>(s1) * sc1) / members;          // safe_extras.py:21-23
            const double exx = (static_cast<double>(s2) * sc2) / members;           // safe_extras.py:25-26
            const double sd = sqrt(exx - mean * mean);                              // safe_extras.py:27
            double zs = mean / sd;                                                  // safe_extras.py:28
            if (sd == 0.0) zs = __longlong_as_double(0x7FF8000000000000ll);         // safe_extras.py:29
            if (members < 3.0) zs = __longlong_as_double(0x7FF8000000000000ll);     // safe_extras.py:30
            const unsigned int add = (static_cast<unsigned int>(zs >= o) << 16) | static_cast<unsigned int>(zs <= o);
            if (add) atomicAdd(&gl_counts[col * n_padr + u], add);
        }
    }
}

// ns[node][column] = observed score (safe.py:496-499): obs64 is [column][padded row]
__global__ __launch_bounds__(256) void k_mfma_obs_ns(const long long *__restrict__ obs64, int64_t n_padr, const int32_t *__restrict__ rowmap,
                                                     const double *__restrict__ col_scale, int64_t mloc, double *__restrict__ ns) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t u0 = static_cast<int64_t>(blockIdx.x) * 32, c0 = static_cast<int64_t>(blockIdx.y) * 32;
    for (int i = 0; i < 4; ++i) {
        const int64_t c = c0 + ty + 8 * i;
        tile[ty + 8 * i][tx] = c < mloc ? static_cast<double>(obs64[c * n_padr + u0 + tx]) * col_scale[c] : 0.0;
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) {
        const int32_t node = rowmap[u0 + ty + 8 * i];
        const int64_t c = c0 + tx;
        if (node >= 0 && c < mloc) ns[static_cast<int64_t>(node) * mloc + c] = tile[tx][ty + 8 * i];
    }
}
}  // namespace

void nbr_free_blocks(safe_nbr *nbr) {
    (void)hipFree(nbr->bs_order);
    (void)hipFree(nbr->bs_rowmap);
    (void)hipFree(nbr->bs_rowcnt);
    (void)hipFree(nbr->bs_grpmax);
    (void)hipFree(nbr->bs_bits4);
    (void)hipFree(nbr->bs_bits4p);
    nbr->bs_rowcnt = nbr->bs_grpmax = nullptr;
    nbr->bs_bits4 = nbr->bs_bits4p = nullptr;
    (void)hipFree(nbr->bs_ptr);
    (void)hipFree(nbr->bs_kb);
    (void)hipFree(nbr->bs_bits);
    nbr->bs_order = nbr->bs_rowmap = nbr->bs_ptr = nbr->bs_kb = nullptr;
    nbr->bs_bits = nullptr;
    nbr->blocks_ready = false;
    nbr->counts_tasks_grp = -1;
}

bool mfma_applicable(const safe_ctx *ctx, const safe_nbr *nbr, const safe_attr *attr, const safe_perms *perms, bool z) {
    (void)ctx;
    (void)attr;
    const char *force = getenv("SAFE_HIP_FORCE_PATH");
    if (force && (!strcmp(force, "gather") || !strcmp(force, "lds"))) return false;
    const char *z_env = getenv("SAFE_HIP_MFMA_Z");                       // =0: z-scores stay on the f64 kernels
    if (z && z_env && !strcmp(z_env, "0")) return false;
    if (perms->count < 1 || perms->count > 65535) return false;
    // a score is a sum of `members` fixed-point values below 2^46, combined and compared as 64-bit integers: neighborhoods of
    // 2^16 members and more could pass 2^62 (the f64 kernels take those)
    if (nbr->n > (1ll << 30) / 32 || nbr->max_count >= (1 << 16)) return false;
    if (force && !strcmp(force, "mfma")) return true;
    return nbr->n >= 256;                       // below one row group the LDS-resident f64 kernel is the better fit
}

// Runs the permutation test of columns [col0, col1) on the MFMA path.  *declined = true (and
// SAFE_OK) when the attribute values cannot be represented on the fixed-point grid without a
// rounding that could matter; the caller then uses the f64 kernels.
static int launch_mfma_run(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int64_t col0, int64_t col1, bool z,
                           const PermOut &out_in, bool *declined, bool allow_filter, bool *overflowed) {
    *declined = false;
    *overflowed = false;
    SAFE_TRY(build_blocks(nbr));
    PermOut out = out_in;
    const int64_t n = nbr->n, mloc = col1 - col0, P = perms->count;
    // z-scores: tiles of 16 columns (value digits | square digits) and a seventh slice of not-NaN flags
    const int64_t n_ct = ceil_div(mloc, z ? 16 : 32), n_src = nbr->bs_src;
    int64_t row_bytes = n_ct * (z ? MF_NS + 1 : MF_NS) * 32;   // (the largest form; the call's slice count is known after the column statistics)
    const int64_t n_padr = nbr->bs_groups * MF_R;
    const bool f32 = attr->dtype == SAFE_DTYPE_F32;

    // ---- the filtered form (six-slice 'sum' columns: three slices on the matrix cores, see k_permtest_mfma)
    const char *filt_env = getenv("SAFE_HIP_MFMA_FILTER");                // =0: all six slices on the matrix cores
    const bool want_filter = allow_filter && !(filt_env && !strcmp(filt_env, "0"));      // (neighborhoods below 2^16 members: mfma_applicable)
    const int64_t split_off = (want_filter && !z) ? n_ct * (n + 1) * (MF_NS / 2) * 32 : 0;   // high digits behind the low digits
    // z-scores, filtered: the seven-slice layout (observed pass) | high digits + not-NaN slice, 128-byte rows | low digits, 96-byte rows
    const int64_t zf_hi_off = (want_filter && z) ? n_ct * (n + 1) * (MF_NS + 1) * 32 : 0;
    const int64_t zf_lo_off = zf_hi_off + n_ct * (n + 1) * 128;
    if (want_filter && z) row_bytes += n_ct * (128 + 96);        // (the scratch below is sized (n + 1) * row_bytes)

    // ---- column scales and slices
    int n_slices = MF_NS;                        // i8 slices of this call: 2 / 4 / 6 by the bits its columns need (k_mfma_colfinish)
    long long *d_q64 = nullptr;                  // (filtered 'sum' form) the fixed-point values as 64-bit words, [n + 1][mloc]: the resolve kernel's operand
    if (split_off) SAFE_TRY(ctx_scratch(ctx, 16, static_cast<size_t>(n + 1) * mloc * sizeof(long long), reinterpret_cast<void **>(&d_q64)));
    longlong2 *d_z64 = nullptr;                  // (filtered z form) value | square + not-NaN flag, [n + 1][mloc]
    if (zf_hi_off) SAFE_TRY(ctx_scratch(ctx, 17, static_cast<size_t>(n + 1) * mloc * sizeof(longlong2), reinterpret_cast<void **>(&d_z64)));
    unsigned char *d_bs = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 1, static_cast<size_t>(n + 1) * row_bytes, reinterpret_cast<void **>(&d_bs)));
    void *d_colbuf = nullptr;                    // maxbits u64 | sumsq f64 | scale, scale2 f64 | cnt, small, rounded, neg_lowbit u32 | shift, shift2 i32 | bad, need i32
    const size_t colbuf_bytes = static_cast<size_t>(mloc) * (8 + 8 + 16 + 4 + 4 + 4 + 4 + 4 + 4 + 8) + 64;   // (+ zeros, inexact u32)
    SAFE_TRY(ctx_scratch(ctx, 6, colbuf_bytes, &d_colbuf));
    unsigned long long *d_max = static_cast<unsigned long long *>(d_colbuf);
    double *d_sumsq = reinterpret_cast<double *>(d_max + mloc);
    double *d_scale = d_sumsq + mloc;
    unsigned int *d_cnt = reinterpret_cast<unsigned int *>(d_scale + 2 * mloc);
    unsigned int *d_small = d_cnt + mloc, *d_rounded = d_small + mloc, *d_lowbit = d_rounded + mloc;
    unsigned int *d_zero = d_lowbit + mloc, *d_inexact = d_zero + mloc;
    int *d_shift = reinterpret_cast<int *>(d_inexact + mloc);
    int *d_bad = d_shift + 2 * mloc, *d_need = d_bad + 1;
    SAFE_HIP_CHECK(hipMemsetAsync(d_colbuf, 0, colbuf_bytes, ctx->stream));
    {
        const int rows_per_block = 2048;
        const dim3 grid(ceil_div(mloc, 32), ceil_div(n, rows_per_block));
        if (f32)
            hipLaunchKernelGGL(k_mfma_colstats<float>, grid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                               attr->col_stride, col0, mloc, rows_per_block, d_max, d_sumsq, d_cnt, d_lowbit);
        else
            hipLaunchKernelGGL(k_mfma_colstats<double>, grid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                               attr->col_stride, col0, mloc, rows_per_block, d_max, d_sumsq, d_cnt, d_lowbit);
        hipLaunchKernelGGL(k_mfma_colfinish, dim3(ceil_div(mloc, 256)), dim3(256), 0, ctx->stream, d_max, d_lowbit, mloc, d_shift,
                           d_scale, d_bad, d_need);
        const dim3 sgrid(n_ct, ceil_div(n + 1, 32));
        if (z) {
            hipLaunchKernelGGL(k_mfma_colfinish_sq, dim3(ceil_div(mloc, 256)), dim3(256), 0, ctx->stream, d_max, d_lowbit, mloc, d_shift + mloc,
                               d_scale + mloc, d_inexact, d_bad);
            if (f32)
                hipLaunchKernelGGL(k_mfma_slice_z<float>, sgrid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                                   attr->col_stride, col0, mloc, n_ct, d_shift, d_shift + mloc, d_max, d_bs, d_small, d_rounded, d_zero, zf_hi_off, zf_lo_off, d_z64);
            else
                hipLaunchKernelGGL(k_mfma_slice_z<double>, sgrid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                                   attr->col_stride, col0, mloc, n_ct, d_shift, d_shift + mloc, d_max, d_bs, d_small, d_rounded, d_zero, zf_hi_off, zf_lo_off, d_z64);
        } else if (f32)
            hipLaunchKernelGGL(k_mfma_slice<float>, sgrid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                               attr->col_stride, col0, mloc, n_ct, d_shift, d_max, d_need, d_bs, d_small, d_rounded, split_off, d_q64);
        else
            hipLaunchKernelGGL(k_mfma_slice<double>, sgrid, dim3(256), 0, ctx->stream, attr->raw, n, attr->row_stride,
                               attr->col_stride, col0, mloc, n_ct, d_shift, d_max, d_need, d_bs, d_small, d_rounded, split_off, d_q64);
        hipLaunchKernelGGL(k_mfma_colcheck, dim3(ceil_div(mloc, 256)), dim3(256), 0, ctx->stream, d_cnt, d_small, d_rounded,
                           z ? d_zero : static_cast<unsigned int *>(nullptr), d_inexact, mloc, d_bad);
        SAFE_HIP_CHECK(hipGetLastError());
        int verdict[2] = {0, 0};                   // {bad, bits needed}
        SAFE_HIP_CHECK(hipMemcpyAsync(verdict, d_bad, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
        const char *force = getenv("SAFE_HIP_FORCE_PATH");
        if (verdict[0] && !(force && !strcmp(force, "mfma"))) {
            *declined = true;
            return SAFE_OK;
        }
        n_slices = z ? MF_NS + 1 : mfma_slices_for(verdict[1]);
        row_bytes = static_cast<int64_t>(n_slices) * 32;          // tile-major: a row of a tile is n_slices x 32 bytes, tiles (n + 1) rows apart
    }
    const bool zfilt = want_filter && z;
    const bool filt = want_filter && !z && n_slices == MF_NS;
    // its own kernel (64 rows per wave, two workgroups per CU) unless a group is too long for its LDS list / a neighborhood too
    // large for its 32-bit thresholds; SAFE_HIP_MFMA_FORM=general: the general kernel's FM = 2 (A/B)
    const char *form_env = getenv("SAFE_HIP_MFMA_FORM");
    const bool filt_own = filt && nbr->bs_max_group_blocks <= MF_F_MAXBLK && nbr->max_count < 2048 && !(form_env && !strcmp(form_env, "general"));
    const int core_slices = filt ? MF_NS / 2 : zfilt ? MF_NS / 2 + 1 : n_slices;         // slices the matrix cores multiply
    if (filt) row_bytes = static_cast<int64_t>(core_slices) * 32;
    const unsigned char *d_bs_lo = zfilt ? d_bs + zf_lo_off : d_bs, *d_bs_hi = zfilt ? d_bs + zf_hi_off : d_bs + split_off;

    // ---- tasks: (row group, column tile), one queue per XCD keyed by column tile so the slice
    //      rows of a tile are pulled into one L2; inside a queue tile-major, heavy groups first
    std::vector<int32_t> g_order(nbr->bs_groups);
    std::iota(g_order.begin(), g_order.end(), 0);
    const std::vector<int32_t> &bp = nbr->h_bs_ptr;
    std::stable_sort(g_order.begin(), g_order.end(), [&](int32_t a, int32_t b) { return bp[a + 1] - bp[a] > bp[b + 1] - bp[b]; });
    std::vector<int2> tasks;
    tasks.reserve(nbr->bs_groups * n_ct);
    int32_t q_off[9] = {0};
    for (int qx = 0; qx < 8; ++qx) {
        for (int64_t ct = qx; ct < n_ct; ct += 8)
            for (int32_t g : g_order) tasks.push_back(make_int2(g, static_cast<int>(ct)));
        q_off[qx + 1] = static_cast<int32_t>(tasks.size());
    }
    int64_t span = 1;
    const std::vector<int64_t> starts = perm_launch_starts(perms, &span);
    const int64_t n_launch = static_cast<int64_t>(starts.size()) - 1;
    void *ws = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 3, tasks.size() * sizeof(int2) + 16 * sizeof(int32_t) + (8 * n_launch + 8 + 16) * sizeof(unsigned int), &ws));
    int2 *d_tasks = static_cast<int2 *>(ws);
    int32_t *d_qoff = reinterpret_cast<int32_t *>(d_tasks + tasks.size());
    unsigned int *d_qctr = reinterpret_cast<unsigned int *>(d_qoff + 16);
    unsigned int *d_counts = nullptr;
    SAFE_TRY(ctx_scratch(ctx, 0, static_cast<size_t>(n_padr) * mloc * sizeof(unsigned int), reinterpret_cast<void **>(&d_counts)));
    int32_t *d_src[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b)
        SAFE_TRY(ctx_scratch(ctx, 4 + b, static_cast<size_t>(span + 1) * n_src * sizeof(int32_t), reinterpret_cast<void **>(&d_src[b])));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_tasks, tasks.data(), tasks.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemcpyAsync(d_qoff, q_off, sizeof(q_off), hipMemcpyHostToDevice, ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_qctr, 0, (8 * n_launch + 8 + 16) * sizeof(unsigned int), ctx->stream));
    SAFE_HIP_CHECK(hipMemsetAsync(d_counts, 0, static_cast<size_t>(n_padr) * mloc * sizeof(unsigned int), ctx->stream));
    // filtered form: exact observed scores, the undecided compares of a launch (one list per stream parity), their counters
    long long *d_obs64 = nullptr;
    ulonglong2 *d_amb[2] = {nullptr, nullptr};
    int32_t *d_src_id = nullptr;
    unsigned int *d_amb_cnt = nullptr;
    unsigned int amb_cap = 0;
    const bool any_filt = filt || zfilt;
    if (any_filt) {
        if (filt) SAFE_TRY(ctx_scratch(ctx, 12, static_cast<size_t>(n_padr) * mloc * sizeof(long long), reinterpret_cast<void **>(&d_obs64)));
        const double per_launch = static_cast<double>(n) * static_cast<double>(mloc) * static_cast<double>(span);
        amb_cap = static_cast<unsigned int>(std::min(67108864.0, std::max(1048576.0, per_launch / 512.0)));
        if (const char *e = getenv("SAFE_HIP_MFMA_FILTER_CAP")) amb_cap = static_cast<unsigned int>(std::max(1, atoi(e)));   // (tests: force the fall-back)
        for (int b = 0; b < 2; ++b)
            SAFE_TRY(ctx_scratch(ctx, 13 + b, static_cast<size_t>(amb_cap) * sizeof(ulonglong2), reinterpret_cast<void **>(&d_amb[b])));
        void *small = nullptr;
        SAFE_TRY(ctx_scratch(ctx, 15, static_cast<size_t>(n_src) * sizeof(int32_t) + static_cast<size_t>(n_launch) * sizeof(unsigned int), &small));
        d_src_id = static_cast<int32_t *>(small);
        d_amb_cnt = reinterpret_cast<unsigned int *>(d_src_id + n_src);
        SAFE_HIP_CHECK(hipMemsetAsync(d_amb_cnt, 0, static_cast<size_t>(n_launch) * sizeof(unsigned int), ctx->stream));
    }

    const size_t lds_bytes = 2 * static_cast<size_t>(4 * mf_ks(core_slices)) + MF_MAXBLK * sizeof(int32_t) + 16 * 512 * sizeof(long long);
    const void *kfn_obs = reinterpret_cast<const void *>(k_permtest_mfma<false, MF_NS / 2, false, true, 0, true, 1>);
    const void *kfn_zobs = reinterpret_cast<const void *>(k_permtest_mfma<false, MF_NS + 1, true>);     // z-scores, all seven slices
    const size_t lds_zobs = 2 * static_cast<size_t>(4 * mf_ks(MF_NS + 1)) + MF_MAXBLK * sizeof(int32_t) + 16 * 512 * sizeof(long long);
    const void *kfn = zfilt           ? reinterpret_cast<const void *>(k_permtest_mfma<false, MF_NS / 2 + 1, true, true, 0, true, 2>)
                      : z             ? kfn_zobs
                      : filt          ? reinterpret_cast<const void *>(k_permtest_mfma<false, MF_NS / 2, false, true, 0, true, 2>)
                      : n_slices == 2 ? reinterpret_cast<const void *>(k_permtest_mfma<false, 2>)
                      : n_slices == 4 ? reinterpret_cast<const void *>(k_permtest_mfma<false, 4>)
                                      : reinterpret_cast<const void *>(k_permtest_mfma<false, MF_NS>);
    SAFE_HIP_CHECK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    if (filt) SAFE_HIP_CHECK(hipFuncSetAttribute(kfn_obs, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    if (zfilt) SAFE_HIP_CHECK(hipFuncSetAttribute(kfn_zobs, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_zobs)));
#ifdef SAFE_HIP_DIAG
    static const int mfma_dbg = getenv("SAFE_HIP_MFMA_DBG") ? atoi(getenv("SAFE_HIP_MFMA_DBG")) : 0;
    if (mfma_dbg) safe_warn_diagnostic("SAFE_HIP_MFMA_DBG");
#else
    constexpr int mfma_dbg = 0;
#endif
    // SAFE_HIP_MFMA_FORM=f: round 5's kernel (register-staged gather); default: k_permtest_mfma_g (LDS-DMA gather)
    const bool form_f = form_env && !strcmp(form_env, "f");
    const void *kfn_own = !form_f                            ? reinterpret_cast<const void *>(k_permtest_mfma_g<0>)
                                                             : reinterpret_cast<const void *>(k_permtest_mfma_f<0>);
#ifdef SAFE_HIP_DIAG
#define MF_F_DIAG(D) if (form_f && mfma_dbg == D) kfn_own = reinterpret_cast<const void *>(k_permtest_mfma_f<D>);
    MF_F_DIAG(1) MF_F_DIAG(2) MF_F_DIAG(4) MF_F_DIAG(8) MF_F_DIAG(15) MF_F_DIAG(31) MF_F_DIAG(47) MF_F_DIAG(79) MF_F_DIAG(271) MF_F_DIAG(127) MF_F_DIAG(383) MF_F_DIAG(511) MF_F_DIAG(512) MF_F_DIAG(1024)
#undef MF_F_DIAG
#define MF_G_DIAG(D) if (!form_f && mfma_dbg == D) kfn_own = reinterpret_cast<const void *>(k_permtest_mfma_g<D>);
    MF_G_DIAG(2) MF_G_DIAG(4) MF_G_DIAG(8) MF_G_DIAG(64) MF_G_DIAG(66) MF_G_DIAG(512) MF_G_DIAG(1024)
#undef MF_G_DIAG
#endif
    // filtered z-scores: their own kernel in the same shape (SAFE_HIP_MFMA_FORM=general keeps the general kernel's FM = 2)
    const bool zfilt_own = zfilt && nbr->bs_max_group_blocks <= MF_GZ_MAXBLK && !(form_env && (!strcmp(form_env, "general") || !strcmp(form_env, "f")));
    const void *kfn_gz = (form_env && !strcmp(form_env, "gz64")) ? reinterpret_cast<const void *>(k_permtest_mfma_gz<0, false>)   // (A/B: the f64 test)
                                                                 : reinterpret_cast<const void *>(k_permtest_mfma_gz<0>);
#ifdef SAFE_HIP_DIAG
#define MF_GZ_DIAG(D) if (mfma_dbg == D) kfn_gz = reinterpret_cast<const void *>(k_permtest_mfma_gz<D>);
    MF_GZ_DIAG(2) MF_GZ_DIAG(4) MF_GZ_DIAG(8) MF_GZ_DIAG(64) MF_GZ_DIAG(512)
#undef MF_GZ_DIAG
#endif
    const size_t lds_gz = 2 * static_cast<size_t>(4 * 32 * (MF_NS / 2 + 1) * 32) + MF_GZ_MAXBLK * sizeof(int32_t) + 16 * 256 * sizeof(long long) + 8 * 256 * sizeof(uint32_t) + 16;
    if (zfilt_own) SAFE_HIP_CHECK(hipFuncSetAttribute(kfn_gz, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_gz)));
    const size_t lds_own = form_f ? 2 * static_cast<size_t>(4 * (MF_NS / 2) * MF_SS) + MF_F_MAXBLK * sizeof(int32_t) + 32 * 256 * sizeof(int32_t) + 8 * 256 * sizeof(uint32_t)
                                  : 2 * static_cast<size_t>(4 * 32 * (MF_NS / 2) * 32) + MF_F_MAXBLK * sizeof(int32_t) + 32 * 256 * sizeof(int32_t) + 8 * 256 * sizeof(uint32_t) + 16;
    const uint4 *bits_own = form_f ? nbr->bs_bits4 : nbr->bs_bits4p;
    if (filt_own)
        SAFE_HIP_CHECK(hipFuncSetAttribute(kfn_own, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_own)));
    ctx->last_slices = n_slices;
    ctx->last_core_slices = core_slices;
    ctx->last_undecided = 0;
    // The kernel is persistent and takes a CU whole (256 VGPRs x 2 waves per SIMD): table kernels of the next span that are
    // queued while it runs (aux stream: scan rounds, row emission) get a CU only as its workgroups retire and trickle through
    // the whole tail of the launch -- harmless for the result, but they then show launch-long durations in a kernel trace
    // (half of the "GPU time" of a profile).  Leaving CUs free does not help unless every XCD has one (workgroups are dealt to
    // XCDs round-robin), which costs 3 % of the matrix-core throughput.  Long launches therefore make the table stream wait for
    // their predecessor instead (below): the tables of a span are needed only when that launch has ended.
    const int64_t blocks = std::min<int64_t>(static_cast<int64_t>(tasks.size()), ctx->num_cu);
    if (zfilt) row_bytes = 128;                                // the filtered z form's rows: three value | square slices + the not-NaN slice
    const int64_t tile_bytes = (n + 1) * row_bytes;
    const bool long_launches = static_cast<double>(n) * static_cast<double>(mloc) * static_cast<double>(span) >= 2e9;   // >~ 20 ms each
    // z-scores: the counters compare against the observed score itself, which may be NaN (k_counts_finalize<true> reads it)
    if (z && !out.ns) SAFE_TRY(ctx_scratch(ctx, 2, static_cast<size_t>(n) * mloc * sizeof(double), reinterpret_cast<void **>(&out.ns)));
    ctx->last_kernel.name = "k_permtest_mfma";
    ctx->last_kernel.total_ms = 0.0;
    ctx->last_kernel.busy_ms = 0.0;
    ctx->last_kernel.launches = 0;
    hipEvent_t *ev = nullptr, *plain = nullptr;                   // pooled on the context
    SAFE_TRY(ctx_events(ctx, true, 2 * n_launch, &ev));
    SAFE_TRY(ctx_events(ctx, false, 2, &plain));
    hipEvent_t ready = plain[0], side_done = plain[1];
    if (filt) {
        // the exact observed scores: low digits stored, high digits added << 24 (two short launches over the identity map)
        hipLaunchKernelGGL(k_mfma_src, dim3(ceil_div(n_src, 256), 1), dim3(256), 0, ctx->stream, nbr->bs_order, n_src, n,
                           static_cast<const int32_t *>(nullptr), 0, d_src_id, 0);
        for (int pass = 0; pass < 2; ++pass) {
            const unsigned char *bs_p = pass ? d_bs_hi : d_bs_lo;
            const int32_t *src_c = d_src_id;
            int n_q = 1;
            unsigned int *qctr_c = d_qctr + 8 * n_launch + 8 + 8 * pass;
            double *ns_c = nullptr;
            HypLookup no_lookup{};
            MfmaFilt fa;
            fa.obs64 = d_obs64;
            fa.obs_shift = pass ? 24 : 0;
            void *args[] = {(void *)&bs_p, (void *)&row_bytes, (void *)&tile_bytes, (void *)&src_c, (void *)&n_src, (void *)&n_q, (void *)&nbr->bs_ptr,
                            (void *)&nbr->bs_kb, (void *)&nbr->bs_bits, (void *)&d_tasks, (void *)&d_qoff, (void *)&qctr_c, (void *)&mloc,
                            (void *)&d_counts, (void *)&n_padr, (void *)&nbr->bs_rowmap, (void *)&d_scale, (void *)&ns_c, (void *)&no_lookup, (void *)&fa};
            SAFE_HIP_CHECK(hipLaunchKernel(kfn_obs, dim3(blocks), dim3(512), args, lds_bytes, ctx->stream));
        }
        if (out.ns)
            hipLaunchKernelGGL(k_mfma_obs_ns, dim3(n_padr / 32, ceil_div(mloc, 32)), dim3(256), 0, ctx->stream, d_obs64, n_padr, nbr->bs_rowmap,
                               d_scale, mloc, out.ns);
        SAFE_HIP_CHECK(hipGetLastError());
    }
    if (zfilt) {
        // the observed z-scores: one pass of the seven-slice kernel over the identity map writes them to out.ns (exact sums, the
        // reference's formula); the filtered launches compare against them
        hipLaunchKernelGGL(k_mfma_src, dim3(ceil_div(n_src, 256), 1), dim3(256), 0, ctx->stream, nbr->bs_order, n_src, n,
                           static_cast<const int32_t *>(nullptr), 0, d_src_id, 0);
        const unsigned char *bs_p = d_bs;
        int64_t rb7 = (MF_NS + 1) * 32, tb7 = (n + 1) * rb7;
        const int32_t *src_c = d_src_id;
        int n_q = 1;
        unsigned int *qctr_c = d_qctr + 8 * n_launch + 8;
        double *ns_c = out.ns;
        HypLookup no_lookup{};
        MfmaFilt fa;
        void *args[] = {(void *)&bs_p, (void *)&rb7, (void *)&tb7, (void *)&src_c, (void *)&n_src, (void *)&n_q, (void *)&nbr->bs_ptr,
                        (void *)&nbr->bs_kb, (void *)&nbr->bs_bits, (void *)&d_tasks, (void *)&d_qoff, (void *)&qctr_c, (void *)&mloc,
                        (void *)&d_counts, (void *)&n_padr, (void *)&nbr->bs_rowmap, (void *)&d_scale, (void *)&ns_c, (void *)&no_lookup, (void *)&fa};
        SAFE_HIP_CHECK(hipLaunchKernel(kfn_zobs, dim3(blocks), dim3(512), args, lds_zobs, ctx->stream));
        SAFE_HIP_CHECK(hipGetLastError());
    }
    SAFE_HIP_CHECK(hipEventRecord(ready, ctx->stream));
    SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ready, 0));
    for (int64_t c = 0; c < n_launch; ++c) {
        const int64_t p_base = starts[c], p_limit = starts[c + 1], cnt = p_limit - p_base;
        hipStream_t ks = (c & 1) ? ctx->side_stream : ctx->stream;
        if (long_launches && c >= 1) SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->aux_stream, ev[2 * (c - 1) + 1], 0));
        SAFE_TRY(perms_wait(perms, p_limit, ks));
#ifdef SAFE_HIP_DIAG
        static const int dbg_window = getenv("SAFE_HIP_MFMA_DBG_WINDOW") ? atoi(getenv("SAFE_HIP_MFMA_DBG_WINDOW")) : 0;
        if (dbg_window > 0) safe_warn_diagnostic("SAFE_HIP_MFMA_DBG_WINDOW");
#else
        constexpr int dbg_window = 0;
#endif
        hipLaunchKernelGGL(k_mfma_src, dim3(ceil_div(n_src, 256), cnt + 1), dim3(256), 0, ks, nbr->bs_order, n_src, n, perms->table,
                           p_base, d_src[c & 1], dbg_window);
        SAFE_HIP_CHECK(hipEventRecord(ev[2 * c], ks));
        {
            // classic: row 0 of the source maps is the identity (the observed score is formed once per task); filtered form:
            // the observed scores are in d_obs64 and every q is a permutation
            const int32_t *src_c = any_filt ? d_src[c & 1] + n_src : d_src[c & 1];
            int n_q = static_cast<int>(any_filt ? cnt : cnt + 1);
            unsigned int *qctr_c = d_qctr + 8 * c;
            double *ns_c = (c == 0 && !any_filt) ? out.ns : static_cast<double *>(nullptr);
            HypLookup no_lookup{};
            no_lookup.dbg = mfma_dbg;        // 1: no transposes / LDS stores of the gathered rows, 4: no barrier per super-step, 8: no score completion
            MfmaFilt fa;
            if (any_filt) {
                fa.obs64 = d_obs64;
                fa.zobs = out.ns;
                fa.rowcnt = nbr->bs_rowcnt;
                fa.amb = d_amb[c & 1];
                fa.amb_count = d_amb_cnt + c;
                fa.amb_cap = amb_cap;
                fa.p_base = static_cast<int>(p_base);
#ifdef SAFE_HIP_DIAG
                if (mfma_dbg == 512) {
                    if (!ctx->diag_prof) SAFE_HIP_CHECK(hipMalloc(&ctx->diag_prof, 64 * sizeof(unsigned long long)));
                    if (c == 0) SAFE_HIP_CHECK(hipMemsetAsync(ctx->diag_prof, 0, 64 * sizeof(unsigned long long), ks));
                    fa.prof = static_cast<unsigned long long *>(ctx->diag_prof);
                }
#endif
            }
            const unsigned char *bs_main = any_filt ? d_bs_hi : d_bs;
            if (zfilt_own) {
                const int64_t blocks_own = std::min<int64_t>(static_cast<int64_t>(tasks.size()), 2 * static_cast<int64_t>(ctx->num_cu));
                void *args[] = {(void *)&bs_main, (void *)&tile_bytes, (void *)&src_c, (void *)&n_src, (void *)&n_q, (void *)&nbr->bs_ptr,
                                (void *)&nbr->bs_kb, (void *)&nbr->bs_bits4p, (void *)&d_tasks, (void *)&d_qoff, (void *)&qctr_c,
                                (void *)&mloc, (void *)&d_counts, (void *)&n_padr, (void *)&nbr->bs_rowmap, (void *)&d_scale, (void *)&fa};
                SAFE_HIP_CHECK(hipLaunchKernel(kfn_gz, dim3(blocks_own), dim3(256), args, lds_gz, ks));
            } else if (filt_own) {
                const int64_t blocks_own = std::min<int64_t>(static_cast<int64_t>(tasks.size()), 2 * static_cast<int64_t>(ctx->num_cu));
                void *args[] = {(void *)&bs_main, (void *)&tile_bytes, (void *)&src_c, (void *)&n_src, (void *)&n_q, (void *)&nbr->bs_ptr,
                                (void *)&nbr->bs_kb, (void *)&bits_own, (void *)&nbr->bs_grpmax, (void *)&d_tasks, (void *)&d_qoff, (void *)&qctr_c,
                                (void *)&mloc, (void *)&d_counts, (void *)&n_padr, (void *)&fa};
                SAFE_HIP_CHECK(hipLaunchKernel(kfn_own, dim3(blocks_own), dim3(256), args, lds_own, ks));
            } else {
            void *args[] = {(void *)&bs_main, (void *)&row_bytes, (void *)&tile_bytes, (void *)&src_c, (void *)&n_src, (void *)&n_q, (void *)&nbr->bs_ptr,
                            (void *)&nbr->bs_kb, (void *)&nbr->bs_bits, (void *)&d_tasks, (void *)&d_qoff, (void *)&qctr_c, (void *)&mloc,
                            (void *)&d_counts, (void *)&n_padr, (void *)&nbr->bs_rowmap, (void *)&d_scale, (void *)&ns_c, (void *)&no_lookup, (void *)&fa};
            SAFE_HIP_CHECK(hipLaunchKernel(kfn, dim3(blocks), dim3(512), args, lds_bytes, ks));
            }
        }
        SAFE_HIP_CHECK(hipGetLastError());
        SAFE_HIP_CHECK(hipEventRecord(ev[2 * c + 1], ks));
        if (any_filt) {
            if (zfilt)
                hipLaunchKernelGGL(k_mfma_resolve_z, dim3(4 * ctx->num_cu), dim3(256), 0, ks, d_amb[c & 1], d_amb_cnt + c, amb_cap, out.ns, mloc, n_padr,
                                   nbr->bs_rowmap, nbr->row_ptr, nbr->col, perms->table, n, d_bs_lo, d_bs_hi, d_z64, d_scale, d_counts);
            else
                hipLaunchKernelGGL(k_mfma_resolve, dim3(4 * ctx->num_cu), dim3(256), 0, ks, d_amb[c & 1], d_amb_cnt + c, amb_cap, d_obs64, n_padr,
                                   nbr->bs_rowmap, nbr->row_ptr, nbr->col, perms->table, n, d_bs_lo, tile_bytes, split_off, d_q64, mloc, d_counts);
            SAFE_HIP_CHECK(hipGetLastError());
            if (c == 0 && long_launches && n_launch > 1) {
                // pilot: data with many equal scores (sparse columns, few distinct values) leaves the high digits little to
                // decide -- if the first launch sent more than 2 in 1000 compares to the resolve kernel, stop here and let the
                // caller run all six slices (a launch of this size is long: the wait is nothing beside it)
                unsigned int seen0 = 0;
                SAFE_HIP_CHECK(hipMemcpyAsync(&seen0, d_amb_cnt, sizeof(seen0), hipMemcpyDeviceToHost, ks));
                SAFE_HIP_CHECK(safe_stream_sync(ks));
                if (static_cast<double>(seen0) > 2e-3 * static_cast<double>(n) * static_cast<double>(mloc) * static_cast<double>(cnt)) {
                    ctx->last_undecided = seen0;
                    *overflowed = true;
                    return SAFE_OK;
                }
            }
        }
        if (c >= 1) {
            // the source-map buffer of span c-1 is reused by span c+1: same stream, ordered
        }
    }
    SAFE_HIP_CHECK(hipEventRecord(side_done, ctx->side_stream));
    SAFE_HIP_CHECK(hipStreamWaitEvent(ctx->stream, side_done, 0));
    std::vector<unsigned int> amb_seen(any_filt ? n_launch : 0, 0u);
    if (any_filt) SAFE_HIP_CHECK(hipMemcpyAsync(amb_seen.data(), d_amb_cnt, amb_seen.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    SAFE_TRY(enrich_finalize_counts(ctx, d_counts, n_padr, nbr->bs_rowmap, mloc, P, out, z ? out.ns : nullptr));
    if (!z) {                                                 // (z-score counters depend on NaN observed scores: not exported)
        ctx->packed_counts = d_counts;
        ctx->packed_n_pad = n_padr;
        ctx->packed_m = mloc;
        ctx->packed_perms = P;
        ctx->packed_layout = 1;
    }
    SAFE_HIP_CHECK(hipEventRecord(ctx->k0, ctx->stream));
    SAFE_HIP_CHECK(hipEventRecord(ctx->k1, ctx->stream));
    SAFE_HIP_CHECK(safe_stream_sync(ctx->stream));
    SAFE_TRY(kernel_stat_from_events(ctx, ev, n_launch));
#ifdef SAFE_HIP_DIAG
    if (mfma_dbg == 512 && ctx->diag_prof) {
        unsigned long long prof[64];
        SAFE_HIP_CHECK(hipMemcpy(prof, ctx->diag_prof, sizeof(prof), hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) {
            const double it = static_cast<double>(std::max<unsigned long long>(prof[w * 8 + 4], 1));
            fprintf(stderr, "mfma_f wave %d: %.0f iterations; cycles per iteration: loads %.0f, k-loop %.0f, completion %.0f, barrier %.0f\n", w, it,
                    prof[w * 8] / it, prof[w * 8 + 1] / it, prof[w * 8 + 2] / it, prof[w * 8 + 3] / it);
        }
    }
#endif
    for (unsigned int seen : amb_seen) {
        ctx->last_undecided += seen;
        if (seen > amb_cap) *overflowed = true;          // a launch left more undecided compares than its list holds
    }
    return SAFE_OK;
}

int launch_mfma(safe_ctx *ctx, safe_nbr *nbr, safe_attr *attr, safe_perms *perms, int64_t col0, int64_t col1, bool z,
                const PermOut &out_in, bool *declined) {
    bool overflowed = false;
    SAFE_TRY(launch_mfma_run(ctx, nbr, attr, perms, col0, col1, z, out_in, declined, true, &overflowed));
    if (overflowed) {
        // data with many (near-)equal scores: the filter decides too little -- the whole call again with all slices on the
        // matrix cores (counters, scores and outputs are rewritten from scratch)
        safe_trace("matrix-core filter: too many undecided compares, running the six-slice form");
        const int64_t undecided = ctx->last_undecided;
        if (out_in.enriched)                                  // (the abandoned pass may have counted its hits already)
            SAFE_HIP_CHECK(hipMemsetAsync(out_in.enriched, 0, static_cast<size_t>(col1 - col0) * sizeof(unsigned int), ctx->stream));
        SAFE_TRY(launch_mfma_run(ctx, nbr, attr, perms, col0, col1, z, out_in, declined, false, &overflowed));
        ctx->last_undecided = -undecided;                // (negative: the filtered pass was abandoned)
    }
    return SAFE_OK;
}
