// cczero_conv4.h -- the tower convolution on the GROUP-OF-16 activation layout, skipping the taps that leave the board.
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// 110 of the 810 (pixel, tap) pairs of a 10 x 9 board point off the board (13.6 %): k_conv3x3_c256 multiplies a zero row for
// them, because a 16-pixel MFMA column block of consecutive pixels of ONE board never has the same tap off the board for all
// 16 pixels. This form changes what the 16 columns are:
//
//   * layout "G16": activation row index = (g * 90 + pos) * 16 + j for board 16 g + j -- 16 consecutive rows = ONE board
//     position ("cell") of 16 boards. An MFMA column block is a cell: a tap is on the board for all 16 columns or for none,
//     which the workgroup knows in SCALAR registers (9 masks of 16 bits per 256-row tile), so the MFMAs of an off-board
//     (cell, tap) are skipped by a scalar branch. No per-lane validity flags, no v_cndmask, no zero rows; the slab offset
//     of a tap is 16 * (9 dy + dx) rows for every lane.
//   * work split as in cczero_conv3.h (the only one in which every wave skips the same MFMAs at the same time, so that the
//     saving is not lost at a barrier): 8 waves = 8 x 32 output channels, every wave covers all 16 cells of the tile
//     (2 x 16 accumulator tiles); weights straight from global memory into registers two half-steps ahead; the activation
//     slab (tile + 10 cells either side = 576 rows of 128 B per 64-channel chunk, double-buffered: 144 KB) by LDS-DMA;
//     two barriers per chunk.
//   * K order per accumulator = that of k_conv3x3_c256 (chunks of 64 x 9 taps x 2 halves of 32) minus the skipped taps,
//     whose products are exact zeros there: the same values (up to the sign of a zero).
#pragma once
#include "../../chinesechesszero_amd/csrc/cczero_conv.h"

namespace ccz {

constexpr int kC4Cells = 36;                              // slab cells: the tile's 16 + 10 either side
constexpr int kC4SlabBytes = kC4Cells * 2048;             // 73,728 B per 64-channel chunk
constexpr int kC4ERow = 528;                              // epilogue: bytes per pixel row (512 + pad)
constexpr int kC4Lds = 2 * kC4SlabBytes;                  // 147,456 B
static_assert(256 * kC4ERow <= kC4Lds, "epilogue image must fit the slabs");
#ifndef C4_LOOK
#define C4_LOOK 4
#define C4_WIN 6
#endif
constexpr int kC4Look = C4_LOOK;                          // pixel fragments requested ahead of the MFMAs that use them
constexpr int kC4Win = C4_WIN;                            // register slots of the rolling window; must divide 288
static_assert(288 % kC4Win == 0 && kC4Look < kC4Win && kC4Look <= 8, "window geometry");

struct C4Ctx {
    unsigned char *lds;
    const _Float16 *X;
    int xsrc[9];          // per staging pass: element offset of this thread's 16-byte source in X (chunk 0)
    int wave_dst;         // w * 1024
    int lane0;            // r * 128 + ((q4 ^ (r & 7)) << 4): this lane's fragment inside a cell of the slab (k-half 0; half 1: ^ 64)
    unsigned m[9];        // SCALAR: bit n of m[t] = tap t of cell n of this tile stays on the board
    const _Float16 *wa;   // this lane's weight source: row 32 w + (lane & 15), k-chunk lane >> 4 (tap 0, chunk 0, half 0)
    int cin, cmask;
};

// LDS offset of this lane's k-half-0 fragment of cell 0 at tap TAP in slab buffer `buf`
template <int TAP> __device__ __forceinline__ int c4_tap(const C4Ctx &c, int buf)
{
    constexpr int delta = 9 * (TAP / 3 - 1) + (TAP % 3 - 1);
    int l0 = c.lane0;
    asm volatile("" : "+v"(l0)); // computed per tap inside the loop: hoisted for 9 taps x 2 buffers it costs registers
    return l0 + (buf * kC4SlabBytes + (kCvHalo + delta) * 2048);
}

template <int KH, int N> __device__ __forceinline__ cv_half8 c4_read_x(const C4Ctx &c, int tapbase)
{
    return *(const cv_half8 *)(c.lds + (tapbase ^ (KH << 6)) + N * 2048);
}

template <int U2> __device__ __forceinline__ void c4_load_w(const C4Ctx &c, int chunk2, cv_half8 (&a)[2])
{
    const _Float16 *s = c.wa + (U2 >> 1) * c.cin + chunk2 * 64 + (U2 & 1) * 32;
    a[0] = *(const cv_half8 *)s;
    a[1] = *(const cv_half8 *)(s + 16l * (9 * c.cin));
}

template <int U>
__device__ __forceinline__ void c4_halfstep(const C4Ctx &c, cv_f32x4 (&acc)[2][16], int chunk, int &tapbase, const cv_half8 (&acur)[2],
                                             cv_half8 (&aload)[2], cv_half8 (&bw)[kC4Win])
{
    constexpr int KH = U & 1, TAP = U >> 1;
    constexpr int Un = (U + 1) % 18, KHn = Un & 1, TAPn = Un >> 1;
    unsigned char *const lds = c.lds;

    if constexpr (U == 2) { // nobody still reads the buffer the staging pieces issued from here on overwrite
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (U >= 2 && U <= 10) { // the next chunk's slab: 9 pieces per thread, one per half-step
        constexpr int pass = U - 2;
        const int nxt = (chunk + 1) & c.cmask; // past the last chunk: re-stage chunk 0 into the free buffer (keeps the code static)
#ifdef C4_HALFSLAB /* timing ablation: half of the slab pieces */
        if constexpr (pass & 1)
#endif
        cv_glds16(c.X + (c.xsrc[pass] + nxt * 64), lds + ((chunk + 1) & 1) * kC4SlabBytes + pass * 8192 + c.wave_dst);
    }
    {
        constexpr int U2 = (U + 2) % 18;
        c4_load_w<U2>(c, (chunk + (U + 2 >= 18 ? 1 : 0)) & c.cmask, aload);
    }
    if constexpr (U == 17) { // chunk boundary: the slab pieces (older than the four youngest loads) have landed for every wave
        cv_wait_vm<4>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned mask = c.m[TAP];
    asm volatile("" : "+s"(mask)); // the 16 tests stay scalar bit tests next to their branches (hoisted out of the chunk loop they become 144 lane masks)
#ifdef C4_NOGUARD /* timing ablation: every MFMA executes (wrong results at the board edges) */
#define C4_GUARD(m, n) true
#else
#define C4_GUARD(m, n) ((m) & (1u << (n)))
#endif
#define C4_TILE(N)                                                                                                        \
    {                                                                                                                     \
        constexpr int slot = (U * 16 + N) % kC4Win, slot_rd = (U * 16 + N + kC4Look) % kC4Win;                            \
        if constexpr (N == 16 - kC4Look && KH == 1) tapbase = c4_tap<TAPn>(c, (chunk + (U == 17 ? 1 : 0)) & 1);          \
        if constexpr (N + kC4Look < 16) bw[slot_rd] = c4_read_x<KH, (N + kC4Look) % 16>(c, tapbase);                      \
        else bw[slot_rd] = c4_read_x<KHn, (N + kC4Look) % 16>(c, tapbase);                                                \
        if (C4_GUARD(mask, N)) {                                                                                          \
            acc[0][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[0], bw[slot], acc[0][N], 0, 0, 0);                    \
            acc[1][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[1], bw[slot], acc[1][N], 0, 0, 0);                    \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
    C4_TILE(0) C4_TILE(1) C4_TILE(2) C4_TILE(3) C4_TILE(4) C4_TILE(5) C4_TILE(6) C4_TILE(7)
    C4_TILE(8) C4_TILE(9) C4_TILE(10) C4_TILE(11) C4_TILE(12) C4_TILE(13) C4_TILE(14) C4_TILE(15)
#undef C4_TILE
}

template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_v4(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                        const float *__restrict__ bias, const _Float16 *R,
                                                        _Float16 *Y, int M, int relu, int cin)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kC4Lds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int tile = __builtin_amdgcn_readfirstlane((relu & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x);
    const long p0 = (long)tile * kCvBM;
    relu &= 1;

    C4Ctx c;
    c.lds = lds;
    c.X = X;
    c.wave_dst = w * 1024;
    c.cin = cin;
    c.cmask = (cin >> 6) - 1;
    c.lane0 = r * 128 + ((q4 ^ (r & 7)) << 4);
    {
        const int srow = tid >> 3, cpos = tid & 7;
        const int schunk = cpos ^ (srow & 7); // slab row r holds source chunk c at position c ^ (r & 7)
#pragma unroll
        for (int it = 0; it < 9; ++it) {
            long p = p0 - kCvHalo * 16 + it * 64 + srow;
            p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
            c.xsrc[it] = (int)(p * cin + schunk * 8);
        }
    }
    c.wa = W + (long)(w * 32 + r) * (9 * cin) + q4 * 8;
    // ---- prologue: slab of chunk 0 (DMA), the weights of half-steps 0 and 1; the setup below runs while they are in flight
#pragma unroll
    for (int it = 0; it < 9; ++it) cv_glds16(X + c.xsrc[it], lds + it * 8192 + c.wave_dst);
    cv_half8 a0[2], a1[2], a2[2];
    c4_load_w<0>(c, 0, a0);
    c4_load_w<1>(c, 0, a1);
    {   // which taps of which cells stay on the board: scalar
        unsigned r0 = 0, r9 = 0, f0 = 0, f8 = 0;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int pos = (tile * 16 + n) % 90, rank = pos / 9, file = pos - rank * 9;
            r0 |= (rank == 0 ? 1u : 0u) << n;
            r9 |= (rank == 9 ? 1u : 0u) << n;
            f0 |= (file == 0 ? 1u : 0u) << n;
            f8 |= (file == 8 ? 1u : 0u) << n;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            const unsigned off = (dy < 0 ? r0 : dy > 0 ? r9 : 0u) | (dx < 0 ? f0 : dx > 0 ? f8 : 0u);
            c.m[t] = __builtin_amdgcn_readfirstlane(0xffffu & ~off);
        }
    }

    // the accumulators start at the bias: nothing left to add in the epilogue
    cv_f32x4 acc[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 bv = *(const float4 *)(bias + w * 32 + i * 16 + 4 * q4);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            acc[i][n][0] = bv.x; acc[i][n][1] = bv.y; acc[i][n][2] = bv.z; acc[i][n][3] = bv.w;
        }
    }

    cv_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    cv_half8 bw[kC4Win];
    int tapbase = c4_tap<0>(c, 0);
    bw[0] = c4_read_x<0, 0>(c, tapbase);
    if constexpr (kC4Look > 1) bw[1] = c4_read_x<0, 1>(c, tapbase);
    if constexpr (kC4Look > 2) bw[2] = c4_read_x<0, 2>(c, tapbase);
    if constexpr (kC4Look > 3) bw[3] = c4_read_x<0, 3>(c, tapbase);
    if constexpr (kC4Look > 4) bw[4] = c4_read_x<0, 4>(c, tapbase);
    if constexpr (kC4Look > 5) bw[5] = c4_read_x<0, 5>(c, tapbase);
    if constexpr (kC4Look > 6) bw[6] = c4_read_x<0, 6>(c, tapbase);
    if constexpr (kC4Look > 7) bw[7] = c4_read_x<0, 7>(c, tapbase);
    for (int chunk = 0; chunk <= c.cmask; ++chunk) {
#define C4_H(u, cur, ld) c4_halfstep<u>(c, acc, chunk, tapbase, cur, ld, bw)
        C4_H(0, a0, a2); C4_H(1, a1, a0); C4_H(2, a2, a1); C4_H(3, a0, a2); C4_H(4, a1, a0); C4_H(5, a2, a1);
        C4_H(6, a0, a2); C4_H(7, a1, a0); C4_H(8, a2, a1); C4_H(9, a0, a2); C4_H(10, a1, a0); C4_H(11, a2, a1);
        C4_H(12, a0, a2); C4_H(13, a1, a0); C4_H(14, a2, a1); C4_H(15, a0, a2); C4_H(16, a1, a0); C4_H(17, a2, a1);
#undef C4_H
    }
    cv_wait_vm<0>(); // the wrapped-around DMA and weight loads must land before the LDS is reused / released

    // ---- epilogue: every wave writes its 32 channels x 256 rows into the [row][channel] image in LDS; then wave w owns
    // rows 32 w .. 32 w + 31 and moves whole 512-byte rows (residual in, output out)
    const int prow = lane >> 5, piece = lane & 31;
    const long pbase = p0 + w * 32 + prow;
    const bool full = p0 + kCvBM <= M;
    cv_half8 rv[16];
    if (RES && full) {
#pragma unroll
        for (int it = 0; it < 16; ++it) rv[it] = *(const cv_half8 *)(R + (pbase + it * 2) * kCvC + piece * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier(); // every wave is done reading the slabs
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = w * 32 + i * 16 + 4 * q4;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            cv_half4 o;
            o[0] = (_Float16)acc[i][n][0];
            o[1] = (_Float16)acc[i][n][1];
            o[2] = (_Float16)acc[i][n][2];
            o[3] = (_Float16)acc[i][n][3];
            *(cv_half4 *)(lds + (n * 16 + r) * kC4ERow + col * 2) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const cv_half8 zero = (cv_half8)(_Float16)0;
        const unsigned char *eb = lds + (w * 32 + prow) * kC4ERow + piece * 16;
        if (full) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                cv_half8 v = *(const cv_half8 *)(eb + it * 2 * kC4ERow);
                if (RES) v = v + rv[it];
                if (relu) v = __builtin_elementwise_max(v, zero);
                *(cv_half8 *)(Y + (pbase + it * 2) * kCvC + piece * 8) = v;
            }
        } else {
            for (int it = 0; it < 16; ++it) {
                const long p = pbase + it * 2;
                if (p >= M) break;
                cv_half8 v = *(const cv_half8 *)(eb + it * 2 * kC4ERow);
                if (RES) v = v + *(const cv_half8 *)(R + p * kCvC + piece * 8);
                if (relu) v = __builtin_elementwise_max(v, zero);
                *(cv_half8 *)(Y + p * kCvC + piece * 8) = v;
            }
        }
    }
}

} // namespace ccz
