// cczero_conv_small.h -- the tower convolution for SMALL batches (one game at a time: MCTS_AI, the UCI loop; up to a few dozen boards).
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// k_conv3x3_c256 (cczero_conv.h) works on 256-pixel tiles, one workgroup per compute unit: a single board (90 pixels) is one
// partial tile on ONE of 256 CUs (44 us per layer), which is why batches under 192 boards used to fall back to MIOpen + an
// epilogue pass (11-12 us per layer at one board, two launches). This kernel spreads a small batch over the chip instead:
//
//   * one workgroup = 4 waves = a 16-output-channel x 64-pixel block, one 16-pixel tile per wave: 2 x 16 = 32 workgroups for
//     one board, every one on its own CU.
//   * the same MFMA (v_mfma_f32_16x16x32_f16), the same fragment composition (weights = A operand: row lane & 15, k-chunk
//     lane >> 4; pixels = B operand) and the same K order (input-channel chunks of 32 x 9 taps, accumulators
//     starting at the bias, fp16 rounding before the residual add) as k_conv3x3_c256: every output element is produced by the
//     same sequence of operations on the same operands, so the result is BIT-IDENTICAL to the tile kernel's -- a board's tower
//     activations do not depend on the size of the batch it is evaluated in. (That rules out splitting K over waves: the
//     72 MFMAs of an output tile are one dependent chain, ~1.4 us; everything else is arranged around that chain.)
//   * staging: the four waves together bring the block's whole weight slice (16 channels x 9 taps x all input channels = 72
//     fragments of 1 KB, 73 KB) and its activation slab (64 pixels + a 10-pixel halo either side, 43 KB, XOR-swizzled rows)
//     into LDS with batched plain loads -- 18 + 11 sixteen-byte loads per lane in flight, ~two memory round trips -- and meet at
//     ONE barrier; the K loop then touches LDS only, fragments prefetched one half-step ahead. (A first version streamed the
//     weights from global memory per wave: a single wave keeps too few bytes in flight, 13 us per layer.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cczero_conv.h"

namespace ccz {

constexpr int kSmPix = 64;                          // pixels per workgroup (4 waves x 16)
constexpr int kSmRows = kSmPix + 2 * kCvHalo;       // slab rows
__host__ __device__ constexpr int sm_lds_bytes(int cin) { return (cin >> 6) * 18 * 1024 + (kSmRows + 1) * cin * 2; }

// LDS: [weights: n_half fragments of 1 KB, lane-linear | slab rows of CIN fp16, 16-byte chunk c of row r at position
// (c & ~swz) | ((c ^ r) & swz) | one zero row]
template <bool RES, int CIN>
__global__ __launch_bounds__(256) void k_conv3x3_small(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                         const float *__restrict__ bias, const _Float16 *R, _Float16 *Y, int M,
                                                         int relu)
{
    constexpr int cin = CIN, n_half = (cin >> 6) * 18;
    constexpr int row_bytes = cin * 2, cpr = cin >> 3; // bytes and 16-byte chunks per slab row
    constexpr int swz = cpr >= 16 ? 15 : 7;
    constexpr int w_bytes = n_half * 1024, slab_off = w_bytes, zero_off = slab_off + kSmRows * row_bytes;
    __shared__ __attribute__((aligned(16))) unsigned char lds[sm_lds_bytes(CIN)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const long p0 = (long)blockIdx.x * kSmPix;
    const int co0 = blockIdx.y * 16;

    // ---- staging in TWO halves of the input channels (k order: chunk-major, so half 0 = half-steps 0 .. n_half / 2 - 1): every load of
    // both halves is requested before the first LDS write; half 0 is written and published as soon as ITS loads have landed (loads
    // return in request order: the wait is a counted vmcnt, the barrier a raw s_barrier) and its MFMAs run while half 1 is still
    // arriving -- the 72-step MFMA chain of a tower layer is ~1.1 us, the tail of the staging about as long.
    constexpr int kWIters = (n_half * 64 + 255) / 256;      // weight pieces per thread: 18 (tower), 5 (stem); fragment h = piece >> 6: ascending
    constexpr int kWHalf = n_half / 2 * 64 / 256;            // ... of which half 0: 9 (tower); the stem (4.5) is staged in one piece
    constexpr bool kTwo = (n_half / 2 * 64) % 256 == 0 && cpr >= 16;
    constexpr int hc = cpr / 2;                               // 16-byte columns of a slab row per half
    constexpr int kPiecesH = kSmRows * hc, kItersH = (kPiecesH + 255) / 256; // slab pieces of one half, per thread
    // accumulator starts at the bias, as in k_conv3x3_c256 (requested FIRST: loads return in order, and the first MFMA waits for it)
    cv_f32x4 acc;
    {
        const float4 bv = *(const float4 *)(bias + co0 + 4 * q4);
        acc[0] = bv.x; acc[1] = bv.y; acc[2] = bv.z; acc[3] = bv.w;
    }
    cv_half8 wreg[kWIters];
    cv_half8 sreg[2][kItersH];
    // No guards: a thread without a piece of its own in the last pass repeats the last piece (same bytes to the same LDS address) --
    // a guarded load is a branch, and behind a branch the compiler's counted waits turn into vmcnt(0).
    auto get_w = [&](int j) { // weights: fragment h (half-step), lane l = row l & 15, k-chunk l >> 4 -> LDS h * 1024 + l * 16 (what the K loop reads back)
        int i = j * 256 + tid;
        i = i < n_half * 64 ? i : n_half * 64 - 1;
        const int h = i >> 6, l = i & 63, c32 = h / 9, tap = h - c32 * 9;
        wreg[j] = *(const cv_half8 *)(W + (long)(co0 + (l & 15)) * (9 * cin) + tap * cin + c32 * 32 + (l >> 4) * 8);
    };
    auto get_s = [&](int g, int j) { // slab: rows p0 - 10 .. p0 + 64 + 10 (clamped into the tensor: a clamped row is only ever read by a masked tap); half g = columns g * hc ..
        int i = j * 256 + tid;
        i = i < kPiecesH ? i : kPiecesH - 1;
        const int row = i / hc, c = g * hc + i % hc;
        long p = p0 - kCvHalo + row;
        p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
        sreg[g][j] = *(const cv_half8 *)(X + p * cin + c * 8);
    };
    // request order = the order the halves are needed in
#pragma unroll
    for (int j = 0; j < (kTwo ? kWHalf : kWIters); ++j) get_w(j);
#pragma unroll
    for (int j = 0; j < kItersH; ++j) get_s(0, j);
    __builtin_amdgcn_sched_barrier(0); // (left alone the scheduler pairs the two halves' slab loads -- same rows -- and half 0 is complete only when everything is)
#pragma unroll
    for (int j = (kTwo ? kWHalf : kWIters); j < kWIters; ++j) get_w(j);
#pragma unroll
    for (int j = 0; j < kItersH; ++j) get_s(1, j);
    auto put_w = [&](int j) {
        int i = j * 256 + tid;
        i = i < n_half * 64 ? i : n_half * 64 - 1;
        *(cv_half8 *)(lds + i * 16) = wreg[j];
    };
    auto put_s = [&](int g, int j) {
        int i = j * 256 + tid;
        i = i < kPiecesH ? i : kPiecesH - 1;
        const int row = i / hc, c = g * hc + i % hc;
        *(cv_half8 *)(lds + slab_off + row * row_bytes + (((c & ~swz) | ((c ^ row) & swz)) << 4)) = sreg[g][j];
    };
    if (tid < row_bytes / 4) *(uint32_t *)(lds + zero_off + tid * 4) = 0u;

    // validity of the nine taps for this lane's pixel
    const long pix = p0 + wv * 16 + r;
    unsigned vmask = 0;
    {
        const int pos = (int)(pix % 90), rank = pos / 9, file = pos - rank * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (rank + dy >= 0 && rank + dy <= 9 && file + dx >= 0 && file + dx <= 8) vmask |= 1u << t;
        }
    }
    // the residual of this lane's 4 channels, requested behind the operands (it is needed last)
    const long at = pix * kCvC + co0 + 4 * q4;
    cv_half4 res = (cv_half4)(_Float16)0;
    if constexpr (RES) res = *(const cv_half4 *)(R + (pix < M ? at : (long)(M - 1) * kCvC + co0 + 4 * q4)); // (no branch: a pixel past the tensor reads the last row's and stores nothing)

    // ---- K loop from LDS: half-step h = (chunk of 32 input channels, tap); the fragments of h + 1 are requested before the MFMA of h
    auto frag_b = [&](int h) {
        const int c32 = h / 9, tap = h - c32 * 9;
        const int delta = 9 * (tap / 3 - 1) + (tap % 3 - 1);
        const int c = c32 * 4 + q4; // 16-byte chunk of this lane's k-range inside the slab row
        const int row = kCvHalo + wv * 16 + r + delta;
        const bool ok = (vmask >> tap) & 1u;
        const int off = ok ? slab_off + row * row_bytes + (((c & ~swz) | ((c ^ row) & swz)) << 4) : zero_off;
        return *(const cv_half8 *)(lds + off);
    };
    const bool idle = p0 + wv * 16 >= M; // this wave's pixel tile lies past the tensor: it helps with the staging and the barriers only
    auto k_loop = [&](int h0, int h1) {
        if (idle) return;
        cv_half8 a = *(const cv_half8 *)(lds + h0 * 1024 + lane * 16), b = frag_b(h0);
#pragma unroll
        for (int h = h0; h < h1; ++h) {
            const cv_half8 an = *(const cv_half8 *)(lds + (h + 1 < h1 ? h + 1 : h) * 1024 + lane * 16);
            const cv_half8 bn = frag_b(h + 1 < h1 ? h + 1 : h);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            a = an;
            b = bn;
        }
    };
    auto publish = [&]() { // this wave's LDS writes are done (lgkmcnt), then every wave's: no vmcnt(0) here -- the other half is still in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (kTwo) {
#pragma unroll
        for (int j = 0; j < kWHalf; ++j) put_w(j);
#pragma unroll
        for (int j = 0; j < kItersH; ++j) put_s(0, j);
        publish();
        k_loop(0, n_half / 2);
        __builtin_amdgcn_sched_barrier(0); // (the second half's LDS writes -- and the wait for its loads -- stay behind the first half's MFMAs)
#pragma unroll
        for (int j = kWHalf; j < kWIters; ++j) put_w(j);
#pragma unroll
        for (int j = 0; j < kItersH; ++j) put_s(1, j);
        publish();
        k_loop(n_half / 2, n_half);
    } else {
#pragma unroll
        for (int j = 0; j < kWIters; ++j) put_w(j);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < kItersH; ++j) put_s(g, j);
        publish();
        k_loop(0, n_half);
    }
    if (idle) return;

    // ---- epilogue: lane holds output channels co0 + 4 q4 .. + 3 of pixel p0 + 16 wv + r
    if (pix < M) {
        cv_half4 o;
        o[0] = (_Float16)acc[0];
        o[1] = (_Float16)acc[1];
        o[2] = (_Float16)acc[2];
        o[3] = (_Float16)acc[3];
        if (RES) o = o + res;
        if (relu) o = __builtin_elementwise_max(o, (cv_half4)(_Float16)0);
        *(cv_half4 *)(Y + at) = o;
    }
}

} // namespace ccz
