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

    // ---- staging, every load of a batch in flight before the first LDS write of that batch
    {   // weights: fragment h (half-step), lane l = row l & 15, k-chunk l >> 4 -> LDS h * 1024 + l * 16 (what the K loop reads back)
        constexpr int kW = n_half * 64 / 256; // 18 (tower) or 4.5 -> 5 (stem) pieces per thread
        constexpr int kWIters = (n_half * 64 + 255) / 256;
        cv_half8 v[kWIters];
#pragma unroll
        for (int j = 0; j < kWIters; ++j) {
            const int i = j * 256 + tid;
            if (i < n_half * 64) {
                const int h = i >> 6, l = i & 63, c32 = h / 9, tap = h - c32 * 9;
                v[j] = *(const cv_half8 *)(W + (long)(co0 + (l & 15)) * (9 * cin) + tap * cin + c32 * 32 + (l >> 4) * 8);
            }
        }
        (void)kW;
#pragma unroll
        for (int j = 0; j < kWIters; ++j) {
            const int i = j * 256 + tid;
            if (i < n_half * 64) *(cv_half8 *)(lds + i * 16) = v[j];
        }
    }
    {   // slab: rows p0 - 10 .. p0 + 64 + 10 (clamped into the tensor: a clamped row is only ever read by a masked tap)
        constexpr int kPieces = kSmRows * cpr, kIters = (kPieces + 255) / 256;
        cv_half8 v[kIters];
#pragma unroll
        for (int j = 0; j < kIters; ++j) {
            const int i = j * 256 + tid;
            if (i < kPieces) {
                const int row = i / cpr, c = i % cpr;
                long p = p0 - kCvHalo + row;
                p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
                v[j] = *(const cv_half8 *)(X + p * cin + c * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < kIters; ++j) {
            const int i = j * 256 + tid;
            if (i < kPieces) {
                const int row = i / cpr, c = i % cpr;
                *(cv_half8 *)(lds + slab_off + row * row_bytes + (((c & ~swz) | ((c ^ row) & swz)) << 4)) = v[j];
            }
        }
    }
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
    // accumulator starts at the bias, as in k_conv3x3_c256
    cv_f32x4 acc;
    {
        const float4 bv = *(const float4 *)(bias + co0 + 4 * q4);
        acc[0] = bv.x; acc[1] = bv.y; acc[2] = bv.z; acc[3] = bv.w;
    }
    // the residual of this lane's 4 channels, requested before the barrier
    const long at = pix * kCvC + co0 + 4 * q4;
    cv_half4 res = (cv_half4)(_Float16)0;
    if (RES && pix < M) res = *(const cv_half4 *)(R + at);
    __syncthreads();
    if (p0 + wv * 16 >= M) return; // this wave's pixel tile lies past the tensor (it helped with the staging)

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
    cv_half8 a = *(const cv_half8 *)(lds + lane * 16), b = frag_b(0);
#pragma unroll
    for (int h = 0; h < n_half; ++h) {
        const cv_half8 an = *(const cv_half8 *)(lds + (h + 1 < n_half ? h + 1 : h) * 1024 + lane * 16);
        const cv_half8 bn = frag_b(h + 1 < n_half ? h + 1 : h);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        a = an;
        b = bn;
    }

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
