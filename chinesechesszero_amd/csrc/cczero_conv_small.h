// cczero_conv_small.h -- the tower convolution for SMALL batches (one game at a time: MCTS_AI, the UCI loop; up to ~190 boards).
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// k_conv3x3_c256 (cczero_conv.h) works on 256-pixel tiles, one workgroup per compute unit: a single board (90 pixels) is one
// partial tile on ONE of 256 CUs (44 us per layer), which is why batches under 192 boards used to fall back to MIOpen + an
// epilogue pass (12 us per layer at one board, two launches). This kernel spreads a small batch over the chip instead:
//
//   * one workgroup = ONE wave = a 16-output-channel x (16 NT)-pixel block: 16 x NT x ... = 96 workgroups for one board (NT = 1).
//   * the same MFMA (v_mfma_f32_16x16x32_f16), the same fragment composition (weights = A operand: row lane & 15, k-chunk
//     lane >> 4; pixels = B operand) and the same K order (input-channel chunks of 64 x 9 taps x 2 halves of 32, accumulators
//     starting at the bias) as k_conv3x3_c256: every output element is produced by the same sequence of operations on the same
//     operands, so the result is BIT-IDENTICAL to the big kernel's -- a board's tower activations no longer depend on the size
//     of the batch it is evaluated in.
//   * weights stream straight from global memory (L2-resident) into VGPRs, kSmAhead half-steps ahead; the activation slab of
//     the block (16 NT pixels + a 10-pixel halo either side, all input channels) is loaded once into LDS with the XOR swizzle
//     that keeps the 16 rows of a fragment read on different banks; a tap that leaves the board reads a zero row.
//   * no barrier anywhere after the slab has landed (one wave per workgroup).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cczero_conv.h"

namespace ccz {

constexpr int kSmAhead = 18; // weight fragments in flight ahead of the MFMA that uses them: one 64-channel chunk (~600 cycles of
                             // dependent MFMAs at NT = 1, about an L2 round trip)

// LDS: slab rows of `cin` fp16 (512 B for the tower, 128 B for the stem chunk), 16-byte chunk c of row r stored at position
// c ^ (r & 15) (within its group of 16 chunks for 512-byte rows; rows of 8 chunks use c ^ (r & 7)); then one zero row.
template <int NT, bool RES, int CIN>
__global__ __launch_bounds__(64) void k_conv3x3_small(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                        const float *__restrict__ bias, const _Float16 *R, _Float16 *Y, int M,
                                                        int relu)
{
    constexpr int cin = CIN;
    constexpr int kPix = 16 * NT, kRows = kPix + 2 * kCvHalo;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    const int r = lane & 15, q4 = lane >> 4;
    const long p0 = (long)blockIdx.x * kPix;
    const int co0 = blockIdx.y * 16;
    constexpr int row_bytes = cin * 2, cpr = cin >> 3; // bytes and 16-byte chunks per slab row
    constexpr int swz = cpr >= 16 ? 15 : 7;
    constexpr int zero_off = kRows * row_bytes;

    // ---- the slab: rows p0 - 10 .. p0 + kPix + 10 (clamped into the tensor: a clamped row is only ever read by a masked tap).
    // Loads in batches of 9 per lane, all in flight before the first LDS write (one memory round trip per batch: a loop of
    // load -> write pairs took 18 round trips and made a layer 29 us at one board).
    constexpr int kPieces = kRows * cpr, kIters = (kPieces + 63) / 64, kBatch = 9;
#pragma unroll
    for (int it0 = 0; it0 < kIters; it0 += kBatch) {
        cv_half8 v[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int i = (it0 + j) * 64 + lane;
            if (it0 + j < kIters && i < kPieces) {
                const int row = i / cpr, c = i % cpr;
                long p = p0 - kCvHalo + row;
                p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
                v[j] = *(const cv_half8 *)(X + p * cin + c * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int i = (it0 + j) * 64 + lane;
            if (it0 + j < kIters && i < kPieces) {
                const int row = i / cpr, c = i % cpr;
                *(cv_half8 *)(lds + row * row_bytes + (((c & ~swz) | ((c ^ row) & swz)) << 4)) = v[j];
            }
        }
    }
    for (int i = lane; i < row_bytes / 4; i += 64) *(uint32_t *)(lds + zero_off + i * 4) = 0u;

    // validity of the nine taps for this lane's pixel of every pixel tile
    unsigned vmask[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int pos = (int)((p0 + n * 16 + r) % 90), rank = pos / 9, file = pos - rank * 9;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (rank + dy >= 0 && rank + dy <= 9 && file + dx >= 0 && file + dx <= 8) m |= 1u << t;
        }
        vmask[n] = m;
    }

    // accumulators start at the bias, as in k_conv3x3_c256
    cv_f32x4 acc[NT];
    {
        const float4 bv = *(const float4 *)(bias + co0 + 4 * q4);
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc[n][0] = bv.x; acc[n][1] = bv.y; acc[n][2] = bv.z; acc[n][3] = bv.w; }
    }

    // ---- K loop: half-step h = (chunk, tap, half); weights of half-step h + kSmAhead are requested while h computes
    constexpr int n_half = (cin >> 6) * 18;
    const _Float16 *wl = W + (long)(co0 + r) * (9 * cin) + q4 * 8;
    auto wsrc = [&](int h) {
        const int chunk = h / 18, u = h - chunk * 18;
        return wl + (u >> 1) * cin + chunk * 64 + (u & 1) * 32;
    };
    cv_half8 aq[kSmAhead];
#pragma unroll
    for (int i = 0; i < kSmAhead; ++i) aq[i] = *(const cv_half8 *)wsrc(i < n_half ? i : 0);
    __syncthreads(); // one wave: the LDS writes above are complete and visible

    for (int h0 = 0; h0 < n_half; h0 += kSmAhead) {
#pragma unroll
        for (int j = 0; j < kSmAhead; ++j) {
            const int h = h0 + j;
            if (h < n_half) { // n_half (18 or 72) is a multiple of kSmAhead: always true; keeps the tail safe if that changes
                const int chunk = h / 18, u = h - chunk * 18, tap = u >> 1, kh = u & 1;
                const int delta = 9 * (tap / 3 - 1) + (tap % 3 - 1);
                const cv_half8 a = aq[j];
                const int hn = h + kSmAhead;
                aq[j] = *(const cv_half8 *)wsrc(hn < n_half ? hn : 0);
                const int c = chunk * 8 + kh * 4 + q4; // 16-byte chunk of this lane's k-range inside the slab row
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int row = kCvHalo + n * 16 + r + delta;
                    const bool ok = (vmask[n] >> tap) & 1u;
                    const int off = ok ? row * row_bytes + (((c & ~swz) | ((c ^ row) & swz)) << 4) : zero_off;
                    const cv_half8 b = *(const cv_half8 *)(lds + off);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[n], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds output channels co0 + 4 q4 .. + 3 of pixel p0 + 16 n + r
    const cv_half4 zero = (cv_half4)(_Float16)0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const long p = p0 + n * 16 + r;
        if (p < M) {
            cv_half4 o;
            o[0] = (_Float16)acc[n][0];
            o[1] = (_Float16)acc[n][1];
            o[2] = (_Float16)acc[n][2];
            o[3] = (_Float16)acc[n][3];
            const long at = p * kCvC + co0 + 4 * q4;
            if (RES) o = o + *(const cv_half4 *)(R + at);
            if (relu) o = __builtin_elementwise_max(o, zero);
            *(cv_half4 *)(Y + at) = o;
        }
    }
}

} // namespace ccz
