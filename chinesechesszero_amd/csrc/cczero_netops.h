// cczero_netops.h -- fused elementwise epilogues for the evaluator's residual tower (NHWC fp16).
//
// The tower stays in PyTorch-ROCm (MIOpen convolutions); what MIOpen leaves unfused is the per-channel
// bias, the residual add and the ReLU, which PyTorch runs as three separate full-tensor passes
// (bias 42 us + add 78 us + clamp 40 us per 189 MB activation at B = 4096). These two kernels do each
// epilogue in ONE pass, 16 B per lane, in packed fp16 arithmetic (same rounding sequence as the separate ops).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccz {

typedef _Float16 half8_t __attribute__((ext_vector_type(8))); // 16 B: one lane's share of a row

// y[r, c] = relu(y[r, c] + bias[c])                     (in place)
// y[r, c] = relu(y[r, c] + bias[c] + res[r, c])         (res != nullptr)
template <bool RES>
__global__ __launch_bounds__(256) void k_bias_act(half8_t *__restrict__ y, const half8_t *__restrict__ bias,
                                                  const half8_t *__restrict__ res, long n_vec, int c_vec)
{
    const long stride = (long)gridDim.x * blockDim.x;
    const half8_t zero = (half8_t)(_Float16)0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
        half8_t t = y[i] + bias[(int)(i % c_vec)]; // v_pk_add_f16: rounds to fp16 like the separate bias pass
        if (RES) t = t + res[i];
        y[i] = __builtin_elementwise_max(t, zero);
    }
}

// The evaluator input [B, 17, 7, 10, 9] fp16 (reference net.py:174-177) has 21 planes that can be non-zero on the
// search path (groups 7, 15, 16 = planes 49..55 and 105..118, net.py:160-173): pack them as NHWC rows of 64 channels
// (21 live + 43 zeros), the stem input of the tower convolution kernel. One workgroup per board.
// rows != nullptr (planned evaluator boundary, ccz_eval_plan): output row i is board rows[i], for i < *n_rows only.
// flags bit 0 (g16): output rows in the group-of-16 layout (row (b / 16 * 90 + p) * 16 + b % 16, cczero_conv_g16.h) instead of
// b * 90 + p; bit 1: the caller's buffer already holds zeros in channels 24..63 (a persistent buffer zeroed once): only the three
// 16-byte chunks that can be non-zero are written (48 of 128 bytes per row).
// The two plane runs of a board (630 + 1260 fp16, both 4-byte aligned) are staged in LDS with dword loads and transposed from
// there (round 3 gathered them with 2-byte global loads: 7.7 M of them per step, 22 us).
// One WAVE per board (round 4, second form; the first ran 256 threads and a workgroup barrier per board: with four waves a board
// the chip holds half the batch at a time and every workgroup sits through two memory round trips and a barrier, 14 us at 3,660
// boards): 15 dword loads per lane in flight, the LDS hand-over needs no barrier inside a wave, and all boards are resident at once.
constexpr int kPackThreads = 64;
__global__ __launch_bounds__(kPackThreads) void k_pack_live_planes(const _Float16 *__restrict__ leaf, half8_t *__restrict__ out, int n_boards,
                                                                   const int *__restrict__ rows, const int *__restrict__ n_rows, int flags)
{
    __shared__ uint32_t s_w[946]; // planes 49..55 (315 dwords) then 105..118 (630 dwords); + 1 pad
    const long b = blockIdx.x;
    const int lane = threadIdx.x;
    long sb = b;
    if (rows) {
        const int nr = *n_rows;
        sb = rows[b]; // (requested with the count, not after it: rows[] has an entry for every board of the batch)
        if (b >= nr) return;
    }
    const uint32_t *src = (const uint32_t *)(leaf + sb * (119 * 90)); // a board is 21,420 B: dword-aligned
    constexpr int kIt = (945 + kPackThreads - 1) / kPackThreads;
    uint32_t v[kIt];
#pragma unroll
    for (int j = 0; j < kIt; ++j) {
        const int i = lane + j * kPackThreads;
        v[j] = i < 945 ? src[i < 315 ? (49 * 45) + i : (105 * 45) + (i - 315)] : 0u;
    }
#pragma unroll
    for (int j = 0; j < kIt; ++j) {
        const int i = lane + j * kPackThreads;
        if (i < 945) s_w[i] = v[j];
    }
    __syncthreads(); // (one wave: no s_barrier is emitted, only the wait for the LDS writes)
    const _Float16 *s_h = (const _Float16 *)s_w; // live channel ch (0..20), pixel p at s_h[ch * 90 + p]
    const bool g16 = flags & 1;
    const int ncp = (flags & 2) ? 3 : 8;
    for (int i = lane; i < 90 * ncp; i += kPackThreads) {
        const int p = i / ncp, cpos = i - p * ncp;
        half8_t o = (half8_t)(_Float16)0;
        if (cpos < 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = cpos * 8 + e;
                if (ch < 21) o[e] = s_h[ch * 90 + p];
            }
        }
        const long row = g16 ? ((b >> 4) * 90 + p) * 16 + (b & 15) : b * 90 + p;
        out[row * 8 + cpos] = o;
    }
}

} // namespace ccz
