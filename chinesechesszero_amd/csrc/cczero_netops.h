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
// (21 live + 43 zeros), the stem input of the tower convolution kernel. One workgroup per board, 16 B per lane.
// rows != nullptr (planned evaluator boundary, ccz_eval_plan): output row i is board rows[i], for i < *n_rows only.
// g16: output rows in the group-of-16 layout (row (b / 16 * 90 + p) * 16 + b % 16, cczero_conv_g16.h) instead of b * 90 + p.
__global__ __launch_bounds__(256) void k_pack_live_planes(const _Float16 *__restrict__ leaf, half8_t *__restrict__ out, int n_boards,
                                                          const int *__restrict__ rows, const int *__restrict__ n_rows, int g16)
{
    const long b = blockIdx.x;
    long sb = b;
    if (rows) {
        if (b >= *n_rows) return;
        sb = rows[b];
    }
    const _Float16 *src = leaf + sb * (119 * 90);
    for (int i = threadIdx.x; i < 90 * 8; i += 256) {
        const int p = i >> 3, cpos = i & 7;
        half8_t v = (half8_t)(_Float16)0;
        if (cpos < 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = cpos * 8 + e;
                if (ch < 21) v[e] = src[(ch < 7 ? 49 + ch : 98 + ch) * 90 + p];
            }
        }
        const long row = g16 ? ((b >> 4) * 90 + p) * 16 + (b & 15) : b * 90 + p;
        out[row * 8 + cpos] = v;
    }
}

} // namespace ccz
