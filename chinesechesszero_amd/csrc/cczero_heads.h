// cczero_heads.h -- the evaluator's tail as three kernels: both 1x1 head convolutions, the fully connected layers, tanh.
//
//   policy = relu(bn(conv1x1(x, 256 -> 17)));  logits = fc(flatten(policy), 1530 -> 2086)            (reference net.py:96-99)
//   value  = relu(bn(conv1x1(x, 256 -> 7)));   v = tanh(fc2(relu(fc1(flatten(value), 630 -> 256)), 256 -> 1))   (net.py:101-109)
//
// Round 3 ran this part through torch (one GEMM for both 1x1 convolutions, a permuting copy out of the group-of-16 row order,
// four slicing / casting copies, two hipBLASLt GEMMs, tanh): 0.21 ms of a 21.7 ms step, always on all 4096 rows, and the
// one piece of the evaluator whose result for a board could depend on the batch it sat in (a library GEMM picks its kernel and
// its split of K by the problem size). Here:
//
//   * k_head_conv1x1 reads the tower's rows ONCE (189 MB at 4096 boards: the HBM / Infinity-Cache-bound part) in whatever row
//     order the tower ran in and writes [board][pos][17] and [board][pos][7] fp16 in BOARD order: the g16 -> board permutation,
//     bias, ReLU and the policy / value split happen in its epilogue. One wave per 16-row cell, weights (24 x 256, padded to 32)
//     held in registers as MFMA A fragments for the wave's whole life, the next cell's rows in flight while this one multiplies.
//   * k_fc_f16 is a plain tiled MFMA GEMM C[m, n] = act(bias[n] + sum_k A[m, k] W[n, k]) (128 x 128 tiles, K through a four-stage
//     LDS ring filled by global_load_lds, XOR-swizzled rows) for the policy FC (K = 1530 padded to 1536) and the first value FC (K = 630 padded to 640).
//   * k_value_out: the 256 -> 1 layer + tanh, one wave per board.
//
// All three take the device-side live-row count of the planned evaluator boundary (ccz_eval_plan): rows past it are not computed.
// Every output element is one fixed chain of MFMAs over k = 0, 32, 64, ... whatever the batch size, tile or wave it lands in, so a
// board's logits and value do not depend on the batch it is evaluated in (tests/test_gpu_evaluator_depth.py) -- which is what the
// evaluation cache assumes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cczero_conv.h"
#include "cczero_device.h"

namespace ccz {

constexpr int kHdPol = 17, kHdVal = 7;        // head channels (net.py:12 PLAYS, net.py:11 PIECES)
constexpr int kHdPolStride = 1536;            // fp16 elements per board of the policy-head output: 90 * 17 = 1530 + 6 zeros
constexpr int kHdValStride = 640;             // ... of the value-head output: 90 * 7 = 630 + 10 zeros

// X: tower rows [n_rows][256] fp16 (NHWC: row = board * 90 + pos; G16: row = (g * 90 + pos) * 16 + j for board 16 g + j).
// Wh: [32][256] fp16, rows 0..16 policy, 17..23 value, 24..31 zero; bh: float [32]. pol / val: [boards][kHdPolStride / kHdValStride].
// live: device count of live boards (rows of boards past it are skipped) or nullptr.
template <bool G16>
__global__ __launch_bounds__(256) void k_head_conv1x1(const _Float16 *__restrict__ X, const _Float16 *__restrict__ Wh,
                                                        const float *__restrict__ bh, _Float16 *__restrict__ pol,
                                                        _Float16 *__restrict__ val, int n_boards, const int *__restrict__ live)
{
    const int lane = threadIdx.x & 63, r = lane & 15, q4 = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    int nb = n_boards;
    if (live) { const int l = *live; nb = l < nb ? l : nb; }
    const long n_rows = G16 ? (long)((nb + 15) >> 4) * 1440 : (long)nb * 90;
    const int n_cells = (int)((n_rows + 15) >> 4);
    if (wave >= n_cells) return;
    // weights as A fragments: [row block m][k-step s]: row m * 16 + r, k = s * 32 + q4 * 8 .. + 7
    cv_half8 a[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 8; ++s) a[m][s] = *(const cv_half8 *)(Wh + (m * 16 + r) * 256 + s * 32 + q4 * 8);
    float4 bias[2];
    bias[0] = *(const float4 *)(bh + 4 * q4);
    bias[1] = *(const float4 *)(bh + 16 + 4 * q4);
    auto load = [&](int cell, cv_half8 (&b)[8]) {
        long row = (long)cell * 16 + r;
        row = row < n_rows ? row : n_rows - 1; // (a clamped row is computed and not stored)
        const _Float16 *src = X + row * 256 + q4 * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) b[s] = *(const cv_half8 *)(src + s * 32);
    };
    cv_half8 b[8], bn[8];
    load(wave, b);
    for (int cell = wave; cell < n_cells; cell += n_waves) {
        const int next = cell + n_waves;
        if (next < n_cells) load(next, bn);
        cv_f32x4 acc0, acc1;
        acc0[0] = bias[0].x; acc0[1] = bias[0].y; acc0[2] = bias[0].z; acc0[3] = bias[0].w;
        acc1[0] = bias[1].x; acc1[1] = bias[1].y; acc1[2] = bias[1].z; acc1[3] = bias[1].w;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][s], b[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][s], b[s], acc1, 0, 0, 0);
        }
        // lane: channels 4 q4 .. + 3 (acc0) and 16 + 4 q4 .. + 3 (acc1) of row cell * 16 + r
        int board, pos;
        if (G16) {
            const int g = cell / 90;
            pos = cell - g * 90;
            board = g * 16 + r;
        } else {
            const long row = (long)cell * 16 + r;
            board = (int)(row / 90);
            pos = (int)(row - (long)board * 90);
        }
        if (board < nb) {
            _Float16 *po = pol + (long)board * kHdPolStride + pos * kHdPol;
            _Float16 *vo = val + (long)board * kHdValStride + pos * kHdVal;
#pragma unroll
            for (int e = 0; e < 4; ++e) po[4 * q4 + e] = (_Float16)fmaxf(acc0[e], 0.0f);
            if (q4 == 0) {
                po[16] = (_Float16)fmaxf(acc1[0], 0.0f);
                vo[0] = (_Float16)fmaxf(acc1[1], 0.0f);
                vo[1] = (_Float16)fmaxf(acc1[2], 0.0f);
                vo[2] = (_Float16)fmaxf(acc1[3], 0.0f);
            } else if (q4 == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vo[3 + e] = (_Float16)fmaxf(acc1[e], 0.0f);
            }
        }
        if (next < n_cells) {
#pragma unroll
            for (int s = 0; s < 8; ++s) b[s] = bn[s];
        }
    }
}

// C[m, n] = act(bias[n] + sum_k A[m, k] W[n, k]); A [M][lda] fp16, W [ceil(N / 128) * 128][K] fp16 (rows past N zero), K a
// multiple of 64, bias float [ceil(N / 128) * 128], C [M][ldc] fp16 (N and ldc even). grid 8 * ceil(tiles / 8), tiles = ceil(N / 128) * ceil(M / 128), one-dimensional.
// 128 x 128 output tile per workgroup, 4 waves of 64 x 64 (4 x 4 MFMA tiles); K in steps of 32 through a ring of FOUR 16 KB LDS
// stages filled by global_load_lds three steps ahead (the first version staged through registers one step ahead: at 16 MFMAs per
// wave and step the loop waited a memory round trip per step, 48 us for the policy layer). Rows are 64 bytes in LDS: chunk c
// of row r sits at position c ^ ((r >> 2) & 3) -- applied to the SOURCE address of the DMA, whose LDS side is lane-linear, and to
// the fragment reads, which are then conflict-free. One raw s_barrier per step, counted vmcnt (never 0 inside the loop).
constexpr int kFcBM = 128, kFcBN = 128, kFcBK = 32, kFcStages = 4;
constexpr int kFcTile = 128 * kFcBK * 2;  // bytes of one operand tile of a stage: 128 rows of 64 B
constexpr int kFcStage = 2 * kFcTile;     // [W tile | A tile]
constexpr int kFcERow = 144;              // epilogue transpose: bytes per row of a wave's 64 x 64 fp16 block (128 + pad)
static_assert(4 * 64 * kFcERow <= kFcStages * kFcStage, "the epilogue blocks live in the operand ring");
#ifdef CCZ_FC_DIAG // ablation switches of the diagnostic build (make ab NAME=fcdiag ABFLAGS=-DCCZ_FC_DIAG; profiles/fc_microbench.py)
#define FC_DBG(bit) (dbg & (bit))
#else
#define FC_DBG(bit) 0
#endif
template <bool RELU>
__global__ __launch_bounds__(256) void k_fc_f16(const _Float16 *__restrict__ A, int lda, const _Float16 *__restrict__ W,
                                                  const float *__restrict__ bias, _Float16 *__restrict__ C, int ldc, int M, int N,
                                                  int K, const int *__restrict__ live, [[maybe_unused]] int dbg)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kFcStages * kFcStage];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wv & 1, wm = wv >> 1;
    int Ml = M;
    if (live) { const int l = *live; Ml = l < Ml ? l : Ml; }
    // XCD-aware tile order. Workgroup b runs on XCD b mod 8 (round-robin dispatch), every XCD has its own 4 MB L2, and with the plain
    // (n, m) grid every tile pair fetched both operands from past the L2: PMC 190 MB per launch of the policy layer against 36 MB
    // algorithmic -- the kernel ran at the fabric's ~4 TB/s, not at the matrix pipe (ablation: MFMA-only 23 us, DMA-only 49 us). Here
    // the tiles are put in an order in which neighbours share operands -- blocks of four m tiles, inside a block n-major (the four m
    // tiles of one n tile next to each other) -- and XCD x takes the x-th CONTIGUOUS eighth of that order: it then holds <= 2 blocks
    // of A rows (8 x 393 KB) and walks W once. Eighths differ by at most one tile (61-62 of 493: one round on the 64 workgroup slots
    // of an XCD; a first attempt that gave XCD x the m tiles x, x + 8, ... put 68 tiles on five XCDs and lost 6 us to the second round).
    const int mt = (Ml + kFcBM - 1) / kFcBM, nt = (N + kFcBN - 1) / kFcBN, tiles = mt * nt;
    int n0, m0;
    {
        const int b = blockIdx.x, x = b & 7, per = tiles >> 3, rem = tiles & 7;
        if ((b >> 3) >= per + (x < rem ? 1 : 0)) return;
        const int t = x * per + (x < rem ? x : rem) + (b >> 3);
        const int full = (mt >> 2) * 4 * nt;          // tiles of the complete four-m-tile blocks
        const int blk = t < full ? t / (4 * nt) : (mt >> 2);
        const int r_ = t < full ? t % (4 * nt) : t - full;
        const int cnt = t < full ? 4 : mt - 4 * blk;
        n0 = (r_ / cnt) * kFcBN;
        m0 = (4 * blk + r_ % cnt) * kFcBM;
    }
    // DMA: per stage and operand 512 sixteen-byte granules = 2 per thread. Instruction j of wave wv fills LDS granules
    // (j * 4 + wv) * 64 + lane (lane-linear); granule p holds row p >> 2, position p & 3 = source chunk (p & 3) ^ ((row >> 2) & 3)
    const _Float16 *wsrc[2], *asrc[2];
    int wdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (j * 4 + wv) * 16 + (lane >> 2), c = (lane & 3) ^ ((lane >> 4) & 3);
        wsrc[j] = W + (long)(n0 + row) * K + c * 8;
        int m = m0 + row;
        m = m < Ml ? m : Ml - 1; // (rows past the live ones are computed from a clamped row and not stored)
        asrc[j] = A + (long)m * lda + c * 8;
        wdst[j] = (j * 4 + wv) * 1024;
    }
    auto issue = [&](int kt) {
        unsigned char *st = lds + (kt & (kFcStages - 1)) * kFcStage;
        const int k0 = kt * kFcBK;
        cv_glds16(wsrc[0] + k0, st + wdst[0]);
        cv_glds16(wsrc[1] + k0, st + wdst[1]);
        cv_glds16(asrc[0] + k0, st + kFcTile + wdst[0]);
        cv_glds16(asrc[1] + k0, st + kFcTile + wdst[1]);
    };
    // accumulators start at the bias: acc[i][j] = n tile i (rows n0 + wn * 64 + i * 16 + 4 q4 .. + 3) x m tile j (column m0 + wm * 64 + j * 16 + r)
    cv_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 bv = *(const float4 *)(bias + n0 + wn * 64 + i * 16 + 4 * q4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] = bv.x; acc[i][j][1] = bv.y; acc[i][j][2] = bv.z; acc[i][j][3] = bv.w; }
    }
    const int nk = K / kFcBK;
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    // this lane's fragment offset inside a tile: row (wn | wm) * 64 + i * 16 + r, position q4 ^ ((r >> 2) & 3)
    const int foff = r * 64 + ((q4 ^ ((r >> 2) & 3)) << 4);
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed when at most the loads of the two younger stages are outstanding (4 per stage and thread)
        if (kt + 2 < nk) cv_wait_vm<8>();
        else if (kt + 1 < nk) cv_wait_vm<4>();
        else cv_wait_vm<0>();
        __builtin_amdgcn_sched_barrier(0);
        if (!FC_DBG(16)) __builtin_amdgcn_s_barrier(); // ... for every wave's part of it; and every wave is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 3 < nk && !FC_DBG(4)) issue(kt + 3); // into the slot of stage kt - 1
        const unsigned char *wb = lds + (kt & (kFcStages - 1)) * kFcStage + wn * 4096 + foff;
        const unsigned char *ab = lds + (kt & (kFcStages - 1)) * kFcStage + kFcTile + wm * 4096 + foff;
        cv_half8 fa[4], fb[4];
        if (!FC_DBG(2) || kt == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *(const cv_half8 *)(wb + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *(const cv_half8 *)(ab + j * 1024);
        }
        if (!FC_DBG(1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][0] += (cv_f32x4){(float)fa[i][0], (float)fb[i][0], 0.0f, 0.0f}; // keep the reads alive
        }
    }
    if (FC_DBG(8)) { if (acc[0][0][0] == 12345.678f) C[0] = (_Float16)1; return; }
    // epilogue: a lane holds 4 consecutive n of ONE row m per tile -- stored straight from the registers that is 64 different
    // cache lines per store instruction, 32 instructions per wave, and the CU's one address unit spent ~8 us on them (ablation:
    // profiles/r04_fc_microbench.json). So each wave transposes its 64 x 64 block through its own 9 KB of LDS (rows of 144 bytes:
    // the 8-byte writes of 16 rows fall on different banks) and writes whole 128-byte row segments, two rows per instruction.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier(); // every wave is done reading the operand stages
    __builtin_amdgcn_sched_barrier(0);
    {
        unsigned char *tb = lds + wv * (64 * kFcERow);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
                if (RELU) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
                cv_half4 o;
                o[0] = (_Float16)v0; o[1] = (_Float16)v1; o[2] = (_Float16)v2; o[3] = (_Float16)v3;
                *(cv_half4 *)(tb + (j * 16 + r) * kFcERow + (i * 16 + 4 * q4) * 2) = o;
            }
        wave_sync(); // (the block is this wave's own)
        const int half = lane >> 5, dw = lane & 31;
        const int n = n0 + wn * 64 + 2 * dw;
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int row = it * 2 + half, m = m0 + wm * 64 + row;
            const uint32_t v = *(const uint32_t *)(tb + row * kFcERow + dw * 4);
            if (m < Ml && n + 1 < N) *(uint32_t *)(C + (long)m * ldc + n) = v;
        }
    }
}

// ---- the same GEMM for a HANDFUL of rows (M <= 16: one game at a time -- 1 + 10 scout rows per evaluator call) -------------------------
// k_fc_f16 gives such a batch ONE m tile: 17 workgroups for the policy layer, each walking 48 k-steps through its LDS ring, a memory
// round trip every few steps: 19 us (policy) + 10.5 us (value fc1) of a 640-us evaluator call (profiles/r06_single_board_timeline.json).
// Here one WAVE owns a 16 (n) x 16 (m) output tile for the whole K: ceil(N / 16) one-wave workgroups spread over the chip (131 for the
// policy layer), operands straight from global memory into registers, sixteen k-steps per batch and the next batch in flight while
// this one multiplies. Same MFMA, same operand fragments (W rows = A operand, activation rows = B operand), accumulator starting at
// the bias, k ascending in steps of 32: the same bits as k_fc_f16 (tests/test_gpu_conv.py::test_both_fc_kernels_give_the_same_bits).
constexpr int kFsU = 16; // k-steps per batch of loads (2 x 16 x 16 B per lane)
template <bool RELU>
__global__ __launch_bounds__(64) void k_fc_skinny_f16(const _Float16 *__restrict__ A, int lda, const _Float16 *__restrict__ W,
                                                        const float *__restrict__ bias, _Float16 *__restrict__ C, int ldc, int M, int N,
                                                        int K, const int *__restrict__ live)
{
    const int lane = threadIdx.x, r = lane & 15, q4 = lane >> 4;
    int Ml = M;
    if (live) { const int l = *live; Ml = l < Ml ? l : Ml; }
    if (Ml <= 0) return;
    const int n0 = blockIdx.x * 16;
    const int m = r < Ml ? r : Ml - 1; // (rows past the live ones are computed from a clamped row and not stored)
    const _Float16 *wp = W + (long)(n0 + r) * K + q4 * 8;   // W has ceil(N / 128) * 128 rows: n0 + r is one of them
    const _Float16 *ap = A + (long)m * lda + q4 * 8;
    cv_f32x4 acc;
    {
        const float4 bv = *(const float4 *)(bias + n0 + 4 * q4);
        acc[0] = bv.x; acc[1] = bv.y; acc[2] = bv.z; acc[3] = bv.w;
    }
    const int nk = K / kFcBK;
    cv_half8 w0[kFsU], x0[kFsU], w1[kFsU], x1[kFsU];
    auto load = [&](cv_half8 (&w)[kFsU], cv_half8 (&x)[kFsU], int c0) {
#pragma unroll
        for (int u = 0; u < kFsU; ++u) {
            const int kt = c0 + u < nk ? c0 + u : nk - 1; // (past the end: the last step again, not multiplied)
            w[u] = *(const cv_half8 *)(wp + kt * kFcBK);
            x[u] = *(const cv_half8 *)(ap + kt * kFcBK);
        }
    };
    auto mul = [&](const cv_half8 (&w)[kFsU], const cv_half8 (&x)[kFsU], int c0) {
#pragma unroll
        for (int u = 0; u < kFsU; u += 2) // K is a multiple of 64: k-steps come in pairs
            if (c0 + u < nk) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], x[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u + 1], x[u + 1], acc, 0, 0, 0);
            }
    };
    load(w0, x0, 0);
    for (int c0 = 0; c0 < nk; c0 += 2 * kFsU) {
        if (c0 + kFsU < nk) load(w1, x1, c0 + kFsU);
        mul(w0, x0, c0);
        if (c0 + 2 * kFsU < nk) load(w0, x0, c0 + 2 * kFsU);
        if (c0 + kFsU < nk) mul(w1, x1, c0 + kFsU);
    }
    // a lane holds n = n0 + 4 q4 .. + 3 of row m = r
    if (r < Ml) {
        float v0 = acc[0], v1 = acc[1], v2 = acc[2], v3 = acc[3];
        if (RELU) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
        cv_half4 o;
        o[0] = (_Float16)v0; o[1] = (_Float16)v1; o[2] = (_Float16)v2; o[3] = (_Float16)v3;
        const int n = n0 + 4 * q4;
        uint32_t *dst = (uint32_t *)(C + (long)r * ldc + n); // n and ldc are even: 4-byte aligned
        typedef _Float16 fs_half2 __attribute__((ext_vector_type(2)));
        const uint32_t lo = __builtin_bit_cast(uint32_t, (fs_half2){o[0], o[1]}), hi = __builtin_bit_cast(uint32_t, (fs_half2){o[2], o[3]});
        if (n + 1 < N) dst[0] = lo;
        if (n + 3 < N) dst[1] = hi;
    }
}

// ---- the same GEMM for the policy layer's shape (many rows, N in the thousands): 256 x 144 tiles, 128-byte rows --------------------------
// k_fc_f16 moves 64 bytes of a row (32 k) per step: HALF a cache line -- the other half is wanted a step later, when the 25-50 KB
// that passed through the CU's 32 KB L1 in between have long evicted it, so every line crosses the L2 -> L1 path (64 B/clk/CU)
// twice: 2 x 16 KB x 2 workgroups per CU and step = 1,024 cycles against 256 of MFMA (measured 0.42 us per step; the first form of this kernel,
// 256 x 144 tiles with 64-byte rows, 0.62 us per step at 2 x 25.6 KB). Here a stage is 64 k: whole 128-byte lines, each fetched
// once -- 51.2 KB per stage = 800 cycles of L1 fill under 1,152 cycles of MFMA. The tile is 256 (m) x 144 (n) per workgroup of EIGHT
// waves: 2086 columns are 15 tiles of 144 (against 17 of 128, the last one 38 wide), a batch of 4096 rows 16 x 15 = 240 tiles --
// ONE round on 256 CUs whatever the live count. Wave (wm, wn) owns rows 64 wm .. + 63 and n-fragments 0..4 (wn = 0) or 5..8
// (wn = 1): 9 or 8 fragment reads per 20 or 16 MFMAs, and the two waves of a SIMD (wave ids s and s + 4) hold 36 MFMAs per k-step
// between them. Three stages of 51.2 KB (153.6 KB of LDS); the K loop is software-pipelined across the stage barrier (below).
// Operands, fragment composition and the chain over k = 0, 32, 64, ... are k_fc_f16's: every output element is the same bits from
// either kernel (tests/test_gpu_conv.py::test_both_fc_kernels_give_the_same_bits).
constexpr int kFwBM = 256, kFwBN = 144, kFwStages = 3;
constexpr int kFwATile = kFwBM * 128, kFwWTile = kFwBN * 128; // bytes per stage: 256 + 144 rows of 128 B (64 k)
constexpr int kFwStage = kFwATile + kFwWTile;                 // [A tile | W tile] 51,200 B
constexpr int kFwERow = 176;                                  // epilogue transpose: bytes per row of a wave's 64 x 80 fp16 block (160 + pad)
static_assert(8 * 64 * kFwERow <= kFwStages * kFwStage, "the epilogue blocks live in the operand ring");
static_assert(kFwStages * kFwStage <= 160 * 1024, "LDS");

template <bool RELU, int NF, int I0>
__device__ __forceinline__ void fw_wave(unsigned char *lds, const _Float16 *const (&src)[7], int wv, int wm, int lane,
                                        const float *__restrict__ bias, int bias_last, _Float16 *__restrict__ C, int ldc, int Ml, int N, int K,
                                        int m0, int n0, [[maybe_unused]] int dbg)
{
    const int r = lane & 15, q4 = lane >> 4;
    // DMA of one stage: 50 wave-instructions of 64 sixteen-byte granules = 8 whole rows each: q = 0..31 the A tile, q = 32..49 the W
    // tile; wave wv issues q = wv + 8 t (t = 0..5) and waves 0, 1 also q = 48 + wv: 7 or 6 loads per lane and stage. dma(st, t): the
    // wave's t-th instruction of stage st (t = 0..3: A rows, 4..6: W rows)
    auto dma = [&](int st_, int t) {
        unsigned char *st = lds + (st_ % kFwStages) * kFwStage;
        if (FC_DBG(4) && st_ > 2) return;
        if (t < 4) cv_glds16(src[t] + st_ * 64, st + (wv + 8 * t) * 1024);
        else if (t < 6 || wv < 2) cv_glds16(src[t] + st_ * 64, st + kFwATile + (wv + 8 * (t - 4)) * 1024);
    };
    auto wait_vm2 = [&](int whole) { // `whole` (0..2) stages of this lane's loads may stay outstanding
        if (wv < 2) {
            if (whole >= 2) cv_wait_vm<14>();
            else if (whole == 1) cv_wait_vm<7>();
            else cv_wait_vm<0>();
        } else {
            if (whole >= 2) cv_wait_vm<12>();
            else if (whole == 1) cv_wait_vm<6>();
            else cv_wait_vm<0>();
        }
    };
    // acc[i][j] = n-fragment I0 + i (rows n0 + (I0 + i) * 16 + 4 q4 .. + 3) x m-fragment j (column m0 + wm * 64 + j * 16 + r), starting at the bias
    cv_f32x4 acc[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        int at = n0 + (I0 + i) * 16 + 4 * q4;
        at = at < bias_last ? at : bias_last;
        const float4 bv = *(const float4 *)(bias + at);
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j][0] = bv.x; acc[i][j][1] = bv.y; acc[i][j][2] = bv.z; acc[i][j][3] = bv.w; }
    }
    // fragment of row 16 f + r, k-sub ks (32 k): 16-byte chunk 4 ks + q4 of the row, stored at position chunk ^ ((row >> 1) & 7)
    // (conflict-free for the four 16-lane groups of ds_read_b128: 16 different (row parity, position) pairs each)
    const int rsw = (r >> 1) & 7;
    const int foff0 = r * 128 + ((q4 ^ rsw) << 4), foff1 = r * 128 + (((4 + q4) ^ rsw) << 4);
    // (The compiler puts `s_waitcnt lgkmcnt(0)` in front of the first MFMA of a fragment set that was loaded in the previous loop
    // iteration, which also waits for the reads issued just before it. Issuing the reads as inline asm with hand-counted lgkmcnt(NF + 4)
    // removed that wait and measured the same, 32.5 against 31.7 us: the exposed LDS latency is not what the loop waits for.)
    auto frags = [&](int st_, int ks, cv_half8 (&fa)[NF], cv_half8 (&fb)[4]) {
        const unsigned char *base = lds + (st_ % kFwStages) * kFwStage + (ks ? foff1 : foff0);
        if (FC_DBG(2) && st_ > 0) return;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fb[jj] = *(const cv_half8 *)(base + (wm * 4 + jj) * 2048);
#pragma unroll
        for (int i = 0; i < NF; ++i) fa[i] = *(const cv_half8 *)(base + kFwATile + (I0 + i) * 2048);
    };
    // One k-sub: NF groups of four MFMAs; in front of group g the wave issues DMA instructions t0 + 2 g and t0 + 2 g + 1 (while < nd). An LDS-DMA
    // costs its wave 60-185 cycles of issue (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"): the first form issued a stage's 6-7 in
    // a burst behind the barrier -- both waves of a SIMD at once, the matrix pipe idle meanwhile: 1.1 us per stage against 0.5 us of
    // MFMA. Spread out, one wave's DMA issue hides under the other wave's MFMAs.
    auto mfmas = [&](const cv_half8 (&fa)[NF], const cv_half8 (&fb)[4], int st_, int t0, int nd) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if (2 * i < nd) dma(st_, t0 + 2 * i);
            if (2 * i + 1 < nd) dma(st_, t0 + 2 * i + 1);
            if (FC_DBG(1)) { acc[i][0] += (cv_f32x4){(float)fa[i][0], (float)fb[i & 3][0], 0.0f, 0.0f}; continue; }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[jj], acc[i][jj], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // The K loop, software-pipelined across the stage barrier: the fragments of the NEXT k-sub are requested before the MFMAs of the
    // current one, so LDS latency and the barrier's skew hide under 20 / 16 MFMAs. Once every wave holds the second half of stage s
    // in registers (the barrier in the middle of iteration s) the slot of stage s is free: stage s + 3 is issued into it during the
    // second k-sub of iteration s (one or two DMA instructions in front of each group of MFMAs), a good 1.5 iterations before its
    // first fragment is read (spreading the stage over two half-iterations, the W part one iteration ahead only, measured the same: 31.8 against 31.7 us).
    const int ns = K / 64;
#ifdef CCZ_FC_DIAG // dbg bit 5: wave 0 leaves cycle stamps in the pad columns of its first row (ldc >= N + 16): entry, first fragments, loop end, tile end
#define FW_STAMP(k) if ((dbg & 32) && wv == 0 && lane == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); uint32_t *o_ = (uint32_t *)(C + (long)m0 * ldc + N + 4 * (k) + (n0 / kFwBN) * 0); if (n0 == 0) { o_[0] = (uint32_t)t_; o_[1] = (uint32_t)(t_ >> 32); } }
#else
#define FW_STAMP(k)
#endif
    FW_STAMP(0)
#pragma unroll
    for (int t = 0; t < 7; ++t) dma(0, t);
    if (ns > 1) {
#pragma unroll
        for (int t = 0; t < 7; ++t) dma(1, t);
    }
    if (ns > 2) {
#pragma unroll
        for (int t = 0; t < 7; ++t) dma(2, t);
    }
    cv_half8 fa0[NF], fb0[4], fa1[NF], fb1[4];
    wait_vm2(ns - 1 < 2 ? ns - 1 : 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    frags(0, 0, fa0, fb0);
    FW_STAMP(1)
    for (int s_ = 0; s_ < ns; ++s_) {
        frags(s_, 1, fa1, fb1);
        mfmas(fa0, fb0, 0, 0, 0);
        if (s_ + 1 < ns) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // this wave's fragments of stage s are in registers
            wait_vm2(s_ + 2 < ns ? 1 : 0);                       // stage s + 1 has landed (all of s + 2 may be outstanding)
            __builtin_amdgcn_sched_barrier(0);
            if (!FC_DBG(16)) __builtin_amdgcn_s_barrier(); // stage s + 1 is complete (every wave's part of it); every wave is done reading stage s
            __builtin_amdgcn_sched_barrier(0);
            frags(s_ + 1, 0, fa0, fb0);
        }
        mfmas(fa1, fb1, s_ + 3, 0, s_ + 3 < ns ? 7 : 0);
    }
    FW_STAMP(2)
    if (FC_DBG(8)) { if (acc[0][0][0] == 12345.678f) C[0] = (_Float16)1; return; }
    // epilogue: the wave's 64 (m) x 16 NF (n) block through its own LDS block, whole row segments out (k_fc_f16's epilogue)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier(); // every wave is done reading the operand stages
    __builtin_amdgcn_sched_barrier(0);
    unsigned char *tb = lds + wv * (64 * kFwERow);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            float v0 = acc[i][j][0], v1 = acc[i][j][1], v2 = acc[i][j][2], v3 = acc[i][j][3];
            if (RELU) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
            cv_half4 o;
            o[0] = (_Float16)v0; o[1] = (_Float16)v1; o[2] = (_Float16)v2; o[3] = (_Float16)v3;
            *(cv_half4 *)(tb + (j * 16 + r) * kFwERow + (i * 16 + 4 * q4) * 2) = o;
        }
    wave_sync(); // (the block is this wave's own)
    constexpr int NW = NF * 8; // dwords per row of the block
#pragma unroll 4
    for (int it = 0; it < NW; ++it) {
        const int L = it * 64 + lane, row = L / NW, dw = L - row * NW;
        const int m = m0 + wm * 64 + row, n = n0 + I0 * 16 + 2 * dw;
        const uint32_t v = *(const uint32_t *)(tb + row * kFwERow + dw * 4);
        if (m < Ml && n + 1 < N) *(uint32_t *)(C + (long)m * ldc + n) = v;
    }
    FW_STAMP(3)
#undef FW_STAMP
}

// Arguments as k_fc_f16 (W rows padded to a multiple of 128 with zeros: rows past that are never read -- clamped). grid 8 * ceil(tiles / 8).
template <bool RELU>
__global__ __launch_bounds__(512) void k_fc_wide_f16(const _Float16 *__restrict__ A, int lda, const _Float16 *__restrict__ W,
                                                       const float *__restrict__ bias, _Float16 *__restrict__ C, int ldc, int M, int N,
                                                       int K, const int *__restrict__ live, [[maybe_unused]] int dbg)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kFwStages * kFwStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv & 3, wn = wv >> 2;
    int Ml = M;
    if (live) { const int l = *live; Ml = l < Ml ? l : Ml; }
    // XCD-aware tile order (see k_fc_f16): blocks of TWO m tiles, inside a block n-major; XCD x takes the x-th contiguous eighth:
    // at 16 x 15 tiles an XCD works on two m tiles (1.5 MB of A rows) and walks W once
    const int mt = (Ml + kFwBM - 1) / kFwBM, nt = (N + kFwBN - 1) / kFwBN, tiles = mt * nt;
    int n0, m0;
    {
        const int b = blockIdx.x, x = b & 7, per = tiles >> 3, rem = tiles & 7;
        if ((b >> 3) >= per + (x < rem ? 1 : 0)) return;
        const int t = x * per + (x < rem ? x : rem) + (b >> 3);
        const int full = (mt >> 1) * 2 * nt;
        const int blk = t < full ? t / (2 * nt) : (mt >> 1);
        const int r_ = t < full ? t % (2 * nt) : t - full;
        const int cnt = t < full ? 2 : 1;
        n0 = (r_ / cnt) * kFwBN;
        m0 = (2 * blk + r_ % cnt) * kFwBM;
    }
    // per-lane DMA sources (fw_wave::issue): granule p = lane of instruction q holds row 8 q + (p >> 3), position p & 7 = source chunk
    // (p & 7) ^ ((row >> 1) & 7); (row >> 1) & 7 = (4 (q & 1) + (lane >> 4)) & 7 and q & 1 = wv & 1 for all of a wave's instructions
    const int wrows = ((N + 127) >> 7) << 7;
    const int c = (lane & 7) ^ ((4 * (wv & 1) + (lane >> 4)) & 7), lr = lane >> 3;
    const _Float16 *src[7];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int m = m0 + (wv + 8 * t) * 8 + lr;
        m = m < Ml ? m : Ml - 1; // (rows past the live ones are computed from a clamped row and not stored)
        src[t] = A + (long)m * lda + c * 8;
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int n = n0 + (wv + 8 * t) * 8 + lr; // (t = 2: waves 0, 1 only)
        n = n < wrows ? n : wrows - 1;
        src[4 + t] = W + (long)n * K + c * 8;
    }
    if (wn == 0) fw_wave<RELU, 5, 0>(lds, src, wv, wm, lane, bias, wrows - 4, C, ldc, Ml, N, K, m0, n0, dbg);
    else fw_wave<RELU, 4, 5>(lds, src, wv, wm, lane, bias, wrows - 4, C, ldc, Ml, N, K, m0, n0, dbg);
}

// v[m] = tanh(fp16(b2 + sum_k h[m, k] w2[k])), h [M][256] fp16 (the first value FC's ReLU output), one wave per board
// (reference net.py:107-109: value_fc2 -> tanh; the fp16 rounding in front of tanh is the fp16 linear layer's output rounding).
__global__ __launch_bounds__(256) void k_value_out(const _Float16 *__restrict__ h, const _Float16 *__restrict__ w2, float b2,
                                                    float *__restrict__ v, int M, const int *__restrict__ live)
{
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    // (the row is requested together with the live count: rows up to M exist whether they are live or not)
    const cv_half4 x = *(const cv_half4 *)(h + (long)m * 256 + lane * 4), w = *(const cv_half4 *)(w2 + lane * 4);
    int Ml = M;
    if (live) { const int l = *live; Ml = l < Ml ? l : Ml; }
    if (m >= Ml) return;
    float s = (float)x[0] * (float)w[0];
    s += (float)x[1] * (float)w[1];
    s += (float)x[2] * (float)w[2];
    s += (float)x[3] * (float)w[3];
    s = wave_sum_f32(s) + b2;
    if (lane == 0) v[m] = tanhf((float)(_Float16)s);
}

} // namespace ccz
