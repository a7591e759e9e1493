// cczero_conv3.h -- the tower convolution without a workgroup barrier in its K loop (round 3 form).
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// Same operation, same tile (256 consecutive pixels x all 256 output channels per workgroup), same MFMA
// (v_mfma_f32_16x16x32_f16), same slab staging and K order (4 chunks of 64 input channels x 9 taps x 2 halves of 32)
// as k_conv3x3_c256 (cczero_conv.h). What changes is who owns the weights:
//
//   * 8 waves as 8 (output channels) x 1 (pixels): a wave computes 32 output channels for ALL 256 pixels of the tile
//     (2 x 16 accumulator tiles = 128 VGPRs). Its weight fragments -- 2 per half-step, 16 rows x 64 B each -- are its
//     own: it loads them straight from global memory (L2-resident: 1.18 MB per layer) into VGPRs, two half-steps ahead.
//     No weight ring in LDS, no DMA of weights, and therefore NO barrier per half-step: in k_conv3x3_c256 every wave
//     reads weight rows that other waves' DMA brought in, which costs one s_barrier per half-step and runs the two waves
//     of a SIMD in lockstep (its no-barrier ablation: 1,267 against 1,577 cycles per half-step).
//   * the pixel operand still comes from the activation slab in LDS (DMA'd once per chunk, read nine times), now 16
//     fragment reads per half-step and wave (each feeds two MFMAs) through a rolling window of registers. The only
//     workgroup barriers left in the K loop are two per chunk: the slab of the next chunk has landed (half-step 17), and
//     nobody still reads the buffer the next staging pieces overwrite (half-step 2): 8 per tile instead of 72.
//   * epilogue: the whole 256 x 256 output block is transposed through LDS (the slabs are dead by then), so that the
//     residual is read and the output written as whole 512-byte pixel rows.
//
// RESULT (profiles/r03_conv_v3.json, 4096 boards, one layer in isolation, interleaved on one device): bit-identical output
// (same summation order per accumulator), but 393-407 us against 355-361 us for k_conv3x3_c256. Ablations of this kernel:
// neither operand (MFMA + slab DMA + epilogue only) 297-301 us; without the pixel-fragment reads 341; without the validity
// selects (16 v_cndmask per half-step + 32 v_and / v_cmp per tap) 377; without the weight loads 366; weight loads that always
// hit L1 389. The costs add up: the loop is ISSUE-bound, not latency-bound (a deeper fragment window, 6 or 8 instead of 4,
// gains 3 %), and a 32 x 256 block per wave needs 18 operand instructions per 32 MFMAs where the 128 x 64 block of
// k_conv3x3_c256 needs 14-15 plus one barrier. The barrier is cheaper than the instructions that replace it. Not shipped.
#pragma once
#include "../../chinesechesszero_amd/csrc/cczero_conv.h"

namespace ccz {

constexpr int kC3Slab = 0;                                // LDS: [slab 0 | slab 1 | 16 zero rows, 2 KB apart]
constexpr int kC3ZeroOff = 2 * kCvABytes;
constexpr int kC3ERow = 528;                              // epilogue: bytes per pixel row (512 + pad)
constexpr int kC3Lds = 256 * kC3ERow;                     // 135,168 B (the K loop needs 2 x 36,864 + 15 x 2,048 + 128)
static_assert(kC3ZeroOff + 15 * 2048 + 128 <= kC3Lds, "zero rows must fit");
#ifndef C3_LOOK
#define C3_LOOK 4
#define C3_WIN 6
#endif
constexpr int kC3Look = C3_LOOK;                          // pixel fragments requested ahead of the MFMAs that use them
constexpr int kC3Win = C3_WIN;                            // register slots of the rolling window; must divide 288 (fragments per chunk)
static_assert(288 % kC3Win == 0 && kC3Look < kC3Win && kC3Look <= 8, "window geometry");

struct C3Ctx {
    unsigned char *lds;
    const _Float16 *X;
    int xsrc[5];          // per staging pass: element offset of this thread's 16-byte source in X (chunk 0)
    int wave_dst;         // w * 1024
    int q4;               // lane >> 4
    int brow;             // slab row of this lane's pixel of tile 0 at tap offset 0 (tile n: + 16 n)
    unsigned tv[5];       // tv[t >> 1] >> (16 * (t & 1)): bit n = tap t of this lane's pixel of tile n stays on the board
    const _Float16 *wa;   // this lane's weight source: row 32 w + (lane & 15), k-chunk lane >> 4 (tap 0, chunk 0, half 0)
    int cin, cmask;
};

// slab addressing and validity of one tap, shared by all 16 pixel tiles and both k-halves of the tap. The 16 validity
// flags live in scalar register pairs (lane masks): one v_cndmask per fragment read selects between the slab row and the
// zero row. Computed once per tap INSIDE the loop (the asm barrier keeps the compiler from hoisting 9 x 16 lane masks).
struct C3Tap {
    int base; // LDS offset of the row of tile 0 at this tap
    int sw16; // ((row & 7) ^ q4) << 4: swizzled position of k-chunk q4
    bool ok[16];
};

template <int TAP> __device__ __forceinline__ C3Tap c3_tap(const C3Ctx &c, int abase)
{
    constexpr int delta = 9 * (TAP / 3 - 1) + (TAP % 3 - 1);
    int br = c.brow;
    unsigned tw = c.tv[TAP >> 1];
    asm volatile("" : "+v"(br), "+v"(tw));
    const int row = br + delta;
    C3Tap t;
    t.base = abase + row * 128;
    t.sw16 = ((row & 7) ^ c.q4) << 4;
#ifndef C3_ABL_NOMASK
#pragma unroll
    for (int n = 0; n < 16; ++n) t.ok[n] = (tw & (1u << (n + 16 * (TAP & 1)))) != 0u;
#endif
    return t;
}

// pixel fragment of tile N, k-half KH: a tap that leaves the board reads a zero row
template <int KH, int N> __device__ __forceinline__ cv_half8 c3_read_x(const C3Ctx &c, const C3Tap &t)
{
#ifdef C3_ABL_NOMASK
    const int off = t.base + (t.sw16 ^ (KH << 6));
#else
    const int off = t.ok[N] ? t.base + (t.sw16 ^ (KH << 6)) : kC3ZeroOff;
#endif
    return *(const cv_half8 *)(c.lds + off + N * 2048);
}

// weight fragments of half-step U2 of chunk `chunk2`: two 16-row tiles, straight from global memory
template <int U2> __device__ __forceinline__ void c3_load_w(const C3Ctx &c, int chunk2, cv_half8 (&a)[2])
{
#ifdef C3_ABL_WSAME
    const _Float16 *s = c.wa + (chunk2 & 0); // every load hits the same (L1-resident) lines: the issue cost alone
#else
    const _Float16 *s = c.wa + (U2 >> 1) * c.cin + chunk2 * 64 + (U2 & 1) * 32;
#endif
    a[0] = *(const cv_half8 *)s;
    a[1] = *(const cv_half8 *)(s + 16l * (9 * c.cin));
}

// One half-step = one k-step of 32 for the wave's 32 x 256 block: 16 pixel fragments x 2 weight fragments = 32 MFMAs.
// acur: weights of this half-step; aload: receives the weights of half-step U + 2. bw: rolling window of pixel fragments;
// on entry the fragments of tiles 0 .. kC3Look - 1 of this half-step are in flight or landed, on exit those of the next.
template <int U>
__device__ __forceinline__ void c3_halfstep(const C3Ctx &c, cv_f32x4 (&acc)[2][16], int chunk, C3Tap &tap, const cv_half8 (&acur)[2],
                                             cv_half8 (&aload)[2], cv_half8 (&bw)[kC3Win])
{
    constexpr int KH = U & 1;
    constexpr int Un = (U + 1) % 18, KHn = Un & 1, TAPn = Un >> 1;
    unsigned char *const lds = c.lds;

    if constexpr (U == 2) {
        // The slab pieces issued from here on overwrite the buffer the PREVIOUS chunk was read from, and nothing else keeps a
        // fast wave from running ahead of a slow one that is still in that chunk's last half-step: second (and last)
        // barrier of a chunk. (Every wave that arrives here has its reads of that buffer behind it: their MFMAs are issued.)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    constexpr int pass = cv_act_pass(U);
    if constexpr (pass >= 0) { // the next chunk's slab, one piece per thread in half-steps 2, 4, 6, 8, 10
        const int nxt = (chunk + 1) & c.cmask; // past the last chunk: re-stage chunk 0 into the free buffer (harmless, keeps the code static)
        cv_glds16(c.X + (c.xsrc[pass] + nxt * 64), lds + kC3Slab + ((chunk + 1) & 1) * kCvABytes + (pass < 4 ? pass * 64 : 224) * 128 + c.wave_dst);
    }
#ifndef C3_ABL_NOW
    {
        constexpr int U2 = (U + 2) % 18;
        c3_load_w<U2>(c, (chunk + (U + 2 >= 18 ? 1 : 0)) & c.cmask, aload);
    }
#endif
    if constexpr (U == 17) {
        // chunk boundary: every wave's slab pieces of the next chunk must have landed before anybody reads them. The
        // pieces are older than the four youngest loads (the weights of the next two half-steps).
        cv_wait_vm<4>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    // tile N: request the fragment kC3Look tiles ahead (the last kC3Look requests belong to the next half-step: from there on
    // `tap` is the next tap's -- the current tap's flags are dead by then and the new ones take their registers), then the
    // two MFMAs of this tile
#ifdef C3_ABL_NOX
#define C3_READ(N, KHX)
#else
#define C3_READ(N, KHX) bw[slot_rd] = c3_read_x<KHX, (N + kC3Look) % 16>(c, tap);
#endif
#define C3_TILE(N)                                                                                                        \
    {                                                                                                                     \
        constexpr int slot = (U * 16 + N) % kC3Win, slot_rd = (U * 16 + N + kC3Look) % kC3Win;                            \
        if constexpr (N == 16 - kC3Look && KH == 1) tap = c3_tap<TAPn>(c, kC3Slab + ((chunk + (U == 17 ? 1 : 0)) & 1) * kCvABytes); \
        if constexpr (N + kC3Look < 16) { C3_READ(N, KH) } else { C3_READ(N, KHn) }                                       \
        acc[0][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[0], bw[slot], acc[0][N], 0, 0, 0);                        \
        acc[1][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[1], bw[slot], acc[1][N], 0, 0, 0);                        \
        __builtin_amdgcn_sched_barrier(0); /* keep the window: left alone, the scheduler requests each fragment just */  \
    }                                      /* one tile ahead of its MFMAs (lowest register pressure) */
    C3_TILE(0) C3_TILE(1) C3_TILE(2) C3_TILE(3) C3_TILE(4) C3_TILE(5) C3_TILE(6) C3_TILE(7)
    C3_TILE(8) C3_TILE(9) C3_TILE(10) C3_TILE(11) C3_TILE(12) C3_TILE(13) C3_TILE(14) C3_TILE(15)
#undef C3_TILE
#undef C3_READ
}


template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_v3(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                       const float *__restrict__ bias, const _Float16 *R,
                                                       _Float16 *Y, int M, int relu, int cin)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kC3Lds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    // flags bit 1: tiles in descending order (the tiles written last by the previous layer are then read first)
    const long p0 = (long)((relu & 2) ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * kCvBM;
    relu &= 1;

    C3Ctx c;
    c.lds = lds;
    c.X = X;
    c.wave_dst = w * 1024;
    c.q4 = q4;
    c.cin = cin;
    c.cmask = (cin >> 6) - 1;
    {
        const int srow = tid >> 3, cpos = tid & 7;
        const int schunk = cpos ^ (srow & 7); // slab row r holds source chunk c at position c ^ (r & 7)
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            long p = p0 - kCvHalo + (it < 4 ? it * 64 : 224) + srow;
            p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
            c.xsrc[it] = (int)(p * cin + schunk * 8);
        }
    }
    c.wa = W + (long)(w * 32 + r) * (9 * cin) + q4 * 8;
    // ---- prologue: slab of chunk 0 (DMA), the zero rows, the weights of half-steps 0 and 1; the per-lane setup below
    // runs while they are in flight
    if (tid < 512) *(uint32_t *)(lds + kC3ZeroOff + (tid >> 5) * 2048 + (tid & 31) * 4) = 0u;
#pragma unroll
    for (int it = 0; it < 5; ++it) cv_glds16(X + c.xsrc[it], lds + kC3Slab + (it < 4 ? it * 64 : 224) * 128 + c.wave_dst);
    cv_half8 a0[2], a1[2], a2[2];
    c3_load_w<0>(c, 0, a0);
    c3_load_w<1>(c, 0, a1);
    c.brow = kCvHalo + r;
#pragma unroll
    for (int j = 0; j < 5; ++j) c.tv[j] = 0u;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int pos = (int)((p0 + n * 16 + r) % 90), rank = pos / 9, file = pos - rank * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (rank + dy >= 0 && rank + dy <= 9 && file + dx >= 0 && file + dx <= 8) c.tv[t >> 1] |= 1u << (n + 16 * (t & 1));
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

    cv_half8 bw[kC3Win];
    C3Tap tap = c3_tap<0>(c, kC3Slab);
    bw[0] = c3_read_x<0, 0>(c, tap);
    if constexpr (kC3Look > 1) bw[1] = c3_read_x<0, 1>(c, tap);
    if constexpr (kC3Look > 2) bw[2] = c3_read_x<0, 2>(c, tap);
    if constexpr (kC3Look > 3) bw[3] = c3_read_x<0, 3>(c, tap);
    if constexpr (kC3Look > 4) bw[4] = c3_read_x<0, 4>(c, tap);
    if constexpr (kC3Look > 5) bw[5] = c3_read_x<0, 5>(c, tap);
    if constexpr (kC3Look > 6) bw[6] = c3_read_x<0, 6>(c, tap);
    if constexpr (kC3Look > 7) bw[7] = c3_read_x<0, 7>(c, tap);
#ifdef C3_ABL_NOX
#pragma unroll
    for (int i = 0; i < kC3Win; ++i) bw[i] = c3_read_x<0, 0>(c, tap);
#endif
#ifdef C3_ABL_NOW
    c3_load_w<2>(c, 0, a2);
#endif
    for (int chunk = 0; chunk <= c.cmask; ++chunk) {
#define C3_H(u, cur, ld) c3_halfstep<u>(c, acc, chunk, tap, cur, ld, bw)
        C3_H(0, a0, a2); C3_H(1, a1, a0); C3_H(2, a2, a1); C3_H(3, a0, a2); C3_H(4, a1, a0); C3_H(5, a2, a1);
        C3_H(6, a0, a2); C3_H(7, a1, a0); C3_H(8, a2, a1); C3_H(9, a0, a2); C3_H(10, a1, a0); C3_H(11, a2, a1);
        C3_H(12, a0, a2); C3_H(13, a1, a0); C3_H(14, a2, a1); C3_H(15, a0, a2); C3_H(16, a1, a0); C3_H(17, a2, a1);
#undef C3_H
    }
    cv_wait_vm<0>(); // the wrapped-around DMA and weight loads must land before the LDS is reused / released

    // ---- epilogue: every wave writes its 32 channels x 256 pixels into the [pixel][channel] image in LDS; then wave w
    // owns pixels 32 w .. 32 w + 31 and moves whole 512-byte rows (residual in, output out)
    const int prow = lane >> 5, piece = lane & 31;
    const long pbase = p0 + w * 32 + prow;
    const bool full = p0 + kCvBM <= M; // whole tile inside the tensor (always, when boards * 90 is a multiple of 256)
    cv_half8 rv[16];
    if (RES && full) { // the residual rows are requested before the transposition, which hides their latency
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
            *(cv_half4 *)(lds + (n * 16 + r) * kC3ERow + col * 2) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
        const cv_half8 zero = (cv_half8)(_Float16)0;
        const unsigned char *eb = lds + (w * 32 + prow) * kC3ERow + piece * 16;
        if (full) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                cv_half8 v = *(const cv_half8 *)(eb + it * 2 * kC3ERow);
                if (RES) v = v + rv[it];
                if (relu) v = __builtin_elementwise_max(v, zero);
                *(cv_half8 *)(Y + (pbase + it * 2) * kCvC + piece * 8) = v;
            }
        } else {
            for (int it = 0; it < 16; ++it) {
                const long p = pbase + it * 2;
                if (p >= M) break;
                cv_half8 v = *(const cv_half8 *)(eb + it * 2 * kC3ERow);
                if (RES) v = v + *(const cv_half8 *)(R + p * kCvC + piece * 8);
                if (relu) v = __builtin_elementwise_max(v, zero);
                *(cv_half8 *)(Y + p * kCvC + piece * 8) = v;
            }
        }
    }
}

} // namespace ccz
