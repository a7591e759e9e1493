// cczero_conv.h -- the evaluator's tower convolution as ONE kernel: 3x3, 256 -> 256 channels on the
// 10 x 9 board, NHWC fp16, fp32 accumulate, + bias [+ residual] + ReLU in the epilogue.
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// (reference net.py:20-43: conv3x3 -> BN -> [+x] -> ReLU with BN folded into w / bias; 80 of the 83
// convolutions of the 40-block tower have exactly this shape.) An implicit GEMM built around the board:
//
//   * D[co, pixel] = W[co, k] . X[k, pixel], k = (tap, ci): weights are the MFMA A operand (rows ->
//     accumulator registers), pixels the B operand (columns -> lanes), so each lane ends up holding 4
//     consecutive output channels of ONE pixel per accumulator tile.
//   * one workgroup = 256 consecutive pixels x all 256 output channels; 8 waves as 2 (co) x 4 (pixels),
//     a 128 x 64 accumulator block per wave = 8 x 4 tiles of v_mfma_f32_16x16x32_f16 (128 VGPRs). The 16x16x32
//     shape, not 32x32x16: this loop is power-limited and the chip holds a higher clock on it at equal cycles per
//     FLOP (MI355X_MICROARCH.md, DVFS give-back item 7); measured -6.6 % time for the same tile and pipeline.
//   * K order = 4 input-channel chunks of 64 (outer) x 2 halves of 32 x 9 taps (inner) = 72 half-steps, i.e. chunks of 32
//     input channels x 9 taps: the order all three convolution kernels add in (cczero_conv_g16.h, cczero_conv_small.h). One
//     half-step is ONE k-step of the MFMA: 8 weight fragments (lane l: row l & 15, k-chunk l >> 4 of a 64-byte
//     weight row), 4 pixel fragments (pixel l & 15, k-chunk 4*KH + (l >> 4) of the 128-byte slab row), 32 MFMAs.
//     The activation slab of a chunk (the tile's 256 pixels + a 10-pixel halo either side, 64 channels) is
//     staged ONCE into LDS and read nine times at row offsets 9*dy + dx; a tap that leaves the board reads a
//     zero row instead (per-lane 9-bit validity mask, no arithmetic on the data). Only the weights are
//     staged per half-step, into a ring of five 16 KB half-tiles, three half-steps ahead.
//   * both operands arrive by global_load_lds (16 B per lane, no VGPR round trip) into XOR-swizzled rows
//     (swizzle applied to the SOURCE address and to the read address): conflict-free ds_read_b128 for both
//     operands at any tap offset (weights: chunk c of row r at c ^ (-(r >> 2) & 3); slab: c ^ (r & 7)).
//   * software pipeline, one barrier per half-step: [DMA issue | read weight tiles 4-7 | 16 MFMA on tiles 0-3]
//     barrier [read the next half-step's weight tiles 0-3 and pixels | 16 MFMA on tiles 4-7]; DMA loads stay in
//     flight across barriers (raw s_barrier + counted s_waitcnt vmcnt, never 0 inside the loop).
//   * epilogue: each wave transposes its 64 pixel x 128 channel block through its own LDS region, so that the
//     residual is read and the output written as whole 256-byte pixel-row segments.
//
// Measured alternatives that did NOT pay (profiles/conv_ab.py, one device, interleaved; figures = change in throughput;
// on the 32x32x16 form):
// staggering the DMA issue of the two wave groups (-4.5 %), running the groups half a half-step apart (-2 %),
// s_setprio around the MFMA groups (-11 %), sched_group_barrier-pinned interleave (-1.5 %), fragments read a
// whole half-step ahead into three register sets (0 %), 16 zero rows (one per bank slot) for the masked lanes
// (-1 %), a ping-pong form with two barriers per 8-MFMA group and the wave groups one barrier apart (-15 %); on this
// form: one barrier per TWO half-steps (the ring of 5 allows it) (-1.6 %), a split barrier on an LDS arrival counter
// (arrive after the DMA wait, 8 MFMAs, then poll) instead of s_barrier (-23 %), several consecutive tiles per workgroup with
// the next tile's first slab and weights staged during the last chunk (no prologue for later tiles) and the epilogue
// transposed through the one free slab buffer, one pixel tile at a time (-12..-16 %: the four serialized epilogue
// passes and their register pressure cost more than the prologue they save), ONE wave per SIMD with a 128 x 128 block per
// wave (256 accumulator registers, a third fewer LDS fragment bytes per MFMA) (-14 % as compiled by hipcc, -20 % with the
// DMA and the reads pinned between the MFMAs by sched_group_barrier: a single instruction stream per SIMD does not hide its
// own DMA issue and waits without hand scheduling). Round 2 (profiles/r02_conv_ab.json): static s_setprio 1 for waves 4-7
// (-0.2 % time, noise), iglp_opt(1) also for the region behind the barrier (-1.7 % median, equal minimum: noise), that region
// pinned as (2 MFMA, 1 DS read) x 8 by sched_group_barrier (+6 % time). The compiled loop is already tight: between two
// barriers 32 MFMAs, 12 ds_read_b128, 2-3 DMA, 8-24 VALU and <= 7 instructions in front of the first MFMA (one region per
// chunk carries the ring-slot SALU arithmetic); what is left is DMA issue cost and the lockstep of the two waves of a SIMD.
// A Winograd F(2x2,3x3) form was sized and dropped: 16 weight matrices instead of 9 and 4 pixels per tile make it need ~8x the
// weight bytes per flop of this tiling at the largest accumulator block the register file holds (~290 GB/s per CU from L2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccz {

typedef _Float16 cv_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_half4 __attribute__((ext_vector_type(4)));

constexpr int kCvC = 256;             // channels in = channels out
constexpr int kCvBM = 256;            // pixels per workgroup
constexpr int kCvHalo = 10;           // |9*dy + dx| <= 10
constexpr int kCvARows = 288;         // staged slab rows (276 used)
constexpr int kCvABytes = kCvARows * 128;
constexpr int kCvWBytes = 256 * 64;   // one half-step of weights: 256 output channels x 32 k
constexpr int kCvRing = 5;
constexpr int kCvAhead = 3;           // weight half-tiles in flight ahead of the one being read
constexpr int kCvAOff = kCvRing * kCvWBytes; // LDS: [weight ring | slab 0 | slab 1 | zero row]; the ring first, so that
                                             // the four weight reads of a k-sub share one address register (+ immediates)
constexpr int kCvZeroOff = kCvAOff + 2 * kCvABytes;
constexpr int kCvERow = 272;          // epilogue transpose: bytes per pixel row of a wave's 64 x 128 block (256 + pad)
static_assert(8 * 64 * kCvERow <= kCvZeroOff, "epilogue transpose must fit the operand buffers");

typedef __attribute__((address_space(3))) void *cv_lds_ptr;
typedef const __attribute__((address_space(1))) void *cv_glb_ptr;

__device__ __forceinline__ void cv_glds16(const void *src, unsigned char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((cv_glb_ptr)src, (cv_lds_ptr)lds_wave_base, 16, 0, 0);
}

// the activation-slab DMA of the next chunk is spread over half-steps 2, 4, 6, 8, 10 of the current one
__host__ __device__ constexpr int cv_act_pass(int u) { return (u >= 2 && u <= 10 && !(u & 1)) ? (u - 2) / 2 : -1; }
// DMA loads younger than the half-tile the NEXT half-step reads (issue order per half-step: slab piece, 2 weight loads):
// 2 x 2 weight loads + the slab pieces of this and the previous half-step
__host__ __device__ constexpr int cv_vmcnt(int u) { return 4 + (cv_act_pass(u) >= 0 ? 1 : 0) + (cv_act_pass((u + 17) % 18) >= 0 ? 1 : 0); }
template <int N> __device__ __forceinline__ void cv_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// slab addressing of one tap, shared by both pixel tiles and all four k-subs of the tap
struct CvTap {
    int base; // LDS offset of the row of tile 0 at this tap
    int sw16; // ((row & 7) ^ q4) << 4: swizzled position of k-chunk q4
};

#ifdef CCZ_STAMPS
#define CV_DBG(c, bit) ((c).dbg & (bit))
// diagnostic build: per-workgroup cycle stamps of waves 0 and 4, [block][2][16] (profiles/conv_stamps.py)
__device__ unsigned long long g_cv_stamps[2048 * 2 * 16];
__device__ __forceinline__ unsigned long long cv_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#else
#define CV_DBG(c, bit) 0
#endif

typedef float cv_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCvLds = kCvZeroOff + 3 * 2048 + 128; // zero rows at kCvZeroOff + n * 2048: the pixel-tile offset stays an immediate

struct CvCtx {
    unsigned char *lds;
    const _Float16 *X;
    int xsrc[5];       // per staging pass: element offset of this thread's 16-byte source in X (chunk 0)
    const _Float16 *wsrc; // this thread's 16-byte source in W (row pass 0, tap 0, chunk 0, half 0)
    int wave_dst;      // w * 1024
    int q4;            // lane >> 4: this lane's k-chunk inside a 32-k step
    int a_off;         // weight fragment offset inside a ring slot (tile 0; tile m: + 1024 m)
    int brow;          // slab row of this lane's pixel of tile 0 at tap offset 0 (tile n: + 16 n)
    unsigned vmask[4]; // per pixel tile: bit t set = tap t stays on the board
    int cin;           // input channels per pixel of X and per (co, tap) row of W: 256 (tower) or 64 (stem, one chunk)
    int cmask;         // number of 64-channel chunks - 1
    int dbg;           // diagnostic build (-DCCZ_STAMPS) only: ablation switches from bits 8.. of the relu argument
};

template <int TAP> __device__ __forceinline__ CvTap cv_tap(const CvCtx &c, int abase)
{
    constexpr int delta = 9 * (TAP / 3 - 1) + (TAP % 3 - 1);
    int br = c.brow;
    asm volatile("" : "+v"(br)); // keep this arithmetic in the loop: hoisted for all nine taps it costs ~40 VGPRs
    const int row = br + delta;
    CvTap t;
    t.base = abase + row * 128;
    t.sw16 = ((row & 7) ^ c.q4) << 4;
    return t;
}

template <int HI> __device__ __forceinline__ void cv_read_w(const CvCtx &c, int ring_slot, cv_half8 (&a)[4])
{
    if (CV_DBG(c, 64)) return;
    const unsigned char *wa = c.lds + (ring_slot * kCvWBytes + c.a_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const cv_half8 *)(wa + (HI * 4 + i) * 1024);
}

template <int TAP, int KH> __device__ __forceinline__ void cv_read_x(const CvCtx &c, const CvTap &t, cv_half8 (&b)[4])
{
    if (CV_DBG(c, 64)) return;
    const int off0 = t.base + (t.sw16 ^ (KH << 6));
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const bool ok = (c.vmask[n] >> TAP) & 1u; // a tap that leaves the board reads a zero row
        const int off = ok ? off0 : kCvZeroOff;
        b[n] = *(const cv_half8 *)(c.lds + off + n * 2048);
    }
}

template <int HI>
__device__ __forceinline__ void cv_mfma16(const CvCtx &c, cv_f32x4 (&acc)[8][4], const cv_half8 (&a)[4], const cv_half8 (&b)[4])
{
    if (!CV_DBG(c, 8)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[HI * 4 + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[n], acc[HI * 4 + i][n], 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
        for (int n = 0; n < 4; ++n) asm volatile("" ::"v"(b[n]));
    }
}

template <int U>
__device__ __forceinline__ void cv_halfstep(const CvCtx &c, cv_f32x4 (&acc)[8][4], int chunk, int &ring_rd, int &ring_wr, CvTap &tap,
                                              cv_half8 (&alo)[4], cv_half8 (&ahi)[4], cv_half8 (&bcur)[4], cv_half8 (&bnxt)[4])
{
    unsigned char *const lds = c.lds;

    constexpr int pass = cv_act_pass(U);
    if constexpr (pass >= 0) if (!CV_DBG(c, 2)) {
        const int nxt = (chunk + 1) & c.cmask; // past the last chunk: re-stage chunk 0 into the free buffer (keeps every count static)
        cv_glds16(c.X + (c.xsrc[pass] + nxt * 64), lds + kCvAOff + ((chunk + 1) & 1) * kCvABytes + (pass < 4 ? pass * 64 : 224) * 128 + c.wave_dst);
    }
    if (!CV_DBG(c, 1)) {
        constexpr int U2 = (U + kCvAhead) % 18;
        const int chunk2 = (chunk + (U + kCvAhead >= 18 ? 1 : 0)) & c.cmask;
        const _Float16 *s = c.wsrc + (U2 % 9) * c.cin + chunk2 * 64 + (U2 / 9) * 32;
        unsigned char *d = lds + ring_wr * kCvWBytes + c.wave_dst;
        cv_glds16(s, d);
        cv_glds16(s + 128l * (9 * c.cin), d + 8192);
    }
    cv_read_w<1>(c, ring_rd, ahi);
    cv_mfma16<0>(c, acc, alo, bcur);
#ifndef CCZ_STAMPS // (the diagnostic build's runtime ablation branches make the pass abort in hipcc 7.2)
    __builtin_amdgcn_iglp_opt(1); // LLVM's single-wave MFMA / DS interleave for this scheduling region: -3..-6 % (same bits)
#endif

    ring_rd = ring_rd + 1 == kCvRing ? 0 : ring_rd + 1;
    ring_wr = ring_wr + 1 == kCvRing ? 0 : ring_wr + 1;
    cv_wait_vm<cv_vmcnt(U)>();
    constexpr int Un = (U + 1) % 18;
    __builtin_amdgcn_sched_barrier(0);
    if (!CV_DBG(c, 32)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    tap = cv_tap<(Un % 9)>(c, kCvAOff + ((chunk + (U == 17 ? 1 : 0)) & 1) * kCvABytes);
    cv_read_w<0>(c, ring_rd, alo);
    cv_read_x<(Un % 9), (Un / 9)>(c, tap, bnxt);
    cv_mfma16<1>(c, acc, ahi, bcur);
}

template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_c256(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                          const float *__restrict__ bias, const _Float16 *R,
                                                          _Float16 *Y, int M, int relu, int cin, const int *live_rows, int row0)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kCvLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int wm = w >> 2, wn = w & 3;
    int tiles = gridDim.x;
    if (live_rows) {
        // Planned evaluator boundary (ccz_eval_plan): only the first *live_rows boards of the batch hold rows to compute, a
        // number that stays on the device. The batch is cut into n_parts EQUAL ranges of the LIVE rows (multiples of 8 boards;
        // the last tile of a range is partial); this launch is range `part` of them (the argument carries part | n_parts << 16)
        // and finds its boards itself. The grid is sized for the largest possible range; workgroups beyond the live tiles leave at once.
        const int part = row0 & 0xffff, n_parts = row0 >> 16;
        const int L = *live_rows;
        const int per = ((L + n_parts - 1) / n_parts + 7) / 8 * 8;
        const int first = part * per;
        int live = L - first;
        live = live < 0 ? 0 : (live > per ? per : live);
        live = live > M / 90 ? M / 90 : live;
        M = live * 90;
        tiles = (M + kCvBM - 1) / kCvBM;
        if ((int)blockIdx.x >= tiles) return;
        const long off = (long)first * 90 * kCvC;
        X += (long)first * 90 * cin;
        Y += off;
        if (RES) R += off;
    }
    // flags bit 1: tiles in descending order (the tiles written last by the previous layer are then read first)
    const long p0 = (long)((relu & 2) ? tiles - 1 - blockIdx.x : blockIdx.x) * kCvBM;

#ifdef CCZ_STAMPS
    const unsigned long long st_prolog = cv_stamp();
#endif
    CvCtx c;
    c.lds = lds;
    c.X = X;
    c.wave_dst = w * 1024;
    c.q4 = q4;
    c.cin = cin;
    c.cmask = (cin >> 6) - 1;
    c.dbg = relu >> 8;
    relu &= 1;
    {
        const int srow = tid >> 3, cpos = tid & 7;
        const int schunk = cpos ^ (srow & 7); // slab row r holds source chunk c at position c ^ (r & 7)
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            long p = p0 - kCvHalo + (it < 4 ? it * 64 : 224) + srow;
            p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
            c.xsrc[it] = (int)(p * cin + schunk * 8);
        }
        // weight half-tile: 64-byte rows; position wpos of row holds source chunk wpos ^ f(row), f = (-(row >> 2)) & 3:
        // conflict-free for the 16 rows x 4 chunks block one ds_read_b128 of this MFMA shape covers
        const int wrow = tid >> 2, wpos = tid & 3;
        c.wsrc = W + (long)wrow * (9 * cin) + ((wpos ^ ((0 - (wrow >> 2)) & 3)) * 8);
    }
    // ---- prologue: slab of chunk 0, weight half-tiles 0..2, the four zero rows; the DMA is issued first, the
    // per-lane setup below runs while it is in flight
    if (tid < 128) *(uint32_t *)(lds + kCvZeroOff + (tid >> 5) * 2048 + (tid & 31) * 4) = 0u;
#pragma unroll
    for (int it = 0; it < 5; ++it) cv_glds16(X + c.xsrc[it], lds + kCvAOff + (it < 4 ? it * 64 : 224) * 128 + c.wave_dst);
#pragma unroll
    for (int u = 0; u < kCvAhead; ++u) {
        const _Float16 *s = c.wsrc + (u % 9) * cin + (u / 9) * 32;
        unsigned char *d = lds + u * kCvWBytes + c.wave_dst;
        cv_glds16(s, d);
        cv_glds16(s + 128l * (9 * cin), d + 8192);
    }
    c.a_off = (wm * 128 + r) * 64 + ((q4 ^ ((0 - (r >> 2)) & 3)) << 4);
    c.brow = kCvHalo + wn * 64 + r;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int pos = (int)((p0 + wn * 64 + n * 16 + r) % 90), rank = pos / 9, file = pos - rank * 9;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (rank + dy >= 0 && rank + dy <= 9 && file + dx >= 0 && file + dx <= 8) m |= 1u << t;
        }
        c.vmask[n] = m;
    }

    // the accumulators start at the bias (loaded in the shadow of the prologue DMA): nothing left to add in the epilogue
    cv_f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 bv = *(const float4 *)(bias + wm * 128 + i * 16 + 4 * q4);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            acc[i][n][0] = bv.x; acc[i][n][1] = bv.y; acc[i][n][2] = bv.z; acc[i][n][3] = bv.w;
        }
    }

    cv_wait_vm<4>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int ring_rd = 0, ring_wr = kCvAhead;
    cv_half8 alo[4], ahi[4], b0[4], b1[4];
    CvTap tap = cv_tap<0>(c, kCvAOff);
    cv_read_w<0>(c, 0, alo);
    cv_read_x<0, 0>(c, tap, b0);
#ifdef CCZ_STAMPS
    const unsigned long long st_loop0 = cv_stamp(), st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int chunk = 0; chunk <= c.cmask; ++chunk) {
#define CV_HE(u) cv_halfstep<u>(c, acc, chunk, ring_rd, ring_wr, tap, alo, ahi, b0, b1)
#define CV_HO(u) cv_halfstep<u>(c, acc, chunk, ring_rd, ring_wr, tap, alo, ahi, b1, b0)
        CV_HE(0); CV_HO(1); CV_HE(2); CV_HO(3); CV_HE(4); CV_HO(5); CV_HE(6); CV_HO(7); CV_HE(8);
        CV_HO(9); CV_HE(10); CV_HO(11); CV_HE(12); CV_HO(13); CV_HE(14); CV_HO(15); CV_HE(16); CV_HO(17);
#undef CV_HE
#undef CV_HO
    }
    cv_wait_vm<0>(); // the wrapped-around DMA loads must land before the LDS is reused / released
#ifdef CCZ_STAMPS
    const unsigned long long st_loop1 = cv_stamp(), st_real1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: lane = pixel l & 15 of tile n, registers e = output channels 16 m + 4 (l >> 4) + e
    const int prow = lane >> 4, piece = lane & 15;
    const long pbase = p0 + wn * 64 + prow;
    const long gcol = wm * 128 + piece * 8;
    const bool full = p0 + kCvBM <= M; // whole tile inside the tensor (always, when boards * 90 is a multiple of 256)
    cv_half8 rv[16];
    if (RES && full) { // the residual rows are requested before the transposition, which hides their latency
#pragma unroll
        for (int it = 0; it < 16; ++it) rv[it] = *(const cv_half8 *)(R + (pbase + it * 4) * kCvC + gcol);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    unsigned char *const eb = lds + w * (64 * kCvERow);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int col = m * 16 + 4 * q4;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            cv_half4 o;
            o[0] = (_Float16)acc[m][n][0];
            o[1] = (_Float16)acc[m][n][1];
            o[2] = (_Float16)acc[m][n][2];
            o[3] = (_Float16)acc[m][n][3];
            *(cv_half4 *)(eb + (n * 16 + r) * kCvERow + col * 2) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
        const cv_half8 zero = (cv_half8)(_Float16)0;
        if (full) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                cv_half8 v = *(const cv_half8 *)(eb + (it * 4 + prow) * kCvERow + piece * 16);
                if (RES) v = v + rv[it];
                if (relu) v = __builtin_elementwise_max(v, zero);
                *(cv_half8 *)(Y + (pbase + it * 4) * kCvC + gcol) = v;
            }
        } else {
            for (int it = 0; it < 16; ++it) {
                const long p = pbase + it * 4;
                if (p >= M) break;
                cv_half8 v = *(const cv_half8 *)(eb + (it * 4 + prow) * kCvERow + piece * 16);
                if (RES) v = v + *(const cv_half8 *)(R + p * kCvC + gcol);
                if (relu) v = __builtin_elementwise_max(v, zero);
                *(cv_half8 *)(Y + p * kCvC + gcol) = v;
            }
        }
    }
#ifdef CCZ_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long st_end = cv_stamp();
    if (blockIdx.x < 2048 && (tid == 0 || tid == 256)) {
        unsigned long long *o = g_cv_stamps + (blockIdx.x * 2 + (tid >> 8)) * 16;
        o[0] = st_loop0; o[1] = st_loop1; o[2] = st_end; o[3] = st_real0; o[4] = st_real1;
        o[10] = st_prolog;
    }
#endif
}

} // namespace ccz
