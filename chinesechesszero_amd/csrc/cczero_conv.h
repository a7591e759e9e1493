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
//     consecutive output channels of ONE pixel per register group.
//   * one workgroup = 256 consecutive pixels x all 256 output channels; 8 waves as 2 (co) x 4 (pixels),
//     a 128 x 64 accumulator block (4 x 2 tiles of v_mfma_f32_32x32x16_f16) per wave.
//   * K order = 4 input-channel chunks of 64 (outer) x 9 taps x 2 halves of 32 (inner) = 72 half-steps.
//     The activation slab of a chunk (the tile's 256 pixels + a 10-pixel halo either side, 64 channels) is
//     staged ONCE into LDS and read nine times at row offsets 9*dy + dx; a tap that leaves the board reads a
//     zero row instead (per-lane 9-bit validity mask, no arithmetic on the data). Only the weights are
//     staged per half-step, into a ring of five 16 KB half-tiles, three half-steps ahead.
//   * both operands arrive by global_load_lds (16 B per lane, no VGPR round trip) into XOR-swizzled rows
//     (swizzle applied to the SOURCE address and to the read address): conflict-free ds_read_b128.
//   * software pipeline, one barrier per half-step: the fragments of the next 8-MFMA group are read while the
//     current group issues; DMA loads stay in flight across barriers (raw s_barrier + counted s_waitcnt vmcnt,
//     never 0 inside the loop).
//   * epilogue: each wave transposes its 64 pixel x 128 channel block through its own LDS region, so that the
//     residual is read and the output written as whole 256-byte pixel-row segments.
//
// Measured alternatives that did NOT pay (profiles/conv_ab.py, one device, interleaved): staggering the DMA issue of
// the two wave groups (-4.5 %), running the groups half a half-step apart (-2 %), s_setprio around the MFMA groups
// (-11 %), sched_group_barrier-pinned interleave (-1.5 %), fragments read a whole half-step ahead into three
// register sets (0 %), 16 zero rows (one per bank slot) for the masked lanes (-1 %), a ping-pong form with two barriers
// per 8-MFMA group and the wave groups one barrier apart (-15 %).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccz {

typedef _Float16 cv_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 cv_half4 __attribute__((ext_vector_type(4)));
typedef float cv_f32x16 __attribute__((ext_vector_type(16)));

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
constexpr int kCvLds = kCvZeroOff + 128;
constexpr int kCvERow = 272;          // epilogue transpose: bytes per pixel row of a wave's 64 x 128 block (256 + pad)
static_assert(8 * 64 * kCvERow <= kCvLds, "epilogue transpose must fit the operand buffers");

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

struct CvCtx {
    unsigned char *lds;
    const _Float16 *X;
    int xsrc[5];             // per staging pass: element offset of this thread's 16-byte source in X (chunk 0)
    const _Float16 *wsrc;    // this thread's 16-byte source in W (row pass 0, tap 0, chunk 0, half 0)
    int wave_dst;            // w * 1024
    int h;                   // lane >> 5
    int a_off[2];            // weight fragment offsets inside a ring slot (k-sub 0 / 1)
    int brow;                // slab row of this lane's pixel of tile j = 0 at tap offset 0 (tile 1: + 32)
    unsigned vmask[2];
    int dbg;                 // diagnostic build (-DCCZ_STAMPS) only: ablation switches from bits 8.. of the relu argument
};
// slab addressing of one tap, shared by both pixel tiles and all four k-subs of the tap
struct CvTap {
    int base; // LDS offset of the row of tile 0 at this tap
    int sw16; // (((row >> 1) & 7) ^ h) << 4: swizzled position of chunk h
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

template <int TAP> __device__ __forceinline__ CvTap cv_tap(const CvCtx &c, int abase)
{
    constexpr int delta = 9 * (TAP / 3 - 1) + (TAP % 3 - 1);
    int br = c.brow;
    asm volatile("" : "+v"(br)); // keep this arithmetic in the loop: hoisted for all nine taps it costs ~40 VGPRs
    const int row = br + delta;
    CvTap t;
    t.base = abase + row * 128;
    t.sw16 = (((row >> 1) & 7) ^ c.h) << 4;
    return t;
}

// fragments of k-sub Q (16 k) of half-step (TAP, KH): 4 weight tiles (A operand) + 2 pixel tiles (B operand)
template <int TAP, int KH, int Q>
__device__ __forceinline__ void cv_read_frags(const CvCtx &c, const CvTap &t, int ring_slot, cv_half8 (&a)[4], cv_half8 (&b)[2])
{
    const unsigned char *const lds = c.lds;
    if (CV_DBG(c, 64)) return; // ablation: keep the stale fragments
    if (!CV_DBG(c, 256)) {
        const unsigned char *wa = lds + (ring_slot * kCvWBytes + c.a_off[Q]);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *(const cv_half8 *)(wa + i * 2048);
    }
    if (CV_DBG(c, 512)) return;
    const int off0 = t.base + (t.sw16 ^ ((4 * KH + 2 * Q) << 4));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const bool ok = (c.vmask[j] >> TAP) & 1u; // a tap that leaves the board reads the zero row
        const int off = ok ? off0 + j * 4096 : kCvZeroOff;
        b[j] = *(const cv_half8 *)(lds + off);
    }
}

__device__ __forceinline__ void cv_mfma8(const CvCtx &c, cv_f32x16 (&acc)[4][2], const cv_half8 (&a)[4], const cv_half8 (&b)[2])
{
    if (!CV_DBG(c, 8)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[i]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(b[j]));
    }
}

// One half-step (32 k of one tap): on entry (a0, b0) hold its k-sub 0 fragments and `tap` its slab addressing.
//   DMA issue (slab piece of the next chunk, weights 3 half-steps ahead) | read k-sub 1 -> (a1, b1) | 8 MFMA on (a0, b0)
//   | counted vmcnt + barrier: the NEXT half-step's weights are now visible | read next k-sub 0 -> (a0, b0) | 8 MFMA on (a1, b1)
template <int U>
__device__ __forceinline__ void cv_halfstep(const CvCtx &c, cv_f32x16 (&acc)[4][2], int chunk, int &ring_rd, int &ring_wr, CvTap &tap,
                                            cv_half8 (&a0)[4], cv_half8 (&b0)[2], cv_half8 (&a1)[4], cv_half8 (&b1)[2])
{
    constexpr int TAP = U >> 1, KH = U & 1;
    unsigned char *const lds = c.lds;

    constexpr int pass = cv_act_pass(U);
    if constexpr (pass >= 0) if (!CV_DBG(c, 2)) {
        const int nxt = (chunk + 1) & 3; // the last chunk re-stages chunk 0 into the free buffer (keeps every count static)
        cv_glds16(c.X + (c.xsrc[pass] + nxt * 64), lds + kCvAOff + ((chunk + 1) & 1) * kCvABytes + (pass < 4 ? pass * 64 : 224) * 128 + c.wave_dst);
    }
    if (!CV_DBG(c, 1)) {
        // weights of half-step (this + kCvAhead); past the end of the tile the loads wrap to the start (unused)
        constexpr int U2 = (U + kCvAhead) % 18;
        const int chunk2 = (chunk + (U + kCvAhead >= 18 ? 1 : 0)) & 3;
        const _Float16 *s = c.wsrc + (U2 >> 1) * kCvC + chunk2 * 64 + (U2 & 1) * 32;
        unsigned char *d = lds + ring_wr * kCvWBytes + c.wave_dst;
        cv_glds16(s, d);
        cv_glds16(s + 128l * (9 * kCvC), d + 8192);
    }
    cv_read_frags<TAP, KH, 1>(c, tap, ring_rd, a1, b1);
    cv_mfma8(c, acc, a0, b0);

    ring_rd = ring_rd + 1 == kCvRing ? 0 : ring_rd + 1;
    ring_wr = ring_wr + 1 == kCvRing ? 0 : ring_wr + 1;
    cv_wait_vm<cv_vmcnt(U)>();
    __builtin_amdgcn_sched_barrier(0);
    if (!CV_DBG(c, 32)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    constexpr int Un = (U + 1) % 18;
    if constexpr (KH == 1) tap = cv_tap<(Un >> 1)>(c, kCvAOff + ((chunk + (U == 17 ? 1 : 0)) & 1) * kCvABytes);
    cv_read_frags<(Un >> 1), (Un & 1), 0>(c, tap, ring_rd, a0, b0);
    cv_mfma8(c, acc, a1, b1);
}

template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_c256(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                      const float *__restrict__ bias, const _Float16 *R,
                                                      _Float16 *Y, int M, int relu)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kCvLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = w >> 2, wn = w & 3;
    const long p0 = (long)blockIdx.x * kCvBM;

    CvCtx c;
    c.lds = lds;
    c.X = X;
    c.wave_dst = w * 1024;
    c.h = h;
    c.dbg = relu >> 8;
    relu &= 1;
    {
        // activation slab: 128-byte rows, 8 lanes per row, 64 rows per pass; chunk position cpos holds source chunk cpos ^ ((row >> 1) & 7)
        const int srow = tid >> 3, cpos = tid & 7;
        const int schunk = cpos ^ ((srow >> 1) & 7);
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            long p = p0 - kCvHalo + (it < 4 ? it * 64 : 224) + srow;
            p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p); // clamped rows are only ever read by masked taps / unstored pixels
            c.xsrc[it] = (int)(p * kCvC + schunk * 8);
        }
        // weight half-tile: 64-byte rows, 4 lanes per row, 128 rows per pass; position cpos holds source chunk cpos ^ ((row >> 2) & 3)
        const int wrow = tid >> 2, wpos = tid & 3;
        c.wsrc = W + (long)wrow * (9 * kCvC) + ((wpos ^ ((wrow >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) c.a_off[q] = (wm * 128 + r) * 64 + (((2 * q + h) ^ ((r >> 2) & 3)) << 4);
    c.brow = kCvHalo + wn * 64 + r;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pos = (int)((p0 + wn * 64 + j * 32 + r) % 90), rank = pos / 9, file = pos - rank * 9;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (rank + dy >= 0 && rank + dy <= 9 && file + dx >= 0 && file + dx <= 8) m |= 1u << t;
        }
        c.vmask[j] = m;
    }

    cv_f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#ifdef CCZ_STAMPS
    const unsigned long long st_prolog = cv_stamp();
#endif
    // ---- prologue: slab of chunk 0, weight half-tiles 0..2
    if (tid < 32) *(uint32_t *)(lds + kCvZeroOff + tid * 4) = 0u;
#pragma unroll
    for (int it = 0; it < 5; ++it) cv_glds16(X + c.xsrc[it], lds + kCvAOff + (it < 4 ? it * 64 : 224) * 128 + c.wave_dst);
#pragma unroll
    for (int u = 0; u < kCvAhead; ++u) {
        const _Float16 *s = c.wsrc + (u >> 1) * kCvC + (u & 1) * 32;
        unsigned char *d = lds + u * kCvWBytes + c.wave_dst;
        cv_glds16(s, d);
        cv_glds16(s + 128l * (9 * kCvC), d + 8192);
    }
    cv_wait_vm<4>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int ring_rd = 0, ring_wr = kCvAhead;
    cv_half8 a0[4], b0[2], a1[4], b1[2];
    CvTap tap = cv_tap<0>(c, kCvAOff);
    cv_read_frags<0, 0, 0>(c, tap, 0, a0, b0);
#ifdef CCZ_STAMPS
    const unsigned long long st_loop0 = cv_stamp(), st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int chunk = 0; chunk < 4; ++chunk) {
#define CV_HS(u) cv_halfstep<u>(c, acc, chunk, ring_rd, ring_wr, tap, a0, b0, a1, b1)
        CV_HS(0); CV_HS(1); CV_HS(2); CV_HS(3); CV_HS(4); CV_HS(5); CV_HS(6); CV_HS(7); CV_HS(8);
        CV_HS(9); CV_HS(10); CV_HS(11); CV_HS(12); CV_HS(13); CV_HS(14); CV_HS(15); CV_HS(16); CV_HS(17);
#undef CV_HS
    }
    cv_wait_vm<0>(); // the wrapped-around DMA loads must land before the LDS is reused / released
#ifdef CCZ_STAMPS
    const unsigned long long st_loop1 = cv_stamp(), st_real1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue. Accumulator layout: lane = pixel (column r of tile j), register group g = output channels
    // 8g + 4h .. + 3 of 32-row tile i. Wave-private 272-byte-row LDS image of the 64 x 128 block, then whole rows out.
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier(); // every wave is done with the operand buffers
    __builtin_amdgcn_sched_barrier(0);
    unsigned char *const eb = lds + w * (64 * kCvERow);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = i * 32 + 8 * g + 4 * h;
            const float4 bv = *(const float4 *)(bias + wm * 128 + col);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                cv_half4 o;
                o[0] = (_Float16)(acc[i][j][4 * g + 0] + bv.x);
                o[1] = (_Float16)(acc[i][j][4 * g + 1] + bv.y);
                o[2] = (_Float16)(acc[i][j][4 * g + 2] + bv.z);
                o[3] = (_Float16)(acc[i][j][4 * g + 3] + bv.w);
                *(cv_half4 *)(eb + (j * 32 + r) * kCvERow + col * 2) = o;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // wave-private region: no barrier needed
    __builtin_amdgcn_sched_barrier(0);
    {
        const int prow = lane >> 4, piece = lane & 15;
        const long pbase = p0 + wn * 64 + prow;
        const long gcol = wm * 128 + piece * 8;
        const cv_half8 zero = (cv_half8)(_Float16)0;
        if (p0 + kCvBM <= M) { // whole tile inside the tensor (always, when boards * 90 is a multiple of 256)
            cv_half8 rv[16];
            if (RES) {
#pragma unroll
                for (int it = 0; it < 16; ++it) rv[it] = *(const cv_half8 *)(R + (pbase + it * 4) * kCvC + gcol);
            }
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
