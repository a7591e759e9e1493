// cczero_conv2.h -- second form of the tower convolution (same contract as k_conv3x3_c256, cczero_conv.h):
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// TWO INDEPENDENT 4-WAVE WORKGROUPS PER CU instead of one 8-wave workgroup: a workgroup computes 256 pixels x 128 output
// channels (one wave per SIMD, 128 co x 64 px per wave = the same 8 x 4 accumulator tiles of v_mfma_f32_16x16x32_f16 as
// before), the two workgroups of a CU are scheduled independently, so that (1) a per-half-step barrier couples four waves
// instead of eight and the partner workgroup's wave on the same SIMD keeps the matrix pipe busy meanwhile, (2) the prologue
// and epilogue of one workgroup run in the shadow of the other's K loop. To fit two workgroups into 160 KB of LDS the
// activation slab is staged in 32-channel chunks (18 KB, double buffered) and the weight ring holds four 8-KB half-tiles.
//
// LDS images ("blocks"): 16 rows x 32 k (fp16) = 1 KB = exactly what ONE global_load_lds_dwordx4 wave-instruction writes,
// laid out piece-major: byte (row, piece) at piece * 256 + row * 16 (piece = 8 k). A fragment read (lane l: row l & 15,
// piece l >> 4) of 16 consecutive rows starting at ANY row offset touches 16 distinct 16-byte bank groups per ds_read_b128
// lane group (each hardware lane group holds every l & 15 exactly once), so both operands are conflict-free for every tap
// offset without a swizzle, and the tiles of one operand differ by instruction immediates (n * 1024).
//   K order = 8 chunks of 32 input channels (outer) x 9 taps (inner) = 72 half-steps, one MFMA k-step each.
//
// STATUS: EXPERIMENTAL, NOT SHIPPED (compiled only with -DCCZ_CONV2: `make -C chinesechesszero_amd/csrc ab NAME=v2 ABFLAGS=-DCCZ_CONV2`,
// then flags bit 2 of ccz_conv3x3_c256_f16 selects it). Correct (same error against float32 as k_conv3x3_c256, 1 fp16 ulp
// apart from it: another summation order), LDS bank-conflict cycles 5 % of LDS-active against 19 %, but SLOWER where it
// counts (profiles/r02_conv_v2.json, one box): 4096 boards 383-388 us against 366 per layer in isolation, 341 against 323
// in the workload (best of five group/chain settings); 1024 boards 96.7 against 110.5 in isolation (finer tail) but 7.15
// against 6.90 ms per step in the workload, where the two concurrent chains already fill the tail. +18 % VALU per wave (tap
// addressing every half-step instead of every second one, ~9 v_readlane of spilled scalars) and +6 % wave cycles; a three
// half-tile lead of the weight DMA instead of two changed nothing, and neither did removing the tap masks (-12 VALU per
// half-step, timing ablation -DC2_NOMASK): the deficit is not instruction issue, so a leaner addressing scheme would not close it.
#pragma once
#include "../../chinesechesszero_amd/csrc/cczero_conv.h"

namespace ccz {

constexpr int kC2BM = 256;                               // pixels per workgroup
constexpr int kC2BN = 128;                               // output channels per workgroup
constexpr int kC2Blocks = 18;                            // slab blocks of 16 rows: 288 rows, 276 used (256 + 2 x 10 halo)
constexpr int kC2SlabBytes = kC2Blocks * 1024;           // one 32-channel slab
constexpr int kC2WBytes = 8 * 1024;                      // one half-step of weights: 128 co x 32 k
constexpr int kC2Ring = 4;
#ifndef C2_AHEAD
#define C2_AHEAD 2
#endif
constexpr int kC2Ahead = C2_AHEAD;                       // 2: half-tile g + 2 goes into slot (g - 2) % 4, last read a whole half-step ago;
                                                         // 3: into slot (g - 1) % 4, free once every wave has passed barrier g - 1
constexpr int kC2AOff = kC2Ring * kC2WBytes;             // [ring 32 KB | slab 0 | slab 1 | 4 zero blocks]
constexpr int kC2ZeroOff = kC2AOff + 2 * kC2SlabBytes;
constexpr int kC2Lds = kC2ZeroOff + 4 * 1024;            // 73,728 B: two workgroups per CU
static_assert(4 * 64 * kCvERow <= kC2ZeroOff, "epilogue transpose must fit the operand buffers");
static_assert(2 * kC2Lds <= 160 * 1024, "two workgroups per CU");

// the slab of the next chunk is issued as one block per wave at taps 1..5 of the current chunk
__host__ __device__ constexpr int c2_slab_pass(int tap) { return (tap >= 1 && tap <= 5) ? tap - 1 : -1; }
// loads younger than the weight half-tile the NEXT half-step reads (issue order per half-step: slab block, 2 weight loads):
// this half-step's two weight loads and its slab block
__host__ __device__ constexpr int c2_vmcnt(int tap) { return 2 * (kC2Ahead - 1) + (c2_slab_pass(tap) >= 0 ? 1 : 0) + (kC2Ahead > 2 && c2_slab_pass((tap + 8) % 9) >= 0 ? 1 : 0); }

struct C2Ctx {
    unsigned char *lds;
    const _Float16 *X;
    int xsrc[5];          // per slab block this wave stages: element offset of this lane's 16-byte source in X (chunk 0)
    const _Float16 *wsrc; // this lane's 16-byte source in W for co-tile 2w (tap 0, chunk 0); co-tile 2w + 1: + 16 rows
    int wdst;             // LDS offset of co-tile 2w inside a ring slot
    int a_off;            // weight fragment offset inside a ring slot (co-tile 0; tile m: + 1024 m)
    int brow;             // slab row of this lane's pixel of tile 0 at tap offset 0 (tile n: + 16 n)
    int q4off;            // (lane >> 4) * 256
    unsigned vmask[4];    // per pixel tile: bit t set = tap t stays on the board
    int cin;              // 256 (tower) or 64 (stem)
    int nchunk;           // cin / 32
    int w;                // wave index in the workgroup (scalar)
};

template <int TAP> __device__ __forceinline__ int c2_tap_off(const C2Ctx &c, int slab_base)
{
    constexpr int delta = 9 * (TAP / 3 - 1) + (TAP % 3 - 1);
    int br = c.brow;
    asm volatile("" : "+v"(br)); // keep this arithmetic in the loop (hoisted for all nine taps it costs registers)
    const int R = br + delta;
    return slab_base + ((R >> 4) << 10) + ((R & 15) << 4) + c.q4off;
}

template <int HI> __device__ __forceinline__ void c2_read_w(const C2Ctx &c, int slot, cv_half8 (&a)[4])
{
    const unsigned char *wa = c.lds + (slot * kC2WBytes + c.a_off);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const cv_half8 *)(wa + (HI * 4 + i) * 1024);
}

template <int TAP> __device__ __forceinline__ void c2_read_x(const C2Ctx &c, int off0, cv_half8 (&b)[4])
{
    const int zoff = kC2ZeroOff + (off0 & 0x3f0); // same bank group as the real row: piece * 256 + (row & 15) * 16
#pragma unroll
    for (int n = 0; n < 4; ++n) {
#ifdef C2_NOMASK // timing ablation only (wrong results at the board edges)
        const int off = off0;
        (void)zoff;
#else
        const bool ok = (c.vmask[n] >> TAP) & 1u; // a tap that leaves the board reads a zero block
        const int off = ok ? off0 : zoff;
#endif
        b[n] = *(const cv_half8 *)(c.lds + off + n * 1024);
    }
}

template <int HI>
__device__ __forceinline__ void c2_mfma16(cv_f32x4 (&acc)[8][4], const cv_half8 (&a)[4], const cv_half8 (&b)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[HI * 4 + i][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[n], acc[HI * 4 + i][n], 0, 0, 0);
}

// one half-step = tap TAP of input-channel chunk `chunk`; (bcur, bnxt) alternate between consecutive half-steps
template <int TAP>
__device__ __forceinline__ void c2_halfstep(const C2Ctx &c, cv_f32x4 (&acc)[8][4], int chunk, int &ring_rd, int &ring_wr, int &xoff,
                                              cv_half8 (&alo)[4], cv_half8 (&ahi)[4], cv_half8 (&bcur)[4], cv_half8 (&bnxt)[4])
{
    unsigned char *const lds = c.lds;
    constexpr int pass = c2_slab_pass(TAP);
    if constexpr (pass >= 0) {
        int nxt = chunk + 1;
        nxt = nxt == c.nchunk ? 0 : nxt; // past the last chunk: re-stage chunk 0 into the free buffer (keeps every count static)
        const int blk = c.w + 4 * pass < kC2Blocks ? c.w + 4 * pass : kC2Blocks - 1;
        cv_glds16(c.X + (c.xsrc[pass] + nxt * 32), lds + kC2AOff + ((chunk + 1) & 1) * kC2SlabBytes + blk * 1024);
    }
    {
        constexpr int T2 = (TAP + kC2Ahead) % 9;
        int chunk2 = chunk + (TAP + kC2Ahead >= 9 ? 1 : 0);
        chunk2 = chunk2 == c.nchunk ? 0 : chunk2;
        const _Float16 *s = c.wsrc + T2 * c.cin + chunk2 * 32;
        unsigned char *d = lds + ring_wr * kC2WBytes + c.wdst;
        cv_glds16(s, d);
        cv_glds16(s + 16l * (9 * c.cin), d + 1024);
    }
    c2_read_w<1>(c, ring_rd, ahi);
    c2_mfma16<0>(acc, alo, bcur);
    __builtin_amdgcn_iglp_opt(1);

    ring_rd = (ring_rd + 1) & (kC2Ring - 1);
    ring_wr = (ring_wr + 1) & (kC2Ring - 1);
    cv_wait_vm<c2_vmcnt(TAP)>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    constexpr int Tn = (TAP + 1) % 9;
    xoff = c2_tap_off<Tn>(c, kC2AOff + ((chunk + (TAP == 8 ? 1 : 0)) & 1) * kC2SlabBytes);
    c2_read_w<0>(c, ring_rd, alo);
    c2_read_x<Tn>(c, xoff, bnxt);
    c2_mfma16<1>(acc, ahi, bcur);
}

template <bool RES>
__global__ __launch_bounds__(256, 2) void k_conv3x3_v2(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                          const float *__restrict__ bias, const _Float16 *R,
                                                          _Float16 *Y, int M, int relu, int cin, int ntiles)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kC2Lds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    // block -> (pixel tile, output-channel half): blocks b and b + 8 share an XCD (round-robin dispatch), so the two halves of
    // a pixel tile are placed 8 blocks apart: the second one finds the activation tile in that XCD's L2
    const int bx = blockIdx.x;
    const int grp = bx >> 4, in = bx & 15;
    int tile = grp * 8 + (in & 7);
    const int half = in >> 3;
    if (tile >= ntiles) return;
    if (relu & 2) tile = ntiles - 1 - tile; // flags bit 1: tiles in descending order (as k_conv3x3_c256)
    const long p0 = (long)tile * kC2BM;

    C2Ctx c;
    c.lds = lds;
    c.X = X;
    c.cin = cin;
    c.nchunk = cin >> 5;
    c.q4off = q4 << 8;
    c.w = w;
    relu &= 1;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        int b = w + 4 * it;
        b = b < kC2Blocks ? b : kC2Blocks - 1; // waves 2, 3 stage block 17 twice (same bytes): every wave issues five loads
        long p = p0 - kCvHalo + b * 16 + r;
        p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
        c.xsrc[it] = (int)(p * cin + q4 * 8);
    }
    c.wsrc = W + (long)(half * kC2BN + 2 * w * 16 + r) * (9 * cin) + q4 * 8;
    c.wdst = 2 * w * 1024;

    // ---- prologue: zero blocks, slab of chunk 0, weight half-tiles 0..2; per-lane setup runs while the DMA is in flight
    for (int i = tid; i < 1024; i += 256) *(uint32_t *)(lds + kC2ZeroOff + i * 4) = 0u;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        int b = w + 4 * it;
        b = b < kC2Blocks ? b : kC2Blocks - 1;
        cv_glds16(X + c.xsrc[it], lds + kC2AOff + b * 1024);
    }
#pragma unroll
    for (int u = 0; u < kC2Ahead; ++u) {
        const _Float16 *s = c.wsrc + u * cin;
        unsigned char *d = lds + u * kC2WBytes + c.wdst;
        cv_glds16(s, d);
        cv_glds16(s + 16l * (9 * cin), d + 1024);
    }
    c.a_off = (q4 << 8) + (r << 4);
    c.brow = kCvHalo + w * 64 + r;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int pos = (int)((p0 + w * 64 + n * 16 + r) % 90), rank = pos / 9, file = pos - rank * 9;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3 - 1, dx = t % 3 - 1;
            if (rank + dy >= 0 && rank + dy <= 9 && file + dx >= 0 && file + dx <= 8) m |= 1u << t;
        }
        c.vmask[n] = m;
    }

    cv_f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 bv = *(const float4 *)(bias + half * kC2BN + i * 16 + 4 * q4);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            acc[i][n][0] = bv.x; acc[i][n][1] = bv.y; acc[i][n][2] = bv.z; acc[i][n][3] = bv.w;
        }
    }

    cv_wait_vm<2 * (kC2Ahead - 1)>(); // slab 0 and weight half-tile 0 have landed (the later half-tiles may still be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int ring_rd = 0, ring_wr = kC2Ahead;
    cv_half8 alo[4], ahi[4], b0[4], b1[4];
    int xoff = c2_tap_off<0>(c, kC2AOff);
    c2_read_w<0>(c, 0, alo);
    c2_read_x<0>(c, xoff, b0);
    for (int chunk = 0; chunk < c.nchunk; chunk += 2) {
#define C2_E(t, ch) c2_halfstep<t>(c, acc, ch, ring_rd, ring_wr, xoff, alo, ahi, b0, b1)
#define C2_O(t, ch) c2_halfstep<t>(c, acc, ch, ring_rd, ring_wr, xoff, alo, ahi, b1, b0)
        C2_E(0, chunk); C2_O(1, chunk); C2_E(2, chunk); C2_O(3, chunk); C2_E(4, chunk); C2_O(5, chunk); C2_E(6, chunk); C2_O(7, chunk); C2_E(8, chunk);
        C2_O(0, chunk + 1); C2_E(1, chunk + 1); C2_O(2, chunk + 1); C2_E(3, chunk + 1); C2_O(4, chunk + 1); C2_E(5, chunk + 1); C2_O(6, chunk + 1);
        C2_E(7, chunk + 1); C2_O(8, chunk + 1);
#undef C2_E
#undef C2_O
    }
    cv_wait_vm<0>(); // the wrapped-around DMA loads must land before the LDS is reused / released

    // ---- epilogue (as k_conv3x3_c256): lane = pixel l & 15 of tile n, registers e = output channels 16 m + 4 (l >> 4) + e
    const int prow = lane >> 4, piece = lane & 15;
    const long pbase = p0 + w * 64 + prow;
    const long gcol = half * kC2BN + piece * 8;
    const bool full = p0 + kC2BM <= M;
    cv_half8 rv[16];
    if (RES && full) {
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
}

} // namespace ccz
