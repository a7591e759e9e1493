// cczero_conv_g16e.h -- the EDGE tiles of the group-of-16 tower convolution (round 4).
//
// k_conv3x3_g16 (cczero_conv_g16.h) cuts a 16-board group into five tiles of two ranks. The first tile has no rank above its first
// rank, the last none below its second: there the wave row that owns the edge rank multiplies zeroed slab rows in three of the nine
// taps -- 6.7 % of all MFMAs -- and nothing cheap removes them: which wave it is, is a run-time property (tile class x wave row), a
// branch around the MFMAs costs the kernel its register allocation (256 of 256, no slack), and hidden in inline asm it is 34 %
// slower (profiles/r04_conv_g16.json). This kernel makes the dead taps a COMPILE-TIME property of a tile instead:
//
//   * an edge tile = the SAME edge rank (rank 0, or rank 9) of TWO groups: wave row 0 owns it for group A, wave row 1 for group B.
//     Both wave rows then lose the same three taps (rank 0: dy = -1, rank 9: dy = +1): a chunk is SIX half-steps -- six weight
//     half-tiles, six barriers, 192-208 MFMAs per wave -- instead of nine, and no slab row is ever zeroed.
//   * top and bottom run the same code: the slab of a wave row holds ITS group's two ranks next to the edge (ranks 0, 1 resp. 8, 9)
//     as sub-ranks 0 and 1; the six live taps read sub-rank 0 in their first three half-steps and sub-rank 1 in the last three
//     (top: dy = 0 then +1; bottom: dy = -1 then 0 -- ascending tap order either way: each accumulator adds the same products in
//     the same order as in the other kernels); what differs is a run-time tap offset into the packed weights and the row offsets.
//   * the other eight ranks of a group are four ordinary two-rank tiles of k_conv3x3_g16 (mode 1: ranks 1-2, 3-4, 5-6, 7-8, every
//     neighbour rank exists). Per pair of groups: 8 + 2 tiles where there were 10, the two edge tiles ~30 % shorter.
//   * it is a second kernel (its own register allocation: 256 VGPRs, no spill) and therefore a second launch per layer, right behind
//     the middle launch in the same stream (csrc/cczero.hip conv3x3_launch: no gap between the two in the kernel trace).
//
// Everything else -- ring of five weight half-tiles three ahead, one raw s_barrier per half-step with counted vmcnt, pixel fragments
// shared by the three dx taps of a dy and refilled in place, LDS-transposed epilogue -- is k_conv3x3_g16's. The next chunk's slab
// is staged in half-steps 0-2 (2 + 2 + 1 pieces) instead of taps 1-5: its first fragments are read in half-step 5.
#pragma once
#include "cczero_conv_g16.h"

namespace ccz {

__host__ __device__ constexpr int g5e_nslab(int h) { return h == 0 ? 2 : h == 1 ? 2 : h == 2 ? 1 : 0; } // slab pieces issued in half-step h
// DMA loads younger than the weight half-tile the NEXT half-step reads (issue order per half-step: slab pieces, 2 weight loads)
__host__ __device__ constexpr int g5e_vmcnt(int h) { return 4 + g5e_nslab(h) + g5e_nslab((h + 5) % 6); }

// One half-step of an edge tile: live tap H = J % 6 of a 32-channel chunk (tap index H + toff in the packed weights), J = its
// index inside the unrolled pair of chunks (A register set = J & 1, slab buffer = J / 6). vb[BUF] = this lane's fragment base of
// the wave row's sub-rank 0 in slab buffer BUF; sub-rank d = H / 3 is at + 9 * 1024 d.
template <int J>
__device__ __forceinline__ void g5e_step(const G5Ctx &c, cv_f32x4 (&acc)[4][9], int chunk, int &ring_rd, int &ring_wr,
                                          cv_half8 (&a0)[4], cv_half8 (&a1)[4], cv_half8 (&b)[9], int toff)
{
    constexpr int H = J % 6, BUF = J / 6;
    constexpr int Hn = (H + 1) % 6, BUFn = (H == 5) ? 1 - BUF : BUF;
    cv_half8 (&acur)[4] = (J & 1) ? a1 : a0;
    cv_half8 (&anxt)[4] = (J & 1) ? a0 : a1;
    unsigned char *const lds = c.lds;
    // as in g5_step: the nine fragments of a sub-rank serve its three dx taps (cell N multiplies b[N + dx]) and are refilled in
    // place for the next sub-rank during the dx = +1 tap; the order is pinned, one scheduling region per cell
#define G5E_CELL(N)                                                                                                       \
    {                                                                                                                     \
        if constexpr (H % 3 == 2)                                                                                         \
            b[N] = *(const cv_half8 *)(lds + c.vb[BUFn] + ((Hn / 3) * 9 + N) * 1024);                                     \
        if constexpr (g5_on_board<H, N>()) {                                                                              \
            constexpr int NB = N + H % 3 - 1;                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
                acc[i][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[i], b[NB], acc[i][N], 0, 0, 0);                   \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
#define G5E_CELLS(LO, HI)                                                                                                 \
    if constexpr (LO <= 0 && 0 < HI) G5E_CELL(0) if constexpr (LO <= 1 && 1 < HI) G5E_CELL(1) if constexpr (LO <= 2 && 2 < HI) G5E_CELL(2) \
    if constexpr (LO <= 3 && 3 < HI) G5E_CELL(3) if constexpr (LO <= 4 && 4 < HI) G5E_CELL(4) if constexpr (LO <= 5 && 5 < HI) G5E_CELL(5) \
    if constexpr (LO <= 6 && 6 < HI) G5E_CELL(6) if constexpr (LO <= 7 && 7 < HI) G5E_CELL(7) if constexpr (LO <= 8 && 8 < HI) G5E_CELL(8)
    constexpr int H2 = (H + kG5Ahead) % 6;
    const int chunk2 = (chunk + (H + kG5Ahead >= 6 ? 1 : 0)) & c.cmask;
    G5E_CELLS(0, 1)
    if constexpr (g5e_nslab(H) > 0) { // the next chunk's slab: passes 0,1 | 2,3 | 4 in half-steps 0 | 1 | 2
        const int nxt = (chunk + 1) & c.cmask; // past the last chunk: re-stage chunk 0 into the free buffer (keeps every count static)
        constexpr int p0 = H * 2;
        cv_glds16(c.X + (c.xoff[p0] + (unsigned)(nxt * 32)), lds + kG5AOff + (1 - BUF) * kG5SlabBytes + (p0 < 4 ? p0 * 8192 + c.wave_dst : c.wave_dst4));
        if constexpr (g5e_nslab(H) > 1)
            cv_glds16(c.X + (c.xoff[p0 + 1] + (unsigned)(nxt * 32)), lds + kG5AOff + (1 - BUF) * kG5SlabBytes + ((p0 + 1) * 8192 + c.wave_dst));
        __builtin_amdgcn_sched_barrier(0);
    }
    G5E_CELLS(1, 2)
    {
        unsigned wo = c.woff;
        asm volatile("" : "+v"(wo)); // the address is formed here, per half-step (as in g5_step)
        const unsigned o = wo + (unsigned)((H2 + toff + 9 * chunk2) * 8192); // half-tile (chunk2, tap H2 + toff): one contiguous 16 KB block
        const unsigned o2 = o + 4096u;
        unsigned char *const d = lds + ring_wr * kG5WBytes + c.wave_dst;
        cv_glds16(c.W + o, d);
        __builtin_amdgcn_sched_barrier(0);
        G5E_CELLS(2, 3)
        cv_glds16(c.W + o2, d + 8192);
        __builtin_amdgcn_sched_barrier(0);
    }
    G5E_CELLS(3, kG5Split)

    ring_rd = ring_rd + 1 == kG5Ring ? 0 : ring_rd + 1;
    cv_wait_vm<g5e_vmcnt(H)>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    {
        const unsigned char *wa = lds + (ring_rd * kG5WBytes + c.a_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) anxt[i] = *(const cv_half8 *)(wa + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
    G5E_CELLS(kG5Split, 9)
    ring_wr = ring_wr + 1 == kG5Ring ? 0 : ring_wr + 1;
#undef G5E_CELLS
#undef G5E_CELL
}

// X, Y, R: rows in the group-of-16 layout, M rows (a multiple of 1440); W packed (k_pack_conv_weights_g16). grid = 2 * ceil(groups
// / 2): workgroup e = pair e / 2 of groups, side e & 1 (0: rank 0, 1: rank 9). live_rows / row0: as k_conv3x3_g16 (planned boundary).
template <bool RES, bool HEADS>
__device__ __forceinline__ void g5e_tile(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                         const float *__restrict__ bias, const _Float16 *R, _Float16 *Y, int M,
                                         int relu, int cin, const int *live_rows, int row0, const G5Heads &ha)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kG5Lds];
    [[maybe_unused]] long first_board = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int wm = w & 3, wn = w >> 2;
    int groups = M / 1440;
    if (live_rows) { // the same cut of the live groups into n_parts equal ranges as k_conv3x3_g16
        const int part = row0 & 0xffff, n_parts = row0 >> 16;
        const int G = (*live_rows + 15) >> 4;
        const int per = (G + n_parts - 1) / n_parts;
        const int first = part * per;
        int live = G - first;
        live = live < 0 ? 0 : (live > per ? per : live);
        live = live > groups ? groups : live;
        groups = live;
        const long off = (long)first * 1440 * kCvC;
        X += (long)first * 1440 * cin;
        if (!HEADS) Y += off; // (HEADS: there is no output tensor)
        if (RES) R += off;
        first_board = (long)first * 16;
    }
    const int e = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
    if (e >= 2 * ((groups + 1) >> 1)) return;
    const int side = e & 1, gA = (e >> 1) * 2;
    const bool dup = gA + 1 >= groups;                       // an odd group count: the last pair is one group twice (wave row 1 stores nothing)
    const int gB = dup ? gA : gA + 1;
    const int toff = __builtin_amdgcn_readfirstlane(side ? 0 : 3); // bottom: taps 0..5 (dy = -1, 0); top: taps 3..8 (dy = 0, +1)
    const long srcA = (long)gA * 1440 + (side ? 8 : 0) * 144;      // the two ranks next to the edge: slab sub-ranks 0, 1
    const long srcB = (long)gB * 1440 + (side ? 8 : 0) * 144;
    const long outA = (long)gA * 1440 + (side ? 9 : 0) * 144;      // the edge rank itself
    const long outB = (long)gB * 1440 + (side ? 9 : 0) * 144;
    relu &= 1;

    G5Ctx c;
    c.lds = lds;
    c.X = X;
    c.W = W;
    c.wave_dst = w * 1024;
    c.lane16 = lane * 16;
    c.wave_dst4 = (w < 4 ? 4 : 3) * 8192 + w * 1024;
    c.cin = cin;
    c.cmask = (cin >> 5) - 1;
    {
        // slab row sr: 0..287 = group A's rows srcA + sr, 288..575 = group B's; 64-byte rows, position pos of row sr holds source
        // chunk pos ^ f(sr), f = (-(sr >> 2)) & 3 (the swizzle of k_conv3x3_g16)
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int piece = (it < 4 || w < 4) ? it * 512 + tid : 3 * 512 + tid; // waves 4-7 repeat pass 3 (same bytes, same place)
            const int sr = piece >> 2, pos = piece & 3;
            const int schunk = pos ^ ((0 - (sr >> 2)) & 3);
            const long p = sr < 288 ? srcA + sr : srcB + (sr - 288);
            c.xoff[it] = (unsigned)(p * cin + schunk * 8);
        }
        c.woff = (unsigned)(tid * 8);
    }
    // ---- prologue: slab of chunk 0, weight half-tiles of the first three live taps
#pragma unroll
    for (int it = 0; it < 5; ++it) cv_glds16(X + c.xoff[it], lds + kG5AOff + (it < 4 ? it * 8192 + c.wave_dst : c.wave_dst4));
#pragma unroll
    for (int u = 0; u < kG5Ahead; ++u) {
        const unsigned o = c.woff + (unsigned)((u + toff) * 8192);
        unsigned char *d = lds + u * kG5WBytes + c.wave_dst;
        cv_glds16(W + o, d);
        cv_glds16(W + (o + 4096u), d + 8192);
    }
    const int lane1 = r * 64 + ((q4 ^ ((0 - (r >> 2)) & 3)) << 4);
    c.a_off = wm * 4096 + lane1;                               // rows 64 wm + 16 i + r of the half-tile
    c.vb[0] = kG5AOff + wn * 18 * 1024 + lane1;                // slab cell 18 wn + 9 d + n, row r of it
    c.vb[1] = c.vb[0] + kG5SlabBytes;

    cv_f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 bv = *(const float4 *)(bias + wm * 64 + i * 16 + 4 * q4);
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            acc[i][n][0] = bv.x; acc[i][n][1] = bv.y; acc[i][n][2] = bv.z; acc[i][n][3] = bv.w;
        }
    }

    cv_wait_vm<4>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int ring_rd = 0, ring_wr = kG5Ahead;
    cv_half8 a0[4], a1[4], b[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = *(const cv_half8 *)(lds + c.a_off + i * 1024);
#pragma unroll
    for (int n = 0; n < 9; ++n) b[n] = *(const cv_half8 *)(lds + c.vb[0] + n * 1024); // sub-rank 0
    for (int chunk = 0; chunk <= c.cmask; chunk += 2) {
#define G5E_S(j) g5e_step<j>(c, acc, chunk + (j) / 6, ring_rd, ring_wr, a0, a1, b, toff)
        G5E_S(0); G5E_S(1); G5E_S(2); G5E_S(3); G5E_S(4); G5E_S(5);
        G5E_S(6); G5E_S(7); G5E_S(8); G5E_S(9); G5E_S(10); G5E_S(11);
#undef G5E_S
    }
    cv_wait_vm<0>(); // the wrapped-around DMA loads must land before the LDS is reused / released

    // ---- epilogue (k_conv3x3_g16's): the [row][channel] image in LDS, rows 0..143 = group A's edge rank, 144..287 = group B's;
    // wave w owns image rows 36 w .. 36 w + 35 and moves whole 512-byte rows (residual in, output out)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = wm * 64 + i * 16 + 4 * q4;
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            cv_half4 o;
            o[0] = (_Float16)acc[i][n][0];
            o[1] = (_Float16)acc[i][n][1];
            o[2] = (_Float16)acc[i][n][2];
            o[3] = (_Float16)acc[i][n][3];
            *(cv_half4 *)(lds + ((wn * 9 + n) * 16 + r) * kG5ERow + col * 2) = o;
        }
    }
    const int prow = lane >> 5, piece = lane & 31;
    const long pbase = (w < 4 ? outA : outB) + (w & 3) * 36 + prow;
    const bool store = !(dup && w >= 4);
    cv_half8 rv[18];
    if (RES) {
#pragma unroll
        for (int it = 0; it < 18; ++it) rv[it] = *(const cv_half8 *)(R + (pbase + it * 2) * kCvC + piece * 8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (store) {
        const cv_half8 zero = (cv_half8)(_Float16)0;
        unsigned char *eb = lds + (w * 36 + prow) * kG5ERow + piece * 16;
#pragma unroll
        for (int it = 0; it < 18; ++it) {
            cv_half8 v = *(const cv_half8 *)(eb + it * 2 * kG5ERow);
            if (RES) v = v + rv[it];
            if (relu) v = __builtin_elementwise_max(v, zero);
            if constexpr (HEADS) *(cv_half8 *)(eb + it * 2 * kG5ERow) = v;   // the finished row stays in the image; Y is not written
            else *(cv_half8 *)(Y + (pbase + it * 2) * kCvC + piece * 8) = v;
        }
    }
    if constexpr (HEADS) { // the heads on the finished rows (cczero_conv_g16.h g5_heads_phase): image rows 0..143 group A's edge rank, 144..287 group B's
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const long board0[2] = {first_board + (long)gA * 16, first_board + (long)gB * 16};
        const int pos0[2] = {side ? 81 : 0, side ? 81 : 0};
        g5_heads_phase(lds, w, lane, ha, board0, pos0, dup);
    }
}

template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_g16_edge(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                            const float *__restrict__ bias, const _Float16 *R, _Float16 *Y, int M,
                                                            int relu, int cin, const int *live_rows, int row0)
{
    g5e_tile<RES, false>(X, W, bias, R, Y, M, relu, cin, live_rows, row0, G5Heads{});
}

__global__ __launch_bounds__(512) void k_conv3x3_g16_edge_heads(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                                  const float *__restrict__ bias, const _Float16 *R, _Float16 *Y,
                                                                  int M, int relu, int cin, const int *live_rows, int row0, G5Heads ha)
{
    if (live_rows) ha.nb = *live_rows; // planned boundary: the pointers are the whole batch's, M only the capacity of this part
    g5e_tile<true, true>(X, W, bias, R, Y, M, relu, cin, live_rows, row0, ha);
}

} // namespace ccz
