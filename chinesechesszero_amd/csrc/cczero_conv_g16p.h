// cczero_conv_g16p.h -- k_conv3x3_g16 as a PERSISTENT workgroup: several tiles per workgroup, no prologue after the first one.
//
// k_conv3x3_g16 (cczero_conv_g16.h) runs one tile per workgroup and one workgroup per CU: every tile pays its prologue (slab of chunk 0
// + three weight half-tiles by LDS-DMA, then a barrier) and its epilogue (LDS-transposed stores) with the matrix pipe idle. Here a
// workgroup walks a list of tiles (flag CCZ_CONV_G16_PERSISTENT, grid = the number of workgroups):
//
//   * the loop's wrap-around slab staging -- k_conv3x3_g16 re-stages its OWN chunk 0 during the last chunk to keep every DMA count
//     static -- stages the NEXT tile's chunk 0 instead: everything that depends on the tile is a SCALAR (G5Ctx::sx / sxn / zo / zon), the
//     per-thread part of every address is tile-independent, so the unrolled loop body is the same code with two s_cselects in it;
//   * the epilogue image is ONE rank (144 rows x 528 B = 76 KB) instead of the whole tile (152 KB = all of the LDS), in two passes, and
//     the LDS is laid out so that it leaves ring slots 0-2 and slab 0 alone (cczero_conv_g16.h kP5*): the next tile's first three weight
//     half-tiles are DMA'd into slots 0-2 as soon as every wave has left the loop and land while the stores drain;
//   * a slab rank that does not exist (above rank 0, below rank 9: zeroed after it has landed) is staged from the tile's own edge rank
//     instead of from the neighbour group's: the same bytes in LDS afterwards, nothing fetched from another group, no address clamp.
//
// Same operands, same order of additions: the same bits as k_conv3x3_g16 (tests/test_gpu_conv.py).
// XCD-aware order as before: XCD x owns the x-th contiguous eighth of the tiles; its workgroups (blockIdx % 8 == x) take them round
// robin, so that the tiles running at the same time are neighbours.
#pragma once
#include "cczero_conv_g16.h"

namespace ccz {

struct P5Tile {
    int k;   // ranks 2k, 2k + 1 of its group (middle mode: 2 = no edge)
    int p0;  // first tensor row
};

__device__ __forceinline__ P5Tile p5_tile_of(int t, int tiles, int flags)
{
    if (flags & 2) t = tiles - 1 - t;
    P5Tile r;
    if (flags & 4) {
        r.k = 2;
        r.p0 = (t >> 2) * 1440 + 144 + (t & 3) * kG5Rows;
    } else {
        r.k = t % 5;
        r.p0 = t * kG5Rows;
    }
    return r;
}

template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_g16_pers(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                              const float *__restrict__ bias, const _Float16 *R,
                                                              _Float16 *Y, int M, int relu, int cin, const int *live_rows, int row0)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kG5Lds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int wm = w & 3, wn = w >> 2;
    const int mid = relu & 4;
    int tiles = M / 1440 * (mid ? 4 : 5);
    if (live_rows) { // planned evaluator boundary: as g5_tile
        const int part = row0 & 0xffff, n_parts = row0 >> 16;
        const int G = (*live_rows + 15) >> 4;
        const int per = (G + n_parts - 1) / n_parts;
        const int first = part * per;
        int live = G - first;
        live = live < 0 ? 0 : (live > per ? per : live);
        live = live > M / 1440 ? M / 1440 : live;
        M = live * 1440;
        tiles = live * (mid ? 4 : 5);
        const long off = (long)first * 1440 * kCvC;
        X += (long)first * 1440 * cin;
        Y += off;
        if (RES) R += off;
    }
    // this workgroup's tiles: XCD x = blockIdx % 8 owns tiles [start, start + cnt); its workgroups take them round robin
    int t_cur, t_step, t_end;
    {
        const int b = blockIdx.x, x = b & 7, y = b >> 3, nwg = gridDim.x;
        const int ny = (nwg - x + 7) >> 3;              // workgroups on this XCD
        const int per = tiles >> 3, rem = tiles & 7;
        const int start = x * per + (x < rem ? x : rem), cnt = per + (x < rem ? 1 : 0);
        if (y >= cnt) return;
        t_cur = __builtin_amdgcn_readfirstlane(start + y);
        t_step = __builtin_amdgcn_readfirstlane(ny);
        t_end = __builtin_amdgcn_readfirstlane(start + cnt);
    }
    const int tflags = relu & 6;
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
    {   // tile- and pass-independent: row tid / 4 of a staging pass, 16-byte position tid & 3 of it (swizzle as in g5_tile)
        const int sr = tid >> 2, pos = tid & 3;
        const int schunk = pos ^ ((0 - (sr >> 2)) & 3);
        c.xoff[0] = (unsigned)(sr * cin + schunk * 8);
    }
    c.woff = (unsigned)(tid * 8);
    c.m_w0 = __builtin_amdgcn_readfirstlane(w == 0 ? -1 : 0);
    c.m_w3 = __builtin_amdgcn_readfirstlane(w >= 3 ? -1 : 0);
    c.rows4 = __builtin_amdgcn_readfirstlane(w < 4 ? 512 : 384);
    // zero stores (as g5_tile): k = 0 -> piece 0 of every wave and piece 1 of wave 0; k = 4 -> piece 3 of waves 3..7, piece 4 of waves 0..3
    c.za[0] = __builtin_amdgcn_readfirstlane(0 * 8192 + w * 1024 + 1);
    c.za[1] = __builtin_amdgcn_readfirstlane(w == 0 ? 1 * 8192 + w * 1024 + 1 : 0);
    c.zb[0] = __builtin_amdgcn_readfirstlane((w >= 3 ? 3 : 4) * 8192 + w * 1024 + 1);
    c.zb[1] = __builtin_amdgcn_readfirstlane(w == 3 ? 4 * 8192 + w * 1024 + 1 : 0);
    {
        const P5Tile tl = p5_tile_of(t_cur, tiles, tflags);
        c.p0 = __builtin_amdgcn_readfirstlane(tl.p0);
        c.k = __builtin_amdgcn_readfirstlane(tl.k);
    }

    // ---- prologue of the FIRST tile: slab of chunk 0 into slab 0, weight half-tiles 0..2 into ring slots 0..2
    cv_glds16(X + (c.xoff[0] + p5_slab_src<0>(c, c.p0, c.k)), lds + kP5Slab0 + 0 * 8192 + c.wave_dst);
    cv_glds16(X + (c.xoff[0] + p5_slab_src<1>(c, c.p0, c.k)), lds + kP5Slab0 + 1 * 8192 + c.wave_dst);
    cv_glds16(X + (c.xoff[0] + p5_slab_src<2>(c, c.p0, c.k)), lds + kP5Slab0 + 2 * 8192 + c.wave_dst);
    cv_glds16(X + (c.xoff[0] + p5_slab_src<3>(c, c.p0, c.k)), lds + kP5Slab0 + 3 * 8192 + c.wave_dst);
    cv_glds16(X + (c.xoff[0] + p5_slab_src<4>(c, c.p0, c.k)), lds + kP5Slab0 + c.wave_dst4);
#pragma unroll
    for (int u = 0; u < kG5Ahead; ++u) {
        const unsigned o = c.woff + (unsigned)(u * 8192);
        unsigned char *d = lds + u * kG5WBytes + c.wave_dst;
        cv_glds16(W + o, d);
        cv_glds16(W + (o + 4096u), d + 8192);
    }
    const int lane1 = r * 64 + ((q4 ^ ((0 - (r >> 2)) & 3)) << 4);
    c.a_off = wm * 4096 + lane1;
    c.vb[0] = kP5Slab0 + wn * 9 * 1024 + lane1;
    c.vb[1] = kP5Slab1 + wn * 9 * 1024 + lane1;

    if (w == 0) *(float4 *)(lds + kP5Bias + lane * 16) = *(const float4 *)(bias + lane * 4); // later tiles start from this copy
    cv_wait_vm<4>();
    p5_zero_ranks(c, 0, false);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    for (;;) {
        // the tile after this one (scalars only; without one: this tile again -- valid addresses, never read)
        const int t_nxt = t_cur + t_step;
        const bool has_next = t_nxt < t_end;
        {
            const P5Tile tn = p5_tile_of(has_next ? t_nxt : t_cur, tiles, tflags);
            c.p0n = __builtin_amdgcn_readfirstlane(tn.p0);
            c.kn = __builtin_amdgcn_readfirstlane(tn.k);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier(); // slab 0 and ring slots 0..2 of this tile are complete (every wave waited for its own DMA)
        __builtin_amdgcn_sched_barrier(0);

        int ring_rd = 0, ring_wr = kG5Ahead;
        cv_half8 a0[4], a1[4], b[9];
        cv_f32x4 acc[4][9]; // the accumulators start at the bias (the layer's 256 biases sit in the last KB of the LDS)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 bv = *(const float4 *)(lds + kP5Bias + wm * 256 + i * 64 + (c.lane16 >> 8) * 16);
#pragma unroll
            for (int n = 0; n < 9; ++n) {
                acc[i][n][0] = bv.x; acc[i][n][1] = bv.y; acc[i][n][2] = bv.z; acc[i][n][3] = bv.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = *(const cv_half8 *)(lds + c.a_off + i * 1024);
#pragma unroll
        for (int n = 0; n < 9; ++n) b[n] = *(const cv_half8 *)(lds + c.vb[0] + n * 1024); // the rank above this wave's: dy = -1
#define P5_S(j) g5_step<j, true>(c, acc, chunk + (j) / 9, ring_rd, ring_wr, a0, a1, b)
        for (int chunk = 0; chunk <= c.cmask; chunk += 2) {
            P5_S(0); P5_S(1); P5_S(2); P5_S(3); P5_S(4); P5_S(5); P5_S(6); P5_S(7); P5_S(8);
            P5_S(9); P5_S(10); P5_S(11); P5_S(12); P5_S(13); P5_S(14); P5_S(15); P5_S(16); P5_S(17);
        }
#undef P5_S
        cv_wait_vm<0>(); // the next tile's slab and the wrapped-around weight loads (ring slots 2..4) have landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier(); // every wave is done reading the ring and slab 1
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) { // the next tile's first three weight half-tiles (the same weights) into ring slots 0..2
            unsigned wo = c.woff;
            asm volatile("" : "+v"(wo)); // formed here: as invariants of the tile loop the six addresses are spilled
#pragma unroll
            for (int u = 0; u < kG5Ahead; ++u) {
                const unsigned o = wo + (unsigned)(u * 8192);
                unsigned char *d = lds + u * kG5WBytes + c.wave_dst;
                cv_glds16(W + o, d);
                cv_glds16(W + (o + 4096u), d + 8192);
            }
        }
        int le = c.lane16; // every per-lane value of the epilogue is formed here, from the one lane constant the loop keeps anyway
        asm volatile("" : "+v"(le));
        const int r = (le >> 4) & 15, q4 = le >> 8, prow = le >> 9, piece = (le >> 4) & 31;
        // ---- epilogue, one rank (= one wave row) per pass: the four waves of row h write their 64 channels x 144 rows into the image,
        // then wave w moves image rows 18 w .. 18 w + 17 as whole 512-byte rows (residual in, output out)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (wn == h) {
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
                        *(cv_half4 *)(lds + kP5Img + (n * 16 + r) * kG5ERow + col * 2) = o;
                    }
                }
            }
            const long pbase = (long)c.p0 + h * 144 + w * 18 + prow;
            cv_half8 rv[9];
            if (RES) {
#pragma unroll
                for (int it = 0; it < 9; ++it) rv[it] = *(const cv_half8 *)(R + (pbase + it * 2) * kCvC + piece * 8);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            {
                const cv_half8 zero = (cv_half8)(_Float16)0;
                const unsigned char *eb = lds + kP5Img + (w * 18 + prow) * kG5ERow + piece * 16;
#pragma unroll
                for (int it = 0; it < 9; ++it) {
                    cv_half8 v = *(const cv_half8 *)(eb + it * 2 * kG5ERow);
                    if (RES) v = v + rv[it];
                    if (relu) v = __builtin_elementwise_max(v, zero);
                    *(cv_half8 *)(Y + (pbase + it * 2) * kCvC + piece * 8) = v;
                }
            }
            if (h == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier(); // the image is free for the second rank
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!has_next) break;
        // ---- next tile: its slab 0 was staged (and zeroed) during the last chunk, its weights were requested above
        t_cur = t_nxt;
        c.p0 = c.p0n;
        c.k = c.kn;
        // the weight DMA is older than this pass's 9 stores (vector memory returns in order): it has landed when at most those are
        // outstanding; this wave's image reads are complete (their data went into the stores)
        cv_wait_vm<9>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

} // namespace ccz
