// cczero_conv_g16.h -- the tower convolution for LARGE batches: group-of-16 activation layout, whole-rank tiles.
//
//   y[p, co] = relu( bias[co] + sum_{tap, ci} w[co, tap, ci] * x[p + 9*dy + dx, ci]  [+ res[p, co]] )
//
// (reference net.py:20-43: conv3x3 -> BN -> [+x] -> ReLU with BN folded into w / bias.) k_conv3x3_c256 (cczero_conv.h) works on
// NHWC rows in board order; 110 of the 810 (pixel, tap) pairs of a 10 x 9 board point off the board and it multiplies a zero row
// for them, because 16 consecutive pixels of ONE board never have the same tap off the board. This kernel changes what the
// 16 columns of an MFMA block are:
//
//   * layout "G16" (CCZ_CONV_G16): activation row = (g * 90 + pos) * 16 + j for board 16 g + j -- 16 consecutive rows = one
//     board position ("cell") of 16 boards = one 16-column MFMA block: a tap is on the board for all 16 columns or for none.
//   * tile = 2 whole ranks of one 16-board group = 18 cells = 288 rows x all 256 output channels; 5 tiles per group,
//     4096 boards = 1280 tiles = 5.0 rounds on 256 CUs (k_conv3x3_c256: 1440 tiles of 256 rows = 5.6 rounds).
//   * 8 waves = 4 (64 output channels) x 2 (one rank = 9 cells each): 4 x 9 accumulator tiles per wave (144 VGPRs). The file
//     of a cell is a compile-time constant of the accumulator tile: a tap with dx = -1 skips cell 0, dx = +1 skips cell 8 --
//     32 instead of 36 MFMAs in six of nine taps, the same for every wave, so nothing of it is lost at the barrier (-7.4 %
//     MFMAs). The slab rows of the ranks above rank 0 and below rank 9 are zeroed once per chunk by the threads that staged
//     them, so dy needs no test. No per-lane validity flags, no v_cndmask, no per-tap address arithmetic: every LDS address
//     in the loop is a lane constant + an immediate.
//   * weights come PACKED (k_pack_conv_weights_g16 / ccz_pack_conv_weights_g16_f16, once per weight set): [ci / 32][tap][co][32]
//     with the LDS swizzle already applied, so that a half-tile is one contiguous 16 KB block.
//   * K order = chunks of 32 input channels x 9 taps (one MFMA k-step each; all three convolution kernels add in this order,
//     so their results are the same values -- up to the sign of a zero, where this one skips a product of zeros): 72
//     half-steps per tower layer, one weight half-tile (256 output channels x 32 k, 16 KB) per half-step through a ring of
//     five (three ahead, one barrier per half-step, LDS-DMA with counted vmcnt, as in k_conv3x3_c256). The slab of a chunk
//     = the tile's 2 ranks + one rank either side = 36 cells = 576 rows of 64 B, double-buffered; ring + slabs = 152 KB.
//   * pixel fragments: the three taps of one dy read the SAME nine cells of slab rank (rank + dy) -- cell N needs cell N + dx, and
//     N + dx = -1 / 9 are exactly the skipped pairs -- so the nine fragments of a rank stay in registers for three half-steps and
//     are refilled once per dy, IN PLACE, during the dx = +1 tap (b[N] is dead as soon as cell N - 1 has issued its MFMAs there):
//     27 fragment reads per chunk and wave instead of 78, one register set. The order is pinned (one scheduling region per
//     cell): left to the scheduler the refills move up and the kernel spills.
//   * XCD-aware tile order: workgroup b runs on XCD b % 8 and the five tiles of a group read each other's ranks as halo, so XCD x
//     takes the x-th contiguous eighth of the tiles (HBM traffic per launch 398 MB -> 283-288 MB = algorithmic).
//
// Measured (profiles/r03_conv_g16.json; 4096 boards, one layer in isolation, interleaved on one device): 310-316 us against
// 355-363 us for k_conv3x3_c256; in the workload 301 us per layer (58 % of the dense fp16 peak), 22.14 against 23.69 ms per step
// on one box. The loop is power-limited like its predecessor's (DESIGN.md sections 2 and 10): 3 / 5 / 7 cells in front of the
// barrier and the weight DMA behind it all measure the same; what paid, step by step, was removing work -- the MFMAs of the
// off-board taps (-3...-5 %), the halo fetches (-2.4 %), 46 % of the LDS fragment reads (-3 %), the scattered weight reads (-2.2 %),
// half of the zero stores (-0.9 %). Not everything that removes LDS traffic pays: an epilogue straight from registers (v_permlane16_swap
// pairs two tiles so that a lane stores 16 contiguous bytes; no LDS image, no barrier) is bit-identical and 5 % slower -- its stores
// are sixteen 64-byte segments per instruction instead of whole 512-byte rows.
#pragma once
#include "cczero_conv.h"

namespace ccz {

constexpr int kG5Rows = 288;                               // rows per tile (18 cells x 16 boards)
constexpr int kG5SlabRows = 576;                           // 36 cells
constexpr int kG5SlabBytes = kG5SlabRows * 64;             // 36,864 B per 32-channel chunk
constexpr int kG5WBytes = 256 * 64;                        // one half-step of weights
constexpr int kG5Ring = 5, kG5Ahead = 3;
constexpr int kG5AOff = kG5Ring * kG5WBytes;               // LDS: [weight ring | slab 0 | slab 1]
constexpr int kG5Dump = kG5AOff + 2 * kG5SlabBytes;        // 8 x 1 KB: where a wave's zero stores go when it has no row to zero
constexpr int kG5Lds = kG5Dump + 8 * 1024;                 // 163,840 B = all of it
constexpr int kG5ERow = 528;                               // epilogue image: bytes per row (512 + pad)
static_assert(kG5Rows * kG5ERow <= kG5Dump, "epilogue image must fit the operand buffers");

struct G5Ctx {
    unsigned char *lds;
    const _Float16 *X, *W;   // uniform bases: every DMA is base (scalar) + 32-bit element offset (one VGPR)
    unsigned xoff[5];        // per staging pass: this thread's 16-byte source in X (chunk 0), row clamped into the tensor
    unsigned woff;           // this thread's 16-byte weight source in W (row pass 0, tap 0, chunk 0)
    int zo[2], zd[2];        // SCALAR: where this wave's two zero stores go for slab 0, and the step to slab 1 (0 for the dump area)
    int wave_dst;            // w * 1024
    int lane16;              // (lane & 63) * 16
    int wave_dst4;           // LDS offset of this wave's piece in staging pass 4 (waves 4-7 repeat their pass-3 piece)
    int a_off;               // weight fragment offset inside a ring slot (tile 0; tile i: + 1024 i)
    int vb[2];               // this lane's pixel-fragment base in slab 0 / 1 (cell 0 of the wave's rank at tap offset 0 = + 9 * 1024)
    int cin, cmask;          // input channels; number of 32-channel chunks - 1
    // persistent form only (cczero_conv_g16p.h; all SCALAR): xoff[] is then tile-independent and everything that depends on the tile is
    // derived at its use from these four numbers (a handful of SALU per DMA: kept as ready-made offsets they cost 16 SGPRs and spill)
    int p0, k;               // the CURRENT tile: first tensor row; ranks 2k, 2k + 1 of its group (2 = no edge rank next to it)
    int p0n, kn;             // the NEXT tile of this workgroup (the current one again when there is none)
    int m_w0, m_w3, rows4;   // wave masks: -1 for wave 0 / waves 3..7, else 0; slab row of staging pass 4 (512; waves 4-7: 384 = pass 3 again)
    int za[2], zb[2];        // this wave's zero stores j = 0, 1 for a tile with k = 0 / k = 4: offset inside a slab, +1 so that 0 = none
};

// LDS layout of the persistent form: [ring slots 0-2 | slab 0 | ring slots 3-4 | slab 1 | dump]. Between two tiles of a workgroup the next
// tile's first three weight half-tiles sit in slots 0-2 and its first slab in slab 0; what is left -- slots 3-4, slab 1 and the dump area
// = one contiguous 76 KB -- holds the epilogue image of ONE rank (144 rows of 528 B), so the epilogue runs in two passes.
constexpr int kP5Slab0 = 3 * kG5WBytes;                     // 49,152
constexpr int kP5Ring3 = kP5Slab0 + kG5SlabBytes;           // 86,016: ring slots 3 and 4; also the epilogue image
constexpr int kP5Slab1 = kP5Ring3 + 2 * kG5WBytes;          // 118,784
constexpr int kP5Img = kP5Ring3;
constexpr int kP5Bias = kG5Lds - 1024;                      // the layer's 256 biases (float): behind the image, in the last KB of the dump area
static_assert(kP5Img + 144 * kG5ERow <= kP5Bias, "persistent layout: the bias copy sits behind the epilogue image");
static_assert(kP5Slab1 + kG5SlabBytes == kG5Dump, "persistent layout: the dump area stays where it is");
static_assert(kP5Img + 144 * kG5ERow <= kG5Lds, "persistent layout: one rank's epilogue image must fit behind slab 0");
__device__ __forceinline__ int p5_ring(int slot) { return slot * kG5WBytes + (slot >= 3 ? kG5SlabBytes : 0); }

__host__ __device__ constexpr bool g5_slab_tap(int t) { return t >= 1 && t <= 5; }
// DMA loads younger than the weight half-tile the NEXT half-step reads (issue order per half-step: slab piece, 2 weight loads)
constexpr int kG5Split = 7; // cells in front of the barrier (3 / 5 / 7 and the weight DMA behind the barrier: 344-351 us, noise)
__host__ __device__ constexpr int g5_vmcnt(int t) { return 4 + (g5_slab_tap(t) ? 1 : 0) + (g5_slab_tap(t - 1) ? 1 : 0); }

template <int T, int N> __host__ __device__ constexpr bool g5_on_board() // is tap T of the cell with file N on the board (dx only)
{
    return !((T % 3 == 0 && N == 0) || (T % 3 == 2 && N == 8));
}

// The rank above rank 0 (tiles with k = 0) and the rank below rank 9 (k = 4) do not exist: their slab rows were staged from
// clamped addresses (the DMA count stays static) and are overwritten with zeros by the thread that staged them, after its DMA has
// landed and before the barrier that publishes the slab. Rows 0..143 / 432..575 = whole 16-row pieces, so the tests are
// wave-uniform -- and there are no tests in the loop: the two stores always execute, a wave that has nothing to zero aims
// them at its 1 KB dump area behind the slabs (a branch here splits the loop body and costs the register allocation 90 spills).
__device__ __forceinline__ void g5_zero_ranks(const G5Ctx &c, int buf)
{
    int l16 = c.lane16, zero = 0;
    asm volatile("" : "+v"(l16), "+v"(zero)); // formed here: as loop invariants the four addresses and the zero vector hold 8 registers
    typedef int g5_int4 __attribute__((ext_vector_type(4)));
    const g5_int4 z = {zero, zero, zero, zero};
#pragma unroll
    for (int j = 0; j < 2; ++j) *(g5_int4 *)(c.lds + (l16 + (c.zo[j] + buf * c.zd[j]))) = z;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// persistent form. Staging pass `it` of wave w covers slab rows it * 128 + 16 w .. + 15 (waves 4-7 repeat pass 3 in pass 4): rows 0..143
// are the rank above the tile (pass 0 of every wave, pass 1 of wave 0), rows 432..575 the rank below it (pass 3 of waves 3-7, pass 4).
// A rank that does not exist (k = 0: above, k = 4: below; zeroed after it has landed) is staged from the tile's own edge rank.
// The per-thread part of a slab source is the SAME in every pass (xoff[0]: row tid / 4 of the pass's 128, swizzled 16-byte chunk -- the
// swizzle depends on the row modulo 16 only), so the pass is part of the scalar too: one address register instead of five.
// Everything here is SCALAR and BRANCH-FREE (a scalar branch inside the loop body costs the register allocation its balance): the
// conditions on the wave are precomputed masks (G5Ctx::m_*), the conditions on the tile are one compare + select each.
template <int IT> __device__ __forceinline__ unsigned p5_slab_src(const G5Ctx &c, int p0, int k)
{
    const int is0 = k == 0 ? -1 : 0, is4 = k == 4 ? -1 : 0;
    int fix = 0, rows = IT * 128;
    if constexpr (IT == 0) fix = 144 & is0;
    if constexpr (IT == 1) fix = 144 & is0 & c.m_w0;
    if constexpr (IT == 3) fix = -(144 & is4 & c.m_w3);
    if constexpr (IT == 4) {
        fix = -(144 & is4);
        rows = c.rows4; // waves 4-7 repeat their pass-3 piece
    }
    return (unsigned)((p0 - 144 + fix + rows) * c.cin);
}
// LDS offset of this wave's zero store j for slab buffer `buf` of a tile with edge index k (the dump KB when it has nothing to zero)
__device__ __forceinline__ int p5_zero_dst(const G5Ctx &c, int k, int j, int buf)
{
    const int is0 = k == 0 ? -1 : 0, is4 = k == 4 ? -1 : 0;
    const int d = (is0 & c.za[j]) | (is4 & c.zb[j]);       // this wave's piece relative to slab 0, or 0
    const int real = d != 0 ? -1 : 0;
    return kG5Dump + (real & (kP5Slab0 - kG5Dump + d - 1 + (buf ? kP5Slab1 - kP5Slab0 : 0)));
}
// the slab staged during a tile's LAST chunk is the next tile's first one (buffer 0; the chunk count is even)
__device__ __forceinline__ void p5_zero_ranks(const G5Ctx &c, int buf, bool next_tile)
{
    int l16 = c.lane16, zero = 0;
    asm volatile("" : "+v"(l16), "+v"(zero));
    typedef int g5_int4 __attribute__((ext_vector_type(4)));
    const g5_int4 z = {zero, zero, zero, zero};
    const int k = next_tile ? c.kn : c.k;
#pragma unroll
    for (int j = 0; j < 2; ++j) *(g5_int4 *)(c.lds + (l16 + p5_zero_dst(c, k, j, buf))) = z;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// One half-step = tap T of a 32-channel chunk; J = its index inside the unrolled pair of chunks (parity of the A register
// set = J & 1, slab buffer = J / 9).
template <int J, bool PERS = false>
__device__ __forceinline__ void g5_step(const G5Ctx &c, cv_f32x4 (&acc)[4][9], int chunk, int &ring_rd, int &ring_wr,
                                         cv_half8 (&a0)[4], cv_half8 (&a1)[4], cv_half8 (&b)[9])
{
    constexpr int T = J % 9, BUF = J / 9;
    constexpr int Tn = (T + 1) % 9, BUFn = (T == 8) ? 1 - BUF : BUF;
    [[maybe_unused]] constexpr int deltan = 9 * (Tn / 3 - 1) + (Tn % 3 - 1);
    cv_half8 (&acur)[4] = (J & 1) ? a1 : a0;
    cv_half8 (&anxt)[4] = (J & 1) ? a0 : a1;
    unsigned char *const lds = c.lds;

    // the order below is pinned (one scheduling region per cell): a fragment register is refilled right AFTER the MFMAs that read
    // it -- left to the scheduler the refills move up and every fragment needs a second register
#ifdef G5_BFRAG_PER_TAP /* A/B: round 3's first form -- nine fragment reads per tap, each register refilled after its MFMAs */
#define G5_CELL(N)                                                                                                        \
    {                                                                                                                     \
        if constexpr (g5_on_board<T, N>()) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
                acc[i][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[i], b[N], acc[i][N], 0, 0, 0);                    \
        }                                                                                                                 \
        if constexpr (g5_on_board<Tn, N>())                                                                               \
            b[N] = *(const cv_half8 *)(lds + c.vb[BUFn] + (9 + N + deltan) * 1024);                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
#else
    // The three taps of one dy read the SAME nine cells of slab rank (rank + dy): cell N needs cell N + dx. So b[] holds the nine
    // fragments of that rank for three half-steps (cell N multiplies b[N + dx]; N + dx = -1 and 9 are exactly the skipped,
    // off-board pairs) and is refilled once per dy, during the dx = +1 tap: b[N] is dead as soon as cell N - 1 has issued its
    // MFMAs there, so it is reloaded for the next dy right in front of cell N's MFMAs -- 27 fragment reads per chunk instead of 78.
#define G5_CELL(N)                                                                                                        \
    {                                                                                                                     \
        if constexpr (T % 3 == 2)                                                                                         \
            b[N] = *(const cv_half8 *)(lds + c.vb[BUFn] + (9 + N + 9 * (Tn / 3 - 1)) * 1024);                             \
        if constexpr (g5_on_board<T, N>()) {                                                                              \
            constexpr int NB = N + T % 3 - 1;                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
                acc[i][N] = __builtin_amdgcn_mfma_f32_16x16x32_f16(acur[i], b[NB], acc[i][N], 0, 0, 0);                   \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
#endif
#define G5_CELLS(LO, HI)                                                                                                  \
    if constexpr (LO <= 0 && 0 < HI) G5_CELL(0) if constexpr (LO <= 1 && 1 < HI) G5_CELL(1) if constexpr (LO <= 2 && 2 < HI) G5_CELL(2) \
    if constexpr (LO <= 3 && 3 < HI) G5_CELL(3) if constexpr (LO <= 4 && 4 < HI) G5_CELL(4) if constexpr (LO <= 5 && 5 < HI) G5_CELL(5) \
    if constexpr (LO <= 6 && 6 < HI) G5_CELL(6) if constexpr (LO <= 7 && 7 < HI) G5_CELL(7) if constexpr (LO <= 8 && 8 < HI) G5_CELL(8)
    constexpr int T2 = (T + kG5Ahead) % 9;
    const int chunk2 = (chunk + (T + kG5Ahead >= 9 ? 1 : 0)) & c.cmask;
    G5_CELLS(0, 1)
    if constexpr (g5_slab_tap(T)) { // the next chunk's slab: 5 pieces per thread, in taps 1..5
        constexpr int pass = T - 1;
        const int nxt = (chunk + 1) & c.cmask; // past the last chunk: re-stage chunk 0 into the free buffer (keeps every count static)
        if constexpr (PERS) { // ... of the workgroup's NEXT tile (of this one again when there is none)
            const bool last = chunk == c.cmask;
            const unsigned so = p5_slab_src<pass>(c, last ? c.p0n : c.p0, last ? c.kn : c.k) + (unsigned)(nxt * 32);
            cv_glds16(c.X + (c.xoff[0] + so), lds + (BUF ? kP5Slab0 : kP5Slab1) + (pass < 4 ? pass * 8192 + c.wave_dst : c.wave_dst4));
        } else
        cv_glds16(c.X + (c.xoff[pass] + (unsigned)(nxt * 32)), lds + kG5AOff + (1 - BUF) * kG5SlabBytes + (pass < 4 ? pass * 8192 + c.wave_dst : c.wave_dst4));
        __builtin_amdgcn_sched_barrier(0);
    }
    G5_CELLS(1, 2)
    {
        unsigned wo = c.woff;
        asm volatile("" : "+v"(wo)); // the address is formed here, per half-step: hoisted for 9 taps x 2 pieces it costs 36 registers
        const unsigned o = wo + (unsigned)((T2 + 9 * chunk2) * 8192); // half-tile (chunk2, T2): one contiguous 16 KB block
        const unsigned o2 = o + 4096u;                                 // its rows 128..255
        unsigned char *const d = lds + (PERS ? p5_ring(ring_wr) : ring_wr * kG5WBytes) + c.wave_dst;
        cv_glds16(c.W + o, d);
        __builtin_amdgcn_sched_barrier(0);
        G5_CELLS(2, 3)
        cv_glds16(c.W + o2, d + 8192);
        __builtin_amdgcn_sched_barrier(0);
    }
    G5_CELLS(3, kG5Split)

    ring_rd = ring_rd + 1 == kG5Ring ? 0 : ring_rd + 1;
    cv_wait_vm<g5_vmcnt(T)>();
    if constexpr (T == 7 && PERS) p5_zero_ranks(c, 1 - BUF, chunk == c.cmask);
    else if constexpr (T == 7) g5_zero_ranks(c, 1 - BUF); // this thread's slab pieces of the next chunk have landed (all but the youngest weight loads)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    {
        const unsigned char *wa = lds + ((PERS ? p5_ring(ring_rd) : ring_rd * kG5WBytes) + c.a_off);
#pragma unroll
        for (int i = 0; i < 4; ++i) anxt[i] = *(const cv_half8 *)(wa + i * 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
    G5_CELLS(kG5Split, 9)
    ring_wr = ring_wr + 1 == kG5Ring ? 0 : ring_wr + 1;
#undef G5_CELLS
#undef G5_CELL
}

// ---- the heads fused into the LAST tower layer (round 4) ------------------------------------------------------------------------------
// The tower's last layer writes 170 MB of activations that only the two 1x1 head convolutions read (k_head_conv1x1: 46 us, HBM-bound).
// In the HEADS instantiation the layer's epilogue keeps its finished rows (conv + bias + residual, ReLU) in the LDS image instead of
// storing them, and the workgroup multiplies them with the 24 head channels right there: per 16-row cell 16 MFMAs against the 72 x 36
// of the tile's own loop. The output tensor of the layer is never written; the head outputs leave in board order exactly as
// k_head_conv1x1 writes them (cczero_heads.h) -- same operands, same chain of MFMAs over k = 0, 32, ...: the same bits.
struct G5Heads {
    const _Float16 *w32; // [32][256]: rows 0..16 policy, 17..23 value, 24..31 zero
    const float *b32;    // [32]
    _Float16 *pol, *val; // [boards][1536], [boards][640] (cczero_heads.h kHdPolStride / kHdValStride)
    int nb;              // boards that hold rows (compact live count or the batch size): stores past it are skipped
};

// `board0[wnr]` / `pos0[wnr]`: first board (group * 16) and first position (rank * 9) of image rows 144 wnr .. 144 wnr + 143 (wnr = 0, 1);
// `skip1`: image rows 144..287 are a duplicate (edge kernel, odd group count). Called by every wave after the barrier that publishes the
// image of FINISHED rows.
__device__ __forceinline__ void g5_heads_phase(const unsigned char *lds, int w, int lane, const G5Heads &ha, const long (&board0)[2],
                                               const int (&pos0)[2], bool skip1)
{
    const int r = lane & 15, q4 = lane >> 4;
    cv_half8 a[2][8];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int s = 0; s < 8; ++s) a[m][s] = *(const cv_half8 *)(ha.w32 + (m * 16 + r) * 256 + s * 32 + q4 * 8);
    const float4 b0 = *(const float4 *)(ha.b32 + 4 * q4), b1 = *(const float4 *)(ha.b32 + 16 + 4 * q4);
    for (int cell = w; cell < 18; cell += 8) { // 18 cells of 16 rows over 8 waves
        const unsigned char *row = lds + (cell * 16 + r) * kG5ERow + q4 * 16;
        cv_f32x4 acc0, acc1;
        acc0[0] = b0.x; acc0[1] = b0.y; acc0[2] = b0.z; acc0[3] = b0.w;
        acc1[0] = b1.x; acc1[1] = b1.y; acc1[2] = b1.z; acc1[3] = b1.w;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const cv_half8 b = *(const cv_half8 *)(row + s * 64);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][s], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][s], b, acc1, 0, 0, 0);
        }
        const int wnr = cell >= 9 ? 1 : 0, n = cell - 9 * wnr;
        const long board = board0[wnr] + r;
        if (board < ha.nb && !(skip1 && wnr)) {
            const int pos = pos0[wnr] + n;
            _Float16 *po = ha.pol + board * 1536 + pos * 17;
            _Float16 *vo = ha.val + board * 640 + pos * 7;
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
    }
}

// ONE: only the first 32 input channels of a row can be non-zero (the stem: 21 live planes in rows of 64 channels) -- the tile runs the
// nine half-steps of chunk 0 and stops; the second chunk would add products of zeros to accumulators that are never -0 (they start at
// a bias): the same bits (the board-major and small-batch kernels run both chunks; tests compare them bit for bit).
template <bool RES, bool HEADS, bool ONE = false>
__device__ __forceinline__ void g5_tile(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                        const float *__restrict__ bias, const _Float16 *R,
                                        _Float16 *Y, int M, int relu, int cin, const int *live_rows, int row0, const G5Heads &ha)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[kG5Lds];
    [[maybe_unused]] long first_board = 0; // (HEADS) global index of this launch's first board: the live parts offset their pointers
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q4 = lane >> 4;
    const int wm = w & 3, wn = w >> 2; // the two waves of a SIMD share their weight fragments' rows
    int tiles = gridDim.x;
    if (live_rows) {
        // Planned evaluator boundary (ccz_eval_plan): only the first *live_rows boards hold rows to compute, a number that stays
        // on the device. Their 16-board groups are cut into n_parts EQUAL ranges; this launch is range `part` of them (the argument
        // carries part | n_parts << 16) and finds its groups itself. The grid is sized for the largest possible range;
        // workgroups beyond the live tiles leave at once.
        const int part = row0 & 0xffff, n_parts = row0 >> 16;
        const int G = (*live_rows + 15) >> 4;
        const int per = (G + n_parts - 1) / n_parts;
        const int first = part * per;
        int live = G - first;
        live = live < 0 ? 0 : (live > per ? per : live);
        live = live > M / 1440 ? M / 1440 : live;
        M = live * 1440;
        tiles = live * ((relu & 4) ? 4 : 5);
        if ((int)blockIdx.x >= tiles) return;
        const long off = (long)first * 1440 * kCvC;
        X += (long)first * 1440 * cin;
        if (!HEADS) Y += off; // (HEADS: there is no output tensor)
        if (RES) R += off;
        first_board = (long)first * 16;
    }
    // XCD-aware order: workgroup b runs on XCD b % 8 (round-robin dispatch), and the five tiles of a group read each other's
    // ranks as halo -- so XCD x takes the x-th CONTIGUOUS eighth of the tiles, in order: a halo rank is then in that XCD's L2
    // (with tile = b: 398 MB fetched per half-batch launch against 283 MB algorithmic, profiles/pmc_summary.json)
    int tile;
    {
        const int b = blockIdx.x, x = b & 7, per = tiles >> 3, rem = tiles & 7;
        tile = x * per + (x < rem ? x : rem) + (b >> 3);
#ifdef G5_NO_XCD_MAP
        tile = b;
#endif
    }
    // flags bit 1: tiles in descending order (the tiles written last by the previous layer are then read first)
    tile = __builtin_amdgcn_readfirstlane((relu & 2) ? tiles - 1 - tile : tile);
    // flags bit 2 ("middle" mode, round 4): the launch covers ranks 1..8 only, four tiles per group (ranks 1-2, 3-4, 5-6, 7-8: every
    // neighbour rank exists, nothing is zeroed); ranks 0 and 9 are k_conv3x3_g16_edge's (cczero_conv_g16e.h)
    const int mid = relu & 4;
    const int k = mid ? 2 : tile % 5;        // ranks 2k, 2k + 1 of the tile's group (middle mode: any k without an edge)
    const long p0 = mid ? (long)(tile >> 2) * 1440 + 144 + (long)(tile & 3) * kG5Rows
                        : (long)tile * kG5Rows;    // = (group * 90 + 18 k) * 16
    relu &= 1;

    G5Ctx c;
    c.lds = lds;
    c.X = X;
    c.W = W;
    {
        // rows 0..143 (k = 0) = pass 0 of every wave + pass 1 of wave 0; rows 432..575 (k = 4) = pass 3 of waves 3..7 + pass 4 of
        // waves 0..3: two stores per wave cover either (wave 0 resp. wave 3 need both of theirs); anything else goes to the dump area
        const int piece[2] = {k == 0 ? 0 : k == 4 ? (w >= 3 ? 3 : 4) : -1, (k == 0 && w == 0) ? 1 : (k == 4 && w == 3) ? 4 : -1};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            c.zo[j] = __builtin_amdgcn_readfirstlane(piece[j] >= 0 ? kG5AOff + piece[j] * 8192 + w * 1024 : kG5Dump + w * 1024);
            c.zd[j] = __builtin_amdgcn_readfirstlane(piece[j] >= 0 ? kG5SlabBytes : 0);
        }
    }
    c.wave_dst = w * 1024;
    c.lane16 = lane * 16;
    c.wave_dst4 = (w < 4 ? 4 : 3) * 8192 + w * 1024;
    c.cin = cin;
    c.cmask = ONE ? 0 : (cin >> 5) - 1;
    {
        // slab row sr (0..575) = tensor row p0 - 144 + sr: rank 2k - 1 + sr / 144 of the group; 64-byte rows, position pos of
        // row sr holds source chunk pos ^ f(sr), f = (-(sr >> 2)) & 3 (conflict-free for the 16 rows x 4 chunks one ds_read_b128
        // of this MFMA shape covers)
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int piece = (it < 4 || w < 4) ? it * 512 + tid : 3 * 512 + tid; // waves 4-7 repeat pass 3 (same bytes, same place)
            const int sr = piece >> 2, pos = piece & 3;
            const int schunk = pos ^ ((0 - (sr >> 2)) & 3);
            long p = p0 - 144 + sr;
            p = p < 0 ? 0 : (p > (long)M - 1 ? (long)M - 1 : p);
            c.xoff[it] = (unsigned)(p * cin + schunk * 8);
        }
        c.woff = (unsigned)(tid * 8); // packed weights (k_pack_conv_weights_g16): a half-tile is the LDS image itself, read linearly
    }
    // ---- prologue: slab of chunk 0, weight half-tiles 0..2; the per-lane setup below runs while the DMA is in flight
#pragma unroll
    for (int it = 0; it < 5; ++it) cv_glds16(X + c.xoff[it], lds + kG5AOff + (it < 4 ? it * 8192 + c.wave_dst : c.wave_dst4));
#pragma unroll
    for (int u = 0; u < kG5Ahead; ++u) {
        const unsigned o = c.woff + (unsigned)(u * 8192);
        unsigned char *d = lds + u * kG5WBytes + c.wave_dst;
        cv_glds16(W + o, d);
        cv_glds16(W + (o + 4096u), d + 8192);
    }
    const int lane1 = r * 64 + ((q4 ^ ((0 - (r >> 2)) & 3)) << 4);
    c.a_off = wm * 4096 + lane1;                               // rows 64 wm + 16 i + r of the half-tile
    c.vb[0] = kG5AOff + wn * 9 * 1024 + lane1;                 // slab cell 9 wn + n + 9 + delta, row r of it
    c.vb[1] = c.vb[0] + kG5SlabBytes;

    // the accumulators start at the bias
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
    g5_zero_ranks(c, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    int ring_rd = 0, ring_wr = kG5Ahead;
    cv_half8 a0[4], a1[4], b[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = *(const cv_half8 *)(lds + c.a_off + i * 1024);
#ifdef G5_BFRAG_PER_TAP
#pragma unroll
    for (int n = 1; n < 9; ++n) b[n] = *(const cv_half8 *)(lds + c.vb[0] + (9 + n - 10) * 1024); // tap 0: delta = -10, cell 0 is off the board
    b[0] = b[1];
#else
#pragma unroll
    for (int n = 0; n < 9; ++n) b[n] = *(const cv_half8 *)(lds + c.vb[0] + (9 + n - 9) * 1024); // the rank above this wave's: dy = -1
#endif
#define G5_S(j) g5_step<j>(c, acc, chunk + (j) / 9, ring_rd, ring_wr, a0, a1, b)
    if constexpr (ONE) {
        const int chunk = 0; // (the prefetches past half-step 8 wrap around to chunk 0: valid memory, never read)
        G5_S(0); G5_S(1); G5_S(2); G5_S(3); G5_S(4); G5_S(5); G5_S(6); G5_S(7); G5_S(8);
    } else {
        for (int chunk = 0; chunk <= c.cmask; chunk += 2) {
            G5_S(0); G5_S(1); G5_S(2); G5_S(3); G5_S(4); G5_S(5); G5_S(6); G5_S(7); G5_S(8);
            G5_S(9); G5_S(10); G5_S(11); G5_S(12); G5_S(13); G5_S(14); G5_S(15); G5_S(16); G5_S(17);
        }
    }
#undef G5_S
    cv_wait_vm<0>(); // the wrapped-around DMA loads must land before the LDS is reused / released

    // ---- epilogue: every wave writes its 64 channels x 144 rows into the [row][channel] image in LDS; then wave w owns
    // rows 36 w .. 36 w + 35 and moves whole 512-byte rows (residual in, output out)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier(); // every wave is done reading the slabs and the ring
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
    const long pbase = p0 + w * 36 + prow;
    cv_half8 rv[18];
    if (RES) {
#pragma unroll
        for (int it = 0; it < 18; ++it) rv[it] = *(const cv_half8 *)(R + (pbase + it * 2) * kCvC + piece * 8);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
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
    if constexpr (HEADS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier(); // every finished row is in the image
        __builtin_amdgcn_sched_barrier(0);
        // image rows 0..143 = the tile's first rank, 144..287 = its second: tensor row p0 + i = (group * 90 + 9 * rank + cell) * 16 + board
        const long grp = p0 / 1440;
        const int rank0 = (int)((p0 - grp * 1440) / 144);
        const long board0[2] = {first_board + grp * 16, first_board + grp * 16};
        const int pos0[2] = {rank0 * 9, rank0 * 9 + 9};
        g5_heads_phase(lds, w, lane, ha, board0, pos0, false);
    }
}

template <bool RES>
__global__ __launch_bounds__(512) void k_conv3x3_g16(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                         const float *__restrict__ bias, const _Float16 *R,
                                                         _Float16 *Y, int M, int relu, int cin, const int *live_rows, int row0)
{
    g5_tile<RES, false>(X, W, bias, R, Y, M, relu, cin, live_rows, row0, G5Heads{});
}

// The stem (ccz_conv3x3_stem_f16 with CCZ_CONV_G16): rows of 64 channels of which only 0..31 can be non-zero, one chunk (g5_tile ONE)
__global__ __launch_bounds__(512) void k_conv3x3_g16_stem(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                            const float *__restrict__ bias, _Float16 *Y, int M, int relu, int cin,
                                                            const int *live_rows, int row0)
{
    g5_tile<false, false, true>(X, W, bias, nullptr, Y, M, relu, cin, live_rows, row0, G5Heads{});
}

// The last layer of the tower with the heads in its epilogue (always with residual; Y may be null: nothing is stored to it).
__global__ __launch_bounds__(512) void k_conv3x3_g16_heads(const _Float16 *__restrict__ X, const _Float16 *__restrict__ W,
                                                             const float *__restrict__ bias, const _Float16 *R, _Float16 *Y, int M,
                                                             int relu, int cin, const int *live_rows, int row0, G5Heads ha)
{
    if (live_rows) ha.nb = *live_rows; // planned boundary: the pointers are the whole batch's, M only the capacity of this part
    g5_tile<true, true>(X, W, bias, R, Y, M, relu, cin, live_rows, row0, ha);
}

// Weights for k_conv3x3_g16: [co][tap][ci] (the memory of a channels-last [co, ci, 3, 3] tensor) -> [ci / 32][tap][co][32], each
// 64-byte row stored with its four 16-byte chunks in the swizzled order the LDS image uses (position pos of row co holds chunk
// pos ^ ((-(co >> 2)) & 3)): a half-tile (32 input channels of one tap, all 256 output channels) becomes ONE contiguous 16 KB
// block that the DMA copies linearly -- whole 128-byte lines instead of 256 scattered 64-byte pieces 4.6 KB apart (-2.7 % per
// layer). One thread per 16-byte piece.
__global__ __launch_bounds__(256) void k_pack_conv_weights_g16(const cv_half8 *__restrict__ w, cv_half8 *__restrict__ wp, int cin)
{
    const int n = 256 * 9 * (cin >> 3);
    const int i = blockIdx.x * 256 + threadIdx.x;   // destination piece: ((c32 * 9 + tap) * 256 + co) * 4 + pos
    if (i >= n) return;
    const int pos = i & 3, co = (i >> 2) & 255, h = i >> 10, tap = h % 9, c32 = h / 9;
    const int chunk = pos ^ ((0 - (co >> 2)) & 3);
    wp[i] = w[(co * 9 + tap) * (cin >> 3) + c32 * 4 + chunk];
}

} // namespace ccz
