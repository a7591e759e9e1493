// cczero_device.h -- device-side building blocks of the gfx950 lockstep engine.
//
// One 64-lane wavefront owns one board; every workgroup is exactly one wavefront, so
// __syncthreads() is a wave-level LDS fence and is legal in any wave-uniform control flow.
// All floating-point code here is contraction-free (-ffp-contract=off) and uses only IEEE
// + - * / sqrt, so it reproduces the reference's NumPy arithmetic (PUCT, incremental mean) and the
// deterministic sampler bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cczero_tables.h"

namespace ccz {

__constant__ Tables c_tab = make_tables();

enum : int { PAWN = 1, CANNON = 2, ROOK = 3, KNIGHT = 4, BISHOP = 5, ADVISOR = 6, KING = 7 };

constexpr int kMaxLegal = 128;
constexpr int kMaskWords = 66;
constexpr int kChainCap = 128;
constexpr int kPseudoCap = 256;
constexpr uint64_t kTurnKey = 0x9D39247E33776D41ull;

struct __align__(16) NodeA { // 16 B: what PUCT select reads per child in one dwordx4 load
    int32_t N;  // visits           (mcts.py:16)
    float Q;    // value, float32   (mcts.py:15; NumPy float32 arithmetic, mcts.py:63-71)
    float P;    // prior, float32   (mcts.py:17)
    int32_t fc; // pool index of the first child, -1 while unexpanded
};
// nodeB word: move id (low 16 bits) | number of children (high 16 bits)

struct __align__(8) BoardMeta {
    uint64_t key;          // Zobrist key of the root position (incl. side to move)
    int32_t halfmove;      // plies since the last capture
    int32_t chain_len;     // keys since the last capture, incl. the root position
    int32_t ply;           // plies played in the current game == records stored
    uint32_t move_counter; // moves ever played on this slot (RNG counter)
    int32_t n_nodes;       // nodes in use in the active pool half
    uint8_t turn;          // 1 RED, 0 BLACK
    uint8_t over;          // 1: game finished, waiting for harvest
    int8_t winner;         // 1 RED, 0 BLACK, -1 draw
    uint8_t half;          // active pool half
    uint32_t pi_used;      // entries used in the pi record arena
    uint32_t game_no;
};

// One entry of the evaluation cache (ccz_config.eval_cache_log2): what the evaluator returned for a position -- the priors of
// its legal moves in `legal_moves` order (what k_softmax_gather writes to prior128) and the value -- keyed by the Zobrist key.
struct __align__(16) CacheEntry {
    uint64_t key;   // 0 = empty
    float v;
    uint32_t k;     // bits 0..7: number of legal moves; bits 8..31: a 24-bit hash of the legal-move LIST (cache_tag). Both are
                    // checked on a hit: the priors are aligned with that list, and it is information the key does not hold
    float pri[kMaxLegal];
};
static_assert(sizeof(CacheEntry) == 528, "cache entry is 528 bytes");

struct BoardStats {
    unsigned long long sims, moves, games, truncated, sum_depth, sum_children, expansions, terminal, pruned;
    int32_t nodes_peak, depth_peak;
    // evaluation cache, counted per board by the board's own wave (no atomics): leaves probed, hits, leaves served by another
    // board's evaluator row of the same step, entries stored
    unsigned int cache_probes, cache_hits, cache_shared, cache_stores;
    // CCZ_FLAG_CACHE_VERIFY: hits that were sent through the evaluator again, and those whose fresh priors / value differ
    unsigned int cache_verified, cache_mismatch;
};

struct Dev {
    int32_t B, cap, maxd, max_plies, pi_cap;
    int32_t reserve;   // nodes of every pool half kept free at re-root time for the next move's expansions
    float c_puct;
    double eps, alpha, temp;
    uint32_t flags;
    uint64_t seed, board_id_base;
    NodeA *nodeA;      // [B][2][cap]
    uint32_t *nodeB;   // [B][2][cap]
    BoardMeta *meta;   // [B]
    uint8_t *root_sq;  // [B][96]
    uint64_t *chain;   // [B][128]
    uint64_t *chain_chk; // [B][2] bit i: the side to move is in check in chain position i (CCZ_RULE_PERPETUAL_CHECK)
    int32_t *path;     // [B][maxd] selection path: {node, N, Q bits, -} as seen by the select phase, so
                       // that the backup needs no dependent load of the node records
    int32_t *path_len; // [B] depth of the leaf (path holds depth+1 nodes)
    uint16_t *leaf_ids;   // [B][128]
    int32_t *leaf_k;      // [B]
    uint8_t *leaf_status; // [B]
    // ---- evaluation cache (optional: cache == nullptr without it). The evaluator input of the search path depends on the leaf
    // position and side to move only (net.py:160-173: history planes zero), i.e. on leaf_key: a position evaluated before -- by
    // this board, by another one, or by another board in this very step -- need not go through the tower again.
    CacheEntry *cache;    // [cache_mask + 1] direct-mapped by the key
    uint32_t cache_mask;
    int32_t *claim;       // [cache_mask + 1] lowest board index that missed on this slot in the current step (INT_MAX: nobody)
    uint32_t *cslot;      // [B] slot of the pending leaf
    uint32_t *ctag;       // [B] cache_tag of the pending leaf's legal-move list (count | 24-bit hash): compared with the key wherever two leaves are called the same position
    uint8_t *cstate;      // [B] 0 = miss: needs a row of the evaluator, 1 = hit, 2 = no evaluation needed (terminal leaf, none)
    uint8_t *cins;        // [B] this board's fresh evaluation is stored (it is the slot's claim winner)
    uint8_t *cver;        // [B] CCZ_FLAG_CACHE_VERIFY: a hit that is evaluated again and compared with what the table returned
    int32_t *crep;        // [B] the board whose evaluator row this board uses (itself, or the same-key board of lower index)
    int32_t *row_of;      // [B] compact evaluator row holding this board's logits / value (misses)
    float *vleaf;         // [B] leaf value of the pending leaf (step / expand_backup read it when value_dev == NULL)
    uint64_t *leaf_key;   // [B] Zobrist key (pieces + side to move) of the pending leaf: what the evaluator input depends on
    uint8_t *rec_sq;   // [B][max_plies][96] root position before each move
    uint8_t *rec_turn; // [B][max_plies]
    uint8_t *rec_k;    // [B][max_plies]
    uint32_t *rec_off; // [B][max_plies] offset into the pi arena
    uint16_t *rec_ids; // [B][pi_cap]
    float *rec_pi;     // [B][pi_cap]
    BoardStats *stats; // [B]
    int32_t *err;      // [1] sticky error bits
    float *prior128;   // [B][128] priors of the leaf's legal moves (compact evaluator boundary, ccz_*_logits)
    int32_t *half;     // [1] pool half holding every live tree: flipped once per move for ALL boards, so that tree
                       // addresses (root = node 0, its children = nodes 1..k) are known before any load returns
    unsigned long long *stamps; // [B][16] s_memtime stamps; only written by the diagnostic build (-DCCZ_STAMPS)
    // run-time rule tables (ccz_config, ABI 2): the two choices no golden trace can pin against cchess
    const uint16_t *rank;   // [2086] position of move id in `board.legal_moves` order, or nullptr = ascending id
    const uint16_t *unrank; // [2086] inverse of rank
    uint32_t chanpack;      // 3 bits per piece type t (bits 3t..3t+2): plane channel of type t (tools.py:100)
    uint32_t typepack;      // 3 bits per channel c (bits 3c..3c+2): piece type - 1 encoded in channel c
    uint32_t trankpack;     // 3 bits per piece type: major key of `legal_moves` order by the mover's type; 0 = none
    uint32_t rule_flags;    // CCZ_RULE_*: bit 0 = perpetual check loses (game end only), bit 1 = pawn moves restart the
                            // sixty-move clock and the history chain like captures
};
__device__ __forceinline__ int plane_of(const Dev &D, int type) { return (int)((D.chanpack >> (3 * type)) & 7u); }
__device__ __forceinline__ int type_in_plane(const Dev &D, int chan) { return (int)((D.typepack >> (3 * chan)) & 7u) + 1; }

// In-kernel stamps (diagnostic build only: profiles/sim_stamps.py builds libcczero_stamps.so with -DCCZ_STAMPS;
// the shipped kernels contain no stamp). One asm statement: s_memtime + its wait, fenced against reordering.
#ifdef CCZ_STAMPS
#define CCZ_STAMP(D_, b_, lane_, i_)                                                             \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((lane_) == 0) (D_).stamps[(size_t)(b_) * 16 + (i_)] = t_;                            \
    }
#define CCZ_GSTAMP(sp_, lane_, i_)                                                                \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if ((lane_) == 0 && (sp_)) (sp_)[i_] = t_;                                               \
    }
#else
#define CCZ_STAMP(D_, b_, lane_, i_)
#define CCZ_GSTAMP(sp_, lane_, i_)
#endif

#define CCZ_LEAF_SKIP 3

// Bounds-checked diagnostic build (-DCCZ_BOUNDS: `make bounds` -> build/diag/libcczero_bounds.so, never shipped): every
// index into the node pool, the selection path, the game record and the pi arena is range-checked; a stray index sets the
// sticky error bit CCZ_ERR_BOUNDS, remembers its source line in err[1] and is redirected to element 0 (SURVEY section 5:
// ROCm has no compute-sanitizer). The shipped build compiles the checks away.
#ifdef CCZ_BOUNDS
__device__ __forceinline__ long long ccz_checked(int32_t *err, long long i, long long n, int line)
{
    if (i < 0 || i >= n) { atomicOr(err, 128); atomicMax(err + 1, line); return 0; }
    return i;
}
#define CCZ_IDX(D_, i_, n_) ccz_checked((D_).err, (long long)(i_), (long long)(n_), __LINE__)
#else
#define CCZ_IDX(D_, i_, n_) (i_)
#endif

// ------------------------------------------------------------------ small helpers
constexpr uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Zobrist keys as a compile-time table in constant memory: in the descent the indices are wave-uniform,
// so a key is one scalar load instead of ~40 VALU instructions of 64-bit hashing.
struct ZobTable { uint64_t k[16 * 90]; };
constexpr ZobTable make_zob()
{
    ZobTable t{};
    for (int i = 0; i < 16 * 90; ++i) t.k[i] = mix64(0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
    return t;
}
__constant__ ZobTable c_zob = make_zob();
__device__ __forceinline__ uint64_t zob(int pc, int sq) { return c_zob.k[pc * 90 + sq]; }
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// LDS-only synchronisation inside the one-wave workgroup. LDS instructions of a wave execute in program
// order, so a later ds_read sees an earlier ds_write of any lane; all that is needed is that the COMPILER
// keeps that order. __syncthreads() would also drain every outstanding global load and store
// (s_waitcnt vmcnt(0)), which put ~1 us of store latency on the critical path at each of ~15 sync points.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide scan / reduction on the DPP network (row_shr 1,2,4,8 + row_bcast15/31: the gfx9 sequence),
// VALU latency instead of six dependent trips through the LDS crossbar (ds_bpermute).
#define CCZ_DPP(old_, src_, ctrl_, rmask_) __builtin_amdgcn_update_dpp((old_), (src_), (ctrl_), (rmask_), 0xf, false)

__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/)
{
    v += CCZ_DPP(0, v, 0x111, 0xf);
    v += CCZ_DPP(0, v, 0x112, 0xf);
    v += CCZ_DPP(0, v, 0x114, 0xf);
    v += CCZ_DPP(0, v, 0x118, 0xf);
    v += CCZ_DPP(0, v, 0x142, 0xa);
    v += CCZ_DPP(0, v, 0x143, 0xc);
    return v;
}

// inclusive XOR scan of a 64-bit value over the wave; readlane of a 64-bit value
__device__ __forceinline__ uint64_t wave_incl_xor64(uint64_t x)
{
    int lo = (int)(x & 0xffffffffull), hi = (int)(x >> 32);
#define CCZ_XOR_STEP(ctrl_, rmask_) { lo ^= CCZ_DPP(0, lo, ctrl_, rmask_); hi ^= CCZ_DPP(0, hi, ctrl_, rmask_); }
    CCZ_XOR_STEP(0x111, 0xf)
    CCZ_XOR_STEP(0x112, 0xf)
    CCZ_XOR_STEP(0x114, 0xf)
    CCZ_XOR_STEP(0x118, 0xf)
    CCZ_XOR_STEP(0x142, 0xa)
    CCZ_XOR_STEP(0x143, 0xc)
#undef CCZ_XOR_STEP
    return ((uint64_t)(unsigned int)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ uint64_t wave_readlane64(uint64_t x, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)(x & 0xffffffffull), l);
    const int hi = __builtin_amdgcn_readlane((int)(x >> 32), l);
    return ((uint64_t)(unsigned int)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ float wave_max_f32(float x)
{
    const int ninf = __float_as_int(-__builtin_huge_valf());
#define CCZ_FMAX_STEP(ctrl_, rmask_) x = fmaxf(x, __int_as_float(CCZ_DPP(ninf, __float_as_int(x), ctrl_, rmask_)));
    CCZ_FMAX_STEP(0x111, 0xf)
    CCZ_FMAX_STEP(0x112, 0xf)
    CCZ_FMAX_STEP(0x114, 0xf)
    CCZ_FMAX_STEP(0x118, 0xf)
    CCZ_FMAX_STEP(0x142, 0xa)
    CCZ_FMAX_STEP(0x143, 0xc)
#undef CCZ_FMAX_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_sum_f32(float x)
{
#define CCZ_FADD_STEP(ctrl_, rmask_) x += __int_as_float(CCZ_DPP(0, __float_as_int(x), ctrl_, rmask_));
    CCZ_FADD_STEP(0x111, 0xf)
    CCZ_FADD_STEP(0x112, 0xf)
    CCZ_FADD_STEP(0x114, 0xf)
    CCZ_FADD_STEP(0x118, 0xf)
    CCZ_FADD_STEP(0x142, 0xa)
    CCZ_FADD_STEP(0x143, 0xc)
#undef CCZ_FADD_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// maximum over the 64 lanes of a double (no NaNs), returned in every lane
__device__ __forceinline__ double wave_max_f64(double x)
{
    const long long ninf = __double_as_longlong(-__builtin_huge_val());
    const int ilo = (int)(ninf & 0xffffffffll), ihi = (int)(ninf >> 32);
#define CCZ_MAX_STEP(ctrl_, rmask_)                                                         \
    {                                                                                       \
        const long long b = __double_as_longlong(x);                                        \
        const int lo = CCZ_DPP(ilo, (int)(b & 0xffffffffll), ctrl_, rmask_);                \
        const int hi = CCZ_DPP(ihi, (int)(b >> 32), ctrl_, rmask_);                         \
        const double t = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);    \
        x = __builtin_fmax(x, t);                                                           \
    }
    CCZ_MAX_STEP(0x111, 0xf)
    CCZ_MAX_STEP(0x112, 0xf)
    CCZ_MAX_STEP(0x114, 0xf)
    CCZ_MAX_STEP(0x118, 0xf)
    CCZ_MAX_STEP(0x142, 0xa)
    CCZ_MAX_STEP(0x143, 0xc)
#undef CCZ_MAX_STEP
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ------------------------------------------------------------------ rules
// Sliding pieces and king-safety rays work on OCCUPANCY WORDS (one per rank and per file) with
// ctz / bit-reverse instead of square-by-square ray walks: straight-line code, no divergent loops
// (the ray-walk version spent half its instructions on exec-mask bookkeeping).
struct GenScratch {
    uint16_t R[10];   // bit f of R[r]: square (rank r, file f) occupied
    uint16_t F[10];   // bit r of F[f]: square (rank r, file f) occupied  (F[9] unused)
    uint8_t plist[16];
    uint8_t pad_[8];
    __align__(16) uint8_t cand[64 * 16]; // per-lane destination squares
    uint16_t list[kPseudoCap];
    uint32_t mask[kMaskWords + 2];
};

// steps (>= 1, 0 = none) from index `pos` to the first and second set bit of `word` in the +direction
__device__ __forceinline__ void line_scan(uint32_t word, int pos, int &step1, int &step2)
{
    const uint32_t m = word >> (pos + 1);
    step1 = __ffs((int)m);
    const uint32_t m2 = m & (m - 1u);
    step2 = __ffs((int)m2);
}

// Is the king of the side `turn` (1 RED / 0 BLACK), standing on ksq, attacked on the board that
// results from moving `mover` from `from` to `to` (from < 0: the board as it is)? Includes the
// facing-kings rule. sq[90..95] must be 0 (index 95 serves as the "no square" dummy).
__device__ inline bool king_attacked(const uint8_t *sq, const GenScratch &S, int ksq, int from, int to, int mover, int turn)
{
    const int eb = turn ? 8 : 0; // enemy piece-code base
    const int eR = eb + ROOK, eC = eb + CANNON, eK = eb + KING, eN = eb + KNIGHT, eP = eb + PAWN;
    const int r = ksq / 9, f = ksq - 9 * r;
#define AT(S_) ((S_) == from ? 0 : ((S_) == to ? mover : (int)sq[S_]))
    uint32_t rw = S.R[r], fw = S.F[f];
    if (from >= 0) { // occupancy of the king's rank and file after the move
        const int fr = from / 9, ff = from - 9 * fr, tr = to / 9, tf = to - 9 * tr;
        rw = fr == r ? (rw & ~(1u << ff)) : rw;
        rw = tr == r ? (rw | (1u << tf)) : rw;
        fw = ff == f ? (fw & ~(1u << fr)) : fw;
        fw = tf == f ? (fw | (1u << tr)) : fw;
    }
    bool att = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) { // first piece on a ray: rook (or the facing king on the file); second: cannon
        const bool horiz = d >= 2, positive = !(d & 1);
        const int len = horiz ? 9 : 10;
        uint32_t word = horiz ? rw : fw;
        int pos = horiz ? f : r, delta = horiz ? 1 : 9;
        if (!positive) { word = __brev(word) >> (32 - len); pos = len - 1 - pos; delta = -delta; }
        int s1, s2;
        line_scan(word, pos, s1, s2);
        const int q1i = s1 ? ksq + s1 * delta : 95, q2i = s2 ? ksq + s2 * delta : 95;
        const int q1 = AT(q1i), q2 = AT(q2i);
        att |= (q1 == eR) | (!horiz && q1 == eK) | (q2 == eC);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) { // a knight's leg is the king's diagonal neighbour
        const int a = (d & 2) ? -1 : 1, b = (d & 1) ? -1 : 1;
        const int rl = r + a, fl = f + b;
        const bool vl = (unsigned)rl < 10u && (unsigned)fl < 9u;
        const int sl = vl ? fl + 9 * rl : 95;
        const bool open = vl && AT(sl) == 0;
        const int r2 = rl + a, f3 = fl + b;
        const int i2 = (open && (unsigned)r2 < 10u) ? fl + 9 * r2 : 95;
        const int i3 = (open && (unsigned)f3 < 9u) ? f3 + 9 * rl : 95;
        att |= (AT(i2) == eN) | (AT(i3) == eN);
    }
    {
        const int fw_ = turn ? -1 : 1; // direction of travel of the ENEMY pawns
        const int rp = r - fw_;
        const int i0 = (unsigned)rp < 10u ? f + 9 * rp : 95;
        const bool crossed = turn ? (r <= 4) : (r >= 5); // an enemy pawn on the king's rank is over the river
        const int i1 = (crossed && f > 0) ? ksq - 1 : 95, i2 = (crossed && f < 8) ? ksq + 1 : 95;
        att |= (AT(i0) == eP) | (AT(i1) == eP) | (AT(i2) == eP);
    }
#undef AT
    return att;
}

struct GenResult {
    int n_legal;
    int ksq;
    bool insufficient;
    bool overflow;
};

// Legal moves of the side to move as a bitmask (S.mask) and, if ids_out is non-null, as a list in
// `board.legal_moves` order: ascending id (the canonical order of this build) when rank is null --
// then bit i of the mask is move id i --, else ascending rank[id] (bit r of the mask is the move of rank r).
// Lane 4p+d generates direction d of piece p; lane j then tests pseudo-move j for king safety.
// Must be called by all 64 lanes of the wave; sq[90..95] must be 0.
__device__ inline GenResult gen_legal(const uint8_t *sq, int turn, GenScratch &S, uint16_t *ids_out, int lane,
                                   unsigned long long *sp = nullptr, const uint16_t *rank = nullptr, const uint16_t *unrank = nullptr,
                                   uint32_t trankpack = 0u)
{
    GenResult R;
    (void)sp;
    const int ownbit = turn ? 0 : 1;
#define OWN(q) ((q) != 0 && (((q) >> 3) == ownbit))
    const int p0 = sq[lane];
    const int p1 = lane < 26 ? sq[64 + lane] : 0;
    const bool o0 = OWN(p0), o1 = OWN(p1);
    const uint64_t m0 = __ballot(o0), m1 = __ballot(o1);
    const int n0 = __popcll(m0);
    int npieces = n0 + __popcll(m1);
    {
        const int i0 = __popcll(m0 & lanemask_lt(lane)), i1 = n0 + __popcll(m1 & lanemask_lt(lane));
        if (o0 && i0 < 16) S.plist[i0] = (uint8_t)lane;
        if (o1 && i1 < 16) S.plist[i1] = (uint8_t)(64 + lane);
    }
    R.overflow = npieces > 16;
    if (npieces > 16) npieces = 16;
    const int kc = turn ? KING : KING + 8;
    const uint64_t k0 = __ballot(p0 == kc), k1 = __ballot(p1 == kc);
    R.ksq = k0 ? (__ffsll((long long)k0) - 1) : (k1 ? 64 + (__ffsll((long long)k1) - 1) : -1);
    {
        const int t0 = p0 & 7, t1 = p1 & 7;
        const bool a0 = p0 != 0 && t0 >= PAWN && t0 <= KNIGHT, a1 = p1 != 0 && t1 >= PAWN && t1 <= KNIGHT;
        R.insufficient = (__ballot(a0) | __ballot(a1)) == 0ull;
    }
    { // occupancy words: ranks from the square-order ballot, files from a transposed ballot
        const uint64_t lo = __ballot(p0 != 0), hi = __ballot(p1 != 0);
        if (lane < 10) {
            const int s = 9 * lane;
            const uint64_t v = s < 64 ? ((lo >> s) | (s > 55 ? (hi << (64 - s)) : 0ull)) : (hi >> (s - 64));
            S.R[lane] = (uint16_t)(v & 0x1ffu);
        }
        const int t0 = lane, t1 = 64 + lane;
        const int q0 = sq[(t0 / 10) + 9 * (t0 % 10)];
        const int q1 = t1 < 90 ? sq[(t1 / 10) + 9 * (t1 % 10)] : 0;
        const uint64_t tlo = __ballot(q0 != 0), thi = __ballot(q1 != 0);
        if (lane < 9) {
            const int s = 10 * lane;
            const uint64_t v = s < 64 ? ((tlo >> s) | (s > 54 ? (thi << (64 - s)) : 0ull)) : (thi >> (s - 64));
            S.F[lane] = (uint16_t)(v & 0x3ffu);
        }
    }
    for (int w = lane; w < kMaskWords + 2; w += 64) S.mask[w] = 0u;
    wave_sync();
    CCZ_GSTAMP(sp, lane, 10)

    // ---- phase B: pseudo-legal generation, lane = 4*piece + direction
    int cnt = 0, from = 0;
    uint8_t *out = S.cand + lane * 16;
    const int p = lane >> 2, d = lane & 3;
    if (p < npieces) {
        from = S.plist[p];
        const int pc = sq[from], t = pc & 7;
        const int r = from / 9, f = from - 9 * r;
        const int dr = (d == 0) - (d == 1), df = (d == 2) - (d == 3); // orthogonal step
        const int a = (d & 2) ? -1 : 1, b = (d & 1) ? -1 : 1;         // diagonal step
        if (t == ROOK || t == CANNON) {
            const bool horiz = d >= 2, positive = !(d & 1);
            const int len = horiz ? 9 : 10;
            uint32_t word = horiz ? S.R[r] : S.F[f];
            int pos = horiz ? f : r, delta = horiz ? 1 : 9;
            if (!positive) { word = __brev(word) >> (32 - len); pos = len - 1 - pos; delta = -delta; }
            int s1, s2;
            line_scan(word, pos, s1, s2);
            const int empt = s1 ? s1 - 1 : len - 1 - pos; // quiet moves: steps 1..empt
            // destination bytes from + i*delta, i = 1..9 (bytes past `empt` are never read)
            *(uint64_t *)out = (uint64_t)from * 0x0101010101010101ull + (uint64_t)(int64_t)delta * 0x0807060504030201ull;
            out[8] = (uint8_t)(from + 9 * delta);
            cnt = empt;
            const int cs = t == ROOK ? s1 : s2; // a rook takes the first piece, a cannon the second (over its screen)
            if (cs) {
                const int s2q = from + cs * delta, q = sq[s2q];
                if (!OWN(q)) out[cnt++] = (uint8_t)s2q;
            }
        } else if (t == KNIGHT) {
            const int rl = r + dr, fl = f + df;
            if ((unsigned)rl < 10u && (unsigned)fl < 9u && sq[fl + 9 * rl] == 0) {
#pragma unroll
                for (int j = -1; j <= 1; j += 2) {
                    const int r2 = r + 2 * dr + (dr ? 0 : j), f2 = f + 2 * df + (df ? 0 : j);
                    if ((unsigned)r2 < 10u && (unsigned)f2 < 9u) {
                        const int s2 = f2 + 9 * r2, q = sq[s2];
                        if (!OWN(q)) out[cnt++] = (uint8_t)s2;
                    }
                }
            }
        } else if (t == BISHOP) {
            const int r2 = r + 2 * a, f2 = f + 2 * b;
            if ((unsigned)r2 < 10u && (unsigned)f2 < 9u && (turn ? r2 <= 4 : r2 >= 5) && sq[(f + b) + 9 * (r + a)] == 0) {
                const int s2 = f2 + 9 * r2, q = sq[s2];
                if (!OWN(q)) out[cnt++] = (uint8_t)s2;
            }
        } else if (t == ADVISOR || t == KING) {
            const int r2 = t == KING ? r + dr : r + a, f2 = t == KING ? f + df : f + b;
            const bool palace = f2 >= 3 && f2 <= 5 && (turn ? (r2 >= 0 && r2 <= 2) : (r2 >= 7 && r2 <= 9));
            if (palace) {
                const int s2 = f2 + 9 * r2, q = sq[s2];
                if (!OWN(q)) out[cnt++] = (uint8_t)s2;
            }
        } else if (t == PAWN) {
            const int fw = turn ? 1 : -1;
            const bool crossed = turn ? r >= 5 : r <= 4;
            int r2 = -1, f2 = -1;
            if (d == 0) { r2 = r + fw; f2 = f; }
            else if (d == 1 && crossed) { r2 = r; f2 = f - 1; }
            else if (d == 2 && crossed) { r2 = r; f2 = f + 1; }
            if ((unsigned)r2 < 10u && (unsigned)f2 < 9u) {
                const int s2 = f2 + 9 * r2, q = sq[s2];
                if (!OWN(q)) out[cnt++] = (uint8_t)s2;
            }
        }
    }
    CCZ_GSTAMP(sp, lane, 11)
    const int incl = wave_incl_scan(cnt, lane);
    int npseudo = __builtin_amdgcn_readlane(incl, 63);
    const int excl = incl - cnt;
    if (npseudo > kPseudoCap) { R.overflow = true; npseudo = kPseudoCap; }
    for (int i = 0; i < cnt; ++i)
        if (excl + i < kPseudoCap) S.list[excl + i] = (uint16_t)((from << 8) | out[i]);
    wave_sync();

    CCZ_GSTAMP(sp, lane, 12)
    // ---- phase C: king safety, lane per pseudo-move
    for (int j = lane; j < npseudo; j += 64) {
        const int fr = S.list[j] >> 8, to = S.list[j] & 0xff;
        const int mover = sq[fr];
        const int ksq = (mover & 7) == KING ? to : R.ksq;
        if (ksq >= 0 && !king_attacked(sq, S, ksq, fr, to, mover, turn)) {
            uint32_t id = c_tab.inv[fr * 90 + to];
            if (id < (uint32_t)kNMoves) {
                if (rank) id = rank[id];
                atomicOr(&S.mask[id >> 5], 1u << (id & 31));
            }
        }
    }
    wave_sync();

    CCZ_GSTAMP(sp, lane, 13)
    // ---- phase D: count and list in ascending id order
    const uint32_t w0 = S.mask[lane];
    const uint32_t w1 = lane < 2 ? S.mask[64 + lane] : 0u;
    const int c0 = __popc(w0), c1 = __popc(w1);
    const int incl0 = wave_incl_scan(c0, lane);
    const int total0 = __builtin_amdgcn_readlane(incl0, 63);
    const int c64 = __builtin_amdgcn_readlane(c1, 0), c65 = __builtin_amdgcn_readlane(c1, 1);
    R.n_legal = total0 + c64 + c65;
    if (R.n_legal > kMaxLegal) R.overflow = true;
    uint16_t *const ids_final = ids_out;
    if (ids_out && trankpack) ids_out = S.list; // major key by piece type: list by rank into LDS first, partition below
    if (ids_out) {
        int o = incl0 - c0;
        uint32_t w = w0;
        while (w) {
            const int bit = __ffs((int)w) - 1;
            w &= w - 1;
            if (o < kMaxLegal) ids_out[o] = unrank ? unrank[lane * 32 + bit] : (uint16_t)(lane * 32 + bit);
            ++o;
        }
        if (lane < 2) {
            o = total0 + (lane ? c64 : 0);
            w = w1;
            while (w) {
                const int bit = __ffs((int)w) - 1;
                w &= w - 1;
                if (o < kMaxLegal) ids_out[o] = unrank ? unrank[(64 + lane) * 32 + bit] : (uint16_t)((64 + lane) * 32 + bit);
                ++o;
            }
        }
    }
    if (ids_final && trankpack) {
        // stable partition of the rank-ordered list by the major key of the mover's piece type (<= 8 classes): the final
        // position of an entry = entries of smaller classes + earlier entries of its own class (two ballots per class)
        wave_sync();
        const int n = R.n_legal < kMaxLegal ? R.n_legal : kMaxLegal;
        const int i0 = lane, i1 = 64 + lane;
        const int id0 = i0 < n ? S.list[i0] : 0, id1 = i1 < n ? S.list[i1] : 0;
        const int cl0 = i0 < n ? (int)((trankpack >> (3 * (sq[c_tab.from[id0]] & 7))) & 7u) : 8;
        const int cl1 = i1 < n ? (int)((trankpack >> (3 * (sq[c_tab.from[id1]] & 7))) & 7u) : 8;
        int base = 0, p0 = 0, p1 = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint64_t m0 = __ballot(cl0 == c), m1 = __ballot(cl1 == c);
            const int n0c = __popcll(m0);
            if (cl0 == c) p0 = base + __popcll(m0 & lanemask_lt(lane));
            if (cl1 == c) p1 = base + n0c + __popcll(m1 & lanemask_lt(lane));
            base += n0c + __popcll(m1);
        }
        if (i0 < n) ids_final[p0] = (uint16_t)id0;
        if (i1 < n) ids_final[p1] = (uint16_t)id1;
    }
    CCZ_GSTAMP(sp, lane, 14)
#undef OWN
    return R;
}

// load a 96-byte mailbox row into LDS (24 dwords) with the 6 pad bytes forced to zero
__device__ __forceinline__ void load_board(uint8_t *s_sq, const uint8_t *src, int lane)
{
    if (lane < 24) {
        uint32_t v = ((const uint32_t *)src)[lane];
        if (lane == 22) v &= 0x0000ffffu;
        if (lane == 23) v = 0u;
        ((uint32_t *)s_sq)[lane] = v;
    }
}

// ------------------------------------------------------------------ deterministic math (twin of oracle/xq_sample.c)
__device__ inline double det_log(double x)
{
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((long long)((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double acc = 1.0 / 27.0;
    for (int i = 25; i >= 1; i -= 2) acc = acc * s2 + 1.0 / (double)i;
    return (double)e * 0.6931471805599453 + 2.0 * s * acc;
}

__device__ inline double det_exp(double x)
{
    if (x < -708.0) return 0.0;
    const double t = x * 1.4426950408889634 + 0.5;
    const double n = floor(t);
    const double r = x - n * 0.693147180369123816490 - n * 1.90821492927058770002e-10;
    double acc = 1.0;
    for (int i = 14; i >= 1; --i) acc = acc * r / (double)i + 1.0;
    const int ni = (int)n;
    if (ni < -1022) return 0.0;
    return acc * __longlong_as_double((long long)((uint64_t)(ni + 1023) << 52));
}

__device__ inline void philox4x32(uint64_t key, uint64_t ctr_hi, uint64_t ctr_lo, uint32_t o[4])
{
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ inline void uniform2(uint64_t seed, uint64_t board, uint64_t move_no, uint32_t child, uint32_t draw, double &ua, double &ub)
{
    uint32_t o[4];
    const uint64_t lo = (move_no << 32) | ((uint64_t)(child & 0xfffu) << 20) | (uint64_t)(draw & 0xfffffu);
    philox4x32(seed, board, lo, o);
    ua = (double)(2 * ((((uint64_t)o[0] << 32) | o[1]) >> 12) + 1) * 1.1102230246251565e-16;
    ub = (double)(2 * ((((uint64_t)o[2] << 32) | o[3]) >> 12) + 1) * 1.1102230246251565e-16;
}

// Gamma(alpha,1), alpha < 1 (Dirichlet component of mcts.py:220): Marsaglia-Tsang on alpha+1 with
// polar normals and the U^(1/alpha) boost; bounded loop so that every wave terminates.
__device__ inline double det_gamma(uint64_t seed, uint64_t board, uint64_t move_no, uint32_t child, double alpha)
{
    const double d = (alpha + 1.0) - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double ua, ub;
    uint32_t j = 0;
    while (j < 0xffff0u) {
        double x1, x2, s;
        for (;;) {
            uniform2(seed, board, move_no, child, j++, ua, ub);
            x1 = 2.0 * ua - 1.0; x2 = 2.0 * ub - 1.0;
            s = x1 * x1 + x2 * x2;
            if ((s < 1.0 && s > 0.0) || j >= 0xffff0u) break;
        }
        const double z = x1 * sqrt(-2.0 * det_log(s) / s);
        double v = 1.0 + c * z;
        if (v <= 0.0) continue;
        v = v * v * v;
        uniform2(seed, board, move_no, child, j++, ua, ub);
        if (det_log(ua) < 0.5 * z * z + d - d * v + d * det_log(v))
            return d * v * det_exp(det_log(ub) / alpha);
    }
    return 0.0;
}

} // namespace ccz
