// cczero_kernels.h -- the gfx950 kernels of the lockstep self-play engine (one wave per board).
#pragma once
#include "cczero_device.h"

namespace ccz {

constexpr uint16_t kHalfOne = 0x3C00; // fp16 1.0
constexpr int kMaxDepth = 512;     // selection path capacity (Dev.maxd <= kMaxDepth)

__device__ __forceinline__ void set_err(const Dev &D, int bit) { atomicOr(D.err, bit); }

// ------------------------------------------------------------------ new game / set position
// Start position: rank 0 = RNBAKABNR (red), rank 2 cannons b,h, rank 3 pawns a,c,e,g,i; black mirrored.
__device__ __forceinline__ int start_piece(int s)
{
    const int r = s / 9, f = s - 9 * r;
    const int rr = r <= 4 ? r : 9 - r; // distance from the own back rank
    const int add = r <= 4 ? 0 : 8;
    int t = 0;
    if (rr == 0) {
        const int ff = f <= 4 ? f : 8 - f;
        t = ff == 0 ? ROOK : ff == 1 ? KNIGHT : ff == 2 ? BISHOP : ff == 3 ? ADVISOR : KING;
    } else if (rr == 2) {
        t = (f == 1 || f == 7) ? CANNON : 0;
    } else if (rr == 3) {
        t = (f & 1) ? 0 : PAWN;
    }
    return t ? t + add : 0;
}

// Node(None, 1.0) (mcts.py:94) as the only node of board b's tree in pool half `half`, no pending leaf: what a new game
// (init_board) and MCTS.update_with_move(-1) (k_reset_tree) share. One lane.
__device__ __forceinline__ void fresh_root(const Dev &D, int b, int half)
{
    const size_t base = ((size_t)b * 2 + half) * (size_t)D.cap;
    D.nodeA[base] = NodeA{0, 0.0f, 1.0f, -1};
    D.nodeB[base] = 0u;
    D.path_len[b] = 0;
    D.leaf_status[b] = CCZ_LEAF_SKIP;
}

// (re)initialise board b from D.root_sq[b] / the given turn+halfmove: key, chain, fresh tree, empty record
__device__ inline void init_board(const Dev &D, int b, int lane, int turn, int halfmove, bool new_game_no)
{
    const uint8_t *sq = D.root_sq + (size_t)b * 96;
    uint64_t k = 0;
    for (int s = lane; s < 90; s += 64) {
        const int pc = sq[s];
        if (pc) k ^= zob(pc, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) k ^= __shfl_xor(k, o);
    if (turn) k ^= kTurnKey;
    if (lane == 0) {
        BoardMeta m = D.meta[b];
        m.key = k;
        m.halfmove = halfmove;
        m.chain_len = 1;
        m.ply = 0;
        m.n_nodes = 1;
        m.turn = (uint8_t)turn;
        m.over = 0;
        m.winner = -1;
        m.pi_used = 0;
        if (new_game_no) m.game_no += 1;
        m.half = (uint8_t)*D.half;
        D.meta[b] = m;
        D.chain[(size_t)b * kChainCap] = k;
        D.chain_chk[(size_t)b * 2] = 0ull; // (the first position of a chain never lies inside a repetition window)
        D.chain_chk[(size_t)b * 2 + 1] = 0ull;
        fresh_root(D, b, m.half);
    }
}

__global__ __launch_bounds__(64) void k_reset(Dev D, const uint8_t *mask)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (mask && !mask[b]) return;
    uint8_t *sq = D.root_sq + (size_t)b * 96;
    for (int s = lane; s < 96; s += 64) sq[s] = s < 90 ? (uint8_t)start_piece(s) : 0;
    __syncthreads();
    init_board(D, b, lane, 1, 0, true);
}

// MCTS.update_with_move(-1) (mcts.py:176-178): a fresh root on the live pool half; position, history and record stay
__global__ void k_reset_tree(Dev D, const uint8_t *mask)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= D.B || (mask && !mask[b])) return;
    const int half = *D.half;
    fresh_root(D, b, half);
    D.meta[b].n_nodes = 1;
    D.meta[b].half = (uint8_t)half;
}

__global__ __launch_bounds__(64) void k_set_position(Dev D, int b, const uint8_t *sq_in, int turn, int halfmove)
{
    const int lane = threadIdx.x;
    uint8_t *sq = D.root_sq + (size_t)b * 96;
    for (int s = lane; s < 96; s += 64) sq[s] = s < 90 ? sq_in[s] : 0;
    __syncthreads();
    init_board(D, b, lane, turn, halfmove, true);
}

// ------------------------------------------------------------------ leaf evaluation shared by select and finish_move
struct LeafEval {
    int n_legal;
    int status; // CCZ_LEAF_*
    bool tie;
    bool insufficient;
    int rep;       // occurrences of the current position in the history chain (incl. itself)
    int first_occ; // chain index of its earliest occurrence
    int ksq;       // king square of the side to move
};

// s_chain[0..chain_len) holds the keys since the last capture incl. the current position (last)
__device__ inline LeafEval eval_position(const uint8_t *s_sq, int turn, int halfmove, uint64_t key,
                                         const uint64_t *s_chain, int chain_len, GenScratch &S,
                                         uint16_t *ids_out, int lane, bool &overflow, unsigned long long *sp = nullptr,
                                         const uint16_t *rank = nullptr, const uint16_t *unrank = nullptr, uint32_t trankpack = 0u)
{
    const GenResult g = gen_legal(s_sq, turn, S, ids_out, lane, sp, rank, unrank, trankpack);
    overflow = g.overflow;
    int rep = 0, first_occ = -1;
    for (int i0 = 0; i0 < chain_len; i0 += 64) {
        const int i = i0 + lane;
        const uint64_t hit = __ballot(i < chain_len && s_chain[i] == key);
        if (hit && first_occ < 0) first_occ = i0 + __ffsll((long long)hit) - 1;
        rep += __popcll(hit);
    }
    // tools.py:109-123 is_tie = insufficient material or fourfold repetition or sixty moves
    const bool sixty = halfmove >= 120 && g.n_legal > 0;
    LeafEval L;
    L.n_legal = g.n_legal;
    L.insufficient = g.insufficient;
    L.rep = rep;
    L.first_occ = first_occ;
    L.ksq = g.ksq;
    L.tie = g.insufficient || rep >= 4 || sixty;
    // mcts.py:116-126: not end and not tie -> expand ; end and tie -> 0.0 ; else side to move lost
    if (g.n_legal == 0) L.status = L.tie ? CCZ_LEAF_DRAW : CCZ_LEAF_LOSS;
    else L.status = L.tie ? CCZ_LEAF_DRAW : CCZ_LEAF_EXPAND;
    return L;
}

// ------------------------------------------------------------------ K1: select + make-move + movegen + terminal + encode
struct SelectShared {
    __align__(16) uint8_t sq[96];
    uint64_t chain[kChainCap];
    GenScratch S;
    union {
        __align__(16) uint32_t enc[948]; // staging of the 3 live plane groups (945 dwords); used after move generation
        struct {           // deferred make-move of the selection path; dead before move generation starts
            uint16_t mv[kMaxDepth];
            uint8_t from[kMaxDepth], to[kMaxDepth], pc[kMaxDepth], cap[kMaxDepth];
        } pm;
    };
};

// per-board state every phase needs, loaded in ONE round at the top of the kernel
struct Prefetch {
    BoardMeta m;
    uint32_t sqw;      // lane < 24: dword `lane` of the root mailbox
    uint64_t c0;       // chain key `lane` (keys 64.. are fetched on demand: > 64 plies without a capture is rare)
    int half;          // live pool half (one global word, the same for all boards)
    NodeA root;        // node 0 of the live half
    uint32_t rootw;
    NodeA kid;         // lane i: node 1 + i = child i of the root (the root's children always start at node 1)
    uint32_t kidw;
};

// what the expand+backup phase of this launch changed at the top of the tree (the prefetched root and
// root-children records predate those stores)
struct TopPatch {
    bool active;       // a backup ran
    bool root_expanded;// the root itself was the leaf and received children 1..k
    bool kid_expanded; // the depth-1 path node was the leaf and received children n0..n0+k
    int k, first_id, n0;
    int rootN;
    float rootQ;
    int node1, N1;     // path node at depth 1 (if depth >= 1) with its updated N, Q
    float Q1;
    bool has1;
};

__device__ __forceinline__ Prefetch prefetch_board(const Dev &D, int b, int lane)
{
    Prefetch P;
    P.m = D.meta[b];
    P.sqw = lane < 24 ? ((const uint32_t *)(D.root_sq + (size_t)b * 96))[lane] : 0u;
    const uint64_t *ch = D.chain + (size_t)b * kChainCap;
    P.c0 = ch[lane];
    P.half = *D.half;
    const size_t base = ((size_t)b * 2 + P.half) * (size_t)D.cap;
    P.root = D.nodeA[base];
    P.rootw = D.nodeB[base];
    P.kid = D.nodeA[base + 1 + lane];
    P.kidw = D.nodeB[base + 1 + lane];
    return P;
}

// The leaf end of a selection: the moves sh.pm.mv[0 .. depth) are made on the LDS board sh.sq (the root position), the Zobrist keys
// of the path extend the history chain sh.chain, then legal moves / game end (net.py:154-157, mcts.py:116-117) and the evaluator
// input of the leaf are written to the slots of board `b`. Shared by select_phase (b = the board searched) and k_scout (b = a scout
// slot: the leaf is a sibling of another board's pending leaf).
__device__ inline void leaf_tail(const Dev &D, int b, int lane, uint16_t *leaf_in, SelectShared &sh, int depth, int turn, int halfmove,
                                 int chain_len, uint64_t key, bool count_stats)
{
    uint8_t *s_sq = sh.sq;
    uint64_t *s_chain = sh.chain;
    CCZ_STAMP(D, b, lane, 4)
    // ---- replay the selection path on the LDS board: (1) table lookups in parallel, (2) the inherently
    // serial piece shuffling by one lane, (3) Zobrist deltas in parallel + XOR prefix scan
    if (depth > 0) {
        for (int j = lane; j < depth; j += 64) {
            const int mvj = sh.pm.mv[j];
            sh.pm.from[j] = c_tab.from[mvj];
            sh.pm.to[j] = c_tab.to[mvj];
        }
        wave_sync();
        int lastcap = -1;
        const bool pawn_zeroes = (D.rule_flags & 2u) != 0;
        if (lane == 0) {
            for (int j = 0; j < depth; ++j) {
                const int fr = sh.pm.from[j], to = sh.pm.to[j];
                const uint8_t pc = s_sq[fr], cp = s_sq[to];
                s_sq[to] = pc;
                s_sq[fr] = 0;
                sh.pm.pc[j] = pc;
                sh.pm.cap[j] = cp;
                if (cp || (pawn_zeroes && (pc & 7) == PAWN)) lastcap = j; // "lastcap" = last clock-resetting move
            }
        }
        lastcap = __builtin_amdgcn_readfirstlane(lastcap);
        wave_sync();
        // keys: key_j = root_key ^ XOR_{i<=j} delta_i ; a capture at move c restarts the chain at key_c
        const int first = lastcap >= 0 ? lastcap : 0;
        const int new_len = (lastcap >= 0 ? 0 : chain_len) + (depth - first);
        if (new_len > kChainCap) {
            set_err(D, 64);
            if (lane == 0) D.leaf_status[b] = CCZ_LEAF_SKIP;
            return;
        }
        uint64_t carry = key;
        const int off = lastcap >= 0 ? -first : chain_len; // chain slot of move j is off + j
        for (int j0 = 0; j0 < depth; j0 += 64) {
            const int j = j0 + lane;
            uint64_t dlt = 0;
            if (j < depth) {
                const int fr = sh.pm.from[j], to = sh.pm.to[j], pc = sh.pm.pc[j], cp = sh.pm.cap[j];
                dlt = zob(pc, fr) ^ zob(pc, to) ^ kTurnKey;
                if (cp) dlt ^= zob(cp, to);
            }
            dlt = wave_incl_xor64(dlt) ^ carry;
            if (j < depth && j >= first) s_chain[CCZ_IDX(D, off + j, kChainCap)] = dlt;
            carry = wave_readlane64(dlt, 63);
        }
        key = carry;
        halfmove = lastcap >= 0 ? depth - 1 - lastcap : halfmove + depth;
        chain_len = new_len;
        turn ^= depth & 1;
    }
    wave_sync();

    CCZ_STAMP(D, b, lane, 5)
    // ---- leaf: legal moves (net.py:154-157), game end (mcts.py:116-117)
    bool overflow;
#ifdef CCZ_STAMPS
    unsigned long long *sp = D.stamps + (size_t)b * 16;
#else
    unsigned long long *sp = nullptr;
#endif
    const LeafEval L = eval_position(s_sq, turn, halfmove, key, s_chain, chain_len, sh.S,
                                     D.leaf_ids + (size_t)b * kMaxLegal, lane, overflow, sp, D.rank, D.unrank, D.trankpack);
    if (overflow) set_err(D, 4);
    CCZ_STAMP(D, b, lane, 6)
    if (lane == 0) {
        D.path_len[b] = depth;
        D.leaf_key[b] = key;
        D.leaf_k[b] = L.n_legal > kMaxLegal ? kMaxLegal : L.n_legal;
        D.leaf_status[b] = (uint8_t)L.status;
        if (count_stats) {
            BoardStats &st = D.stats[b];
            st.sum_depth += (unsigned long long)depth;
            if (depth > st.depth_peak) st.depth_peak = depth;
        }
    }

    // ---- evaluator input (net.py:160-177): groups 7 (red now), 15 (black now), 16 (side to move).
    // Staged in LDS: fill (zeros / the turn plane), scatter one fp16 1.0 per piece, stream out as dwords.
    CCZ_STAMP(D, b, lane, 7)
    if (leaf_in) {
        uint32_t *enc = sh.enc;
        const uint32_t tv = turn ? 0x3C003C00u : 0u;
#pragma unroll
        for (int it = 0; it < 15; ++it) {
            const int i = lane + 64 * it;
            if (i < 945) enc[i] = i >= 630 ? tv : 0u;
        }
        wave_sync();
        {
            uint16_t *eh = (uint16_t *)enc;
            const int q0 = s_sq[lane];
            const int q1 = lane < 26 ? s_sq[64 + lane] : 0;
            if (q0) eh[(q0 >> 3) * 630 + plane_of(D, q0 & 7) * 90 + lane] = kHalfOne;
            if (q1) eh[(q1 >> 3) * 630 + plane_of(D, q1 & 7) * 90 + 64 + lane] = kHalfOne;
        }
        wave_sync();
        uint32_t *row = (uint32_t *)(leaf_in + (size_t)b * 10710);
#pragma unroll
        for (int it = 0; it < 15; ++it) {
            const int i = lane + 64 * it;
            if (i < 945) row[i + (i < 315 ? 2205 : (i < 630 ? 4725 - 315 : 5040 - 630))] = enc[i];
        }
    }
}

// One wave: PUCT descent from the root of board b, leaf rules, evaluator input. Per tree level there is
// ONE dependent global load round (the children's 16-B NodeA records + their move/count words); the
// chosen child's own N / first_child / count are broadcast from the winning lane, not re-read.
__device__ inline void select_phase(const Dev &D, int b, int lane, uint16_t *leaf_in, SelectShared &sh, const Prefetch &P,
                                    const TopPatch &tp)
{
    uint8_t *s_sq = sh.sq;
    uint64_t *s_chain = sh.chain;
    const BoardMeta m = P.m;
    if (m.over) {
        if (lane == 0) D.leaf_status[b] = CCZ_LEAF_SKIP;
        return;
    }
    const size_t base = ((size_t)b * 2 + P.half) * (size_t)D.cap;
    const NodeA *A = D.nodeA + base;
    const uint32_t *Bn = D.nodeB + base;
    // the root record was requested together with everything else at the top of the kernel; what this
    // launch's backup changed in it is patched in from registers
    NodeA pa = P.root;
    uint32_t nb = P.rootw;
    if (tp.active) {
        pa.N = tp.rootN;
        pa.Q = tp.rootQ;
        if (tp.root_expanded) { pa.fc = 1; nb = (nb & 0xffffu) | ((uint32_t)tp.k << 16); }
    }
    if (lane < 24) {
        uint32_t v = P.sqw;
        if (lane == 22) v &= 0x0000ffffu;
        if (lane == 23) v = 0u;
        ((uint32_t *)s_sq)[lane] = v;
    }
    s_chain[lane] = P.c0;
    if (m.chain_len > 64) s_chain[64 + lane] = D.chain[(size_t)b * kChainCap + 64 + lane];
    wave_sync();

    CCZ_STAMP(D, b, lane, 3)
    int32_t *path = D.path + (size_t)b * D.maxd;
    int depth = 0, turn = m.turn, halfmove = m.halfmove, chain_len = m.chain_len;
    uint64_t key = m.key;
    bool bad = false;
    if (lane == 0) path[0] = 0;

    // ---- PUCT descent (mcts.py:105-111, 41-61)
    for (;;) {
        const int nc = (int)(nb >> 16);
        if (nc == 0) break;
        const double sqrtNp = sqrt((double)pa.N); // np.sqrt(parent.visits): float64
        double best = -__builtin_huge_val();
        int besti = 0x7fffffff, bN = 0, bfc = -1;
        float bQ = 0.0f;
        uint32_t bw = 0;
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int i = c0 + lane;
            if (i < nc) {
                NodeA c;
                uint32_t w;
                if (depth == 0 && c0 == 0 && pa.fc == 1) { // root children: prefetched (+ this launch's backup)
                    c = P.kid;
                    w = P.kidw;
                    if (tp.root_expanded) { c.N = 0; c.Q = 0.0f; c.fc = -1; w = (uint32_t)tp.first_id; }
                    else if (tp.has1 && 1 + i == tp.node1) {
                        c.N = tp.N1;
                        c.Q = tp.Q1;
                        if (tp.kid_expanded) { c.fc = tp.n0; w = (w & 0xffffu) | ((uint32_t)tp.k << 16); }
                    }
                } else {
                    c = A[CCZ_IDX(D, pa.fc + i, D.cap)];
                    w = Bn[CCZ_IDX(D, pa.fc + i, D.cap)];
                }
                // value + c_puct*prob*sqrt(N_parent)/(1+N): float32 product, float64 elsewhere; inf if unvisited
                const double sc = c.N == 0 ? __builtin_huge_val()
                                           : (double)c.Q + (double)(D.c_puct * c.P) * sqrtNp / (double)(1 + c.N);
                if (sc > best) { best = sc; besti = i; bN = c.N; bQ = c.Q; bfc = c.fc; bw = w; }
            }
        }
        // first maximum in insertion order wins (Python max()): wave max of the score, then the lowest index
        // holding it (a lane's running best is its lowest-index maximum; indices < 64 precede the second pass)
        const double top = wave_max_f64(best);
        const bool hit = best == top && besti < nc;
        const uint64_t h0 = __ballot(hit && besti < 64), h1 = __ballot(hit);
        if (h1 == 0ull) { bad = true; set_err(D, 32); break; } // NaN priors: no comparable child
        const int owner = __builtin_amdgcn_readfirstlane((h0 ? __ffsll((long long)h0) : __ffsll((long long)h1)) - 1);
        besti = __builtin_amdgcn_readlane(besti, owner);
        const int child = pa.fc + besti;
        pa.N = __builtin_amdgcn_readlane(bN, owner);
        pa.Q = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bQ), owner));
        pa.fc = __builtin_amdgcn_readlane(bfc, owner);
        nb = (uint32_t)__builtin_amdgcn_readlane((int)bw, owner);
        const int mv = (int)(nb & 0xffffu);
        // board.push(move) (mcts.py:111) is DEFERRED: only the move id is noted here, so that a tree level costs
        // one global load round plus the arg-max and nothing else sits on the critical path
        if (lane == 0) sh.pm.mv[depth] = (uint16_t)mv;
        ++depth;
        if (depth >= D.maxd) { bad = true; set_err(D, 2); break; }
        if (lane == 0) path[CCZ_IDX(D, depth, D.maxd)] = child;
    }
    wave_sync();
    if (bad) {
        if (lane == 0) D.leaf_status[b] = CCZ_LEAF_SKIP;
        return;
    }

    leaf_tail(D, b, lane, leaf_in, sh, depth, turn, halfmove, chain_len, key, true);
}

__global__ __launch_bounds__(64) void k_select(Dev D, uint16_t *leaf_in)
{
    __shared__ SelectShared sh;
    const Prefetch P = prefetch_board(D, blockIdx.x, threadIdx.x);
    TopPatch none;
    none.active = false; none.root_expanded = false; none.kid_expanded = false; none.k = 0; none.first_id = 0; none.n0 = 0;
    none.rootN = 0; none.rootQ = 0.0f; none.node1 = -1; none.N1 = 0; none.Q1 = 0.0f; none.has1 = false;
    select_phase(D, blockIdx.x, threadIdx.x, leaf_in, sh, P, none);
}

// ------------------------------------------------------------------ scouts: the NEXT leaves of a board, before it asks for them
// The reference's first-maximum rule (mcts.py:47-48,59-61: an unvisited child scores +inf, max() returns the first one) makes the
// order in which a node's children are first visited the order of board.legal_moves: after child i of a node X has been expanded,
// the next simulations that reach X expand children i + 1, i + 2, ... So when board r's pending leaf is child i of X, scout slot
// (active + j * active + r) is handed child i + 1 + j of X as ITS pending leaf -- position, legal moves, status, key and evaluator
// input, exactly what the selection writes for a leaf -- and goes through the evaluation cache's plan with it: one evaluator call of
// `1 + scouts` rows answers the leaf AND its next siblings, which board r then finds in the table (ccz_scout, round 6: one game at a
// time -- MCTS_AI, the UCI loop -- is one 90-pixel row per evaluator call, the worst shape for the chip). Scout slots have no tree
// and no game of their own (the simulator kernels run on boards 0 .. active - 1 only); they read board r's root, chain and path.
// Results are unchanged: the table returns what the evaluator returns for the position (tests/test_gpu_scouts.py).
__device__ inline void scout_wave(const Dev &D, uint16_t *leaf_in, int active, int q, int lane, SelectShared &sh)
{
    const int b = active + q;                           // the scout slot
    const int r = q % active;                           // the board it scouts for
    const int ahead = 1 + q / active;                   // how many children past that board's pending leaf
    const Prefetch P = prefetch_board(D, r, lane);
    const int d = D.path_len[r];
    const int st_r = D.leaf_status[r];
    const int32_t *path = D.path + (size_t)r * D.maxd;
    const int pj = path[lane < D.maxd ? lane : 0];      // path nodes 0 .. 63 (deeper paths are not scouted)
    bool none = P.m.over || d < 1 || d >= 64 || st_r == CCZ_LEAF_SKIP || st_r == CCZ_LEAF_NONE;
    const size_t base = ((size_t)r * 2 + P.half) * (size_t)D.cap;
    const NodeA *A = D.nodeA + base;
    const uint32_t *Bn = D.nodeB + base;
    int sib = -1;
    if (!none) {
        const int leaf = __builtin_amdgcn_readlane(pj, __builtin_amdgcn_readfirstlane(d));
        const int par = __builtin_amdgcn_readlane(pj, __builtin_amdgcn_readfirstlane(d - 1));
        const NodeA X = A[CCZ_IDX(D, par, D.cap)];
        const int nc = (int)(Bn[CCZ_IDX(D, par, D.cap)] >> 16);
        sib = leaf + ahead;
        // the sibling exists, and nobody has been there: an expanded or visited child already has its evaluation (or needs none)
        none = X.fc < 0 || leaf < X.fc || sib >= X.fc + nc;
        if (!none) {
            const NodeA S = A[CCZ_IDX(D, sib, D.cap)];
            none = S.fc >= 0 || S.N != 0;
        }
    }
    if (none) {
        if (lane == 0) { D.leaf_status[b] = CCZ_LEAF_NONE; D.leaf_k[b] = 0; D.path_len[b] = 0; }
        return;
    }
    // the LDS board and chain of board r (as select_phase sets them up), the moves down to X, then the sibling's move
    if (lane < 24) {
        uint32_t v = P.sqw;
        if (lane == 22) v &= 0x0000ffffu;
        if (lane == 23) v = 0u;
        ((uint32_t *)sh.sq)[lane] = v;
    }
    sh.chain[lane] = P.c0;
    if (P.m.chain_len > 64) sh.chain[64 + lane] = D.chain[(size_t)r * kChainCap + 64 + lane];
    if (lane >= 1 && lane < d) sh.pm.mv[lane - 1] = (uint16_t)(Bn[CCZ_IDX(D, pj, D.cap)] & 0xffffu);
    if (lane == 0) sh.pm.mv[d - 1] = (uint16_t)(Bn[CCZ_IDX(D, sib, D.cap)] & 0xffffu);
    wave_sync();
    leaf_tail(D, b, lane, leaf_in, sh, d, P.m.turn, P.m.halfmove, P.m.chain_len, P.m.key, false);
}

__global__ __launch_bounds__(64) void k_scout(Dev D, uint16_t *leaf_in, int active)
{
    __shared__ SelectShared sh;
    scout_wave(D, leaf_in, active, blockIdx.x, threadIdx.x, sh);
}

// ------------------------------------------------------------------ K2: expand + backup
// prob: dense [B][2086] priors (compact == false) or compact [B][128] priors aligned with leaf_ids (compact == true)
template <bool COMPACT>
__device__ inline TopPatch expand_backup_phase(const Dev &D, int b, int lane, const float *prob, const float *value,
                                               const BoardMeta &m0, int half)
{
    TopPatch tp;
    tp.active = false; tp.root_expanded = false; tp.kid_expanded = false; tp.k = 0; tp.first_id = 0; tp.n0 = 0; tp.rootN = 0;
    tp.rootQ = 0.0f; tp.node1 = -1; tp.N1 = 0; tp.Q1 = 0.0f; tp.has1 = false;
    // everything that does not depend on another load is requested first
    const int status = D.leaf_status[b];
    const int d = D.path_len[b];
    const int k_leaf = D.leaf_k[b];
    const float v_net = value ? value[b] : D.vleaf[b]; // value == nullptr: the engine-owned leaf values of the planned evaluator boundary
    const int32_t *path = D.path + (size_t)b * D.maxd;
    const uint16_t *ids = D.leaf_ids + (size_t)b * kMaxLegal;
    const int id0 = ids[lane], id1 = ids[64 + lane];
    // compact priors need no second, id-dependent load round: they are requested here with everything else
    float cp0 = 0.0f, cp1 = 0.0f;
    if (COMPACT) { cp0 = prob[(size_t)b * kMaxLegal + lane]; cp1 = prob[(size_t)b * kMaxLegal + 64 + lane]; }
    const int pj = path[lane < D.maxd ? lane : 0];
    if (status == CCZ_LEAF_SKIP) return tp;
    CCZ_STAMP(D, b, lane, 8)
    BoardMeta *mp = D.meta + b;
    const size_t base = ((size_t)b * 2 + half) * (size_t)D.cap;
    NodeA *A = D.nodeA + base;
    uint32_t *Bn = D.nodeB + base;
    const int leaf = __builtin_amdgcn_readlane(pj, __builtin_amdgcn_readfirstlane(d < 64 ? d : 0));
    float v;
    if (status == CCZ_LEAF_EXPAND) {
        // Node.expand (mcts.py:31-39): one child per legal id, ascending id order
        const int k = k_leaf;
        const int n0 = m0.n_nodes;
        v = v_net;
        if (n0 + k > D.cap) {
            set_err(D, 1); // pool exhausted: the leaf stays unexpanded, the value is still backed up
        } else {
            const float *pr = prob + (size_t)b * kNMoves;
            if (lane < k) {
                A[CCZ_IDX(D, n0 + lane, D.cap)] = NodeA{0, 0.0f, COMPACT ? cp0 : pr[COMPACT ? 0 : CCZ_IDX(D, id0, kNMoves)], -1};
                Bn[CCZ_IDX(D, n0 + lane, D.cap)] = (uint32_t)id0;
            }
            if (64 + lane < k) {
                A[CCZ_IDX(D, n0 + 64 + lane, D.cap)] = NodeA{0, 0.0f, COMPACT ? cp1 : pr[COMPACT ? 0 : CCZ_IDX(D, id1, kNMoves)], -1};
                Bn[CCZ_IDX(D, n0 + 64 + lane, D.cap)] = (uint32_t)id1;
            }
            tp.root_expanded = d == 0;
            tp.kid_expanded = d == 1;
            tp.n0 = n0;
            tp.k = k;
            tp.first_id = __builtin_amdgcn_readlane(id0, 0);
            if (lane == 0) {
                const int leaf0 = (int)CCZ_IDX(D, d < 64 ? leaf : path[CCZ_IDX(D, d, D.maxd)], D.cap);
                A[leaf0].fc = n0;
                Bn[leaf0] = (Bn[leaf0] & 0xffffu) | ((uint32_t)k << 16);
                mp->n_nodes = n0 + k;
                BoardStats &st = D.stats[b];
                st.sum_children += (unsigned long long)k;
                st.expansions += 1;
                if (n0 + k > st.nodes_peak) st.nodes_peak = n0 + k;
            }
        }
    } else {
        v = status == CCZ_LEAF_DRAW ? 0.0f : -1.0f; // mcts.py:120-126
        if (lane == 0) D.stats[b].terminal += 1;
    }
    if (lane == 0) {
        D.stats[b].sims += 1;
        D.leaf_status[b] = CCZ_LEAF_SKIP; // consumed: a repeated call must not back the same leaf up twice
    }
    // Node.update_recursive(-leaf_value) (mcts.py:73-78,129): leaf gets -v, its parent +v, ...
    CCZ_STAMP(D, b, lane, 15)
    int myN = 0;
    float myQ = 0.0f;
    for (int j = lane; j <= d; j += 64) {
        const int node = (int)CCZ_IDX(D, j < 64 ? pj : path[CCZ_IDX(D, j, D.maxd)], D.cap);
        const float val = ((d - j) & 1) ? v : -v;
        const int n = A[node].N + 1;
        const float q = A[node].Q;
        // visits += 1 ; value += 1.0*(leaf_value - value)/visits   in float32 (mcts.py:68-71)
        float qn;
        if (D.flags & 4u) {
            // CCZ_FLAG_VALUE_F16: the same expression on a float16 value (reference CUDA path). Every operation is done
            // in float32 and rounded once to float16, which is what NumPy's half loops do (and equals IEEE binary16
            // arithmetic: 24 >= 2*11 + 2); `visits` is a weak Python int, i.e. converted to float16 as well.
            const float vh = (float)(_Float16)val, qh = (float)(_Float16)q;
            float dh = (float)(_Float16)(vh - qh);
            dh = (float)(_Float16)(dh / (float)(_Float16)(float)n);
            qn = (float)(_Float16)(qh + dh);
        } else {
            float delta = val - q;
            delta = delta / (float)n;
            qn = q + delta;
        }
        A[node].N = n;
        A[node].Q = qn;
        if (j < 64) { myN = n; myQ = qn; }
    }
    tp.active = true;
    tp.rootN = __builtin_amdgcn_readlane(myN, 0);
    tp.rootQ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myQ), 0));
    tp.has1 = d >= 1;
    tp.node1 = __builtin_amdgcn_readlane(pj, 1);
    tp.N1 = __builtin_amdgcn_readlane(myN, 1);
    tp.Q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myQ), 1));
    return tp;
}

template <bool COMPACT>
__global__ __launch_bounds__(64) void k_expand_backup(Dev D, const float *prob, const float *value)
{
    const BoardMeta m0 = D.meta[blockIdx.x];
    (void)expand_backup_phase<COMPACT>(D, blockIdx.x, threadIdx.x, prob, value, m0, *D.half);
}

// ------------------------------------------------------------------ fused step: expand+backup of the pending leaf, then the next select
// Saves one launch boundary and the re-read of the board's tree head per simulation. The phases
// touch the same nodes from different lanes, so a workgroup barrier (one wave) separates them.
template <bool COMPACT>
__global__ __launch_bounds__(64) void k_step(Dev D, const float *prob, const float *value, uint16_t *leaf_in)
{
    __shared__ SelectShared sh;
    const int b = blockIdx.x, lane = threadIdx.x;
    CCZ_STAMP(D, b, lane, 0)
    const Prefetch P = prefetch_board(D, b, lane); // root board, chain and meta: untouched by the expand phase
    const TopPatch tp = expand_backup_phase<COMPACT>(D, b, lane, prob, value, P.m, P.half);
    CCZ_STAMP(D, b, lane, 1)
    __threadfence_block();
    __syncthreads();
    CCZ_STAMP(D, b, lane, 2)
    select_phase(D, b, lane, leaf_in, sh, P, tp);
    CCZ_STAMP(D, b, lane, 9)
}

// What a cache entry is checked against besides its 64-bit key: the number of legal moves (8 bits) and a 24-bit hash of the
// legal-move LIST in the order the priors are stored in. The list is derived from the position, not from the key: a position that
// collides with another one on all 64 key bits (2^-40 per probe of an occupied slot, i.e. once in days at 2 x 10^5 probes a
// second) would also have to have the same legal moves in the same order to be served the other position's priors; a list that
// differs in any entry passes with probability 2^-24. Residual per probe: < 2^-64.
__device__ __forceinline__ uint32_t cache_tag(int id0, int id1, int k, int lane)
{
    uint64_t h = 0;
    if (lane < k) h ^= mix64(((uint64_t)(uint32_t)id0 << 8 | (uint32_t)lane) + 0x9E3779B97F4A7C15ull);
    if (64 + lane < k) h ^= mix64(((uint64_t)(uint32_t)id1 << 8 | (uint32_t)(64 + lane)) + 0x9E3779B97F4A7C15ull);
    h = wave_readlane64(wave_incl_xor64(h), 63);
    return ((uint32_t)(h >> 40) << 8) | (uint32_t)(k & 0xff);
}

// ------------------------------------------------------------------ compact evaluator boundary: logits -> priors of the legal moves
// exp(log_softmax(logits))[legal ids] (net.py:202-205) for every board in one pass: the wave reads its 2086
// logits once (coalesced), reduces max and sum on the DPP network, and writes only the <= 128 priors the
// expansion will use, aligned with leaf_ids. Replaces two full [B,2086] torch passes (log_softmax, exp) and
// turns the expansion's scattered 4-byte gather (one 64-B line per legal move) into one coalesced 512-B read.
// PLANNED: the evaluator ran on the compact rows of k_cache_plan; board b's logits sit in row row_of[b] of `logits`, its value in
// vcompact[row_of[b]]; cache hits already hold their priors and value (k_cache_probe); a fresh evaluation whose board won the
// slot's claim is stored in the cache.
template <typename T, bool PLANNED>
__global__ __launch_bounds__(64) void k_softmax_gather(Dev D, const T *logits, const float *vcompact)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) float row[kNMoves + 2];
    // everything the wave needs to know about its board is requested at once (one memory round trip instead of a chain of four:
    // status -> state -> row -> logits); a board that turns out to have nothing to do leaves after it
    const int status = D.leaf_status[b];
    int cst = 0, src = b;
    if (PLANNED) {
        cst = D.cstate[b];
        src = D.row_of[b];
    }
    const int k = D.leaf_k[b];
    const uint16_t *ids = D.leaf_ids + (size_t)b * kMaxLegal;
    const int id0 = ids[lane], id1 = ids[64 + lane];
    if (status != CCZ_LEAF_EXPAND || cst != 0) return; // (cst != 0: a hit -- prior128 / vleaf were filled by the probe)
    const T *srcrow = logits + (size_t)src * kNMoves;
    float mx = -__builtin_huge_valf();
    float sum = 0.0f;
    if constexpr (sizeof(T) == 2) {
        // fp16 logits: a row is 1043 dwords (4-byte aligned: 2086 is even), two logits per load -- half the loads, LDS writes and
        // loop trips of the element-wise form. A lane adds its exponentials in ascending element order (2j, 2j + 1, 2j + 128, ...).
        static_assert(kNMoves % 2 == 0, "dword rows");
        const uint32_t *src32 = (const uint32_t *)srcrow;
        constexpr int kIt = (kNMoves / 2 + 63) / 64;
        uint32_t wreg[kIt]; // the whole row in flight before the first use
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int j = lane + 64 * i;
            wreg[i] = j < kNMoves / 2 ? src32[j] : 0u;
        }
#pragma unroll
        for (int i = 0; i < kIt; ++i) {
            const int j = lane + 64 * i;
            if (j < kNMoves / 2) {
                const uint32_t w = wreg[i];
                const float x0 = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
                const float x1 = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
                *(float2 *)(row + 2 * j) = make_float2(x0, x1);
                mx = fmaxf(mx, fmaxf(x0, x1));
            }
        }
        mx = wave_max_f32(mx);
        for (int j = lane; j < kNMoves / 2; j += 64) {
            const float2 x = *(const float2 *)(row + 2 * j);
            sum += __expf(x.x - mx);
            sum += __expf(x.y - mx);
        }
    } else {
        for (int i = lane; i < kNMoves; i += 64) {
            const float x = (float)srcrow[i];
            row[i] = x;
            mx = fmaxf(mx, x);
        }
        mx = wave_max_f32(mx);
        for (int i = lane; i < kNMoves; i += 64) sum += __expf(row[i] - mx);
    }
    sum = wave_sum_f32(sum);
    wave_sync();
    float *out = D.prior128 + (size_t)b * kMaxLegal;
    const float p0 = lane < k ? __expf(row[id0] - mx) / sum : 0.0f;
    const float p1 = 64 + lane < k ? __expf(row[id1] - mx) / sum : 0.0f;
    if (PLANNED) {
        const float v = vcompact[src];
        if (D.cver[b]) {
            // CCZ_FLAG_CACHE_VERIFY: this leaf HIT the table (the probe left the cached priors / value in prior128 / vleaf) and was
            // sent through the evaluator all the same: the fresh numbers must be the cached ones, bit for bit
            const bool d0 = lane < k && __float_as_uint(out[lane]) != __float_as_uint(p0);
            const bool d1 = 64 + lane < k && __float_as_uint(out[64 + lane]) != __float_as_uint(p1);
            const bool dv = __float_as_uint(D.vleaf[b]) != __float_as_uint(v);
            const bool bad = __ballot(d0 || d1 || dv) != 0ull;
            if (lane == 0) {
                D.stats[b].cache_verified += 1u;
                if (bad) D.stats[b].cache_mismatch += 1u;
            }
        }
        if (lane < k) out[lane] = p0;
        if (64 + lane < k) out[64 + lane] = p1;
        const uint32_t slot = D.cslot[b];
        if (lane == 0) {
            D.vleaf[b] = v;
            D.claim[slot] = 0x7fffffff; // (every board that missed on the slot resets it: idempotent)
        }
        if (D.cins[b]) { // the slot's claim winner stores its evaluation (nobody reads the table before the next launch)
            CacheEntry *e = D.cache + slot;
            const uint32_t tag = cache_tag(id0, id1, k, lane);
            e->pri[lane] = p0;
            e->pri[64 + lane] = p1;
            if (lane == 0) { e->v = v; e->k = tag; e->key = D.leaf_key[b]; D.stats[b].cache_stores += 1u; }
        }
    } else {
        if (lane < k) out[lane] = p0;
        if (64 + lane < k) out[64 + lane] = p1;
    }
}

// ------------------------------------------------------------------ evaluation cache: probe, plan
__device__ __forceinline__ uint32_t cache_slot(uint64_t key, uint32_t mask) { return (uint32_t)(key ^ (key >> 29)) & mask; }

// One wave per board: look the pending leaf up. Hit: its priors and value go straight to prior128 / vleaf. Miss: the board
// bids for the slot (lowest board index wins: deterministic) -- the winner's evaluation will be stored there, and boards that
// missed with the SAME key in this step share the winner's evaluator row (k_cache_plan).
__device__ inline int cache_probe_wave(const Dev &D, int b, int lane) // returns the board's plan state (wave-uniform): 0 miss, 1 hit, 2 nothing to evaluate
{
    const int status = D.leaf_status[b];
    const uint64_t key = D.leaf_key[b]; // (requested together with the status: one round trip less in front of the table access)
    if (status != CCZ_LEAF_EXPAND) {
        if (lane == 0) D.cstate[b] = 2;
        return 2;
    }
    const uint32_t slot = cache_slot(key, D.cache_mask);
    const CacheEntry *e = D.cache + slot;
    const uint64_t ekey = e->key;
    const uint32_t ek = e->k;
    const float ev = e->v, q0 = e->pri[lane], q1 = e->pri[64 + lane];
    const uint16_t *ids = D.leaf_ids + (size_t)b * kMaxLegal;
    const uint32_t tag = cache_tag(ids[lane], ids[64 + lane], D.leaf_k[b], lane);
    const bool hit = ekey == key && ek == tag;
    if (hit) {
        float *out = D.prior128 + (size_t)b * kMaxLegal;
        out[lane] = q0;
        out[64 + lane] = q1;
    }
    // CCZ_FLAG_CACHE_VERIFY: one hit in 128 (chosen by the key, the board and the board's probe count) is ALSO planned as an evaluator row;
    // k_softmax_gather compares its fresh priors / value with what the table just returned. It does not bid for the slot.
    BoardStats &st = D.stats[b];
    const bool verify = hit && (D.flags & 8u) && (mix64(key + 0x632BE59BD9B4E019ull * (uint64_t)st.cache_probes + 0xD1342543DE82EF95ull * (uint64_t)b) & 127ull) == 0ull;
    if (lane == 0) {
        D.cslot[b] = slot;
        D.ctag[b] = tag;
        D.cstate[b] = hit && !verify ? 1 : 0;
        D.cver[b] = verify ? 1 : 0;
        if (hit) D.vleaf[b] = ev;
        else atomicMin(D.claim + slot, b);
        st.cache_probes += 1u;
        if (hit && !verify) st.cache_hits += 1u;
    }
    return hit && !verify ? 1 : 0;
}

__global__ __launch_bounds__(64) void k_cache_probe(Dev D) { (void)cache_probe_wave(D, blockIdx.x, threadIdx.x); }

// One workgroup: representatives and the compaction plan. A miss whose slot was won by a board with the same key uses that
// board's row; every other miss is its own representative (a different key on the same slot is evaluated but not stored).
// miss_rows[0 .. n_miss) = the representatives in ascending board order: the rows the evaluator computes.
__global__ __launch_bounds__(1024) void k_cache_plan(Dev D, int32_t *miss_rows, int32_t *n_miss)
{
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    // 4096 boards per pass: a thread looks after boards b0 + tid + 1024 i, i = 0..3. The three dependent load rounds (state / slot /
    // key -> claim -> the claimant's state and key) are issued for all four boards before any is used, so a pass costs three memory
    // round trips instead of twelve (this kernel is one workgroup: nothing else hides its latency)
    for (int b0 = 0; b0 < D.B; b0 += 4096) {
        int st[4], w[4], rep[4];
        uint32_t slot[4], tag[4];
        uint64_t key[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + i * 1024 + tid;
            st[i] = 3;
            slot[i] = 0;
            tag[i] = 0;
            key[i] = 0;
            if (b < D.B) { st[i] = D.cstate[b]; slot[i] = D.cslot[b]; key[i] = D.leaf_key[b]; tag[i] = D.ctag[b]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = st[i] == 0 ? D.claim[slot[i]] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + i * 1024 + tid;
            rep[i] = b;
            // (w < b: a bidder's claim winner is never above it; a CACHE_VERIFY board did not bid, and must not take the row of a
            // higher board -- possibly of a later pass, not assigned yet -- on a full 64-bit key collision)
            if (st[i] == 0 && w[i] >= 0 && w[i] < b) {
                // the same position = the same 64-bit key AND the same legal-move list (count + 24-bit hash of the list in prior order):
                // what a table hit is checked against (k_cache_probe), now also between two leaves of one step (round 6)
                const bool same = D.cstate[w[i]] == 0 && D.leaf_key[w[i]] == key[i] && D.ctag[w[i]] == tag[i];
                rep[i] = same ? w[i] : b;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + i * 1024 + tid;
            int isrep = 0;
            if (st[i] == 0) {
                D.crep[b] = rep[i];
                D.cins[b] = (uint8_t)(w[i] == b);
                isrep = rep[i] == b;
                if (!isrep) D.stats[b].cache_shared += 1u;
            }
            // exclusive position of every representative: ballot inside the wave, 16 wave totals through LDS
            const uint64_t m = __ballot(isrep);
            const int in_wave = __popcll(m & lanemask_lt(lane)), wave_total = __popcll(m);
            if (lane == 0) s_wave[wv] = wave_total;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = s_wave[j];
                before += j < wv ? t : 0;
                total += t;
            }
            const int base = s_base;
            if (isrep) {
                const int pos = base + before + in_wave;
                D.row_of[b] = pos;
                miss_rows[pos] = b;
            }
            __syncthreads();
            if (tid == 0) s_base = base + total;
            __syncthreads();
        }
        // boards that use another board's row: the representative is the slot's claim winner, i.e. a LOWER board index -- its row
        // was assigned above, in this pass or an earlier one (straight from the registers of this pass: one load round)
        __threadfence_block();
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = b0 + i * 1024 + tid;
            if (st[i] == 0 && rep[i] != b) // (read past this CU's L1: the row was written a moment ago by another wave)
                D.row_of[b] = __hip_atomic_load(D.row_of + rep[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) *n_miss = s_base;
}

// The plan of a step with scouts (ccz_eval_plan_scouted): the evaluator runs -- on ALL slots, row b = slot b, a fixed small batch --
// only when a board that is really searched misses; the scouts of a board that hit (or whose leaf needs no evaluation) are dropped
// for this step: their next siblings are asked for again when the board next misses. No row sharing here (a duplicate position
// costs a row of a batch that is latency-bound anyway). state_out[r] = cstate of real board r (0 = its leaf needs the evaluator).
__device__ inline void plan_scouted_block(const Dev &D, int active, int32_t *miss_rows, int32_t *n_miss, int32_t *state_out, int tid, int nt, int &s_any)
{
    if (tid == 0) s_any = 0;
    __syncthreads();
    for (int b = tid; b < D.B; b += nt) {
        const int st = D.cstate[b];
        if (b < active) {
            if (st == 0) atomicOr(&s_any, 1);
            if (state_out) state_out[b] = st;
        }
    }
    __syncthreads();
    for (int b = tid; b < D.B; b += nt) {
        const int st = D.cstate[b];
        const uint32_t slot = D.cslot[b];
        const int r = b < active ? b : (b - active) % active;
        const bool keep = st == 0 && (b < active || D.cstate[r] == 0);
        if (st == 0) {
            const int w = D.claim[slot];
            if (!keep) {
                if (w == b) D.claim[slot] = 0x7fffffff; // a dropped scout that won its slot gives it back (nobody stores there this step)
            } else {
                D.crep[b] = b;
                D.row_of[b] = b;
                D.cins[b] = (uint8_t)(w == b);
            }
        }
        miss_rows[b] = b;
    }
    __syncthreads();
    // (second pass: cstate of the scouts is rewritten only after every thread has read what it needed of it)
    for (int b = active + tid; b < D.B; b += nt) {
        const int r = (b - active) % active;
        if (D.cstate[b] == 0 && D.cstate[r] != 0) D.cstate[b] = 2;
    }
    if (tid == 0) *n_miss = s_any ? D.B : 0;
}

__global__ __launch_bounds__(256) void k_cache_plan_scouted(Dev D, int active, int32_t *miss_rows, int32_t *n_miss, int32_t *state_out)
{
    __shared__ int s_any;
    plan_scouted_block(D, active, miss_rows, n_miss, state_out, threadIdx.x, 256, s_any);
}

// scout + probe + plan of an engine of up to 16 slots as ONE workgroup (one wave per slot): what a scouted simulation launches behind
// ccz_step_compact -- two launches per simulation instead of four (the step is a chain of small launches: every one costs ~5 us)
constexpr int kScoutFusedMax = 16;
__global__ __launch_bounds__(1024) void k_scout_probe_plan(Dev D, uint16_t *leaf_in, int active, int32_t *miss_rows, int32_t *n_miss, int32_t *state_out)
{
    __shared__ SelectShared sh[kScoutFusedMax];
    __shared__ int s_any;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (w >= active && w < D.B) scout_wave(D, leaf_in, active, w - active, lane, sh[w]);
    __threadfence_block();      // the scout's leaf slots (global memory, written by this wave) are read back by its own probe
    if (w < D.B) (void)cache_probe_wave(D, w, lane);
    __threadfence();            // cstate / cslot / claim of every slot are read by other waves of this workgroup in the plan
    __syncthreads();
    plan_scouted_block(D, active, miss_rows, n_miss, state_out, tid, (int)blockDim.x, s_any);
}

// Simulations of a scouted engine WITHOUT the host in between (ccz_scouted_run): what the host loop launches per simulation -- the
// fused step (expand + backup of the pending leaf, next selection) and scout + probe + plan -- repeated by ONE workgroup (one wave per
// slot, <= 16 slots) for as long as every searched board finds its next leaf in the table. A simulation that hits costs the
// latency chains of its phases instead of two launches, a graph replay and a stream synchronisation (~46 us,
// profiles/r06_single_board.json); the host comes back only when the evaluator has to run, when `budget` simulations are done (a
// caller that reports progress) or when the move's last simulation -- which has no next selection -- is backed up.
//   * the scouts of a board that hit are dropped by the plan (plan_scouted_block), so while the searched boards hit, the scout waves
//     do nothing at all: a searched board's wave expands, backs up, selects and probes ITS leaf; the scouts are handed their leaves
//     only once, for the step that leaves the loop with a miss -- and what that step's plan leaves behind is what k_scout_probe_plan
//     would have left (a dropped scout gives back the slot it claimed: the same table state as never having claimed it).
//   * run[0] = budget (>= 1), run[1] = simulations left in this move including the pending one (>= 1); run[2] <- simulations done
//     here, run[3] <- 1 if a searched board's leaf needs the evaluator now.
// Same phases, same order, same device functions as the separate launches; all hand-offs between waves stay inside the workgroup
// (one CU, one L1): workgroup-scope fences and barriers, as in k_step and k_scout_probe_plan.
template <int MAXW> // 12: engines of up to 12 slots (168 registers per lane: no spills; the default 1 + 10 scouts), 16: up to 16
__global__ __launch_bounds__(64 * MAXW) void k_scouted_run(Dev D, uint16_t *leaf_in, int active, int32_t *miss_rows, int32_t *n_miss, int32_t *state_out,
                                                        int32_t *run)
{
    __shared__ SelectShared sh[MAXW];
    __shared__ int s_any, s_state[MAXW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool real = w < active;
    int budget = __hip_atomic_load(run + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int left = __hip_atomic_load(run + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    budget = budget < 1 ? 1 : (budget > (1 << 20) ? (1 << 20) : budget); // whatever the caller left in the run block, the loop ends
    int done = 0, need = 0;
    for (;;) {
        Prefetch P;
        TopPatch tp;
        if (real) {
            P = prefetch_board(D, w, lane);
            tp = expand_backup_phase<true>(D, w, lane, D.prior128, nullptr, P.m, P.half);
            __threadfence_block();      // (k_step: the phases touch the same nodes from different lanes of the wave)
            __builtin_amdgcn_wave_barrier();
        }
        ++done;
        if (--left <= 0) break;         // the move's last simulation: nothing is selected behind it (ccz_expand_backup_compact)
        if (real) {
            select_phase(D, w, lane, leaf_in, sh[w], P, tp);
            __threadfence_block();      // the leaf (global memory, written by this wave) is read back by its own probe
            const int st = cache_probe_wave(D, w, lane);
            if (lane == 0) s_state[w] = st;
        }
        __threadfence_block();          // the searched boards' leaves, paths, statuses and probe results: read by every wave from here on
        __syncthreads();
        need = 0;
        for (int r = 0; r < active; ++r) need |= s_state[r] == 0 ? 1 : 0;
        if (need || done >= budget) break;
        __syncthreads();                // (s_state is rewritten by the next pass)
    }
    if (left > 0) {
        // the plan of the step that leaves the loop. A miss: the scouts get their leaves and probe, then the plan of all slots; no
        // miss (budget used): the searched boards' states, no evaluator call -- plan_scouted_block with every scout dropped
        if (need) {
            if (!real) {
                scout_wave(D, leaf_in, active, w - active, lane, sh[w]);
                __threadfence_block();
                (void)cache_probe_wave(D, w, lane);
            }
            __threadfence_block();      // cstate / cslot / claim of every slot are read by other waves of this workgroup in the plan
            __syncthreads();
            plan_scouted_block(D, active, miss_rows, n_miss, state_out, tid, (int)blockDim.x, s_any);
        } else {
            if (tid < active) state_out[tid] = s_state[tid];
            if (tid == 0) *n_miss = 0;
        }
    }
    if (tid == 0) {
        run[2] = done;
        run[3] = need;
    }
}

// ------------------------------------------------------------------ pi from root visits (mcts.py:162-166)
// s_vis[k] -> s_pi[k] ; softmax(1/temp * log(N + 1e-10)) with the deterministic log/exp; the
// normalising sum is accumulated sequentially (index order) to match the CPU twin bit for bit.
__device__ inline void root_pi(const int32_t *s_vis, double *s_pi, int k, double temp, int lane)
{
    const double it = 1.0 / temp;
    double mx = -__builtin_huge_val();
    for (int i = lane; i < k; i += 64) {
        const double x = it * det_log((double)s_vis[i] + 1e-10);
        s_pi[i] = x;
        if (x > mx) mx = x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(mx, o); if (t > mx) mx = t; }
    for (int i = lane; i < k; i += 64) s_pi[i] = det_exp(s_pi[i] - mx);
    wave_sync();
    double sum = 0.0;
    if (lane == 0) for (int i = 0; i < k; ++i) sum += s_pi[i];
    sum = __shfl(sum, 0);
    wave_sync();
    for (int i = lane; i < k; i += 64) s_pi[i] = s_pi[i] / sum;
    wave_sync();
}

__device__ __forceinline__ double board_temp(const Dev &D, const BoardMeta &m, const double *temps, int b)
{
    if (temps) return temps[b];
    // game.py:157-159: move_count = ply+1 ; temp if move_count <= 30 else max(0.1, temp*0.5)
    const double half = D.temp * 0.5;
    return (m.ply + 1) <= 30 ? D.temp : (half > 0.1 ? half : 0.1);
}

__global__ __launch_bounds__(64) void k_root_children(Dev D, int32_t *k_out, uint16_t *acts, int32_t *visits,
                                                        float *q, float *prior, int32_t *root_visits,
                                                        const double *temps, double *pi_out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ int32_t s_vis[kMaxLegal];
    __shared__ double s_pi[kMaxLegal];
    const BoardMeta m = D.meta[b];
    const size_t base = ((size_t)b * 2 + *D.half) * (size_t)D.cap;
    const NodeA *A = D.nodeA + base;
    const uint32_t *Bn = D.nodeB + base;
    const NodeA root = A[0];
    int k = (int)(Bn[0] >> 16);
    if (k > kMaxLegal) k = kMaxLegal;
    if (lane == 0) {
        if (k_out) k_out[b] = k;
        if (root_visits) root_visits[b] = root.N;
    }
    for (int i = lane; i < kMaxLegal; i += 64) {
        NodeA c = NodeA{0, 0.0f, 0.0f, -1};
        uint32_t w = 0;
        if (i < k) { c = A[root.fc + i]; w = Bn[root.fc + i]; }
        s_vis[i] = c.N;
        const size_t o = (size_t)b * kMaxLegal + i;
        if (acts) acts[o] = (uint16_t)(w & 0xffffu);
        if (visits) visits[o] = c.N;
        if (q) q[o] = c.Q;
        if (prior) prior[o] = c.P;
    }
    __syncthreads();
    if (pi_out) {
        if (k > 0) root_pi(s_vis, s_pi, k, board_temp(D, m, temps, b), lane);
        for (int i = lane; i < kMaxLegal; i += 64) pi_out[(size_t)b * kMaxLegal + i] = i < k ? s_pi[i] : 0.0;
    }
}

// ------------------------------------------------------------------ K3: once per move
__global__ __launch_bounds__(64) void k_finish_move(Dev D, const int32_t *forced, const double *temps,
                                                      int32_t *moves_out, int keep_tree)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ __align__(16) uint8_t s_sq[96];
    __shared__ uint64_t s_chain[kChainCap];
    __shared__ GenScratch S;
    __shared__ int32_t s_vis[kMaxLegal];
    __shared__ uint16_t s_act[kMaxLegal];
    __shared__ double s_pi[kMaxLegal];
    __shared__ double s_g[kMaxLegal];
    __shared__ int32_t s_src[64], s_dst[64], s_cnt[64];
    __shared__ int s_choice;

    BoardMeta m = D.meta[b];
    if (moves_out && lane == 0) moves_out[b] = -1;
    if (m.over) return;
    const int oh = *D.half, nh = oh ^ 1; // every board moves to the other pool half (the host flips the word afterwards)
    const size_t baseOld = ((size_t)b * 2 + oh) * (size_t)D.cap;
    {   // whatever happens below, the new half holds a valid (empty) tree for this board
        const size_t bn = ((size_t)b * 2 + nh) * (size_t)D.cap;
        if (lane == 0) { D.nodeA[bn] = NodeA{0, 0.0f, 1.0f, -1}; D.nodeB[bn] = 0u; }
    }
    const NodeA *A = D.nodeA + baseOld;
    const uint32_t *Bn = D.nodeB + baseOld;
    const NodeA root = A[0];
    int k = (int)(Bn[0] >> 16);
    const int want = forced ? forced[b] : -1;
    if (k > kMaxLegal) { if (lane == 0) set_err(D, 16); return; }
    if (k == 0 && want < 0) { if (lane == 0) set_err(D, 16); return; } // nothing searched, nothing to sample from
    if (want >= kNMoves) { if (lane == 0) set_err(D, 16); return; }

    load_board(s_sq, D.root_sq + (size_t)b * 96, lane);
    for (int i = lane; i < k; i += 64) {
        s_vis[i] = A[CCZ_IDX(D, root.fc + i, D.cap)].N;
        s_act[i] = (uint16_t)(Bn[CCZ_IDX(D, root.fc + i, D.cap)] & 0xffffu);
    }
    __syncthreads();

    // ---- pi (mcts.py:162-166) and the training record (game.py:195-198)
    if (k > 0) root_pi(s_vis, s_pi, k, board_temp(D, m, temps, b), lane);
    if (m.ply >= D.max_plies || m.pi_used + (uint32_t)k > (uint32_t)D.pi_cap) {
        // documented cap: the game is adjudicated a draw, its records so far stay valid
        if (lane == 0) {
            if (m.ply < D.max_plies) set_err(D, 8);
            else if (D.flags & CCZ_FLAG_STRICT) set_err(D, CCZ_ERR_TRUNCATED); // the reference's game has no ply cap (game.py:155)
            m.over = 1; m.winner = -1;
            D.meta[b] = m;
            D.stats[b].truncated += 1;
            D.stats[b].games += 1;
        }
        return;
    }
    {
        const size_t r = (size_t)b * D.max_plies + CCZ_IDX(D, m.ply, D.max_plies);
        if (lane < 24) ((uint32_t *)(D.rec_sq + r * 96))[lane] = ((const uint32_t *)s_sq)[lane];
        if (lane == 0) { D.rec_turn[r] = m.turn; D.rec_k[r] = (uint8_t)k; D.rec_off[r] = m.pi_used; }
        const size_t po = (size_t)b * D.pi_cap + m.pi_used;
        for (int i = lane; i < k; i += 64) {
            const size_t o = (size_t)b * D.pi_cap + CCZ_IDX(D, m.pi_used + (uint32_t)i, D.pi_cap);
            D.rec_ids[o] = s_act[i];
            D.rec_pi[o] = (float)s_pi[i];
        }
        (void)po;
    }

    // ---- move choice (mcts.py:216-229)
    if (want >= 0) {
        // a forced move that is not a child of the root (root unexpanded, or an opponent's reply the
        // search never saw) gives a fresh root, as MCTS.update_with_move does (mcts.py:176-178)
        int found = -1;
        for (int i0 = 0; i0 < k; i0 += 64) {
            const int i = i0 + lane;
            const uint64_t hit = __ballot(i < k && s_act[i] == want);
            if (hit && found < 0) found = i0 + __ffsll((long long)hit) - 1;
        }
        if (found < 0 && s_sq[c_tab.from[want]] == 0) { if (lane == 0) set_err(D, 16); return; }
        if (lane == 0) s_choice = found;
    } else {
        // move ~ Categorical((1-EPS)*pi + EPS*Dirichlet(ALPHA)) on the board's Philox stream
        const uint64_t gid = D.board_id_base + (uint64_t)b;
        for (int i = lane; i < k; i += 64) s_g[i] = det_gamma(D.seed, gid, m.move_counter, (uint32_t)i, D.alpha);
        __syncthreads();
        if (lane == 0) {
            double gs = 0.0, acc = 0.0, ua, ub;
            for (int i = 0; i < k; ++i) gs += s_g[i];
            for (int i = 0; i < k; ++i) {
                const double dir = gs > 0.0 ? s_g[i] / gs : s_pi[i];
                acc += (1.0 - D.eps) * s_pi[i] + D.eps * dir;
                s_g[i] = acc; // cdf
            }
            uniform2(D.seed, gid, m.move_counter, 0xfffu, 0, ua, ub);
            int idx = 0;
            for (int i = 0; i < k; ++i) if (s_g[i] / acc <= ua) idx = i + 1; // searchsorted(side="right")
            s_choice = idx < k ? idx : k - 1;
        }
    }
    __syncthreads();
    const int ci = s_choice;
    const int mv = ci >= 0 ? (int)s_act[ci] : want;
    if (ci < 0) keep_tree = 0;
    if (moves_out && lane == 0) moves_out[b] = mv;

    // ---- MCTS.update_with_move (mcts.py:168-178): re-root on the chosen child, subtree copied
    // breadth-first into the other pool half (children of a node stay contiguous)
    const size_t baseNew = ((size_t)b * 2 + nh) * (size_t)D.cap;
    NodeA *NA = D.nodeA + baseNew;
    uint32_t *NB = D.nodeB + baseNew;
    int n_new = 1;
    if (keep_tree) {
        if (lane == 0) { NA[0] = A[CCZ_IDX(D, root.fc + ci, D.cap)]; NB[0] = Bn[CCZ_IDX(D, root.fc + ci, D.cap)]; }
        __syncthreads();
        int head = 0, pruned = 0;
        // The kept subtree may use the pool up to `budget`, leaving room for a whole move of new expansions.
        // Breadth-first order copies the tree level by level, so when the budget runs out it is the DEEPEST nodes
        // that lose their children (they become unexpanded leaves again, keeping N, Q, P): bounded memory with a
        // graceful loss at the bottom of very concentrated trees instead of failed expansions later (counted in
        // stats.pruned_subtrees; the reference's Python tree is unbounded, DESIGN.md "caps").
        const int budget = D.cap - D.reserve;
        while (head < n_new) {
            const int i = head + lane;
            const bool valid = i < n_new;
            NodeA rec = NodeA{0, 0.0f, 0.0f, -1};
            int nc = 0;
            if (valid) { rec = NA[CCZ_IDX(D, i, D.cap)]; nc = (int)(NB[CCZ_IDX(D, i, D.cap)] >> 16); }
            const int incl = wave_incl_scan(nc, lane);
            const bool keep = nc > 0 && n_new + incl <= budget;   // a prefix of the lanes: incl is non-decreasing
            const bool drop = nc > 0 && !keep;
            const uint64_t km = __ballot(keep);
            const int total = km ? __builtin_amdgcn_readlane(incl, __builtin_amdgcn_readfirstlane(63 - __builtin_clzll(km))) : 0;
            const int dst = n_new + incl - nc;
            s_cnt[lane] = keep ? nc : 0;
            if (keep) { s_src[lane] = rec.fc; s_dst[lane] = dst; NA[i].fc = dst; }
            if (drop) { NA[i].fc = -1; NB[i] = NB[i] & 0xffffu; }
            pruned += __popcll(__ballot(drop));
            __syncthreads();
            uint64_t todo = km;
            while (todo) {
                const int L = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int src = s_src[L], dd = s_dst[L], cn = s_cnt[L];
                for (int j = lane; j < cn; j += 64) {
                    NA[CCZ_IDX(D, dd + j, D.cap)] = A[CCZ_IDX(D, src + j, D.cap)];
                    NB[CCZ_IDX(D, dd + j, D.cap)] = Bn[CCZ_IDX(D, src + j, D.cap)];
                }
            }
            const int nbatch = n_new - head < 64 ? n_new - head : 64;
            head += nbatch;
            n_new += total;
            __syncthreads();
        }
        if (pruned && lane == 0) {
            D.stats[b].pruned += (unsigned long long)pruned;
            if (D.flags & CCZ_FLAG_STRICT) set_err(D, CCZ_ERR_PRUNED); // the reference's tree is unbounded (mcts.py:31-39)
        }
    }
    if (!keep_tree || n_new == 0) {
        if (lane == 0) { NA[0] = NodeA{0, 0.0f, 1.0f, -1}; NB[0] = 0u; }
        n_new = 1;
    }

    // ---- board.push(move) on the root (game.py:201) and its history
    const int from = c_tab.from[mv], to = c_tab.to[mv];
    const int pc = s_sq[from], cap = s_sq[to];
    __syncthreads();
    if (lane == 0) { s_sq[to] = (uint8_t)pc; s_sq[from] = 0; }
    uint64_t key = m.key ^ zob(pc, from) ^ zob(pc, to) ^ kTurnKey;
    if (cap) key ^= zob(cap, to);
    const int turn = m.turn ^ 1;
    const bool zeroing = cap || ((D.rule_flags & 2u) && (pc & 7) == PAWN); // CCZ_RULE_PAWN_MOVE_RESETS_CLOCK
    int halfmove = zeroing ? 0 : m.halfmove + 1;
    int chain_len = zeroing ? 0 : m.chain_len;
    if (chain_len >= kChainCap) { chain_len = kChainCap - 1; set_err(D, 64); }
    for (int i = lane; i < chain_len; i += 64) s_chain[i] = D.chain[(size_t)b * kChainCap + i];
    if (lane == 0) { s_chain[CCZ_IDX(D, chain_len, kChainCap)] = key; D.chain[(size_t)b * kChainCap + CCZ_IDX(D, chain_len, kChainCap)] = key; }
    ++chain_len;
    __syncthreads();
    if (lane < 24) ((uint32_t *)(D.root_sq + (size_t)b * 96))[lane] = ((const uint32_t *)s_sq)[lane];

    // ---- game end (game.py:208-219): is_game_over() or is_tie(); winner from outcome()
    bool overflow;
    const LeafEval L = eval_position(s_sq, turn, halfmove, key, s_chain, chain_len, S, nullptr, lane, overflow);
    if (overflow) set_err(D, 4);
    // one bit per chain position: the side to move stands in check there (the move that led to it gave check)
    uint64_t chk0 = zeroing ? 0ull : D.chain_chk[(size_t)b * 2], chk1 = zeroing ? 0ull : D.chain_chk[(size_t)b * 2 + 1];
    {
        const bool in_check = L.ksq >= 0 && king_attacked(s_sq, S, L.ksq, -1, -1, 0, turn);
        const int ci = chain_len - 1;
        if (in_check) { if (ci < 64) chk0 |= 1ull << ci; else chk1 |= 1ull << (ci - 64); }
    }
    // CCZ_RULE_PERPETUAL_CHECK: the game ends by fourfold repetition; inside the repetition window --
    // the positions after the earliest occurrence of the repeated position -- a side whose EVERY move gave check while the
    // other side's did not loses. Only outcome().winner changes (game.py:210-216): in the search the same leaf is
    // "end and is_tie" -> 0.0 either way (mcts.py:120-122), so visit counts do not depend on this flag.
    int perpetual_winner = -1;
    // (the host view and the reference order of checks, game.py:208-214: insufficient material, then the sixty-move rule, then the
    // repetition -- a game that the sixty-move rule ends at the same ply is a DRAW, whatever the repetition window holds)
    if ((D.rule_flags & 1u) && L.n_legal > 0 && !L.insufficient && !(halfmove >= 120) && L.rep >= 4) {
        const int last = chain_len - 1;
        bool miss_mover = false, miss_other = false, any_other = false; // mover = the side that just moved (turn ^ 1)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = h * 64 + lane;
            const bool in = i > L.first_occ && i <= last;
            const bool bit = ((h ? chk1 : chk0) >> lane) & 1ull;
            const bool by_mover = ((last - i) & 1) == 0;
            miss_mover |= __ballot(in && by_mover && !bit) != 0ull;
            miss_other |= __ballot(in && !by_mover && !bit) != 0ull;
            any_other |= __ballot(in && !by_mover) != 0ull;
        }
        const bool mover_all = !miss_mover, other_all = any_other && !miss_other;
        if (mover_all && !other_all) perpetual_winner = turn;          // the side that kept checking loses
        else if (other_all && !mover_all) perpetual_winner = turn ^ 1;
    }
    if (lane == 0) {
        D.chain_chk[(size_t)b * 2] = chk0;
        D.chain_chk[(size_t)b * 2 + 1] = chk1;
        m.key = key;
        m.halfmove = halfmove;
        m.chain_len = chain_len;
        m.ply += 1;
        m.move_counter += 1;
        m.n_nodes = n_new;
        m.turn = (uint8_t)turn;
        m.half = (uint8_t)nh;
        m.pi_used += (uint32_t)k;
        BoardStats &st = D.stats[b];
        st.moves += 1;
        if (L.status != CCZ_LEAF_EXPAND) {
            m.over = 1;
            m.winner = L.n_legal == 0 ? (int8_t)(turn ^ 1) : (int8_t)perpetual_winner; // no legal move: side to move loses
            st.games += 1;
        }
        D.meta[b] = m;
        D.leaf_status[b] = CCZ_LEAF_SKIP;
    }
}

// ------------------------------------------------------------------ harvest: finished games -> training rows
// rows of board b start at row_base[b] (< 0: board not harvested). Per game: T samples then, unless
// CCZ_FLAG_NO_MIRROR, their T mirror images (collect.py:112-131: data + data_flip). Grid = (boards, kHarvestSlices):
// only a few dozen games end per move, so the plies of a game are spread over kHarvestSlices blocks (one 256-thread
// block per game left the chip nearly empty: 1.8 ms for 28 k rows; the blocks of boards that are not harvested exit at once).
constexpr int kHarvestSlices = 32;
__global__ __launch_bounds__(256) void k_harvest(Dev D, const long long *row_base, uint16_t *states, float *pi, float *z)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long rb = row_base[b];
    if (rb < 0) return;
    const BoardMeta m = D.meta[b];
    const int T = m.ply;
    const bool quirks = (D.flags & 1u) != 0, mirror = (D.flags & 2u) == 0;
    const uint8_t *rsq = D.rec_sq + (size_t)b * D.max_plies * 96;
    for (int t = blockIdx.y; t < T; t += gridDim.y) {
        const size_t r = (size_t)b * D.max_plies + t;
        // game.py:23-44: index i of the 8-deep history holds the position i plies back (start position
        // before that); reference quirk: every sample aliases the history at the LAST recorded ply
        const int te = quirks ? T - 1 : t;
        const int turn_plane = quirks ? 1 : D.rec_turn[r]; // collect.py:78 reads a board that never advances
        for (int pass = 0; pass < (mirror ? 2 : 1); ++pass) {
            const long long row = rb + (pass ? T : 0) + t;
            uint32_t *srow = (uint32_t *)(states + (size_t)row * 10710);
            for (int i = tid; i < 5355; i += 256) {
                uint32_t v = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int e = 2 * i + h;
                    const int g = e / 630, w = e - g * 630;
                    bool on;
                    if (g == 16) on = turn_plane != 0;
                    else {
                        const int ch = w / 90, s = w - 90 * ch;
                        const int ss = pass ? (s - s % 9) + (8 - s % 9) : s; // np.flip(axis=2): file mirror
                        int tp = te - (g & 7);
                        if (tp < 0) tp = 0;
                        on = rsq[(size_t)CCZ_IDX(D, tp, D.max_plies) * 96 + CCZ_IDX(D, ss, 90)] == type_in_plane(D, ch) + (g >= 8 ? 8 : 0);
                    }
                    if (on) v |= (uint32_t)kHalfOne << (16 * h);
                }
                srow[i] = v;
            }
            float *prow = pi + (size_t)row * kNMoves;
            for (int i = tid; i < kNMoves; i += 256) prow[i] = 0.0f;
            __syncthreads();
            const int k = D.rec_k[r];
            const size_t po = (size_t)b * D.pi_cap + D.rec_off[r];
            for (int i = tid; i < k; i += 256) {
                const int id = (int)CCZ_IDX(D, D.rec_ids[(size_t)b * D.pi_cap + CCZ_IDX(D, D.rec_off[r] + (uint32_t)i, D.pi_cap)], kNMoves);
                prow[pass ? c_tab.flip[id] : id] = D.rec_pi[po + i]; // mcts_prob[flip_map]
            }
            if (tid == 0) // game.py:213-219
                z[row] = m.winner < 0 ? 0.0f : (D.rec_turn[r] == (uint8_t)m.winner ? 1.0f : -1.0f);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ compact game records (the multi-GPU wire format)
// One fixed-size record per PLY of a finished game (include/cczero.h: CCZ_REC_*): position before the move, 16-byte
// header, sparse pi (ids + float32, zero-padded to 128). The plies of a game are contiguous and in order, so a record
// finds its game's first record at (index - t) and the two dense rows it stands for (the sample and its mirror image)
// at 2 * first + t and 2 * first + T + t: expansion needs no prefix sum and no engine state. 880 B per ply against
// 2 x 29,768 B of dense rows: what the all-gather moves (k_expand_records rebuilds the rows on the receiving side).
constexpr int kRecBytes = 880, kRecHdr = 96, kRecIds = 112, kRecPi = 368;
struct __align__(4) PlyHeader {
    uint16_t t, T;      // ply index inside its game, plies of the game
    int8_t winner;      // 1 RED, 0 BLACK, -1 draw
    uint8_t turn, k, flags;
    uint32_t board_id;  // global board id (low 32 bits)
    uint32_t game_no;
};
static_assert(sizeof(PlyHeader) == 16, "ply header is 16 bytes");

// rec_base[b]: index of the first record of board b's game in `out` (< 0: board not harvested)
__global__ __launch_bounds__(256) void k_harvest_records(Dev D, const long long *rec_base, uint8_t *out)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    const long long rb = rec_base[b];
    if (rb < 0) return;
    const BoardMeta m = D.meta[b];
    const int T = m.ply;
    for (int t = blockIdx.y; t < T; t += gridDim.y) {
        const size_t r = (size_t)b * D.max_plies + t;
        uint32_t *rec = (uint32_t *)(out + (size_t)(rb + t) * kRecBytes);
        const int k = D.rec_k[r];
        const size_t po = (size_t)b * D.pi_cap + D.rec_off[r];
        if (tid < 24) {
            uint32_t v = ((const uint32_t *)(D.rec_sq + r * 96))[tid];
            if (tid == 22) v &= 0x0000ffffu;
            if (tid == 23) v = 0u;
            rec[tid] = v;
        } else if (tid == 24) {
            PlyHeader h;
            h.t = (uint16_t)t; h.T = (uint16_t)T; h.winner = m.winner; h.turn = D.rec_turn[r]; h.k = (uint8_t)k; h.flags = 0;
            h.board_id = (uint32_t)(D.board_id_base + (uint64_t)b); h.game_no = m.game_no;
            *(PlyHeader *)(rec + kRecHdr / 4) = h;
        }
        if (tid < 64) { // ids: 128 x u16 = 64 dwords
            const int i0 = 2 * tid, i1 = 2 * tid + 1;
            const uint32_t lo = i0 < k ? D.rec_ids[po + i0] : 0u, hi = i1 < k ? D.rec_ids[po + i1] : 0u;
            rec[kRecIds / 4 + tid] = lo | (hi << 16);
        } else if (tid < 192) {
            const int i = tid - 64;
            ((float *)rec)[kRecPi / 4 + i] = i < k ? D.rec_pi[po + i] : 0.0f;
        }
    }
}

// Records -> dense training rows, exactly what k_harvest writes for the same games (game.py:213-237 z and history,
// collect.py:64-131 preprocess + flip_data). One block per ply record; rows go to a ring of ring_rows rows starting at
// row `head` (head = 0 and ring_rows >= rows: a plain array). flags: CCZ_FLAG_REFERENCE_QUIRKS / CCZ_FLAG_NO_MIRROR;
// typepack: 3 bits per plane channel = piece type - 1 encoded there (ccz_config.plane_of_type inverted).
__global__ __launch_bounds__(256) void k_expand_records(const uint8_t *recs, long long n_plies, uint32_t flags, uint32_t typepack,
                                                          uint16_t *states, float *pi, float *z, long long ring_rows, long long head,
                                                          int32_t *bad)
{
    const long long p = blockIdx.x;
    const int tid = threadIdx.x;
    if (p >= n_plies) return;
    __shared__ __align__(16) uint8_t hist[8][96];
    __shared__ PlyHeader sh;
    const uint8_t *rec = recs + (size_t)p * kRecBytes;
    if (tid == 0) sh = *(const PlyHeader *)(rec + kRecHdr);
    __syncthreads();
    const PlyHeader h = sh;
    const bool quirks = (flags & 1u) != 0, mirror = (flags & 2u) == 0;
    const int t = h.t, T = h.T;
    const long long first = p - t;
    if (first < 0 || t >= T || first + T > n_plies || h.k > kMaxLegal) { // not a whole game in this buffer: nothing is read out of bounds
        if (tid == 0 && bad) atomicAdd(bad, 1);
        return;
    }
    const int te = quirks ? T - 1 : t; // game.py:234-237: every sample aliases the history at the LAST recorded ply
    if (tid < 192) { // the 8-deep history of game.py:23-44: index i = the position i plies back, the first one before that
        const int i = tid / 24, w = tid - 24 * i;
        int tp = te - i;
        if (tp < 0) tp = 0;
        ((uint32_t *)hist[i])[w] = ((const uint32_t *)(recs + (size_t)(first + tp) * kRecBytes))[w];
    }
    __syncthreads();
    const int turn_plane = quirks ? 1 : h.turn; // collect.py:78 reads a board that never advances
    const long long mul = mirror ? 2 : 1;
    for (int pass = 0; pass < (mirror ? 2 : 1); ++pass) {
        long long row = head + mul * first + (pass ? T : 0) + t;
        row = ring_rows > 0 ? row % ring_rows : row;
        uint32_t *srow = (uint32_t *)(states + (size_t)row * 10710);
        for (int i = tid; i < 5355; i += 256) {
            uint32_t v = 0;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int e = 2 * i + hh;
                const int g = e / 630, w = e - g * 630;
                bool on;
                if (g == 16) on = turn_plane != 0;
                else {
                    const int ch = w / 90, s = w - 90 * ch;
                    const int ss = pass ? (s - s % 9) + (8 - s % 9) : s; // np.flip(axis=2): file mirror
                    on = hist[g & 7][ss] == (int)((typepack >> (3 * ch)) & 7u) + 1 + (g >= 8 ? 8 : 0);
                }
                if (on) v |= (uint32_t)kHalfOne << (16 * hh);
            }
            srow[i] = v;
        }
        float *prow = pi + (size_t)row * kNMoves;
        for (int i = tid; i < kNMoves; i += 256) prow[i] = 0.0f;
        __syncthreads();
        const uint16_t *ids = (const uint16_t *)(rec + kRecIds);
        const float *pv = (const float *)(rec + kRecPi);
        for (int i = tid; i < h.k; i += 256) {
            const int id = ids[i];
            if (id < kNMoves) prow[pass ? c_tab.flip[id] : id] = pv[i]; // mcts_prob[flip_map]
        }
        if (tid == 0) z[row] = h.winner < 0 ? 0.0f : (h.turn == (uint8_t)h.winner ? 1.0f : -1.0f); // game.py:213-219
        __syncthreads();
    }
}

// ------------------------------------------------------------------ stateless batch rules
__global__ __launch_bounds__(64) void k_legal_moves(int n, const uint8_t *sq, const uint8_t *turn, const int32_t *halfmove,
                                                      uint32_t *mask, int32_t *count, uint8_t *flags)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n) return;
    __shared__ __align__(16) uint8_t s_sq[96];
    __shared__ GenScratch S;
    load_board(s_sq, sq + (size_t)b * 96, lane);
    __syncthreads();
    const int t = turn[b] ? 1 : 0;
    const GenResult g = gen_legal(s_sq, t, S, nullptr, lane);
    __syncthreads();
    if (mask) for (int w = lane; w < kMaskWords; w += 64) mask[(size_t)b * kMaskWords + w] = S.mask[w];
    if (lane == 0) {
        if (count) count[b] = g.n_legal;
        if (flags) {
            uint8_t f = 0;
            if (g.ksq >= 0 && king_attacked(s_sq, S, g.ksq, -1, -1, 0, t)) f |= 1;
            if (g.insufficient) f |= 2;
            if (halfmove && halfmove[b] >= 120 && g.n_legal > 0) f |= 4;
            if (g.overflow) f |= 128;
            flags[b] = f;
        }
    }
}

__global__ void k_apply_moves(int n, uint8_t *sq, uint8_t *turn, const int32_t *ids, uint8_t *captured)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int id = ids[i];
    if (id < 0 || id >= kNMoves) return;
    uint8_t *s = sq + (size_t)i * 96;
    const int from = c_tab.from[id], to = c_tab.to[id];
    if (captured) captured[i] = s[to];
    s[to] = s[from];
    s[from] = 0;
    turn[i] ^= 1;
}

} // namespace ccz

namespace ccz {
// one thread, after k_finish_move: all live trees now sit in the other pool half
__global__ void k_flip_half(Dev D) { *D.half ^= 1; }
} // namespace ccz
