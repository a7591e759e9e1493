/*
 * cczero.h -- C ABI of libcczero.so: the MI355X (gfx950) lockstep self-play rollout engine.
 *
 * Drop-in boundary for the ONE hot path of Symb0x76/ChineseChessZero:
 * per-move MCTS select / expand / backup (reference mcts.py:101-178), the rules the reference
 * takes from `cchess` (move generation, make-move, game-end and draw predicates; call sites
 * mcts.py:111-126, net.py:154-157, game.py:201-216, tools.py:109-123), leaf encoding for the
 * evaluator (net.py:160-177, tools.py:74-106), pi / Dirichlet-mixed move choice
 * (mcts.py:163-166,216-224), tree reuse (mcts.py:168-178) and the self-play bookkeeping of
 * game.py:133-237 + collect.py:64-131 (training tuples, mirror augmentation).
 *
 * The reference is pure Python and has no FFI; what it has is a call surface
 * (policy_value_fn / MCTS_AI / Game.start_self_play / CollectPipeline). The Python mirror of
 * that surface (chinesechesszero_amd/{mcts,game,collect,net,tools}.py) binds these entry points
 * with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - B boards advance in lockstep; one 64-lane wavefront owns one board.
 *   - every function returns 0 on success, <0 on error (ccz_last_error() has the text); nothing
 *     throws across the ABI.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all
 *     device work is enqueued on it; functions documented "syncs" wait for that stream.
 *   - pointers named *_dev are device pointers owned by the caller (torch tensors); *_host are
 *     host pointers. Everything else is owned by the engine.
 *   - single caller thread per engine, no re-entrancy (same as the reference).
 *   - there is NO CPU fallback: every entry point that computes fails if no gfx950 device is
 *     usable.
 */
#ifndef CCZERO_H
#define CCZERO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCZ_ABI_VERSION 7
#define CCZ_NSQ 90
#define CCZ_SQ_STRIDE 96            /* mailbox row stride in bytes (90 squares + 6 pad)         */
#define CCZ_NMOVES 2086             /* action space, reference tools.py:172-272                 */
#define CCZ_MAX_LEGAL 128           /* upper bound on legal moves of one position               */
#define CCZ_MASK_WORDS 66           /* ceil(2086/32): legal-move bitmask words                  */
#define CCZ_PLANES 10710            /* 17*7*10*9 evaluator input elements, net.py:174-177       */

/* piece codes in the mailbox: 0 empty, red = type, black = type + 8;
 * types PAWN=1 CANNON=2 ROOK=3 KNIGHT=4 BISHOP=5 ADVISOR=6 KING=7 (the engine's own numbering; the plane
 * channel a type is encoded into is ccz_config.plane_of_type, default type-1 as tools.py:100 computes it)
 * colours as in cchess: RED = 1 (True), BLACK = 0 (False). square = file + 9*rank (tools.py:91). */

/* ccz_config.rule_flags: rule variants of the absent `cchess` module that cannot be checked here */
#define CCZ_RULE_PAWN_MOVE_RESETS_CLOCK 2u /* the sixty-move clock (and the repetition history with it) restarts on pawn
                                       moves as well as on captures -- python-chess's `is_zeroing`, which a port may have kept;
                                       default: captures only. A pawn never moves backwards, so no earlier position can recur */
#define CCZ_RULE_PERPETUAL_CHECK 1u /* a game that ends by fourfold repetition is examined for perpetual check: inside the
                                       repetition window (the positions after the EARLIEST occurrence of the repeated position,
                                       up to the current one) a side whose every move gave check, while not every move of the
                                       other side did, LOSES (winner = the other side); both or neither: draw, as without the
                                       flag. Changes outcome().winner only (game.py:210-216, z of the tuples): in the search
                                       the same leaf is `end and is_tie` -> 0.0 either way (mcts.py:120-122), visit counts do
                                       not depend on it. This build's statement of the rule: python-chinese-chess may apply
                                       one [unverified]; default off                                                    */

/* flags for ccz_config.flags */
#define CCZ_FLAG_REFERENCE_QUIRKS 1u /* harvest(): reproduce game.py:234-237 (all samples carry the
                                        final 8-ply history) and collect.py:78 (turn plane all ones) */
#define CCZ_FLAG_NO_MIRROR 2u        /* harvest(): do not append collect.py:112-131 mirror samples  */
#define CCZ_FLAG_VALUE_F16 4u        /* reference CUDA-path quirk: under autocast (net.py:178-189) the value reaches
                                        Node.update as a float16 ndarray and NumPy accumulates Node.value in float16
                                        (mcts.py:63-71 under NEP 50). With this flag the leaf value is rounded to float16
                                        and every operation of the incremental mean rounds to float16; default (off) is
                                        the float32 arithmetic of the reference's CPU path                              */

#define CCZ_FLAG_CACHE_VERIFY 8u     /* debug mode of the evaluation cache (ABI 5): one table hit in 128 is sent through the evaluator
                                        again and its fresh priors / value are compared, bit for bit, with what the table
                                        returned (ccz_stats.cache_verified / cache_verify_mismatches). Results are unchanged;
                                        the evaluator computes ~0.8 % of the hits again. Needs eval_cache_log2 > 0           */

#define CCZ_FLAG_STRICT 16u          /* parity mode (ABI 7): the two places where this engine silently departs from the reference become
                                        sticky error bits instead of counters -- a kept subtree that had to be pruned to fit the node
                                        pool (CCZ_ERR_PRUNED; the reference's tree is unbounded, mcts.py:31-39) and a game adjudicated
                                        at max_plies (CCZ_ERR_TRUNCATED; the reference's game ends by the rules only, game.py:155).
                                        The search results are what they are without the flag; MCTS_AI, the UCI loop and the parity
                                        tests run with it, the throughput paths (bench, collector) count and report instead          */

/* leaf status written by ccz_select_leaves */
#define CCZ_LEAF_EXPAND 0 /* non-terminal: children are created from the evaluator's priors */
#define CCZ_LEAF_DRAW 1   /* game over and is_tie(): leaf value 0.0   (mcts.py:120-122)   */
#define CCZ_LEAF_LOSS 2   /* side to move has no legal move: leaf value -1.0 (mcts.py:123-126) */
#define CCZ_LEAF_NONE 3   /* no pending leaf: board finished, leaf already backed up, or nothing selected yet */

typedef struct ccz_engine ccz_engine;

typedef struct ccz_config {
    int32_t n_boards;      /* B: concurrent boards on this GPU                                   */
    int32_t n_playout;     /* simulations per move (parameters.py:14 PLAYOUT; informational)    */
    float c_puct;          /* parameters.py:8  C_PUCT = 5                                        */
    float eps;             /* parameters.py:10 EPS   = 0.25 (Dirichlet mixing weight)            */
    float alpha;           /* parameters.py:12 ALPHA = 0.2                                       */
    float temp;            /* game.py:133 temp=1.0 ; schedule of game.py:159 applied per board   */
    int32_t max_nodes;     /* tree nodes per board per pool half (0 = 512 x (n_playout + 64))    */
    int32_t max_depth;     /* selection path capacity (0 = 512)                                  */
    int32_t max_plies;     /* recorded plies per game before adjudicating a draw (0 = 2048)      */
    uint32_t flags;        /* CCZ_FLAG_*                                                         */
    uint64_t seed;         /* device-mode sampling: Philox key                                   */
    uint64_t board_id_base;/* global id of board 0 (rank * n_boards): RNG streams independent of GPU count */
    int32_t device;        /* HIP device ordinal                                                 */
    int32_t reserve_nodes; /* pool nodes kept free at re-root time for the next move's expansions
                              (0 = min(128 x n_playout, max_nodes / 2)); the kept subtree is pruned
                              bottom-up to max_nodes - reserve_nodes                            */
    /* ---- the two choices the golden traces cannot pin, as run-time tables (ABI 2) ---------------------
     * move_rank_host: uint16 [2086] host array or NULL. The iteration order of `board.legal_moves`
     *   (net.py:154-157) decides the children's insertion order (mcts.py:37-39), hence which unvisited
     *   child is tried first and how PUCT ties break (mcts.py:47-48,59-61). Legal moves are listed in
     *   ascending move_rank_host[id]; NULL = ascending id (this build's canonical order). Must be a
     *   permutation of 0..2085. Copied at create.
     * type_rank: major sort key by the PIECE TYPE of the mover (index 1..7, entry 0 unused, values 0..7; all zero = no
     *   major key): legal moves are listed in ascending (type_rank[type], move_rank[id]). Bitboard libraries iterate
     *   piece sets, so their order is typically "these piece types first, squares in scan order within" -- e.g. the
     *   python-chess scheme (non-pawn moves by from-square and to-square descending, then pawn moves) is
     *   type_rank = {PAWN: 1, others: 0} with move_rank = the rank of (from, to) in descending order.
     * plane_of_type: channel (0..6) of piece type t = 1..7 inside a colour's 7-plane group, entry 0 unused;
     *   all zero = {-, 0,1,2,3,4,5,6}, i.e. `piece_type - 1` (tools.py:100) under this build's type numbering.
     *   With another cchess PIECE_TYPES numbering only this table changes (reference-trained weights see
     *   their own plane order). Must be a permutation of 0..6. */
    const uint16_t *move_rank_host;
    uint8_t plane_of_type[8];
    uint32_t rule_flags;   /* CCZ_RULE_*                                                         */
    uint32_t eval_cache_log2; /* 0 = none; n = an evaluation cache of 2^n entries of 528 B (ABI 3): see ccz_eval_plan */
    uint8_t type_rank[8];
} ccz_config;

typedef struct ccz_stats {
    int64_t sims;            /* playouts completed since create                                  */
    int64_t moves;           /* moves played                                                     */
    int64_t games;           /* games finished                                                   */
    int64_t truncated_games; /* games adjudicated at max_plies                                   */
    int64_t nodes_peak;      /* max nodes in use on any board                                    */
    int64_t depth_peak;      /* max selection depth seen                                         */
    int64_t sum_depth;       /* sum of leaf depths (for d-bar)                                   */
    int64_t sum_children;    /* sum of children created (for k-bar)                              */
    int64_t expansions;      /* leaves expanded                                                  */
    int64_t terminal_leaves; /* terminal leaves backed up                                        */
    int32_t error_flags;     /* sticky device error bits (CCZ_ERR_*), 0 = healthy                */
    int32_t reserved;        /* -DCCZ_BOUNDS diagnostic build: source line of the last stray index; 0 otherwise */
    int64_t hbm_bytes;       /* device memory held by the engine                                 */
    int64_t pruned_subtrees; /* nodes whose children were dropped at re-root time to keep the kept
                                subtree within the pool budget (0 in normal runs)                 */
    /* evaluation cache (ABI 3; all zero without one) */
    int64_t cache_probes;      /* leaves that needed an evaluation                                */
    int64_t cache_hits;        /* ... served from the cache                                       */
    int64_t cache_shared_rows; /* ... served by the evaluator row of another board of the same step (same position) */
    int64_t cache_stores;      /* entries written                                                 */
    /* CCZ_FLAG_CACHE_VERIFY (ABI 5; zero without it) */
    int64_t cache_verified;          /* table hits that were evaluated again                        */
    int64_t cache_verify_mismatches; /* ... whose fresh priors or value differ from the cached ones: must stay 0 (a colliding
                                        position, a non-deterministic evaluator, or weights changed without ccz_eval_cache_clear) */
} ccz_stats;

#define CCZ_ERR_NODE_POOL 1   /* a board ran out of tree nodes (raise max_nodes)                 */
#define CCZ_ERR_DEPTH 2       /* a selection path exceeded max_depth                             */
#define CCZ_ERR_CHAIN 64      /* more than 128 positions since the last capture (history chain)  */
#define CCZ_ERR_MOVES 4       /* more than CCZ_MAX_LEGAL legal moves / pseudo-move overflow       */
#define CCZ_ERR_RECORD 8      /* pi record arena overflow (game adjudicated)                     */
#define CCZ_ERR_BAD_MOVE 16   /* forced move id invalid / nothing searched and nothing forced    */
#define CCZ_ERR_NAN 32        /* NaN priors: no comparable child during selection                */
#define CCZ_ERR_BOUNDS 128    /* bounds-checked diagnostic build only (-DCCZ_BOUNDS): an index left its array      */
#define CCZ_ERR_PRUNED 256    /* CCZ_FLAG_STRICT: a kept subtree lost nodes at re-root time (raise max_nodes / lower reserve_nodes) */
#define CCZ_ERR_TRUNCATED 512 /* CCZ_FLAG_STRICT: a game reached max_plies and was adjudicated a draw (raise max_plies)            */

/* ---- library ---------------------------------------------------------------------------- */
int ccz_abi_version(void);
const char *ccz_last_error(void);
/* number of usable gfx950 devices (0 if none); never initialises more than the runtime needs */
int ccz_device_count(void);

/* ---- action space (replaces reference tools.py:172-272 tables and tools.py:133-166 flip) - */
/* uci_out: 2086 x 5 bytes, NUL-terminated 4-char strings ("a0a1") */
int ccz_action_table(char *uci_out_host, uint8_t *from_out_host, uint8_t *to_out_host);
int ccz_flip_map(int32_t *flip_out_host); /* collect.py:118-123 */

/* ---- engine lifetime --------------------------------------------------------------------- */
int ccz_create(const ccz_config *cfg, ccz_engine **out);
int ccz_destroy(ccz_engine *e);
/* start new games (standard opening, fresh trees, empty records) on the boards whose mask byte is
 * non-zero; mask_host == NULL means all boards. Replaces cchess.Board() + reset_player()
 * (game.py:148, mcts.py:200-201). */
int ccz_reset(ccz_engine *e, void *stream, const uint8_t *mask_host);
/* set an arbitrary root position on one board (fresh tree, game record restarted); for tests and
 * match play. sq_host: 90 piece codes. */
int ccz_set_position(ccz_engine *e, void *stream, int32_t board, const uint8_t *sq_host,
                     int32_t turn, int32_t halfmove);
/* fresh root `Node(None, 1.0)` on the boards whose mask byte is non-zero (NULL = all), keeping position,
 * history chain, game record and clocks: MCTS.update_with_move(-1) (mcts.py:176-178, what reset_player() and
 * the non-self-play branch of get_action call, mcts.py:200-201,228-229). The pending leaf is dropped. */
int ccz_reset_tree(ccz_engine *e, void *stream, const uint8_t *mask_host);

/* ---- one lockstep simulation = select -> (evaluator) -> expand+backup --------------------- */
/* Replaces the first half of MCTS.playout (mcts.py:101-111) + policy_value_fn's input building
 * (net.py:154-177) for all boards: PUCT descent from each root with make-move, legal-move
 * generation and game-end tests at the leaf, and the evaluator input written to
 * leaf_input_f16_dev [B,17,7,10,9] fp16. Only plane groups 7, 15 and 16 are ever non-zero on this
 * path (net.py:160-173); the engine rewrites those three and assumes the other 14 groups are zero
 * (ccz_zero_leaf_input zeroes the whole tensor once). */
int ccz_select_leaves(ccz_engine *e, void *stream, void *leaf_input_f16_dev);
int ccz_zero_leaf_input(ccz_engine *e, void *stream, void *leaf_input_f16_dev);
/* Replaces the second half of MCTS.playout (mcts.py:113-129): Node.expand with priors
 * prob_dev[b, id] (float32 [B,2086] = exp(log_act_probs), net.py:202-203) for the leaf's legal ids
 * in ascending id order, leaf value value_dev[b] (float32 [B], side to move's view) or the terminal
 * value, and Node.update_recursive up the path. */
int ccz_expand_backup(ccz_engine *e, void *stream, const float *prob_dev, const float *value_dev);

/* Fused form of "ccz_expand_backup for the pending leaf, then ccz_select_leaves for the next
 * simulation" in ONE launch (one launch boundary and one tree-head re-read less per simulation).
 * A move of n simulations is: select, (evaluator, step) x (n-1), evaluator, expand_backup. */
int ccz_step(ccz_engine *e, void *stream, const float *prob_dev, const float *value_dev,
             void *leaf_input_f16_dev);

/* Compact evaluator boundary: the evaluator hands over the policy head's LOGITS (float32 or float16
 * [B,2086], before log_softmax). ccz_gather_priors computes exp(log_softmax(logits)) for the pending leaf's
 * legal ids only (net.py:202-205 use exactly those) in one pass per board into an engine-owned [B,128] row;
 * ccz_step_compact / ccz_expand_backup_compact are ccz_step / ccz_expand_backup reading that row. Two full
 * [B,2086] passes less on the evaluator side and no scattered prior gather inside the simulator kernel. */
int ccz_gather_priors(ccz_engine *e, void *stream, const void *logits_dev, int32_t logits_f16);
int ccz_step_compact(ccz_engine *e, void *stream, const float *value_dev, void *leaf_input_f16_dev);
int ccz_expand_backup_compact(ccz_engine *e, void *stream, const float *value_dev);

/* ---- scouts (round 6, ABI 7): the next leaves of a board, evaluated before it asks for them -------------------------------------
 * The reference's first-maximum rule (mcts.py:47-48,59-61: an unvisited child scores +inf and max() returns the first one) fixes the
 * order in which a node's children are first visited: the order of board.legal_moves. When a board's pending leaf is child i of a
 * node, the simulations that reach that node next will ask for children i + 1, i + 2, ... One game at a time (MCTS_AI, the UCI loop:
 * one 90-pixel row per evaluator call) therefore runs with SCOUT SLOTS: the last n_scouts boards of the engine have no tree and no
 * game; ccz_scout hands slot active + j * active + r the (i + 1 + j)-th child of board r's pending leaf's parent as its pending leaf
 * (position, legal moves, status, key, evaluator input row), ccz_eval_plan_scouted probes the evaluation cache for every slot and
 * plans ONE evaluator call of all n_boards rows (row b = slot b; *n_miss_dev = n_boards) iff a searched board misses, else none
 * (*n_miss_dev = 0; state_dev[r] = 0 miss / 1 hit / 2 no evaluation needed, for the host to branch on); ccz_gather_priors_planned
 * then stores the scouts' evaluations in the table, where board r finds them. The simulator entry points (select / step /
 * expand_backup / finish_move) run on boards 0 .. n_boards - n_scouts - 1 only. Same visit counts, bit for bit: the table returns
 * what the evaluator returns for the position, and the evaluator's result for a row does not depend on the batch it sits in. */
int ccz_set_scouts(ccz_engine *e, int32_t n_scouts);
int ccz_scout(ccz_engine *e, void *stream, void *leaf_input_f16_dev);
int ccz_eval_plan_scouted(ccz_engine *e, void *stream, int32_t *miss_rows_dev, int32_t *n_miss_dev, int32_t *state_dev);
/* ccz_scout + ccz_eval_plan_scouted as ONE launch (engines of up to 16 slots: one workgroup, one wave per slot; more: the three launches) */
int ccz_scout_and_plan(ccz_engine *e, void *stream, void *leaf_input_f16_dev, int32_t *miss_rows_dev, int32_t *n_miss_dev, int32_t *state_dev);
/* Simulations of a scouted engine (at most 16 slots) WITHOUT the host in between: ONE launch, one workgroup, repeats
 * { ccz_step_compact (expand + backup of the pending leaves from the engine's prior / value rows, next selection) ; ccz_scout_and_plan }
 * for as long as every searched board finds its next leaf in the table (or needs no evaluation). It returns to the host when the
 * evaluator has to run, after run_dev[0] simulations (a caller that reports progress), or when the move's last simulation -- which has
 * no next selection: ccz_expand_backup_compact -- is backed up. run_dev: four int32 the device can reach (device or pinned host memory):
 *   [0] in:  budget, >= 1            [1] in:  simulations left in this move, the pending one included, >= 1
 *   [2] out: simulations done here   [3] out: 1 = a searched board's leaf needs the evaluator now (the plan and state_dev are then what
 *                                            ccz_scout_and_plan would have left), 0 = budget or move exhausted
 * Call it where ccz_step_compact would be called: the pending leaves' priors and values are in place (ccz_gather_priors_planned, or a
 * table hit). Same phases in the same order on the same data as the separate launches: same trees, bit for bit
 * (tests/test_gpu_scouts.py). mcts.py:101-129 is the loop this runs; the reference's one-evaluation-per-playout is the path it skips. */
int ccz_scouted_run(ccz_engine *e, void *stream, void *leaf_input_f16_dev, int32_t *miss_rows_dev, int32_t *n_miss_dev, int32_t *state_dev,
                    int32_t *run_dev);

/* Evaluation cache (ccz_config.eval_cache_log2 > 0). On the search path the evaluator sees the leaf POSITION and the side to move
 * only (net.py:160-173: the history planes are zero; mcts.py:214 passes none), i.e. a function of the leaf's Zobrist key. The
 * reference evaluates every leaf on its own (mcts.py:114); here a position that was evaluated before -- by this board (a
 * transposition), by another board (every restarted game walks through openings that earlier games searched), or by another
 * board in this very step -- is not sent through the network again. Results are unchanged bit for bit as long as the evaluator is
 * a deterministic function of the position that does not depend on the row it sits in (tests/test_gpu_evaluator_depth.py) --
 * up to key collisions: a hit needs the 64-bit key, the number of legal moves AND a 24-bit hash of the legal-move list (in the
 * order the priors are stored in) to agree, so a foreign position is served with probability < 2^-64 per probe (about 1e-9 per
 * day at 2 x 10^5 probes a second); CCZ_FLAG_CACHE_VERIFY re-evaluates a sample of the hits and counts disagreements.
 *   ccz_eval_plan: probes the direct-mapped table (2^n entries: key, value, the priors of the legal moves) for every pending
 *     leaf that needs an evaluation; hits receive their priors and value at once. The misses are deduplicated (boards with the
 *     same key share one row) and compacted: miss_rows_dev int32 [B] receives the board index of every row the evaluator has to
 *     compute, ascending; *n_miss_dev (int32 on the device) their number. No host sync.
 *   The evaluator then runs on those rows only: ccz_pack_live_planes_rows_f16 gathers them, the ccz_conv3x3_*_live entry points
 *     skip the tiles beyond the live rows; logits / values come back COMPACT: row i belongs to board miss_rows_dev[i].
 *   ccz_gather_priors_planned: ccz_gather_priors for the misses (each reads the row of its representative), and the fresh
 *     evaluations are stored in the table. Afterwards ccz_step_compact / ccz_expand_backup_compact with value_dev == NULL use
 *     the engine-owned leaf values (hits: from the table; misses: value_compact_dev[row]).
 *   ccz_eval_cache_clear: forget everything (the evaluator's weights changed). */
int ccz_eval_plan(ccz_engine *e, void *stream, int32_t *miss_rows_dev, int32_t *n_miss_dev);
int ccz_gather_priors_planned(ccz_engine *e, void *stream, const void *logits_compact_dev, int32_t logits_f16,
                              const float *value_compact_dev);
int ccz_eval_cache_clear(ccz_engine *e, void *stream);

/* ---- once per move ------------------------------------------------------------------------ */
/* Replaces MCTS.get_move_probs' tail (mcts.py:162-166), MCTS_AI.get_action's choice
 * (mcts.py:216-224), MCTS.update_with_move (mcts.py:168-178) and the per-move part of
 * Game.start_self_play (game.py:159,188-237): pi from root visits at the board's temperature,
 * record (position, turn, pi), choose the move, re-root the tree on the chosen child (subtree
 * kept), make the move on the root position, detect the end of the game and its winner.
 *   forced_moves_dev: int32 [B] or NULL. entry >= 0: play that move id (host-exact numpy sampling
 *     or match play); a forced move that is not a child of the root (root never searched, or an
 *     opponent's reply the tree never saw) gives a fresh root, as update_with_move does
 *     (mcts.py:176-178). entry < 0 or NULL: sample on the device stream Philox(seed, board id).
 *   temps_dev: float64 [B] or NULL (NULL: schedule of game.py:159 from cfg.temp).
 *   moves_out_dev: int32 [B] or NULL, receives the move played.
 *   keep_tree: 1 = self-play tree reuse (mcts.py:222-224); 0 = discard (mcts.py:228-229).
 * All live trees move to the other half of their node pool in this call (one flip for every board),
 * which is what lets the simulation kernel request a root's children before any load has returned. */
int ccz_finish_move(ccz_engine *e, void *stream, const int32_t *forced_moves_dev,
                    const double *temps_dev, int32_t *moves_out_dev, int32_t keep_tree);

/* root children of every board (syncs): k_host int32[B]; acts_host uint16[B*128];
 * visits_host int32[B*128]; q_host/prior_host float[B*128] (may be NULL); root_visits_host int32[B]
 * (may be NULL). What mcts.py:162-163 reads. */
int ccz_root_children(ccz_engine *e, void *stream, int32_t *k_host, uint16_t *acts_host,
                      int32_t *visits_host, float *q_host, float *prior_host, int32_t *root_visits_host);
/* pi of the root at temperature temps_host[b] (float64 [B], or NULL for the schedule) without
 * moving: pi_host float64 [B*128] aligned with ccz_root_children's acts (syncs). */
int ccz_root_pi(ccz_engine *e, void *stream, const double *temps_host, double *pi_host);
/* per-board game state (syncs): over_host uint8[B] (1 = finished, waiting for harvest),
 * winner_host int8[B] (1 RED, 0 BLACK, -1 draw), plies_host int32[B], turn_host uint8[B];
 * any pointer may be NULL. */
int ccz_game_status(ccz_engine *e, void *stream, uint8_t *over_host, int8_t *winner_host,
                    int32_t *plies_host, uint8_t *turn_host);
/* root positions (syncs): sq_host uint8 [B*96] */
int ccz_root_positions(ccz_engine *e, void *stream, uint8_t *sq_host);
/* leaf bookkeeping of the last ccz_select_leaves / ccz_step (syncs; tests): status uint8[B] (CCZ_LEAF_*), k int32[B],
 * ids uint16[B*128], depth int32[B]; any pointer may be NULL. */
int ccz_leaf_info(ccz_engine *e, void *stream, uint8_t *status_host, int32_t *k_host,
                  uint16_t *ids_host, int32_t *depth_host);

/* What the compact / planned evaluator boundary hands the tree for the pending leaves, AFTER ccz_gather_priors[_planned] and
 * before ccz_step_compact / ccz_expand_backup_compact consume it (syncs; tests): prior_host float [B*128] = the engine-owned
 * prior rows, entry i of board b belongs to the i-th id of ccz_leaf_info (entries past k and rows of terminal leaves are
 * unspecified); value_host float [B] = the engine-owned leaf values of the planned boundary (table hits and fresh evaluations
 * alike; only with an evaluation cache, else pass NULL). Either may be NULL. This is the `(act_probs, leaf_value)` pair of
 * mcts.py:114 as the tree will see it: a sequential oracle fed these numbers must grow the same tree. */
int ccz_leaf_priors(ccz_engine *e, void *stream, float *prior_host, float *value_host);

/* Zobrist keys (pieces + side to move: everything the evaluator input of net.py:160-173 depends on) and CCZ_LEAF_* status of
 * the pending leaves, copied device-to-device on `stream` (no sync): keys_dev uint64 [B], status_dev uint8 [B], either may be
 * NULL. Equal keys = equal evaluator input: what a transposition / duplicate-leaf cache would key on (the reference
 * evaluates every leaf separately, mcts.py:114). */
int ccz_leaf_keys(ccz_engine *e, void *stream, uint64_t *keys_dev, uint8_t *status_dev);

/* ---- training tuples ---------------------------------------------------------------------- */
/* number of tuple rows the finished games would produce (syncs). */
int ccz_harvest_rows(ccz_engine *e, void *stream, int64_t *rows_host);
/* Replaces game.py:208-237 (z assignment) + collect.py:64-131 (preprocess, flip_data) for every
 * finished board: writes rows (state fp16 [17,7,10,9], pi float32 [2086], z float32) into the
 * caller's device buffers, game by game (samples, then their mirror images), then starts a new
 * game on those boards. Finished boards are taken in index order while their rows fit capacity_rows;
 * the others stay finished for the next call (loop until ccz_harvest_rows reports 0). Syncs. */
int ccz_harvest(ccz_engine *e, void *stream, void *states_f16_dev, float *pi_dev, float *z_dev,
                int64_t capacity_rows, int64_t *rows_host);

/* ---- compact game records: the wire format of the multi-GPU exchange ------------------------------ */
/* One fixed-size record per PLY of a finished game, plies of a game contiguous and in order:
 *   bytes   0..89   position before the move (piece codes, square = file + 9*rank), 90..95 zero
 *   bytes  96..111  header: uint16 t (ply index), uint16 T (plies of the game), int8 winner (1 RED, 0 BLACK, -1 draw),
 *                   uint8 turn (side to move), uint8 k (entries of pi), uint8 flags (0), uint32 board_id (global),
 *                   uint32 game_no
 *   bytes 112..367  uint16 ids[128]   move ids of the root's children (mcts.py:162), zero-padded
 *   bytes 368..879  float  pi[128]    visit distribution of the move (mcts.py:163-166), zero-padded
 * 880 B per ply stand for the TWO dense rows of 29,768 B (sample + mirror image) that ccz_harvest writes, so the
 * all-gather of finished games (replay.RecordGatherer) moves 1/67 of the bytes; ccz_expand_records rebuilds the rows
 * on the receiving side. What N reference collectors would append to data.h5 (collect.py:146-167). */
#define CCZ_REC_BYTES 880
#define CCZ_REC_HDR 96
#define CCZ_REC_IDS 112
#define CCZ_REC_PI 368
/* ccz_harvest with records as output: the finished boards (index order, while their plies fit capacity_plies; loop
 * until ccz_harvest_rows reports 0) are written to records_dev [capacity_plies x 880 B] and restarted. Syncs. */
int ccz_harvest_records(ccz_engine *e, void *stream, void *records_dev, int64_t capacity_plies, int64_t *plies_host);
/* Records -> dense rows, byte for byte what ccz_harvest writes for the same games (game.py:213-237, collect.py:64-131):
 * state fp16 [17,7,10,9], pi float32 [2086], z float32; per game T samples then (unless CCZ_FLAG_NO_MIRROR) their T mirror
 * images. Stateless: needs no engine (the receiving rank may never have played these games). records_dev must hold WHOLE
 * games (a record whose game is cut is skipped and counted in *bad_records_dev, int32 on the device, may be NULL).
 * flags: CCZ_FLAG_REFERENCE_QUIRKS | CCZ_FLAG_NO_MIRROR; plane_of_type_host: as ccz_config.plane_of_type or NULL.
 * Rows are written to row (head_row + i) % ring_rows of the output arrays (a replay ring resident in HBM);
 * ring_rows = 0: a plain array, row i. Asynchronous on `stream`. */
int ccz_expand_records(void *stream, const void *records_dev, int64_t n_plies, uint32_t flags,
                       const uint8_t *plane_of_type_host, void *states_f16_dev, float *pi_dev, float *z_dev,
                       int64_t ring_rows, int64_t head_row, int32_t *bad_records_dev);

int ccz_get_stats(ccz_engine *e, void *stream, ccz_stats *out); /* syncs */

/* ---- stateless batch rules (parity tests, perft; replaces cchess legal_moves / game-end) ---- */
/* n positions: sq_dev uint8 [n*96], turn_dev uint8 [n], halfmove_dev int32 [n] or NULL.
 * outputs (any may be NULL): mask_dev uint32 [n*66] legal-move bitmask over the 2086 ids,
 * count_dev int32 [n], flags_dev uint8 [n]: bit0 side to move in check, bit1 insufficient material,
 * bit2 sixty-move rule (needs halfmove_dev). */
int ccz_legal_moves(void *stream, int32_t n, const uint8_t *sq_dev, const uint8_t *turn_dev,
                    const int32_t *halfmove_dev, uint32_t *mask_dev, int32_t *count_dev,
                    uint8_t *flags_dev);
/* apply move ids to n positions in place (captures reported in captured_dev uint8[n], may be NULL);
 * replaces cchess.Board.push for batches. */
int ccz_apply_moves(void *stream, int32_t n, uint8_t *sq_dev, uint8_t *turn_dev,
                    const int32_t *move_ids_dev, uint8_t *captured_dev);

/* ---- evaluator epilogue (the net itself stays in PyTorch-ROCm / MIOpen) ------------------------ */
/* One-pass fused epilogue of a tower convolution on NHWC fp16 activations y [rows, channels]:
 * y = relu(y + bias[c])  or, with residual_dev != NULL,  y = relu(y + bias[c] + residual)  (reference
 * net.py:32-43: conv -> BN(folded into conv/bias) -> [+x] -> ReLU). Replaces three separate full-tensor
 * passes (bias, add, clamp) that PyTorch issues around an MIOpen convolution. 16-byte aligned pointers. */
int ccz_bias_act_f16(void *stream, void *y_dev, const void *bias_dev, const void *residual_dev,
                     int64_t rows, int32_t channels);

/* A whole tower convolution in one kernel (MFMA implicit GEMM, hand-written for gfx950): 3x3, padding 1,
 * 256 -> 256 channels over boards of 10 x 9, NHWC fp16 in and out, fp32 accumulate:
 *   y[p, co] = act( bias[co] + sum_{ky,kx,ci} w[co, ky, kx, ci] * x[p + 9*(ky-1) + (kx-1), ci] [+ residual[p, co]] )
 * with taps that leave the board contributing zero; relu is a flag word (CCZ_CONV_*): bit 0 = apply ReLU, bit 1 = process
 * the pixel tiles in descending order (same results; alternating the order from layer to layer reads first what the
 * previous layer wrote last), bit 6 = the activations are in the group-of-16 row layout (below) (reference net.py:20-43,
 * ResBlock conv -> BN(folded) -> [+x] -> ReLU; replaces F.conv2d + ccz_bias_act_f16 for these layers).
 * x, y, residual: [n_pixels, 256] fp16 (n_pixels = boards * 90); w: [256, 3, 3, 256] fp16 (the memory of a
 * channels-last [co, ci, 3, 3] tensor); bias: float32 [256]. y may alias residual, not x. At most 93,206 boards
 * per call (32-bit element offsets).
 * Row layouts. Default: NHWC, row = board * 90 + pos (pos = rank * 9 + file). CCZ_CONV_G16: row = (g * 90 + pos) * 16 + j
 * for board 16 g + j (n_pixels must then be a multiple of 1440 = 16 boards): sixteen consecutive rows are ONE board position
 * of sixteen boards, which lets the kernel skip the taps that leave the board instead of multiplying zeros, on tiles of two
 * whole ranks (csrc/cczero_conv_g16.h: the form the evaluator uses from 640 boards on). With CCZ_CONV_G16 the WEIGHTS
 * are packed too: w_dev is what ccz_pack_conv_weights_g16_f16 wrote (below). All kernels behind these entry points add their
 * products in the same order: a board's result is the same in either layout and at any batch size. */
#define CCZ_CONV_RELU 1
#define CCZ_CONV_DESCENDING 2
#define CCZ_CONV_FORCE_SMALL 16 /* A/B runs and tests: k_conv3x3_small whatever the batch size (default layout only) */
#define CCZ_CONV_FORCE_TILE 32  /* A/B runs and tests: the 256-pixel tile kernel whatever the batch size (default layout only) */
#define CCZ_CONV_G16 64
#define CCZ_CONV_G16_EDGE_TILES 128 /* with CCZ_CONV_G16 (round 4, opt-in): the middle launch (ranks 1..8, four two-rank tiles per group) followed
                                       by the edge-pair launch (ranks 0 and 9 of two groups per tile, six live taps instead of nine:
                                       csrc/cczero_conv_g16e.h) instead of ONE launch of five tiles per group whose edge tiles multiply
                                       zeroed ranks. Same values; -3 % per layer at 4096 boards in isolation; in the workload it pays only
                                       with three launch chains and from ~4096 boards on (+0.7...0.9 % sims/s), where the evaluator
                                       sets it (InferenceNet edge_tiles=auto); smaller launches lose (profiles/r04_conv_g16.json) */
#define CCZ_CONV_G16_PERSISTENT 256 /* with CCZ_CONV_G16 (round 6, opt-in, 256 input channels): the same tiles on a FIXED number of workgroups
                                       that walk tile lists -- no prologue after a workgroup's first tile, the next tile's operands arrive
                                       while the epilogue's stores drain (csrc/cczero_conv_g16p.h). Bits 16..27 of the flag word = number of
                                       workgroups (0 = 256, one per CU). Same values */
int ccz_conv3x3_c256_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev,
                         const void *residual_dev, void *y_dev, int64_t n_pixels, int32_t relu);
/* Weights for the CCZ_CONV_G16 form, once per weight set: w [256, 3, 3, cin] fp16 (cin = 256: tower, 64: stem) ->
 * wp [cin / 32][9][256][32] fp16, the four 16-byte chunks of every 64-byte row in the order of the kernel's LDS image: a
 * half-tile (32 input channels of one tap, all output channels) is then one contiguous 16 KB block. Not in place. */
int ccz_pack_conv_weights_g16_f16(void *stream, const void *w_dev, void *wp_dev, int32_t cin);

/* The stem convolution (reference net.py:59-66,86-88: conv3x3(119 -> 256) -> BN(folded) -> ReLU) with the same kernel,
 * one 64-channel chunk: x64 [n_pixels, 64] fp16 holds the 21 planes that can be non-zero on the search path in
 * channels 0..20 (ccz_pack_live_planes_f16) and zeros above; w: [256, 3, 3, 64] fp16 (input channels in the same
 * order, zero-padded); bias float32 [256]; y [n_pixels, 256] fp16. Exact: zero inputs contribute nothing. With
 * CCZ_CONV_G16 the kernel takes that at its word: channels 32..63 of x64 are NOT READ (one 32-channel chunk instead of two:
 * they hold zeros by this contract, and adding products of zeros changes no bit of an accumulator that starts at a bias). */
int ccz_conv3x3_stem_f16(void *stream, const void *x64_dev, const void *w_dev, const void *bias_f32_dev,
                         void *y_dev, int64_t n_pixels, int32_t relu);
/* Evaluator input [n_boards, 17, 7, 10, 9] fp16 (the layout ccz_select_leaves / ccz_step write, net.py:174-177)
 * -> NHWC rows of 64 channels: planes 49..55 (group 7), 105..118 (groups 15, 16), then zeros (net.py:160-173 leaves
 * every other group zero on the search path). */
int ccz_pack_live_planes_f16(void *stream, const void *leaf_dev, void *x64_dev, int32_t n_boards);
/* The same into the CCZ_CONV_G16 row layout; x64 must hold ceil(n_boards / 16) * 1440 rows (the rows of the boards that pad the
 * last group are left alone). rows_dev / n_rows_dev as for ccz_pack_live_planes_rows_f16 below, or both NULL. */
int ccz_pack_live_planes_g16_f16(void *stream, const void *leaf_dev, void *x64_dev, int32_t n_boards,
                                 const int32_t *rows_dev, const int32_t *n_rows_dev);

/* The same three for the planned evaluator boundary (ccz_eval_plan): the number of rows to compute is a DEVICE value, so that
 * no host sync stands between the plan and the evaluator. ccz_pack_live_planes_rows_f16: output row i = board rows_dev[i] for
 * i < *n_rows_dev (the rest is left alone). The planned forms (rows_dev != NULL, either layout) write channels 0..23 of a row
 * only: x64 must be a buffer whose channels 24..63 are ZERO (zeroed once and reused step after step: what the evaluator does). The *_live convolutions take the pointers of the WHOLE batch: the first *live_rows_dev
 * boards are live and are cut into n_parts equal ranges (multiples of 8 boards; of 16 with CCZ_CONV_G16) of which this launch
 * computes range `part` -- so that concurrent launch chains stay balanced whatever the live count is; n_pixels = the largest
 * range a launch may get (ceil(boards / n_parts) rounded up to 8 (16) boards, x 90): the grid is sized for it, tiles past the
 * live rows exit at once, the last live tile may be partial. */
int ccz_pack_live_planes_rows_f16(void *stream, const void *leaf_dev, void *x64_dev, int32_t n_boards,
                                  const int32_t *rows_dev, const int32_t *n_rows_dev);
int ccz_conv3x3_c256_f16_live(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev,
                              const void *residual_dev, void *y_dev, int64_t n_pixels, int32_t relu,
                              const int32_t *live_rows_dev, int32_t part, int32_t n_parts);
int ccz_conv3x3_stem_f16_live(void *stream, const void *x64_dev, const void *w_dev, const void *bias_f32_dev,
                              void *y_dev, int64_t n_pixels, int32_t relu, const int32_t *live_rows_dev,
                              int32_t part, int32_t n_parts);

/* ---- the evaluator's tail: head convolutions, fully connected layers, tanh (reference net.py:96-109) -------------------------
 * Hand-written like the tower so that (1) only the LIVE rows of a planned batch are computed, (2) the group-of-16 -> board
 * permutation, bias, ReLU and the policy / value split cost no passes of their own, and (3) a board's logits and value are the
 * same bits at every batch size (each output element is one fixed chain of MFMAs over k = 0, 32, 64, ...).
 *
 * ccz_heads_conv1x1_f16: both 1x1 head convolutions (policy_conv 256 -> 17, value_conv 256 -> 7, BatchNorm folded) + ReLU on the
 *   tower's output rows x [n_boards * 90, 256] fp16 (NHWC rows, or the group-of-16 row order with flags = CCZ_CONV_G16: n_boards
 *   a multiple of 16). w32: [32, 256] fp16, rows 0..16 = policy channels, 17..23 = value channels, 24..31 zero; bias32: float
 *   [32]. Outputs in BOARD order: pol [n_boards, 1536] fp16 = (pos, channel)-major 90 x 17 values + 6 pad elements that are
 *   never written (keep them zero), val [n_boards, 640] fp16 = 90 x 7 values + 10 pad. live_boards_dev: device int32 count of
 *   live boards (rows of boards past it are skipped) or NULL.
 * ccz_fc_f16: c[m, n] = act(bias[n] + sum_k a[m, k] * w[n, k]): a [m, lda] fp16, w [ceil(n / 128) * 128, k] fp16 with zero rows
 *   past n, bias float [ceil(n / 128) * 128], c [m, ldc] fp16; k a multiple of 64 (zero-pad the columns of w), n and ldc even;
 *   relu bit 0 applies ReLU (bits 1 / 2 force the 128 x 128-tile kernel / the 256 x 144-tile kernel, which otherwise serves m >= 2048
 *   with n >= 1024; m <= 16 -- one game at a time -- runs on a one-wave-per-16-columns kernel: the same bits from all three -- tests and A/B runs). policy_fc: a = pol, k = 1536, n = 2086; value_fc1: a = val,
 *   k = 640, n = 256, relu. The FC weights'
 *   input columns are in (pos, channel) order (the reference flattens (channel, pos): net.py:98,103).
 * ccz_value_out_f32: v[m] = tanh(fp16(b2 + sum_k h[m, k] * w2[k])), h [m, 256] fp16, w2 fp16 [256] (value_fc2 + tanh, net.py:107-109).
 * live_rows_dev as above (rows past *live_rows_dev are not computed) or NULL. Asynchronous on `stream`. */
#define CCZ_HEAD_POL_STRIDE 1536
#define CCZ_HEAD_VAL_STRIDE 640
int ccz_heads_conv1x1_f16(void *stream, const void *x_dev, const void *w32_dev, const void *bias32_f32_dev, void *pol_dev,
                          void *val_dev, int32_t n_boards, int32_t flags, const int32_t *live_boards_dev);
int ccz_fc_f16(void *stream, const void *a_dev, int32_t lda, const void *w_dev, const void *bias_f32_dev, void *c_dev, int32_t ldc,
               int32_t m, int32_t n, int32_t k, int32_t relu, const int32_t *live_rows_dev);
int ccz_value_out_f32(void *stream, const void *h_dev, const void *w2_dev, float b2, float *v_dev, int32_t m,
                      const int32_t *live_rows_dev);

/* The LAST residual layer of the tower with both head convolutions in its epilogue (group-of-16 rows only: flags must hold
 * CCZ_CONV_G16; bit 0 ReLU, CCZ_CONV_G16_EDGE_TILES as ccz_conv3x3_c256_f16): computes relu(conv3x3(x) + bias + residual) like
 * ccz_conv3x3_c256_f16 but does NOT store it -- each tile multiplies its finished rows with w32 while they are still in LDS and
 * writes pol / val exactly as ccz_heads_conv1x1_f16 would from the stored tensor (same operands, same MFMA chain: the same bits).
 * Saves one write and one read of the activation tensor (2 x 189 MB at 4096 boards). live_rows_dev == NULL: n_pixels rows at
 * x_dev / residual_dev, boards 0 .. n_pixels / 90 - 1 at pol_dev / val_dev (part / n_parts ignored). live_rows_dev != NULL: the
 * pointers of the WHOLE batch and part / n_parts as ccz_conv3x3_c256_f16_live; boards past *live_rows_dev are not written. */
int ccz_conv3x3_c256_heads_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev,
                               const void *residual_dev, const void *head_w32_dev, const void *head_bias32_f32_dev,
                               void *pol_dev, void *val_dev, int64_t n_pixels, int32_t flags,
                               const int32_t *live_rows_dev, int32_t part, int32_t n_parts);

#ifdef __cplusplus
}
#endif
#endif
