/*
 * xq_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the self-play rollout path of Symb0x76/ChineseChessZero:
 *   - the 2086-move action table          (reference tools.py:172-272)
 *   - board -> plane encoding             (reference tools.py:74-106, net.py:151-177)
 *   - the draw predicate                  (reference tools.py:109-123)
 *   - the Xiangqi rules the reference consumes from the third-party `cchess` module
 *     (call sites: mcts.py:111,116,125-126; net.py:154-157; game.py:148,201,208-216)
 *   - the sequential PUCT search          (reference mcts.py:7-233)
 *   - the self-play game loop             (reference game.py:133-237)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY STATUS
 *   search / encoding / table : PINNED -- checked against golden vectors produced by
 *       executing the reference's own mcts.py / tools.py (tests/golden/make_golden.py).
 *   rules (`cchess`)          : PARITY UNPINNED -- python-chinese-chess is un-vendored,
 *       un-pinned (reference README.md:21, .gitignore:3) and absent from this image; the
 *       reference has no tests. The rules below are standard Xiangqi as stated in
 *       DESIGN.md "Rules spec", anchored by the published start-position perft counts
 *       (44 / 1,920 / 79,666 / 3,290,240 / 133,312,995).
 */
#ifndef XQ_ORACLE_H
#define XQ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XQ_NSQ 90          /* square = file + 9*rank  (tools.py:91)            */
#define XQ_NMOVES 2086     /* action space            (tools.py:172-272)        */
#define XQ_MAX_LEGAL 128   /* upper bound on legal moves in one position        */
#define XQ_MAX_CHAIN 128   /* positions since the last capture (<=120 + margin) */
#define XQ_PLANES (17 * 7 * 10 * 9) /* net input elements (net.py:174-177)      */

/* piece types: numbering of python-chinese-chess PIECE_TYPES [unverified, see header] */
enum { XQ_PAWN = 1, XQ_CANNON = 2, XQ_ROOK = 3, XQ_KNIGHT = 4, XQ_BISHOP = 5, XQ_ADVISOR = 6, XQ_KING = 7 };
/* colours as in cchess: RED = True, BLACK = False */
enum { XQ_BLACK = 0, XQ_RED = 1 };
/* piece code in the mailbox: 0 empty, red = type, black = type + 8 */
#define XQ_PC(color, type) ((uint8_t)((color) == XQ_RED ? (type) : ((type) + 8)))
#define XQ_PTYPE(pc) ((pc) & 7)
#define XQ_PCOLOR(pc) (((pc) & 8) ? XQ_BLACK : XQ_RED)

typedef struct {
    uint8_t sq[XQ_NSQ];
    uint8_t turn;      /* side to move                                   */
    uint8_t pad;
} xq_pos;

typedef struct {
    xq_pos pos;
    int32_t halfmove;  /* plies since the last capture                   */
    int32_t ply;       /* plies since the start of the game              */
    /* positions since the last capture, oldest first, INCLUDING the current one */
    int32_t chain_len;
    xq_pos chain[XQ_MAX_CHAIN];
} xq_board;

/* ---- action table (tools.py:172-272) ---- */
void xq_table_init(void);
const char *xq_move_uci(int id);              /* id -> "a0a1"            */
int xq_move_from(int id);
int xq_move_to(int id);
int xq_move_id(int from, int to);             /* -1 if not in the table  */
int xq_flip_id(int id);                       /* file mirror (tools.py:133-166, collect.py:118-123) */

/* ---- rules ---- */
void xq_board_init(xq_board *b);              /* standard start position */
int xq_board_set(xq_board *b, const uint8_t sq[XQ_NSQ], int turn, int halfmove);
void xq_push(xq_board *b, int from, int to);
int xq_in_check(const xq_pos *p, int color);        /* reverse-ray test          */
int xq_in_check_slow(const xq_pos *p, int color);   /* independent: enemy movegen */
int xq_pseudo_moves(const xq_pos *p, int color, uint8_t *from, uint8_t *to);
/* legal move ids in `board.legal_moves` order: ASCENDING id (the canonical order) unless
 * xq_set_move_order installed a rank permutation (uint16[2086]; NULL = default) */
int xq_legal_ids(const xq_board *b, uint16_t *ids);
void xq_set_move_order(const uint16_t *rank);
/* major key by the mover's piece type (uint8[8], index 1..7; NULL = none): order = ascending (type_rank[type], rank[id]) */
void xq_set_type_order(const uint8_t *type_rank);
/* the sixty-move clock and the repetition history restart on pawn moves as well as on captures (default: captures only) */
void xq_set_pawn_move_resets_clock(int on);
/* perpetual check loses (twin of CCZ_RULE_PERPETUAL_CHECK; changes xq_outcome_winner of a fourfold repetition only) */
void xq_set_perpetual_check(int on);
int xq_perpetual_check_winner(const xq_board *b);
/* channel (0..6) of piece type t = 1..7 in decode_board (tools.py:100); NULL = type-1 */
void xq_set_plane_map(const uint8_t *plane_of_type);
int xq_insufficient_material(const xq_board *b);
int xq_repetition_count(const xq_board *b);
int xq_fourfold(const xq_board *b);
int xq_sixty_moves(const xq_board *b, int n_legal);
int xq_is_tie(const xq_board *b, int n_legal);          /* tools.py:109-123 */
int xq_is_game_over(const xq_board *b, int n_legal);
/* outcome winner: 1 RED, 0 BLACK, -1 none(draw) ; only meaningful if game over */
int xq_outcome_winner(const xq_board *b, int n_legal);
uint64_t xq_perft(xq_board *b, int depth);

/* ---- encoding ---- */
/* decode_board (tools.py:74-106): red[7][10][9], black[7][10][9] int8 one-hot */
void xq_decode_board(const xq_pos *p, int8_t *red, int8_t *black);
/* evaluator input on the search path (net.py:160-177): float [17*7*10*9], values 0/1 */
void xq_leaf_planes(const xq_pos *p, float *out);

/* ---- sequential PUCT search (mcts.py) ---- */
typedef struct xq_mcts xq_mcts;
/* evaluator callback == policy_value_fn (net.py:151-205): fills prob[k] for ids[k], returns value */
typedef float (*xq_eval_fn)(void *user, const xq_board *leaf, int k, const uint16_t *ids, float *prob);

xq_mcts *xq_mcts_new(int c_puct, int n_playout);
void xq_mcts_free(xq_mcts *t);
/* the evaluator's value is a float16 ndarray (reference CUDA path: autocast, net.py:178-189): Node.value is then
   accumulated in float16 (NEP 50). Default off = the float32 value of the reference's CPU path. */
void xq_mcts_set_value_f16(xq_mcts *t, int on);
/* wall-clock split of the playouts since the call (bench.py cpu_baseline: "net / rules / tree"); off by default */
void xq_mcts_set_timing(xq_mcts *t, int on);
void xq_mcts_timers(const xq_mcts *t, double *rules_s, double *tree_s);
void xq_mcts_playout(xq_mcts *t, const xq_board *root_board, xq_eval_fn fn, void *user);
/* runs n_playout playouts (mcts.py:131-166); returns k root children; visits/acts/probs sized >= XQ_MAX_LEGAL */
int xq_mcts_get_move_probs(xq_mcts *t, const xq_board *b, double temp, xq_eval_fn fn, void *user,
                           int32_t *acts, int32_t *visits, double *probs);
void xq_mcts_update_with_move(xq_mcts *t, int move_id); /* -1 resets (mcts.py:168-178) */
int xq_mcts_root_children(const xq_mcts *t, int32_t *acts, int32_t *visits, float *q, float *p);
int xq_mcts_root_visits(const xq_mcts *t);
int64_t xq_mcts_node_count(const xq_mcts *t);
/* step-wise twin of the batched engine: select (returns leaf board), then expand/backup */
int xq_mcts_select(xq_mcts *t, const xq_board *root_board, xq_board *leaf_out, int *depth_out);
void xq_mcts_expand_backup(xq_mcts *t, const xq_board *leaf, int k, const uint16_t *ids,
                           const float *prob, float value);

/* ---- deterministic device-mode sampler twin (DESIGN.md "Sampling") ---- */
double xq_det_log(double x);
double xq_det_exp(double x);
void xq_philox4x32(uint64_t key, uint64_t ctr_hi, uint64_t ctr_lo, uint32_t out[4]);
/* pi from visits (mcts.py:163-166) with the deterministic log/exp */
void xq_det_pi(const int32_t *visits, int k, double temp, double *pi);
/* Dirichlet-mixed move choice (mcts.py:216-224) on the per-board Philox stream; returns child index */
int xq_det_sample(uint64_t seed, uint64_t board_id, uint64_t move_no, const double *pi, int k,
                  double eps, double alpha, double *mixed_out);

#ifdef __cplusplus
}
#endif
#endif
