/*
 * ccz_ref.c -- CPU ORACLE (test infrastructure, NOT the product): the host twins of the lockstep entry points of
 * include/cczero.h, with the same signatures on HOST memory ("plus CPU-oracle twins (ccz_ref_*) with identical
 * signatures"). `stream` is ignored; every pointer the device version takes as a device pointer is a host pointer here.
 *
 * Each of the B boards is one sequential search of xq_mcts.c (reference mcts.py restated) on the rules of xq_rules.c; a lockstep
 * "simulation" is a loop over the boards. What each function follows:
 *   ccz_ref_select_leaves  mcts.py:101-111 (descent, board.copy / push) + net.py:154-177 (legal ids, input planes) + mcts.py:116-117
 *   ccz_ref_expand_backup  mcts.py:113-129 (expand | 0.0 | +-1.0, update_recursive)
 *   ccz_ref_finish_move    mcts.py:162-166 (pi), 216-224 (Dirichlet-mixed choice; device mode: the per-board Philox stream of
 *                          xq_sample.c), 168-178 (update_with_move), game.py:159 (temperature schedule), 201 (push), 208-219 (end, winner)
 * The rule tables of ccz_config (move order, plane map, clock / perpetual-check flags) are installed process-wide, as the one
 * `cchess` module of a reference process would be.
 *
 * Only tests/ may load this (tests/test_gpu_ref_twins.py drives libcczero.so and these twins through ONE function and compares).
 */
#include <stdlib.h>
#include <string.h>

#include "../include/cczero.h"
#include "xq_oracle.h"

typedef struct ccz_ref_engine {
    int B, n_playout, max_plies;
    float eps, alpha, temp;
    uint64_t seed, base;
    uint32_t flags;
    xq_mcts **t;
    xq_board *root, *leaf;
    int32_t *leaf_k, *depth, *plies;
    uint16_t *leaf_ids; /* [B][128] */
    uint8_t *status, *over;
    int8_t *winner;
    uint64_t *move_no;
} ccz_ref_engine;

static uint16_t g_rank[XQ_NMOVES];
static uint8_t g_trank[8], g_plane[8];

int ccz_ref_destroy(ccz_ref_engine *e)
{
    int b;
    if (!e) return -1;
    if (e->t) for (b = 0; b < e->B; b++) if (e->t[b]) xq_mcts_free(e->t[b]);
    free(e->t); free(e->root); free(e->leaf); free(e->leaf_k); free(e->depth); free(e->plies); free(e->leaf_ids);
    free(e->status); free(e->over); free(e->winner); free(e->move_no);
    free(e);
    return 0;
}

int ccz_ref_create(const ccz_config *cfg, ccz_ref_engine **out)
{
    ccz_ref_engine *e;
    int b, any = 0;
    if (!cfg || !out || cfg->n_boards < 1) return -1;
    xq_table_init();
    e = (ccz_ref_engine *)calloc(1, sizeof *e);
    if (!e) return -2;
    e->B = cfg->n_boards;
    e->n_playout = cfg->n_playout;
    e->max_plies = cfg->max_plies > 0 ? cfg->max_plies : 2048;
    e->eps = cfg->eps; e->alpha = cfg->alpha; e->temp = cfg->temp;
    e->seed = cfg->seed; e->base = cfg->board_id_base; e->flags = cfg->flags;
    /* the rule tables, process-wide */
    if (cfg->move_rank_host) { memcpy(g_rank, cfg->move_rank_host, sizeof g_rank); xq_set_move_order(g_rank); } else xq_set_move_order(NULL);
    for (b = 0; b < 8; b++) any |= cfg->type_rank[b];
    if (any) { memcpy(g_trank, cfg->type_rank, 8); xq_set_type_order(g_trank); } else xq_set_type_order(NULL);
    for (any = 0, b = 0; b < 8; b++) any |= cfg->plane_of_type[b];
    if (any) { memcpy(g_plane, cfg->plane_of_type, 8); xq_set_plane_map(g_plane); } else xq_set_plane_map(NULL);
    xq_set_pawn_move_resets_clock((cfg->rule_flags & CCZ_RULE_PAWN_MOVE_RESETS_CLOCK) != 0);
    xq_set_perpetual_check((cfg->rule_flags & CCZ_RULE_PERPETUAL_CHECK) != 0);
    e->t = (xq_mcts **)calloc((size_t)e->B, sizeof *e->t);
    e->root = (xq_board *)calloc((size_t)e->B, sizeof *e->root);
    e->leaf = (xq_board *)calloc((size_t)e->B, sizeof *e->leaf);
    e->leaf_k = (int32_t *)calloc((size_t)e->B, sizeof *e->leaf_k);
    e->depth = (int32_t *)calloc((size_t)e->B, sizeof *e->depth);
    e->plies = (int32_t *)calloc((size_t)e->B, sizeof *e->plies);
    e->leaf_ids = (uint16_t *)calloc((size_t)e->B * XQ_MAX_LEGAL, sizeof *e->leaf_ids);
    e->status = (uint8_t *)calloc((size_t)e->B, 1);
    e->over = (uint8_t *)calloc((size_t)e->B, 1);
    e->winner = (int8_t *)calloc((size_t)e->B, 1);
    e->move_no = (uint64_t *)calloc((size_t)e->B, sizeof *e->move_no);
    if (!e->t || !e->root || !e->leaf || !e->leaf_k || !e->depth || !e->plies || !e->leaf_ids || !e->status || !e->over || !e->winner || !e->move_no) {
        ccz_ref_destroy(e);
        return -2;
    }
    for (b = 0; b < e->B; b++) {
        e->t[b] = xq_mcts_new((int)cfg->c_puct, cfg->n_playout);
        if (cfg->flags & CCZ_FLAG_VALUE_F16) xq_mcts_set_value_f16(e->t[b], 1);
        xq_board_init(&e->root[b]);
        e->status[b] = CCZ_LEAF_NONE;
        e->winner[b] = -1;
    }
    *out = e;
    return 0;
}

int ccz_ref_set_position(ccz_ref_engine *e, void *stream, int32_t board, const uint8_t *sq_host, int32_t turn, int32_t halfmove)
{
    (void)stream;
    if (!e || board < 0 || board >= e->B || !sq_host) return -1;
    xq_board_set(&e->root[board], sq_host, turn, halfmove);
    xq_mcts_update_with_move(e->t[board], -1);
    e->over[board] = 0; e->winner[board] = -1; e->plies[board] = 0; e->move_no[board] = 0; e->status[board] = CCZ_LEAF_NONE;
    return 0;
}

/* leaf_input_f16_host: [B][17][7][10][9] IEEE half (0x3C00 = 1.0); only groups 7, 15, 16 can be non-zero (net.py:160-173) */
int ccz_ref_select_leaves(ccz_ref_engine *e, void *stream, void *leaf_input_f16_host)
{
    int b, i;
    float planes[XQ_PLANES];
    uint16_t *out = (uint16_t *)leaf_input_f16_host;
    (void)stream;
    if (!e) return -1;
    for (b = 0; b < e->B; b++) {
        int k, end, tie, d = 0;
        if (e->over[b]) { e->status[b] = CCZ_LEAF_NONE; continue; }
        xq_mcts_select(e->t[b], &e->root[b], &e->leaf[b], &d);
        k = xq_legal_ids(&e->leaf[b], e->leaf_ids + (size_t)b * XQ_MAX_LEGAL);
        end = xq_is_game_over(&e->leaf[b], k);
        tie = xq_is_tie(&e->leaf[b], k);
        e->leaf_k[b] = k;
        e->depth[b] = d;
        e->status[b] = (uint8_t)((!end && !tie) ? CCZ_LEAF_EXPAND : (end && tie) ? CCZ_LEAF_DRAW : CCZ_LEAF_LOSS);
        if (out) {
            xq_leaf_planes(&e->leaf[b].pos, planes);
            for (i = 0; i < XQ_PLANES; i++) out[(size_t)b * XQ_PLANES + i] = planes[i] != 0.0f ? 0x3C00u : 0u;
        }
    }
    return 0;
}

int ccz_ref_expand_backup(ccz_ref_engine *e, void *stream, const float *prob_host, const float *value_host)
{
    int b, i;
    float pr[XQ_MAX_LEGAL];
    (void)stream;
    if (!e || !prob_host || !value_host) return -1;
    for (b = 0; b < e->B; b++) {
        const uint16_t *ids = e->leaf_ids + (size_t)b * XQ_MAX_LEGAL;
        if (e->status[b] == CCZ_LEAF_NONE) continue;
        for (i = 0; i < e->leaf_k[b]; i++) pr[i] = prob_host[(size_t)b * XQ_NMOVES + ids[i]]; /* exp(log_act_probs)[legal] (net.py:202-205) */
        xq_mcts_expand_backup(e->t[b], &e->leaf[b], e->leaf_k[b], ids, pr, value_host[b]);
        e->status[b] = CCZ_LEAF_NONE;
    }
    return 0;
}

int ccz_ref_finish_move(ccz_ref_engine *e, void *stream, const int32_t *forced_moves_host, const double *temps_host,
                        int32_t *moves_out_host, int32_t keep_tree)
{
    int b;
    (void)stream;
    if (!e) return -1;
    for (b = 0; b < e->B; b++) {
        int32_t acts[XQ_MAX_LEGAL], visits[XQ_MAX_LEGAL];
        float q[XQ_MAX_LEGAL], p[XQ_MAX_LEGAL];
        double pi[XQ_MAX_LEGAL], mixed[XQ_MAX_LEGAL];
        int k, mv, n, want = forced_moves_host ? forced_moves_host[b] : -1;
        if (moves_out_host) moves_out_host[b] = -1;
        if (e->over[b]) continue;
        k = xq_mcts_root_children(e->t[b], acts, visits, q, p);
        if (e->plies[b] >= e->max_plies) { /* documented cap: adjudicated a draw */
            e->over[b] = 1; e->winner[b] = -1;
            continue;
        }
        if (want >= 0) mv = want;
        else {
            /* game.py:157-159: temp for the first 30 moves, then max(0.1, temp / 2) */
            const double half = (double)e->temp * 0.5;
            const double temp = temps_host ? temps_host[b] : ((e->plies[b] + 1) <= 30 ? (double)e->temp : (half > 0.1 ? half : 0.1));
            if (k <= 0) return -3;
            xq_det_pi(visits, k, temp, pi);
            mv = acts[xq_det_sample(e->seed, e->base + (uint64_t)b, e->move_no[b], pi, k, (double)e->eps, (double)e->alpha, mixed)];
        }
        xq_mcts_update_with_move(e->t[b], keep_tree ? mv : -1);
        xq_push(&e->root[b], xq_move_from(mv), xq_move_to(mv));
        e->plies[b] += 1;
        e->move_no[b] += 1;
        {
            uint16_t ids[XQ_MAX_LEGAL];
            n = xq_legal_ids(&e->root[b], ids);
        }
        if (xq_is_game_over(&e->root[b], n) || xq_is_tie(&e->root[b], n)) { /* game.py:208 */
            e->over[b] = 1;
            e->winner[b] = (int8_t)xq_outcome_winner(&e->root[b], n);
        }
        e->status[b] = CCZ_LEAF_NONE;
        if (moves_out_host) moves_out_host[b] = mv;
    }
    return 0;
}

int ccz_ref_root_children(ccz_ref_engine *e, void *stream, int32_t *k_host, uint16_t *acts_host, int32_t *visits_host, float *q_host,
                          float *prior_host, int32_t *root_visits_host)
{
    int b, i;
    (void)stream;
    if (!e) return -1;
    for (b = 0; b < e->B; b++) {
        int32_t acts[XQ_MAX_LEGAL], visits[XQ_MAX_LEGAL];
        float q[XQ_MAX_LEGAL], p[XQ_MAX_LEGAL];
        const int k = e->over[b] ? 0 : xq_mcts_root_children(e->t[b], acts, visits, q, p);
        if (k_host) k_host[b] = k;
        if (root_visits_host) root_visits_host[b] = e->over[b] ? 0 : xq_mcts_root_visits(e->t[b]);
        for (i = 0; i < XQ_MAX_LEGAL; i++) {
            const size_t o = (size_t)b * XQ_MAX_LEGAL + (size_t)i;
            if (acts_host) acts_host[o] = i < k ? (uint16_t)acts[i] : 0;
            if (visits_host) visits_host[o] = i < k ? visits[i] : 0;
            if (q_host) q_host[o] = i < k ? q[i] : 0.0f;
            if (prior_host) prior_host[o] = i < k ? p[i] : 0.0f;
        }
    }
    return 0;
}

int ccz_ref_game_status(ccz_ref_engine *e, void *stream, uint8_t *over_host, int8_t *winner_host, int32_t *plies_host, uint8_t *turn_host)
{
    int b;
    (void)stream;
    if (!e) return -1;
    for (b = 0; b < e->B; b++) {
        if (over_host) over_host[b] = e->over[b];
        if (winner_host) winner_host[b] = e->over[b] ? e->winner[b] : (int8_t)-1;
        if (plies_host) plies_host[b] = e->plies[b];
        if (turn_host) turn_host[b] = e->root[b].pos.turn;
    }
    return 0;
}

int ccz_ref_leaf_info(ccz_ref_engine *e, void *stream, uint8_t *status_host, int32_t *k_host, uint16_t *ids_host, int32_t *depth_host)
{
    int b;
    (void)stream;
    if (!e) return -1;
    for (b = 0; b < e->B; b++) {
        if (status_host) status_host[b] = e->status[b];
        if (k_host) k_host[b] = e->status[b] == CCZ_LEAF_NONE ? 0 : e->leaf_k[b];
        if (depth_host) depth_host[b] = e->depth[b];
        if (ids_host) memcpy(ids_host + (size_t)b * XQ_MAX_LEGAL, e->leaf_ids + (size_t)b * XQ_MAX_LEGAL, sizeof(uint16_t) * XQ_MAX_LEGAL);
    }
    return 0;
}

int ccz_ref_root_positions(ccz_ref_engine *e, void *stream, uint8_t *sq_host)
{
    int b;
    (void)stream;
    if (!e || !sq_host) return -1;
    for (b = 0; b < e->B; b++) {
        memset(sq_host + (size_t)b * 96, 0, 96);
        memcpy(sq_host + (size_t)b * 96, e->root[b].pos.sq, XQ_NSQ);
    }
    return 0;
}
