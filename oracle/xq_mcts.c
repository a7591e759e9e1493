/*
 * xq_mcts.c -- CPU ORACLE (test infrastructure, NOT the product). See xq_oracle.h.
 *
 * Sequential PUCT search restating reference mcts.py:7-233, including the NumPy (>=2, NEP 50)
 * dtype behaviour of its arithmetic, which decides visit counts bit for bit:
 *   - Node.value starts as Python int 0; a backup with the net's ndarray(1,1) float32 value makes
 *     it float32 and every later update runs in float32; a node that only ever receives terminal
 *     values (Python floats) keeps a Python double (mcts.py:63-78).
 *   - puct_value (mcts.py:41-52): `c_puct * prob` is float32 (int * np.float32), np.sqrt(int) is
 *     float64, so the product, the quotient and the final sum are float64.
 *   - select (mcts.py:54-61): Python max() keeps the FIRST maximal child in insertion order;
 *     unvisited children score +inf.
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime (the optional playout timers) */
#include "xq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { V_INT0 = 0, V_PYFLOAT = 1, V_F32 = 2 };

/* IEEE binary16 rounding (round to nearest even) of a float, returned as a float: what a NumPy float16 operation
   leaves (its half loops compute in float32 and round once; double rounding is innocuous, 24 >= 2*11+2). */
static float f16_round(float x)
{
    union { float f; uint32_t u; } a;
    uint32_t sign, m;
    a.f = x;
    sign = a.u & 0x80000000u;
    m = a.u & 0x7fffffffu;
    if (m >= 0x7f800000u) return x;                 /* inf / nan */
    if (m >= 0x477ff000u) { a.u = sign | 0x7f800000u; return a.f; } /* >= 65520 rounds to inf */
    if (m < 0x38800000u) {                          /* below 2^-14: half subnormals, quantum 2^-24 */
        float q = nearbyintf(fabsf(x) * 16777216.0f) / 16777216.0f; /* default rounding mode = nearest even */
        return sign ? -q : q;
    }
    m += 0xfffu + ((m >> 13) & 1u);
    m &= ~0x1fffu;
    a.u = sign | m;
    return a.f;
}

typedef struct node {
    struct node *parent;
    int nchild;
    int32_t *act;
    struct node **child;
    int visits;
    int vkind;
    double vd; /* value when V_PYFLOAT */
    float vf;  /* value when V_F32     */
    float prob;
} node;

struct xq_mcts {
    node *root;
    int c_puct;
    int n_playout;
    int value_f16;  /* the evaluator's value is a float16 ndarray (reference CUDA/autocast path, net.py:178-189) */
    node *cur_leaf; /* set by xq_mcts_select */
    int64_t live_nodes;
    /* optional wall-clock split of a playout (bench.py's cpu_baseline leg): seconds spent in the rules (push along the path,
       legal-move generation, game-end / draw predicates) and in the tree (PUCT scan, expansion, backup); the evaluator callback
       is timed by the caller. Off by default: no clock call is made. */
    int timing;
    double t_rules, t_tree;
};

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void xq_mcts_set_timing(xq_mcts *t, int on) { t->timing = on != 0; t->t_rules = t->t_tree = 0.0; }
void xq_mcts_timers(const xq_mcts *t, double *rules_s, double *tree_s)
{
    if (rules_s) *rules_s = t->t_rules;
    if (tree_s) *tree_s = t->t_tree;
}

static node *node_new(xq_mcts *t, node *parent, float prob)
{
    node *n = (node *)calloc(1, sizeof *n);
    if (!n) abort();
    n->parent = parent;
    n->prob = prob;
    t->live_nodes++;
    return n;
}

static void node_free(xq_mcts *t, node *n)
{
    int i;
    if (!n) return;
    for (i = 0; i < n->nchild; i++) node_free(t, n->child[i]);
    free(n->act);
    free(n->child);
    free(n);
    t->live_nodes--;
}

xq_mcts *xq_mcts_new(int c_puct, int n_playout)
{
    xq_mcts *t = (xq_mcts *)calloc(1, sizeof *t);
    if (!t) abort();
    xq_table_init();
    t->c_puct = c_puct;
    t->n_playout = n_playout;
    t->root = node_new(t, NULL, 1.0f); /* Node(None, 1.0)  mcts.py:94 */
    return t;
}

void xq_mcts_free(xq_mcts *t)
{
    if (!t) return;
    node_free(t, t->root);
    free(t);
}

/* Node.puct_value  mcts.py:41-52 */
static double puct_value(const node *n, int c_puct)
{
    float cp;
    double u, q;
    if (n->visits == 0) return INFINITY;
    cp = (float)c_puct * n->prob;                                   /* int * np.float32 -> float32 */
    u = (double)cp * sqrt((double)n->parent->visits) / (double)(1 + n->visits); /* float64        */
    q = n->vkind == V_F32 ? (double)n->vf : (n->vkind == V_PYFLOAT ? n->vd : 0.0);
    return q + u;
}

/* Node.update  mcts.py:63-71 ; leaf value is either an ndarray float32 (is_f32) or a Python float.
   f16: the ndarray is float16 instead (autocast path): every operation of `value += 1.0*(leaf_value - value)/visits`
   then rounds to float16 (NEP 50: Python scalars are weak, so `visits` is converted to float16 too). */
static void node_update(node *n, int is_f32, float lf, double ld, int f16)
{
    n->visits += 1;
    if (f16 && (is_f32 || n->vkind == V_F32)) {
        float v = f16_round(is_f32 ? lf : (float)ld);
        float cur = n->vkind == V_F32 ? n->vf : (n->vkind == V_PYFLOAT ? f16_round((float)n->vd) : 0.0f);
        float delta = f16_round(v - cur);
        delta = f16_round(1.0f * delta);
        delta = f16_round(delta / f16_round((float)n->visits));
        n->vf = f16_round(cur + delta);
        n->vkind = V_F32; /* vf holds a float16-representable number */
    } else if (is_f32 || n->vkind == V_F32) {
        /* any float32 ndarray operand makes the whole expression float32 (Python scalars are weak) */
        float v = is_f32 ? lf : (float)ld;
        float cur = n->vkind == V_F32 ? n->vf : (n->vkind == V_PYFLOAT ? (float)n->vd : 0.0f);
        float delta = v - cur;
        delta = 1.0f * delta;
        delta = delta / (float)n->visits;
        n->vf = cur + delta;
        n->vkind = V_F32;
    } else {
        double cur = n->vkind == V_PYFLOAT ? n->vd : 0.0;
        n->vd = cur + 1.0 * (ld - cur) / (double)n->visits;
        n->vkind = V_PYFLOAT;
    }
}

/* Node.update_recursive  mcts.py:73-78 : parent first, with the negated value */
static void node_update_recursive(node *n, int is_f32, float lf, double ld, int f16)
{
    if (n->parent) node_update_recursive(n->parent, is_f32, -lf, -ld, f16);
    node_update(n, is_f32, lf, ld, f16);
}

void xq_mcts_set_value_f16(xq_mcts *t, int on) { t->value_f16 = on != 0; }

/* descend from the root pushing moves (mcts.py:105-111); returns the leaf */
int xq_mcts_select(xq_mcts *t, const xq_board *root_board, xq_board *leaf_out, int *depth_out)
{
    node *n = t->root;
    int depth = 0;
    double t0 = t->timing ? now_s() : 0.0, tr = 0.0;
    *leaf_out = *root_board; /* board.copy()  mcts.py:151 */
    while (n->nchild != 0) {
        int i, best = 0;
        double bestv = puct_value(n->child[0], t->c_puct);
        for (i = 1; i < n->nchild; i++) {
            double v = puct_value(n->child[i], t->c_puct);
            if (v > bestv) { bestv = v; best = i; } /* strict: first maximum wins */
        }
        if (t->timing) {
            double p0 = now_s();
            xq_push(leaf_out, xq_move_from(n->act[best]), xq_move_to(n->act[best]));
            tr += now_s() - p0;
        } else
            xq_push(leaf_out, xq_move_from(n->act[best]), xq_move_to(n->act[best]));
        n = n->child[best];
        depth++;
    }
    if (t->timing) { t->t_rules += tr; t->t_tree += now_s() - t0 - tr; }
    t->cur_leaf = n;
    if (depth_out) *depth_out = depth;
    return 0;
}

/* second half of playout (mcts.py:113-129) for the leaf chosen by xq_mcts_select */
void xq_mcts_expand_backup(xq_mcts *t, const xq_board *leaf, int k, const uint16_t *ids,
                           const float *prob, float value)
{
    node *n = t->cur_leaf;
    double t0 = t->timing ? now_s() : 0.0, t1;
    int end = xq_is_game_over(leaf, k), tie = xq_is_tie(leaf, k);
    t1 = t->timing ? now_s() : 0.0;
    if (!end && !tie) {
        int i;
        /* Node.expand  mcts.py:31-39 : one child per (action, prob), evaluator order */
        n->act = (int32_t *)malloc(sizeof(int32_t) * (size_t)k);
        n->child = (node **)malloc(sizeof(node *) * (size_t)k);
        if (!n->act || !n->child) abort();
        for (i = 0; i < k; i++) {
            n->act[i] = ids[i];
            n->child[i] = node_new(t, n, prob[i]);
        }
        n->nchild = k;
        node_update_recursive(n, 1, -value, 0.0, t->value_f16);
    } else if (end && tie) {
        node_update_recursive(n, 0, 0.0f, -0.0, t->value_f16); /* leaf_value = 0.0 */
    } else {
        /* winner = RED if outcome().winner else BLACK ; +1 if winner == board.turn  (mcts.py:125-126) */
        int w = xq_outcome_winner(leaf, k);
        int winner = (w == 1) ? XQ_RED : XQ_BLACK; /* None is falsy -> BLACK */
        double lv = winner == leaf->pos.turn ? 1.0 : -1.0;
        node_update_recursive(n, 0, 0.0f, -lv, t->value_f16);
    }
    if (t->timing) { t->t_rules += t1 - t0; t->t_tree += now_s() - t1; }
    t->cur_leaf = NULL;
}

void xq_mcts_playout(xq_mcts *t, const xq_board *root_board, xq_eval_fn fn, void *user)
{
    xq_board leaf;
    uint16_t ids[XQ_MAX_LEGAL];
    float prob[XQ_MAX_LEGAL];
    float value = 0.0f;
    int k, depth;
    xq_mcts_select(t, root_board, &leaf, &depth);
    if (t->timing) {
        double p0 = now_s();
        k = xq_legal_ids(&leaf, ids);
        t->t_rules += now_s() - p0;
    } else
        k = xq_legal_ids(&leaf, ids);
    /* the reference evaluates the net on terminal leaves too and discards the result (mcts.py:114);
       that is results-neutral, so the evaluator is only called where its output is used */
    if (!xq_is_game_over(&leaf, k) && !xq_is_tie(&leaf, k)) value = fn(user, &leaf, k, ids, prob);
    xq_mcts_expand_backup(t, &leaf, k, ids, prob, value);
}

int xq_mcts_root_children(const xq_mcts *t, int32_t *acts, int32_t *visits, float *q, float *p)
{
    int i;
    for (i = 0; i < t->root->nchild; i++) {
        const node *c = t->root->child[i];
        if (acts) acts[i] = t->root->act[i];
        if (visits) visits[i] = c->visits;
        if (q) q[i] = c->vkind == V_F32 ? c->vf : (c->vkind == V_PYFLOAT ? (float)c->vd : 0.0f);
        if (p) p[i] = c->prob;
    }
    return t->root->nchild;
}

int xq_mcts_root_visits(const xq_mcts *t) { return t->root->visits; }
int64_t xq_mcts_node_count(const xq_mcts *t) { return t->live_nodes; }

/* MCTS.get_move_probs  mcts.py:131-166 */
int xq_mcts_get_move_probs(xq_mcts *t, const xq_board *b, double temp, xq_eval_fn fn, void *user,
                           int32_t *acts, int32_t *visits, double *probs)
{
    int i, k;
    double mx = -INFINITY, sum = 0.0;
    for (i = 0; i < t->n_playout; i++) xq_mcts_playout(t, b, fn, user);
    k = xq_mcts_root_children(t, acts, visits, NULL, NULL);
    /* softmax(1.0/temp * np.log(np.array(visits) + 1e-10))  (mcts.py:165, tools.py:126-129) */
    for (i = 0; i < k; i++) {
        probs[i] = 1.0 / temp * log((double)visits[i] + 1e-10);
        if (probs[i] > mx) mx = probs[i];
    }
    for (i = 0; i < k; i++) { probs[i] = exp(probs[i] - mx); sum += probs[i]; }
    for (i = 0; i < k; i++) probs[i] /= sum;
    return k;
}

/* MCTS.update_with_move  mcts.py:168-178 */
void xq_mcts_update_with_move(xq_mcts *t, int move_id)
{
    node *r = t->root, *keep = NULL;
    int i;
    for (i = 0; i < r->nchild; i++)
        if (r->act[i] == move_id) { keep = r->child[i]; r->child[i] = NULL; }
    node_free(t, r);
    if (keep) {
        keep->parent = NULL;
        t->root = keep;
    } else {
        t->root = node_new(t, NULL, 1.0f);
    }
}
