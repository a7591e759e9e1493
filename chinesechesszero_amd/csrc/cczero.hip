// cczero.hip -- C ABI (include/cczero.h) of the gfx950 lockstep self-play rollout engine.
//
// Host side: memory layout in HBM, launches on the caller's stream, error reporting. There is no
// CPU fallback anywhere in this file: without a usable HIP device every compute entry point fails.
#include "../../include/cczero.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "cczero_kernels.h"
#include "cczero_netops.h"
#include "cczero_conv.h"
#include "cczero_conv_small.h"
#include "cczero_conv_g16.h"
#include "cczero_conv_g16e.h"
#include "cczero_conv_g16p.h"
#include "cczero_heads.h"

using namespace ccz;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

constexpr Tables h_tab = make_tables();

template <typename T>
hipError_t dalloc(T **p, size_t n, std::vector<void *> &owned, size_t &total)
{
    void *q = nullptr;
    const size_t bytes = n * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    e = hipMemset(q, 0, bytes ? bytes : 16);
    if (e != hipSuccess) return e;
    owned.push_back(q);
    total += bytes;
    *p = (T *)q;
    return hipSuccess;
}

} // namespace

struct ccz_engine {
    ccz_config cfg;
    Dev d;
    std::vector<void *> owned;
    size_t bytes = 0;
    // staging (device) for the sync'ing accessors
    int32_t *st_k = nullptr, *st_visits = nullptr, *st_rootn = nullptr;
    uint16_t *st_acts = nullptr;
    float *st_q = nullptr, *st_p = nullptr;
    double *st_pi = nullptr, *st_temps = nullptr;
    long long *st_rowbase = nullptr;
    uint8_t *st_mask = nullptr, *st_sq = nullptr;
    std::vector<BoardMeta> h_meta;
    int active = 0; // boards 0 .. active - 1 are searched; the rest are scout slots (ccz_set_scouts); 0 = all
};

#define ACTIVE(e) ((unsigned)((e)->active > 0 ? (e)->active : (e)->d.B))

extern "C" {

int ccz_abi_version(void) { return CCZ_ABI_VERSION; }
const char *ccz_last_error(void) { return g_err; }

int ccz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ccz_action_table(char *uci, uint8_t *from, uint8_t *to)
{
    for (int i = 0; i < kNMoves; ++i) {
        const int fr = h_tab.from[i], t = h_tab.to[i];
        if (uci) {
            uci[i * 5 + 0] = (char)('a' + fr % 9);
            uci[i * 5 + 1] = (char)('0' + fr / 9);
            uci[i * 5 + 2] = (char)('a' + t % 9);
            uci[i * 5 + 3] = (char)('0' + t / 9);
            uci[i * 5 + 4] = 0;
        }
        if (from) from[i] = (uint8_t)fr;
        if (to) to[i] = (uint8_t)t;
    }
    return 0;
}

int ccz_flip_map(int32_t *flip)
{
    if (!flip) return fail(-1, "ccz_flip_map: null output");
    for (int i = 0; i < kNMoves; ++i) flip[i] = h_tab.flip[i];
    return 0;
}

int ccz_create(const ccz_config *cfg, ccz_engine **out)
{
    if (!cfg || !out) return fail(-1, "ccz_create: null argument");
    if (cfg->n_boards <= 0) return fail(-1, "ccz_create: n_boards must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(-3, "ccz_create: no HIP device available (the engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(-1, "ccz_create: device %d out of range (%d devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    ccz_engine *e = new (std::nothrow) ccz_engine();
    if (!e) return fail(-4, "ccz_create: out of host memory");
    e->cfg = *cfg;
    Dev &d = e->d;
    d.B = cfg->n_boards;
    // nodes per pool half: every playout creates <= ~k children; the retained subtree adds to it
    const int n_play = cfg->n_playout > 0 ? cfg->n_playout : 400;
    // nodes per pool half: a move adds <= n_playout expansions of ~40 children; the subtree kept across moves adds to
    // it (concentrated searches keep most of the tree: 97 k nodes were seen after 45 moves at 400 sims). Sized for
    // the HBM at hand (288 GB): 512 nodes per simulation of a move, 39 GB at 4096 boards x 400 sims.
    d.cap = cfg->max_nodes > 0 ? cfg->max_nodes : (n_play + 64) * 512;
    if (d.cap < 130) d.cap = 130; // root + one full expansion (the root's children are prefetched unconditionally)
    d.reserve = n_play * 128 < d.cap / 2 ? n_play * 128 : d.cap / 2;
    if (cfg->reserve_nodes > 0) d.reserve = cfg->reserve_nodes < d.cap - 129 ? cfg->reserve_nodes : d.cap - 129;
    d.maxd = cfg->max_depth > 0 ? cfg->max_depth : kMaxDepth;
    if (d.maxd > kMaxDepth) { delete e; return fail(-1, "ccz_create: max_depth %d exceeds the compiled limit %d", d.maxd, kMaxDepth); }
    if (d.maxd < 64) { delete e; return fail(-1, "ccz_create: max_depth must be >= 64"); }
    d.max_plies = cfg->max_plies > 0 ? cfg->max_plies : 2048;
    d.pi_cap = d.max_plies * 80; // sparse pi entries: opening positions have 44 legal moves, open middlegames 50-70
    d.c_puct = cfg->c_puct;
    d.eps = (double)cfg->eps;
    d.alpha = (double)cfg->alpha;
    d.temp = (double)cfg->temp;
    d.flags = cfg->flags;
    d.seed = cfg->seed;
    d.board_id_base = cfg->board_id_base;
    // ---- run-time rule tables (ABI 2)
    if ((cfg->flags & CCZ_FLAG_CACHE_VERIFY) && !cfg->eval_cache_log2) { delete e; return fail(-1, "ccz_create: CCZ_FLAG_CACHE_VERIFY without an evaluation cache (eval_cache_log2)"); }
    if (cfg->flags & ~(CCZ_FLAG_REFERENCE_QUIRKS | CCZ_FLAG_NO_MIRROR | CCZ_FLAG_VALUE_F16 | CCZ_FLAG_CACHE_VERIFY | CCZ_FLAG_STRICT)) { delete e; return fail(-1, "ccz_create: unknown flags 0x%x", cfg->flags); }
    if (cfg->eval_cache_log2 && (cfg->eval_cache_log2 < 10 || cfg->eval_cache_log2 > 28)) { delete e; return fail(-1, "ccz_create: eval_cache_log2 must be 0 (no cache) or 10..28"); }
    if (cfg->rule_flags & ~(CCZ_RULE_PERPETUAL_CHECK | CCZ_RULE_PAWN_MOVE_RESETS_CLOCK)) { delete e; return fail(-1, "ccz_create: unknown rule_flags 0x%x", cfg->rule_flags); }
    d.rule_flags = cfg->rule_flags;
    {
        uint8_t pot[8] = {0, 0, 1, 2, 3, 4, 5, 6};
        bool all_zero = true;
        for (int t = 0; t < 8; ++t) all_zero = all_zero && cfg->plane_of_type[t] == 0;
        if (!all_zero) {
            unsigned seen = 0;
            for (int t = 1; t <= 7; ++t) {
                const int c = cfg->plane_of_type[t];
                if (c > 6 || (seen >> c & 1u)) { delete e; return fail(-1, "ccz_create: plane_of_type[1..7] must be a permutation of 0..6"); }
                seen |= 1u << c;
                pot[t] = (uint8_t)c;
            }
        }
        d.chanpack = d.typepack = 0;
        for (int t = 1; t <= 7; ++t) {
            d.chanpack |= (uint32_t)pot[t] << (3 * t);
            d.typepack |= (uint32_t)(t - 1) << (3 * pot[t]);
        }
    }
    d.trankpack = 0;
    for (int t = 1; t <= 7; ++t) {
        if (cfg->type_rank[t] > 7) { delete e; return fail(-1, "ccz_create: type_rank[%d] must be 0..7", t); }
        d.trankpack |= (uint32_t)cfg->type_rank[t] << (3 * t);
    }
    std::vector<uint16_t> h_rank, h_unrank;
    if (cfg->move_rank_host) {
        h_rank.assign(cfg->move_rank_host, cfg->move_rank_host + kNMoves);
        h_unrank.assign(kNMoves, 0xffff);
        for (int i = 0; i < kNMoves; ++i) {
            if (h_rank[i] >= kNMoves || h_unrank[h_rank[i]] != 0xffff) { delete e; return fail(-1, "ccz_create: move_rank_host must be a permutation of 0..2085 (entry %d)", i); }
            h_unrank[h_rank[i]] = (uint16_t)i;
        }
    }
    const size_t B = (size_t)d.B;
    hipError_t he = hipSuccess;
#define ALLOC(ptr, n)                                                        \
    if (he == hipSuccess) he = dalloc(&(ptr), (size_t)(n), e->owned, e->bytes)
    ALLOC(d.nodeA, B * 2 * d.cap);
    ALLOC(d.nodeB, B * 2 * d.cap);
    ALLOC(d.meta, B);
    ALLOC(d.root_sq, B * 96);
    ALLOC(d.chain, B * kChainCap);
    ALLOC(d.chain_chk, B * 2);
    ALLOC(d.path, B * d.maxd);
    ALLOC(d.path_len, B);
    ALLOC(d.leaf_ids, B * kMaxLegal);
    ALLOC(d.leaf_k, B);
    ALLOC(d.leaf_status, B);
    ALLOC(d.leaf_key, B);
    d.cache = nullptr;
    d.cache_mask = 0;
    if (cfg->eval_cache_log2) {
        const size_t slots = (size_t)1 << cfg->eval_cache_log2;
        d.cache_mask = (uint32_t)(slots - 1);
        ALLOC(d.cache, slots);
        ALLOC(d.claim, slots);
        ALLOC(d.cslot, B);
        ALLOC(d.ctag, B);
        ALLOC(d.cstate, B);
        ALLOC(d.cins, B);
        ALLOC(d.cver, B);
        ALLOC(d.crep, B);
        ALLOC(d.row_of, B);
        ALLOC(d.vleaf, B);
        if (he == hipSuccess) he = hipMemset(d.claim, 0x7f, slots * 4);
    }
    ALLOC(d.rec_sq, B * d.max_plies * 96);
    ALLOC(d.rec_turn, B * d.max_plies);
    ALLOC(d.rec_k, B * d.max_plies);
    ALLOC(d.rec_off, B * d.max_plies);
    ALLOC(d.rec_ids, B * d.pi_cap);
    ALLOC(d.rec_pi, B * d.pi_cap);
    ALLOC(d.stats, B);
    ALLOC(d.err, 4);
    ALLOC(d.half, 4);
    ALLOC(d.prior128, B * kMaxLegal);
    ALLOC(d.stamps, B * 16);
    ALLOC(e->st_k, B);
    ALLOC(e->st_visits, B * kMaxLegal);
    ALLOC(e->st_rootn, B);
    ALLOC(e->st_acts, B * kMaxLegal);
    ALLOC(e->st_q, B * kMaxLegal);
    ALLOC(e->st_p, B * kMaxLegal);
    ALLOC(e->st_pi, B * kMaxLegal);
    ALLOC(e->st_temps, B);
    ALLOC(e->st_rowbase, B);
    ALLOC(e->st_mask, B);
    ALLOC(e->st_sq, 96);
    uint16_t *d_rank = nullptr, *d_unrank = nullptr;
    if (!h_rank.empty()) {
        ALLOC(d_rank, kNMoves);
        ALLOC(d_unrank, kNMoves);
        if (he == hipSuccess) he = hipMemcpy(d_rank, h_rank.data(), kNMoves * 2, hipMemcpyHostToDevice);
        if (he == hipSuccess) he = hipMemcpy(d_unrank, h_unrank.data(), kNMoves * 2, hipMemcpyHostToDevice);
    }
    d.rank = d_rank;
    d.unrank = d_unrank;
#undef ALLOC
    if (he != hipSuccess) {
        for (void *p : e->owned) (void)hipFree(p);
        const size_t got = e->bytes;
        delete e;
        return fail(-2, "ccz_create: device allocation failed after %zu bytes: %s", got, hipGetErrorString(he));
    }
    e->h_meta.resize(B);
    hipLaunchKernelGGL(k_reset, dim3(d.B), dim3(64), 0, 0, d, (const uint8_t *)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    *out = e;
    return 0;
}

int ccz_destroy(ccz_engine *e)
{
    if (!e) return 0;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void *p : e->owned) (void)hipFree(p);
    delete e;
    return 0;
}

#define NEED(e) do { if (!(e)) return fail(-1, "%s: null engine", __func__); } while (0)

int ccz_reset(ccz_engine *e, void *stream, const uint8_t *mask_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *mask = nullptr;
    if (mask_host) {
        HIP_TRY(hipMemcpyAsync(e->st_mask, mask_host, (size_t)e->d.B, hipMemcpyHostToDevice, s));
        mask = e->st_mask;
    }
    hipLaunchKernelGGL(k_reset, dim3(e->d.B), dim3(64), 0, s, e->d, mask);
    HIP_TRY(hipGetLastError());
    if (mask_host) HIP_TRY(hipStreamSynchronize(s)); // st_mask is reused by the next call
    return 0;
}

int ccz_set_position(ccz_engine *e, void *stream, int32_t board, const uint8_t *sq_host, int32_t turn, int32_t halfmove)
{
    NEED(e);
    if (board < 0 || board >= e->d.B || !sq_host) return fail(-1, "ccz_set_position: bad board index or null squares");
    int kings[2] = {0, 0};
    for (int i = 0; i < 90; ++i) {
        const int pc = sq_host[i];
        if (pc > 15 || pc == 8) return fail(-1, "ccz_set_position: bad piece code %d on square %d", pc, i);
        if ((pc & 7) == KING && pc) kings[pc >> 3]++;
    }
    if (kings[0] != 1 || kings[1] != 1) return fail(-1, "ccz_set_position: each side needs exactly one king");
    if (halfmove < 0) return fail(-1, "ccz_set_position: negative halfmove clock");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(e->st_sq, sq_host, 90, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_set_position, dim3(1), dim3(64), 0, s, e->d, board, (const uint8_t *)e->st_sq, turn ? 1 : 0, halfmove);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_reset_tree(ccz_engine *e, void *stream, const uint8_t *mask_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *mask = nullptr;
    if (mask_host) {
        HIP_TRY(hipMemcpyAsync(e->st_mask, mask_host, (size_t)e->d.B, hipMemcpyHostToDevice, s));
        mask = e->st_mask;
    }
    hipLaunchKernelGGL(k_reset_tree, dim3((e->d.B + 255) / 256), dim3(256), 0, s, e->d, mask);
    HIP_TRY(hipGetLastError());
    if (mask_host) HIP_TRY(hipStreamSynchronize(s)); // st_mask is reused by the next call
    return 0;
}

int ccz_zero_leaf_input(ccz_engine *e, void *stream, void *leaf_input_f16_dev)
{
    NEED(e);
    if (!leaf_input_f16_dev) return fail(-1, "ccz_zero_leaf_input: null buffer");
    HIP_TRY(hipMemsetAsync(leaf_input_f16_dev, 0, (size_t)e->d.B * CCZ_PLANES * 2, (hipStream_t)stream));
    return 0;
}

int ccz_select_leaves(ccz_engine *e, void *stream, void *leaf_input_f16_dev)
{
    NEED(e);
    hipLaunchKernelGGL(k_select, dim3(ACTIVE(e)), dim3(64), 0, (hipStream_t)stream, e->d, (uint16_t *)leaf_input_f16_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_expand_backup(ccz_engine *e, void *stream, const float *prob_dev, const float *value_dev)
{
    NEED(e);
    if (!prob_dev || !value_dev) return fail(-1, "ccz_expand_backup: null prob/value");
    hipLaunchKernelGGL(k_expand_backup<false>, dim3(ACTIVE(e)), dim3(64), 0, (hipStream_t)stream, e->d, prob_dev, value_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_step(ccz_engine *e, void *stream, const float *prob_dev, const float *value_dev, void *leaf_input_f16_dev)
{
    NEED(e);
    if (!prob_dev || !value_dev) return fail(-1, "ccz_step: null prob/value");
    hipLaunchKernelGGL(k_step<false>, dim3(ACTIVE(e)), dim3(64), 0, (hipStream_t)stream, e->d, prob_dev, value_dev,
                       (uint16_t *)leaf_input_f16_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_gather_priors(ccz_engine *e, void *stream, const void *logits_dev, int32_t logits_f16)
{
    NEED(e);
    if (!logits_dev) return fail(-1, "ccz_gather_priors: null logits");
    hipStream_t s = (hipStream_t)stream;
    if (logits_f16)
        hipLaunchKernelGGL((k_softmax_gather<_Float16, false>), dim3(e->d.B), dim3(64), 0, s, e->d, (const _Float16 *)logits_dev, (const float *)nullptr);
    else
        hipLaunchKernelGGL((k_softmax_gather<float, false>), dim3(e->d.B), dim3(64), 0, s, e->d, (const float *)logits_dev, (const float *)nullptr);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_eval_plan(ccz_engine *e, void *stream, int32_t *miss_rows_dev, int32_t *n_miss_dev)
{
    NEED(e);
    if (!e->d.cache) return fail(-1, "ccz_eval_plan: the engine was created without an evaluation cache (ccz_config.eval_cache_log2)");
    if (!miss_rows_dev || !n_miss_dev) return fail(-1, "ccz_eval_plan: null output");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cache_probe, dim3(e->d.B), dim3(64), 0, s, e->d);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_cache_plan, dim3(1), dim3(1024), 0, s, e->d, miss_rows_dev, n_miss_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_set_scouts(ccz_engine *e, int32_t n_scouts)
{
    NEED(e);
    if (n_scouts < 0 || n_scouts >= e->d.B) return fail(-1, "ccz_set_scouts: n_scouts must be in 0 .. n_boards - 1 (got %d of %d boards)", n_scouts, e->d.B);
    if (n_scouts && !e->d.cache) return fail(-1, "ccz_set_scouts: scouts work through the evaluation cache (ccz_config.eval_cache_log2)");
    e->active = n_scouts ? e->d.B - n_scouts : 0;
    return 0;
}

int ccz_scout(ccz_engine *e, void *stream, void *leaf_input_f16_dev)
{
    NEED(e);
    if (e->active <= 0) return fail(-1, "ccz_scout: no scout slots (ccz_set_scouts)");
    if (!leaf_input_f16_dev) return fail(-1, "ccz_scout: null leaf input");
    hipLaunchKernelGGL(k_scout, dim3((unsigned)(e->d.B - e->active)), dim3(64), 0, (hipStream_t)stream, e->d, (uint16_t *)leaf_input_f16_dev, e->active);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_eval_plan_scouted(ccz_engine *e, void *stream, int32_t *miss_rows_dev, int32_t *n_miss_dev, int32_t *state_dev)
{
    NEED(e);
    if (!e->d.cache) return fail(-1, "ccz_eval_plan_scouted: the engine was created without an evaluation cache (ccz_config.eval_cache_log2)");
    if (e->active <= 0) return fail(-1, "ccz_eval_plan_scouted: no scout slots (ccz_set_scouts)");
    if (!miss_rows_dev || !n_miss_dev) return fail(-1, "ccz_eval_plan_scouted: null output");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cache_probe, dim3(e->d.B), dim3(64), 0, s, e->d);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_cache_plan_scouted, dim3(1), dim3(256), 0, s, e->d, e->active, miss_rows_dev, n_miss_dev, state_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_scout_and_plan(ccz_engine *e, void *stream, void *leaf_input_f16_dev, int32_t *miss_rows_dev, int32_t *n_miss_dev, int32_t *state_dev)
{
    NEED(e);
    if (!e->d.cache || e->active <= 0) return fail(-1, "ccz_scout_and_plan: needs an evaluation cache and scout slots (ccz_set_scouts)");
    if (!leaf_input_f16_dev || !miss_rows_dev || !n_miss_dev) return fail(-1, "ccz_scout_and_plan: null leaf input / output");
    if (e->d.B > kScoutFusedMax) {     // more slots than one workgroup has waves: the three launches
        const int rc = ccz_scout(e, stream, leaf_input_f16_dev);
        return rc ? rc : ccz_eval_plan_scouted(e, stream, miss_rows_dev, n_miss_dev, state_dev);
    }
    hipLaunchKernelGGL(k_scout_probe_plan, dim3(1), dim3((unsigned)(64 * e->d.B)), 0, (hipStream_t)stream, e->d, (uint16_t *)leaf_input_f16_dev, e->active,
                       miss_rows_dev, n_miss_dev, state_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_scouted_run(ccz_engine *e, void *stream, void *leaf_input_f16_dev, int32_t *miss_rows_dev, int32_t *n_miss_dev, int32_t *state_dev, int32_t *run_dev)
{
    NEED(e);
    if (!e->d.cache || e->active <= 0 || e->active == e->d.B) return fail(-1, "ccz_scouted_run: needs an evaluation cache and scout slots (ccz_set_scouts)");
    if (e->d.B > kScoutFusedMax) return fail(-1, "ccz_scouted_run: at most %d slots (one workgroup, one wave per slot); this engine has %d", kScoutFusedMax, e->d.B);
    if (!leaf_input_f16_dev || !miss_rows_dev || !n_miss_dev || !state_dev || !run_dev) return fail(-1, "ccz_scouted_run: null leaf input / output / run block");
    if (e->d.B <= 12)
        hipLaunchKernelGGL(k_scouted_run<12>, dim3(1), dim3((unsigned)(64 * e->d.B)), 0, (hipStream_t)stream, e->d, (uint16_t *)leaf_input_f16_dev, e->active,
                           miss_rows_dev, n_miss_dev, state_dev, run_dev);
    else
        hipLaunchKernelGGL(k_scouted_run<kScoutFusedMax>, dim3(1), dim3((unsigned)(64 * e->d.B)), 0, (hipStream_t)stream, e->d, (uint16_t *)leaf_input_f16_dev,
                           e->active, miss_rows_dev, n_miss_dev, state_dev, run_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_gather_priors_planned(ccz_engine *e, void *stream, const void *logits_compact_dev, int32_t logits_f16, const float *value_compact_dev)
{
    NEED(e);
    if (!e->d.cache) return fail(-1, "ccz_gather_priors_planned: the engine was created without an evaluation cache");
    if (!logits_compact_dev || !value_compact_dev) return fail(-1, "ccz_gather_priors_planned: null logits / value");
    hipStream_t s = (hipStream_t)stream;
    if (logits_f16)
        hipLaunchKernelGGL((k_softmax_gather<_Float16, true>), dim3(e->d.B), dim3(64), 0, s, e->d, (const _Float16 *)logits_compact_dev, value_compact_dev);
    else
        hipLaunchKernelGGL((k_softmax_gather<float, true>), dim3(e->d.B), dim3(64), 0, s, e->d, (const float *)logits_compact_dev, value_compact_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_eval_cache_clear(ccz_engine *e, void *stream)
{
    NEED(e);
    if (!e->d.cache) return 0;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(e->d.cache, 0, ((size_t)e->d.cache_mask + 1) * sizeof(CacheEntry), s));
    HIP_TRY(hipMemsetAsync(e->d.claim, 0x7f, ((size_t)e->d.cache_mask + 1) * 4, s));
    return 0;
}

int ccz_step_compact(ccz_engine *e, void *stream, const float *value_dev, void *leaf_input_f16_dev)
{
    NEED(e);
    if (!value_dev && !e->d.cache) return fail(-1, "ccz_step_compact: null value (engine-owned leaf values exist only with an evaluation cache)");
    hipLaunchKernelGGL(k_step<true>, dim3(ACTIVE(e)), dim3(64), 0, (hipStream_t)stream, e->d, (const float *)e->d.prior128, value_dev,
                       (uint16_t *)leaf_input_f16_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_expand_backup_compact(ccz_engine *e, void *stream, const float *value_dev)
{
    NEED(e);
    if (!value_dev && !e->d.cache) return fail(-1, "ccz_expand_backup_compact: null value (engine-owned leaf values exist only with an evaluation cache)");
    hipLaunchKernelGGL(k_expand_backup<true>, dim3(ACTIVE(e)), dim3(64), 0, (hipStream_t)stream, e->d, (const float *)e->d.prior128,
                       value_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_finish_move(ccz_engine *e, void *stream, const int32_t *forced_moves_dev, const double *temps_dev,
                    int32_t *moves_out_dev, int32_t keep_tree)
{
    NEED(e);
    hipLaunchKernelGGL(k_finish_move, dim3(ACTIVE(e)), dim3(64), 0, (hipStream_t)stream, e->d, forced_moves_dev, temps_dev,
                       moves_out_dev, keep_tree ? 1 : 0);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_flip_half, dim3(1), dim3(1), 0, (hipStream_t)stream, e->d);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_root_children(ccz_engine *e, void *stream, int32_t *k_host, uint16_t *acts_host, int32_t *visits_host,
                      float *q_host, float *prior_host, int32_t *root_visits_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const size_t B = (size_t)e->d.B;
    hipLaunchKernelGGL(k_root_children, dim3(e->d.B), dim3(64), 0, s, e->d, e->st_k, e->st_acts, e->st_visits, e->st_q, e->st_p,
                       e->st_rootn, (const double *)nullptr, (double *)nullptr);
    HIP_TRY(hipGetLastError());
    if (k_host) HIP_TRY(hipMemcpyAsync(k_host, e->st_k, B * 4, hipMemcpyDeviceToHost, s));
    if (acts_host) HIP_TRY(hipMemcpyAsync(acts_host, e->st_acts, B * kMaxLegal * 2, hipMemcpyDeviceToHost, s));
    if (visits_host) HIP_TRY(hipMemcpyAsync(visits_host, e->st_visits, B * kMaxLegal * 4, hipMemcpyDeviceToHost, s));
    if (q_host) HIP_TRY(hipMemcpyAsync(q_host, e->st_q, B * kMaxLegal * 4, hipMemcpyDeviceToHost, s));
    if (prior_host) HIP_TRY(hipMemcpyAsync(prior_host, e->st_p, B * kMaxLegal * 4, hipMemcpyDeviceToHost, s));
    if (root_visits_host) HIP_TRY(hipMemcpyAsync(root_visits_host, e->st_rootn, B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_root_pi(ccz_engine *e, void *stream, const double *temps_host, double *pi_host)
{
    NEED(e);
    if (!pi_host) return fail(-1, "ccz_root_pi: null output");
    hipStream_t s = (hipStream_t)stream;
    const size_t B = (size_t)e->d.B;
    const double *temps = nullptr;
    if (temps_host) {
        HIP_TRY(hipMemcpyAsync(e->st_temps, temps_host, B * 8, hipMemcpyHostToDevice, s));
        temps = e->st_temps;
    }
    hipLaunchKernelGGL(k_root_children, dim3(e->d.B), dim3(64), 0, s, e->d, (int32_t *)nullptr, (uint16_t *)nullptr,
                       (int32_t *)nullptr, (float *)nullptr, (float *)nullptr, (int32_t *)nullptr, temps, e->st_pi);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pi_host, e->st_pi, B * kMaxLegal * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

static int fetch_meta(ccz_engine *e, hipStream_t s)
{
    HIP_TRY(hipMemcpyAsync(e->h_meta.data(), e->d.meta, (size_t)e->d.B * sizeof(BoardMeta), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_game_status(ccz_engine *e, void *stream, uint8_t *over_host, int8_t *winner_host, int32_t *plies_host, uint8_t *turn_host)
{
    NEED(e);
    const int rc = fetch_meta(e, (hipStream_t)stream);
    if (rc) return rc;
    for (int b = 0; b < e->d.B; ++b) {
        const BoardMeta &m = e->h_meta[b];
        if (over_host) over_host[b] = m.over;
        if (winner_host) winner_host[b] = m.winner;
        if (plies_host) plies_host[b] = m.ply;
        if (turn_host) turn_host[b] = m.turn;
    }
    return 0;
}

int ccz_root_positions(ccz_engine *e, void *stream, uint8_t *sq_host)
{
    NEED(e);
    if (!sq_host) return fail(-1, "ccz_root_positions: null output");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(sq_host, e->d.root_sq, (size_t)e->d.B * 96, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_leaf_info(ccz_engine *e, void *stream, uint8_t *status_host, int32_t *k_host, uint16_t *ids_host, int32_t *depth_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const size_t B = (size_t)e->d.B;
    if (status_host) HIP_TRY(hipMemcpyAsync(status_host, e->d.leaf_status, B, hipMemcpyDeviceToHost, s));
    if (k_host) HIP_TRY(hipMemcpyAsync(k_host, e->d.leaf_k, B * 4, hipMemcpyDeviceToHost, s));
    if (ids_host) HIP_TRY(hipMemcpyAsync(ids_host, e->d.leaf_ids, B * kMaxLegal * 2, hipMemcpyDeviceToHost, s));
    if (depth_host) HIP_TRY(hipMemcpyAsync(depth_host, e->d.path_len, B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_leaf_priors(ccz_engine *e, void *stream, float *prior_host, float *value_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const size_t B = (size_t)e->d.B;
    if (value_host && !e->d.cache) return fail(-1, "ccz_leaf_priors: engine-owned leaf values exist only with an evaluation cache (pass value_host = NULL)");
    if (prior_host) HIP_TRY(hipMemcpyAsync(prior_host, e->d.prior128, B * kMaxLegal * 4, hipMemcpyDeviceToHost, s));
    if (value_host) HIP_TRY(hipMemcpyAsync(value_host, e->d.vleaf, B * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_leaf_keys(ccz_engine *e, void *stream, uint64_t *keys_dev, uint8_t *status_dev)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    if (keys_dev) HIP_TRY(hipMemcpyAsync(keys_dev, e->d.leaf_key, (size_t)e->d.B * 8, hipMemcpyDeviceToDevice, s));
    if (status_dev) HIP_TRY(hipMemcpyAsync(status_dev, e->d.leaf_status, (size_t)e->d.B, hipMemcpyDeviceToDevice, s));
    return 0;
}

int ccz_harvest_rows(ccz_engine *e, void *stream, int64_t *rows_host)
{
    NEED(e);
    if (!rows_host) return fail(-1, "ccz_harvest_rows: null output");
    const int rc = fetch_meta(e, (hipStream_t)stream);
    if (rc) return rc;
    const int mul = (e->d.flags & CCZ_FLAG_NO_MIRROR) ? 1 : 2;
    int64_t rows = 0;
    for (int b = 0; b < e->d.B; ++b)
        if (e->h_meta[b].over) rows += (int64_t)e->h_meta[b].ply * mul;
    *rows_host = rows;
    return 0;
}

int ccz_harvest(ccz_engine *e, void *stream, void *states_f16_dev, float *pi_dev, float *z_dev, int64_t capacity_rows,
                int64_t *rows_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const int rc = fetch_meta(e, s);
    if (rc) return rc;
    const int B = e->d.B;
    const int mul = (e->d.flags & CCZ_FLAG_NO_MIRROR) ? 1 : 2;
    std::vector<long long> base((size_t)B, -1);
    std::vector<uint8_t> mask((size_t)B, 0);
    int64_t rows = 0;
    bool any = false;
    // finished boards are taken in index order while their rows fit the caller's buffers; the rest stay
    // finished and are picked up by the next call (many boards can reach the ply cap in the same move)
    for (int b = 0; b < B; ++b)
        if (e->h_meta[b].over) {
            const int64_t add = (int64_t)e->h_meta[b].ply * mul;
            if (rows + add > capacity_rows) {
                if (!any) return fail(-5, "ccz_harvest: the first finished game needs %lld rows, capacity %lld", (long long)add, (long long)capacity_rows);
                break;
            }
            base[b] = rows;
            mask[b] = 1;
            any = true;
            rows += add;
        }
    if (rows_host) *rows_host = rows;
    if (!any) return 0;
    if (rows > 0 && (!states_f16_dev || !pi_dev || !z_dev)) return fail(-1, "ccz_harvest: null output buffer");
    if (rows > 0) {
        HIP_TRY(hipMemcpyAsync(e->st_rowbase, base.data(), (size_t)B * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_harvest, dim3(B, kHarvestSlices), dim3(256), 0, s, e->d, (const long long *)e->st_rowbase,
                           (uint16_t *)states_f16_dev, pi_dev, z_dev);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(e->st_mask, mask.data(), (size_t)B, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_reset, dim3(B), dim3(64), 0, s, e->d, (const uint8_t *)e->st_mask);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_harvest_records(ccz_engine *e, void *stream, void *records_dev, int64_t capacity_plies, int64_t *plies_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    const int rc = fetch_meta(e, s);
    if (rc) return rc;
    const int B = e->d.B;
    std::vector<long long> base((size_t)B, -1);
    std::vector<uint8_t> mask((size_t)B, 0);
    int64_t plies = 0;
    bool any = false;
    // as ccz_harvest: finished boards in index order while their plies fit; the rest stay finished for the next call
    for (int b = 0; b < B; ++b)
        if (e->h_meta[b].over) {
            const int64_t add = (int64_t)e->h_meta[b].ply;
            if (plies + add > capacity_plies) {
                if (!any) return fail(-5, "ccz_harvest_records: the first finished game has %lld plies, capacity %lld", (long long)add, (long long)capacity_plies);
                break;
            }
            base[b] = plies;
            mask[b] = 1;
            any = true;
            plies += add;
        }
    if (plies_host) *plies_host = plies;
    if (!any) return 0;
    if (plies > 0 && !records_dev) return fail(-1, "ccz_harvest_records: null output buffer");
    if (plies > 0) {
        HIP_TRY(hipMemcpyAsync(e->st_rowbase, base.data(), (size_t)B * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_harvest_records, dim3(B, kHarvestSlices), dim3(256), 0, s, e->d, (const long long *)e->st_rowbase, (uint8_t *)records_dev);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(e->st_mask, mask.data(), (size_t)B, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_reset, dim3(B), dim3(64), 0, s, e->d, (const uint8_t *)e->st_mask);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ccz_expand_records(void *stream, const void *records_dev, int64_t n_plies, uint32_t flags, const uint8_t *plane_of_type_host,
                       void *states_f16_dev, float *pi_dev, float *z_dev, int64_t ring_rows, int64_t head_row, int32_t *bad_records_dev)
{
    if (n_plies < 0 || ring_rows < 0 || head_row < 0) return fail(-1, "ccz_expand_records: negative size");
    if (n_plies == 0) return 0;
    if (!records_dev || !states_f16_dev || !pi_dev || !z_dev) return fail(-1, "ccz_expand_records: null buffer");
    if (((uintptr_t)records_dev | (uintptr_t)states_f16_dev) & 3) return fail(-1, "ccz_expand_records: buffers must be 4-byte aligned");
    if (n_plies > (int64_t)INT32_MAX) return fail(-1, "ccz_expand_records: too many records for one launch");
    const int64_t rows = n_plies * ((flags & CCZ_FLAG_NO_MIRROR) ? 1 : 2);
    if (ring_rows > 0 && rows > ring_rows) return fail(-1, "ccz_expand_records: %lld rows do not fit a ring of %lld", (long long)rows, (long long)ring_rows);
    if (ring_rows == 0 && head_row != 0) return fail(-1, "ccz_expand_records: head_row needs ring_rows");
    uint8_t pot[8] = {0, 0, 1, 2, 3, 4, 5, 6};
    if (plane_of_type_host) {
        bool all_zero = true;
        for (int t = 0; t < 8; ++t) all_zero = all_zero && plane_of_type_host[t] == 0;
        if (!all_zero) {
            unsigned seen = 0;
            for (int t = 1; t <= 7; ++t) {
                const int c = plane_of_type_host[t];
                if (c > 6 || (seen >> c & 1u)) return fail(-1, "ccz_expand_records: plane_of_type[1..7] must be a permutation of 0..6");
                seen |= 1u << c;
                pot[t] = (uint8_t)c;
            }
        }
    }
    uint32_t typepack = 0;
    for (int t = 1; t <= 7; ++t) typepack |= (uint32_t)(t - 1) << (3 * pot[t]);
    hipLaunchKernelGGL(k_expand_records, dim3((unsigned)n_plies), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)records_dev, (long long)n_plies,
                       flags & (CCZ_FLAG_REFERENCE_QUIRKS | CCZ_FLAG_NO_MIRROR), typepack, (uint16_t *)states_f16_dev, pi_dev, z_dev,
                       (long long)ring_rows, (long long)head_row, bad_records_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_get_stats(ccz_engine *e, void *stream, ccz_stats *out)
{
    NEED(e);
    if (!out) return fail(-1, "ccz_get_stats: null output");
    hipStream_t s = (hipStream_t)stream;
    std::vector<BoardStats> st((size_t)e->d.B);
    int32_t err[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(st.data(), e->d.stats, st.size() * sizeof(BoardStats), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(err, e->d.err, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memset(out, 0, sizeof *out);
    for (const BoardStats &b : st) {
        out->sims += (int64_t)b.sims;
        out->moves += (int64_t)b.moves;
        out->games += (int64_t)b.games;
        out->truncated_games += (int64_t)b.truncated;
        out->sum_depth += (int64_t)b.sum_depth;
        out->sum_children += (int64_t)b.sum_children;
        out->expansions += (int64_t)b.expansions;
        out->terminal_leaves += (int64_t)b.terminal;
        out->pruned_subtrees += (int64_t)b.pruned;
        out->cache_probes += (int64_t)b.cache_probes;
        out->cache_hits += (int64_t)b.cache_hits;
        out->cache_shared_rows += (int64_t)b.cache_shared;
        out->cache_stores += (int64_t)b.cache_stores;
        out->cache_verified += (int64_t)b.cache_verified;
        out->cache_verify_mismatches += (int64_t)b.cache_mismatch;
        if (b.nodes_peak > out->nodes_peak) out->nodes_peak = b.nodes_peak;
        if (b.depth_peak > out->depth_peak) out->depth_peak = b.depth_peak;
    }
    out->error_flags = err[0];
    out->reserved = err[1]; // bounds-checked diagnostic build: source line of the stray index (0 otherwise)
    out->hbm_bytes = (int64_t)e->bytes;
    return 0;
}

int ccz_legal_moves(void *stream, int32_t n, const uint8_t *sq_dev, const uint8_t *turn_dev, const int32_t *halfmove_dev,
                    uint32_t *mask_dev, int32_t *count_dev, uint8_t *flags_dev)
{
    if (n < 0 || !sq_dev || !turn_dev) return fail(-1, "ccz_legal_moves: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_legal_moves, dim3(n), dim3(64), 0, (hipStream_t)stream, n, sq_dev, turn_dev, halfmove_dev, mask_dev,
                       count_dev, flags_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_apply_moves(void *stream, int32_t n, uint8_t *sq_dev, uint8_t *turn_dev, const int32_t *move_ids_dev, uint8_t *captured_dev)
{
    if (n < 0 || !sq_dev || !turn_dev || !move_ids_dev) return fail(-1, "ccz_apply_moves: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_apply_moves, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, sq_dev, turn_dev, move_ids_dev,
                       captured_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_bias_act_f16(void *stream, void *y_dev, const void *bias_dev, const void *residual_dev, int64_t rows, int32_t channels)
{
    if (!y_dev || !bias_dev || rows < 0 || channels <= 0 || (channels & 7)) return fail(-1, "ccz_bias_act_f16: bad arguments (channels must be a multiple of 8)");
    if ((((uintptr_t)y_dev) | ((uintptr_t)bias_dev) | ((uintptr_t)residual_dev)) & 15) return fail(-1, "ccz_bias_act_f16: pointers must be 16-byte aligned");
    const long n_vec = rows * (long)(channels / 8);
    if (n_vec == 0) return 0;
    long blocks = (n_vec + 255) / 256;
    if (blocks > 4096) blocks = 4096; // grid-stride: 16 resident blocks per CU
    if (residual_dev)
        hipLaunchKernelGGL(k_bias_act<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (half8_t *)y_dev, (const half8_t *)bias_dev,
                           (const half8_t *)residual_dev, n_vec, channels / 8);
    else
        hipLaunchKernelGGL(k_bias_act<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (half8_t *)y_dev, (const half8_t *)bias_dev,
                           (const half8_t *)nullptr, n_vec, channels / 8);
    HIP_TRY(hipGetLastError());
    return 0;
}

constexpr int64_t kSmallMaxPixels = 64 * 90; // up to 64 boards (profiles/r03_single_board.json: crossover with the tile kernel)

static int conv3x3_launch(const char *who, void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev, const void *residual_dev, void *y_dev,
                          int64_t n_pixels, int32_t relu, int cin, const int32_t *live_rows_dev = nullptr, int32_t row0 = 0)
{
    if (!x_dev || !w_dev || !bias_f32_dev || !y_dev || n_pixels < 0 || n_pixels % 90 || n_pixels > (int64_t)INT32_MAX / kCvC) /* 32-bit element offsets in the kernel */
        return fail(-1, "%s: bad arguments (n_pixels must be boards * 90, at most 93206 boards per call)", who);
    if ((((uintptr_t)x_dev) | ((uintptr_t)w_dev) | ((uintptr_t)bias_f32_dev) | ((uintptr_t)residual_dev) | ((uintptr_t)y_dev)) & 15)
        return fail(-1, "%s: pointers must be 16-byte aligned", who);
    if (x_dev == y_dev) return fail(-1, "%s: the output may alias the residual but not the input", who);
    if (n_pixels == 0) return 0;
    if (relu & CCZ_CONV_G16) { // rows in the group-of-16 layout: whole-rank tiles, off-board taps skipped (cczero_conv_g16.h)
        if (n_pixels % 1440) return fail(-1, "%s: CCZ_CONV_G16 needs a multiple of 16 boards", who);
        const int groups = (int)(n_pixels / 1440);
        const int fl = relu & 3;
        hipStream_t s = (hipStream_t)stream;
#define CCZ_G16(KERNEL_, GRID_, STREAM_, FLAGS_)                                                                                   \
        do {                                                                                                                       \
            if (residual_dev)                                                                                                      \
                hipLaunchKernelGGL(KERNEL_<true>, dim3((unsigned)(GRID_)), dim3(512), 0, STREAM_, (const _Float16 *)x_dev, (const _Float16 *)w_dev, \
                                   (const float *)bias_f32_dev, (const _Float16 *)residual_dev, (_Float16 *)y_dev, (int)n_pixels, (int)(FLAGS_), cin, (const int *)live_rows_dev, (int)row0); \
            else                                                                                                                   \
                hipLaunchKernelGGL(KERNEL_<false>, dim3((unsigned)(GRID_)), dim3(512), 0, STREAM_, (const _Float16 *)x_dev, (const _Float16 *)w_dev, \
                                   (const float *)bias_f32_dev, (const _Float16 *)nullptr, (_Float16 *)y_dev, (int)n_pixels, (int)(FLAGS_), cin, (const int *)live_rows_dev, (int)row0); \
        } while (0)
        // Without the flag: ONE launch of five two-rank tiles per group. CCZ_CONV_G16_EDGE_TILES (flag bit 7, round 4): the edge ranks (0
        // and 9) of two groups at a time are their own tiles on their own kernel (six live taps instead of nine, cczero_conv_g16e.h) and
        // the middle launch covers ranks 1..8 with four two-rank tiles per group -- two ordinary launches back to back in the caller's
        // stream (no gap between them in the kernel trace). Same values. Measured (profiles/r04_conv_g16.json): -3 % per layer at 4096
        // boards in isolation (1024 middle tiles = four full rounds of 256 CUs, 256 edge tiles = one round ~24 % shorter); in the
        // workload +0.7 % on the step with two launch chains (the extra launch boundary per layer and chain costs more than the six
        // taps save) but -0.7...-0.9 % with three: the evaluator sets the flag from 4096 boards on and runs three chains then. (The
        // edge launch on a helper stream beside the middle one put two event packets per layer on the main stream -- a 12.7 us gap
        // between layers; hipExtAnyOrderLaunch is ignored on gfx9: the trace shows the kernels one after the other.)
        if (cin == 64) { // the stem: the packed live planes sit in channels 0..20 of 64: ONE 32-channel chunk (k_conv3x3_g16_stem), one launch
            hipLaunchKernelGGL(k_conv3x3_g16_stem, dim3((unsigned)(groups * 5)), dim3(512), 0, s, (const _Float16 *)x_dev, (const _Float16 *)w_dev,
                               (const float *)bias_f32_dev, (_Float16 *)y_dev, (int)n_pixels, (int)fl, cin, (const int *)live_rows_dev, (int)row0);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        // CCZ_CONV_G16_PERSISTENT: the same tiles on a fixed number of workgroups that walk tile lists (cczero_conv_g16p.h); bits 16..27 =
        // the number of workgroups (0: one per CU)
        int pers = (relu & CCZ_CONV_G16_PERSISTENT) ? (((relu >> 16) & 0xfff) ? ((relu >> 16) & 0xfff) : 256) : 0;
        if (pers && pers < 8) pers = 8; // every XCD that owns tiles needs a workgroup (tile lists are per XCD)
        if (pers && cin != 256) return fail(-1, "%s: CCZ_CONV_G16_PERSISTENT is the tower shape only (256 input channels)", who);
        if (!(relu & CCZ_CONV_G16_EDGE_TILES) || groups < 2) {
            if (pers) CCZ_G16(k_conv3x3_g16_pers, groups * 5 < pers ? groups * 5 : pers, s, fl);
            else CCZ_G16(k_conv3x3_g16, groups * 5, s, fl);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if (pers) CCZ_G16(k_conv3x3_g16_pers, groups * 4 < pers ? groups * 4 : pers, s, fl | 4);
        else
        CCZ_G16(k_conv3x3_g16, groups * 4, s, fl | 4);   // (edge launch first: measured the same, 194.4 / 194.3 k against 194.8 / 194.0 k sims/s)
        HIP_TRY(hipGetLastError());
        CCZ_G16(k_conv3x3_g16_edge, 2 * ((groups + 1) / 2), s, fl);
        HIP_TRY(hipGetLastError());
#undef CCZ_G16
        return 0;
    }
    // Small batches (one game at a time; up to kSmallMaxPixels): the 16-channel x 64-pixel-block kernel spreads them over the chip
    // instead of filling a few 256-pixel tiles. Same operations in the same order: bit-identical results (cczero_conv_small.h).
    // flags bit 4 forces it, bit 5 forces the tile kernel (A/B runs, tests).
    if (!live_rows_dev && !(relu & 32) && ((relu & 16) || n_pixels <= kSmallMaxPixels)) {
        const dim3 grid((unsigned)((n_pixels + kSmPix - 1) / kSmPix), 16);
        const int rl = relu & 1;
#define CCZ_SMALL(RES_, CIN_)                                                                                                  \
        hipLaunchKernelGGL((k_conv3x3_small<RES_, CIN_>), grid, dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x_dev, (const _Float16 *)w_dev, \
                           (const float *)bias_f32_dev, (const _Float16 *)residual_dev, (_Float16 *)y_dev, (int)n_pixels, rl)
        if (cin == 64) CCZ_SMALL(false, 64);
        else if (residual_dev) CCZ_SMALL(true, 256);
        else CCZ_SMALL(false, 256);
#undef CCZ_SMALL
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const unsigned tiles = (unsigned)((n_pixels + kCvBM - 1) / kCvBM);
#ifndef CCZ_STAMPS
    relu &= 3; // bit 0: ReLU, bit 1: descending tile order; the diagnostic build passes ablation switches in bits 8.. (profiles/conv_microbench.py)
#endif
    if (residual_dev)
        hipLaunchKernelGGL(k_conv3x3_c256<true>, dim3(tiles), dim3(512), 0, (hipStream_t)stream, (const _Float16 *)x_dev, (const _Float16 *)w_dev,
                           (const float *)bias_f32_dev, (const _Float16 *)residual_dev, (_Float16 *)y_dev, (int)n_pixels, (int)relu, cin, (const int *)live_rows_dev, (int)row0);
    else
        hipLaunchKernelGGL(k_conv3x3_c256<false>, dim3(tiles), dim3(512), 0, (hipStream_t)stream, (const _Float16 *)x_dev, (const _Float16 *)w_dev,
                           (const float *)bias_f32_dev, (const _Float16 *)nullptr, (_Float16 *)y_dev, (int)n_pixels, (int)relu, cin, (const int *)live_rows_dev, (int)row0);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_conv3x3_c256_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev, const void *residual_dev, void *y_dev,
                         int64_t n_pixels, int32_t relu)
{
    return conv3x3_launch("ccz_conv3x3_c256_f16", stream, x_dev, w_dev, bias_f32_dev, residual_dev, y_dev, n_pixels, relu, 256);
}

int ccz_conv3x3_c256_heads_f16(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev, const void *residual_dev,
                               const void *head_w32_dev, const void *head_b32_dev, void *pol_dev, void *val_dev, int64_t n_pixels, int32_t flags,
                               const int32_t *live_rows_dev, int32_t part, int32_t n_parts)
{
    const char *who = "ccz_conv3x3_c256_heads_f16";
    if (!x_dev || !w_dev || !bias_f32_dev || !residual_dev || !head_w32_dev || !head_b32_dev || !pol_dev || !val_dev || n_pixels < 0 || n_pixels % 1440 ||
        n_pixels > (int64_t)INT32_MAX / kCvC)
        return fail(-1, "%s: bad arguments (n_pixels must be a multiple of 16 boards * 90, at most 93200 boards per call)", who);
    if (!(flags & CCZ_CONV_G16)) return fail(-1, "%s: rows must be in the group-of-16 layout (CCZ_CONV_G16)", who);
    if ((((uintptr_t)x_dev) | ((uintptr_t)w_dev) | ((uintptr_t)bias_f32_dev) | ((uintptr_t)residual_dev) | ((uintptr_t)head_w32_dev) | ((uintptr_t)head_b32_dev)) & 15)
        return fail(-1, "%s: pointers must be 16-byte aligned", who);
    if ((((uintptr_t)pol_dev) | ((uintptr_t)val_dev)) & 1) return fail(-1, "%s: outputs must be 2-byte aligned", who);
    if (live_rows_dev && (n_parts < 1 || n_parts > 256 || part < 0 || part >= n_parts)) return fail(-1, "%s: bad part / n_parts", who);
    if (n_pixels == 0) return 0;
    const int groups = (int)(n_pixels / 1440);
    const int fl = flags & 3, row0 = live_rows_dev ? (part | (n_parts << 16)) : 0;
    G5Heads ha;
    ha.w32 = (const _Float16 *)head_w32_dev;
    ha.b32 = (const float *)head_b32_dev;
    ha.pol = (_Float16 *)pol_dev;
    ha.val = (_Float16 *)val_dev;
    ha.nb = (int)(n_pixels / 90);
    hipStream_t s = (hipStream_t)stream;
#define CCZ_G16H(KERNEL_, GRID_, FLAGS_)                                                                                                \
    hipLaunchKernelGGL(KERNEL_, dim3((unsigned)(GRID_)), dim3(512), 0, s, (const _Float16 *)x_dev, (const _Float16 *)w_dev, (const float *)bias_f32_dev, \
                       (const _Float16 *)residual_dev, (_Float16 *)nullptr, (int)n_pixels, (int)(FLAGS_), 256, (const int *)live_rows_dev, row0, ha)
    if (!(flags & CCZ_CONV_G16_EDGE_TILES) || groups < 2) {
        CCZ_G16H(k_conv3x3_g16_heads, groups * 5, fl);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    CCZ_G16H(k_conv3x3_g16_heads, groups * 4, fl | 4);
    HIP_TRY(hipGetLastError());
    CCZ_G16H(k_conv3x3_g16_edge_heads, 2 * ((groups + 1) / 2), fl);
    HIP_TRY(hipGetLastError());
#undef CCZ_G16H
    return 0;
}

int ccz_conv3x3_stem_f16(void *stream, const void *x64_dev, const void *w_dev, const void *bias_f32_dev, void *y_dev, int64_t n_pixels, int32_t relu)
{
    return conv3x3_launch("ccz_conv3x3_stem_f16", stream, x64_dev, w_dev, bias_f32_dev, nullptr, y_dev, n_pixels, relu, 64);
}

int ccz_conv3x3_c256_f16_live(void *stream, const void *x_dev, const void *w_dev, const void *bias_f32_dev, const void *residual_dev, void *y_dev,
                              int64_t n_pixels, int32_t relu, const int32_t *live_rows_dev, int32_t part, int32_t n_parts)
{
    if (!live_rows_dev || n_parts < 1 || n_parts > 256 || part < 0 || part >= n_parts) return fail(-1, "ccz_conv3x3_c256_f16_live: null live-row count or bad part / n_parts");
    return conv3x3_launch("ccz_conv3x3_c256_f16_live", stream, x_dev, w_dev, bias_f32_dev, residual_dev, y_dev, n_pixels, relu, 256, live_rows_dev, part | (n_parts << 16));
}

int ccz_conv3x3_stem_f16_live(void *stream, const void *x64_dev, const void *w_dev, const void *bias_f32_dev, void *y_dev, int64_t n_pixels, int32_t relu,
                              const int32_t *live_rows_dev, int32_t part, int32_t n_parts)
{
    if (!live_rows_dev || n_parts < 1 || n_parts > 256 || part < 0 || part >= n_parts) return fail(-1, "ccz_conv3x3_stem_f16_live: null live-row count or bad part / n_parts");
    return conv3x3_launch("ccz_conv3x3_stem_f16_live", stream, x64_dev, w_dev, bias_f32_dev, nullptr, y_dev, n_pixels, relu, 64, live_rows_dev, part | (n_parts << 16));
}

int ccz_pack_live_planes_rows_f16(void *stream, const void *leaf_dev, void *x64_dev, int32_t n_boards, const int32_t *rows_dev, const int32_t *n_rows_dev)
{
    if (!leaf_dev || !x64_dev || n_boards < 0 || !rows_dev || !n_rows_dev) return fail(-1, "ccz_pack_live_planes_rows_f16: bad arguments");
    if ((uintptr_t)x64_dev & 15) return fail(-1, "ccz_pack_live_planes_rows_f16: output must be 16-byte aligned");
    if (n_boards == 0) return 0;
    hipLaunchKernelGGL(k_pack_live_planes, dim3((unsigned)n_boards), dim3(kPackThreads), 0, (hipStream_t)stream, (const _Float16 *)leaf_dev, (half8_t *)x64_dev, (int)n_boards,
                       (const int *)rows_dev, (const int *)n_rows_dev, 2); // planned form: only the chunks that can be non-zero are written
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_pack_conv_weights_g16_f16(void *stream, const void *w_dev, void *wp_dev, int32_t cin)
{
    if (!w_dev || !wp_dev || w_dev == wp_dev || (cin != 64 && cin != 256)) return fail(-1, "ccz_pack_conv_weights_g16_f16: bad arguments (cin must be 64 or 256, not in place)");
    if ((((uintptr_t)w_dev) | ((uintptr_t)wp_dev)) & 15) return fail(-1, "ccz_pack_conv_weights_g16_f16: pointers must be 16-byte aligned");
    const int n = 256 * 9 * (cin >> 3);
    hipLaunchKernelGGL(k_pack_conv_weights_g16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const cv_half8 *)w_dev, (cv_half8 *)wp_dev, (int)cin);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_pack_live_planes_g16_f16(void *stream, const void *leaf_dev, void *x64_dev, int32_t n_boards, const int32_t *rows_dev, const int32_t *n_rows_dev)
{
    if (!leaf_dev || !x64_dev || n_boards < 0 || (!rows_dev) != (!n_rows_dev)) return fail(-1, "ccz_pack_live_planes_g16_f16: bad arguments");
    if ((uintptr_t)x64_dev & 15) return fail(-1, "ccz_pack_live_planes_g16_f16: output must be 16-byte aligned");
    if (n_boards == 0) return 0;
    hipLaunchKernelGGL(k_pack_live_planes, dim3((unsigned)n_boards), dim3(kPackThreads), 0, (hipStream_t)stream, (const _Float16 *)leaf_dev, (half8_t *)x64_dev, (int)n_boards,
                       (const int *)rows_dev, (const int *)n_rows_dev, rows_dev ? 3 : 1); // planned form: as ccz_pack_live_planes_rows_f16
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_pack_live_planes_f16(void *stream, const void *leaf_dev, void *x64_dev, int32_t n_boards)
{
    if (!leaf_dev || !x64_dev || n_boards < 0) return fail(-1, "ccz_pack_live_planes_f16: bad arguments");
    if ((uintptr_t)x64_dev & 15) return fail(-1, "ccz_pack_live_planes_f16: output must be 16-byte aligned");
    if (n_boards == 0) return 0;
    hipLaunchKernelGGL(k_pack_live_planes, dim3((unsigned)n_boards), dim3(kPackThreads), 0, (hipStream_t)stream, (const _Float16 *)leaf_dev, (half8_t *)x64_dev, (int)n_boards,
                       (const int *)nullptr, (const int *)nullptr, 0);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef CCZ_STAMPS
// diagnostic build only: cycle stamps of the last ccz_conv3x3_c256_f16 launch (uint64 [2048][2][16])
int ccz_debug_conv_stamps(unsigned long long *out_host)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_cv_stamps), sizeof(unsigned long long) * 2048 * 2 * 16));
    return 0;
}
// diagnostic build only: per-board s_memtime stamps of the last k_step launch (uint64 [B*16])
int ccz_debug_stamps(ccz_engine *e, void *stream, unsigned long long *out_host)
{
    NEED(e);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemcpyAsync(out_host, e->d.stamps, (size_t)e->d.B * 16 * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
#endif

int ccz_heads_conv1x1_f16(void *stream, const void *x_dev, const void *w32_dev, const void *bias32_f32_dev, void *pol_dev, void *val_dev,
                          int32_t n_boards, int32_t flags, const int32_t *live_boards_dev)
{
    if (!x_dev || !w32_dev || !bias32_f32_dev || !pol_dev || !val_dev || n_boards < 0) return fail(-1, "ccz_heads_conv1x1_f16: bad arguments");
    if ((((uintptr_t)x_dev) | ((uintptr_t)w32_dev) | ((uintptr_t)bias32_f32_dev)) & 15) return fail(-1, "ccz_heads_conv1x1_f16: x, weights and bias must be 16-byte aligned");
    if (flags & ~CCZ_CONV_G16) return fail(-1, "ccz_heads_conv1x1_f16: unknown flags 0x%x", flags);
    const bool g16 = (flags & CCZ_CONV_G16) != 0;
    if (g16 && (n_boards & 15)) return fail(-1, "ccz_heads_conv1x1_f16: the group-of-16 layout holds whole groups of 16 boards (n_boards = %d)", n_boards);
    if (n_boards == 0) return 0;
    const long cells = ((long)n_boards * 90 + 15) / 16;
    const unsigned blocks = (unsigned)((cells + 3) / 4 < 768 ? (cells + 3) / 4 : 768); // 768 x 4 waves = 3 per SIMD on 256 CUs: all resident, one round
    if (g16)
        hipLaunchKernelGGL(k_head_conv1x1<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x_dev, (const _Float16 *)w32_dev,
                           (const float *)bias32_f32_dev, (_Float16 *)pol_dev, (_Float16 *)val_dev, (int)n_boards, (const int *)live_boards_dev);
    else
        hipLaunchKernelGGL(k_head_conv1x1<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)x_dev, (const _Float16 *)w32_dev,
                           (const float *)bias32_f32_dev, (_Float16 *)pol_dev, (_Float16 *)val_dev, (int)n_boards, (const int *)live_boards_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_fc_f16(void *stream, const void *a_dev, int32_t lda, const void *w_dev, const void *bias_f32_dev, void *c_dev, int32_t ldc,
               int32_t m, int32_t n, int32_t k, int32_t relu, const int32_t *live_rows_dev)
{
    if (!a_dev || !w_dev || !bias_f32_dev || !c_dev || m < 0 || n <= 0 || k <= 0) return fail(-1, "ccz_fc_f16: bad arguments");
    if ((k & 63) || (lda & 7) || lda < k || (n & 1) || (ldc & 1) || ldc < n) return fail(-1, "ccz_fc_f16: k must be a multiple of 64, lda a multiple of 8 and >= k, n and ldc even, ldc >= n");
    if ((((uintptr_t)a_dev) | ((uintptr_t)w_dev) | ((uintptr_t)bias_f32_dev)) & 15 || ((uintptr_t)c_dev & 3)) return fail(-1, "ccz_fc_f16: a, w, bias must be 16-byte aligned, c 4-byte aligned");
    if (m == 0) return 0;
    // many rows x thousands of columns (the policy layer of a big batch): the 256 x 144 tile kernel; relu bit 1 forces the 128 x 128 kernel,
    // bit 2 the wide one (tests, A/B) -- same bits from either
    // (policy shape on one box, us: M 4096: 31.8 against 53.4; 3712: 31.6 / 41.5; 2048: 27.1 / 29.3; 1024: 25.9 / 22.1 -- profiles/r04_fc_microbench_wide.json)
    if (!(relu & 2) && ((relu & 4) || (m >= 2048 && n >= 1024))) {
        const int wtiles = ((n + kFwBN - 1) / kFwBN) * ((m + kFwBM - 1) / kFwBM);
        const dim3 wgrid((unsigned)(8 * ((wtiles + 7) / 8)));
        if (relu & 1)
            hipLaunchKernelGGL(k_fc_wide_f16<true>, wgrid, dim3(512), 0, (hipStream_t)stream, (const _Float16 *)a_dev, (int)lda, (const _Float16 *)w_dev,
                               (const float *)bias_f32_dev, (_Float16 *)c_dev, (int)ldc, (int)m, (int)n, (int)k, (const int *)live_rows_dev, (int)(relu >> 8));
        else
            hipLaunchKernelGGL(k_fc_wide_f16<false>, wgrid, dim3(512), 0, (hipStream_t)stream, (const _Float16 *)a_dev, (int)lda, (const _Float16 *)w_dev,
                               (const float *)bias_f32_dev, (_Float16 *)c_dev, (int)ldc, (int)m, (int)n, (int)k, (const int *)live_rows_dev, (int)(relu >> 8));
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // a handful of rows (one game at a time: 1 + 10 scout rows): one wave per 16 columns, operands straight from memory -- same bits
    if (m <= 16 && !(relu & 6)) {
        const dim3 sgrid((unsigned)((n + 15) / 16));
        if (relu & 1)
            hipLaunchKernelGGL(k_fc_skinny_f16<true>, sgrid, dim3(64), 0, (hipStream_t)stream, (const _Float16 *)a_dev, (int)lda, (const _Float16 *)w_dev,
                               (const float *)bias_f32_dev, (_Float16 *)c_dev, (int)ldc, (int)m, (int)n, (int)k, (const int *)live_rows_dev);
        else
            hipLaunchKernelGGL(k_fc_skinny_f16<false>, sgrid, dim3(64), 0, (hipStream_t)stream, (const _Float16 *)a_dev, (int)lda, (const _Float16 *)w_dev,
                               (const float *)bias_f32_dev, (_Float16 *)c_dev, (int)ldc, (int)m, (int)n, (int)k, (const int *)live_rows_dev);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const int tiles = ((n + kFcBN - 1) / kFcBN) * ((m + kFcBM - 1) / kFcBM);
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8))); // XCD x = block mod 8 takes the x-th contiguous eighth of the tile order: see k_fc_f16
    if (relu & 1)
        hipLaunchKernelGGL(k_fc_f16<true>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16 *)a_dev, (int)lda, (const _Float16 *)w_dev,
                           (const float *)bias_f32_dev, (_Float16 *)c_dev, (int)ldc, (int)m, (int)n, (int)k, (const int *)live_rows_dev, (int)(relu >> 8));
    else
        hipLaunchKernelGGL(k_fc_f16<false>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16 *)a_dev, (int)lda, (const _Float16 *)w_dev,
                           (const float *)bias_f32_dev, (_Float16 *)c_dev, (int)ldc, (int)m, (int)n, (int)k, (const int *)live_rows_dev, (int)(relu >> 8));
    HIP_TRY(hipGetLastError());
    return 0;
}

int ccz_value_out_f32(void *stream, const void *h_dev, const void *w2_dev, float b2, float *v_dev, int32_t m, const int32_t *live_rows_dev)
{
    if (!h_dev || !w2_dev || !v_dev || m < 0) return fail(-1, "ccz_value_out_f32: bad arguments");
    if ((((uintptr_t)h_dev) | ((uintptr_t)w2_dev)) & 7) return fail(-1, "ccz_value_out_f32: h and w2 must be 8-byte aligned");
    if (m == 0) return 0;
    hipLaunchKernelGGL(k_value_out, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)h_dev, (const _Float16 *)w2_dev, b2, v_dev,
                       (int)m, (const int *)live_rows_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

} // extern "C"
