/*
 * xq_sample.c -- CPU ORACLE (test infrastructure, NOT the product). See xq_oracle.h.
 *
 * CPU twin of the DEVICE-mode move sampler (DESIGN.md "Sampling"). The reference samples with the
 * global, unseeded legacy np.random (mcts.py:216-224); B lockstep boards cannot share that one
 * stream, so device mode replaces it by a counter-based per-board stream (Philox4x32-10) and keeps
 * the reference's formula: move ~ Categorical((1-EPS)*pi + EPS*Dirichlet(ALPHA*1_k)).
 * Everything here uses only + - * / sqrt on IEEE doubles (no libm, no FMA contraction) so that the
 * HIP kernel reproduces it bit for bit. The reference-exact numpy path lives on the host side of
 * the product (mcts.py mirror) and is pinned by golden vectors instead.
 */
#include "xq_oracle.h"

#include <math.h>
#include <string.h>

static uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* natural log, x > 0 normal: x = m * 2^e, m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh((m-1)/(m+1)) */
double xq_det_log(double x)
{
    uint64_t u = d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    double m = u2d((u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL), s, s2, acc;
    int i;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    s = (m - 1.0) / (m + 1.0);
    s2 = s * s;
    acc = 1.0 / 27.0;
    for (i = 25; i >= 1; i -= 2) acc = acc * s2 + 1.0 / (double)i;
    return (double)e * 0.6931471805599453 + 2.0 * s * acc;
}

/* exp(x) for x <= 0 (returns 0 below 2^-1022): x = n ln2 + r, Taylor degree 14 on r */
double xq_det_exp(double x)
{
    double t, n, r, acc;
    int i, ni;
    if (x < -708.0) return 0.0;
    t = x * 1.4426950408889634 + 0.5;
    n = floor(t);
    r = x - n * 0.693147180369123816490 - n * 1.90821492927058770002e-10;
    acc = 1.0;
    for (i = 14; i >= 1; i--) acc = acc * r / (double)i + 1.0;
    ni = (int)n;
    if (ni < -1022) return 0.0;
    return acc * u2d((uint64_t)(ni + 1023) << 52);
}

void xq_philox4x32(uint64_t key, uint64_t ctr_hi, uint64_t ctr_lo, uint32_t out[4])
{
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    int r;
    for (r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* two uniforms in (0,1), 52 random bits each: (2r+1) * 2^-53 */
static void uniform2(uint64_t seed, uint64_t board, uint64_t move_no, uint32_t child, uint32_t draw, double *ua, double *ub)
{
    uint32_t o[4];
    uint64_t lo = (move_no << 32) | ((uint64_t)(child & 0xfffu) << 20) | (uint64_t)(draw & 0xfffffu);
    xq_philox4x32(seed, board, lo, o);
    *ua = (double)(2 * ((((uint64_t)o[0] << 32) | o[1]) >> 12) + 1) * 1.1102230246251565e-16;
    *ub = (double)(2 * ((((uint64_t)o[2] << 32) | o[3]) >> 12) + 1) * 1.1102230246251565e-16;
}

/* Gamma(alpha, 1), alpha < 1: Marsaglia-Tsang for alpha+1 with polar normals, then the U^(1/alpha) boost */
static double det_gamma(uint64_t seed, uint64_t board, uint64_t move_no, uint32_t child, double alpha)
{
    double d = (alpha + 1.0) - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d), ua, ub;
    uint32_t j = 0;
    for (;;) {
        double x1, x2, s, z, v;
        for (;;) {
            uniform2(seed, board, move_no, child, j++, &ua, &ub);
            x1 = 2.0 * ua - 1.0; x2 = 2.0 * ub - 1.0;
            s = x1 * x1 + x2 * x2;
            if (s < 1.0 && s > 0.0) break;
        }
        z = x1 * sqrt(-2.0 * xq_det_log(s) / s);
        v = 1.0 + c * z;
        if (v <= 0.0) continue;
        v = v * v * v;
        uniform2(seed, board, move_no, child, j++, &ua, &ub);
        if (xq_det_log(ua) < 0.5 * z * z + d - d * v + d * xq_det_log(v))
            return d * v * xq_det_exp(xq_det_log(ub) / alpha);
    }
}

void xq_det_pi(const int32_t *visits, int k, double temp, double *pi)
{
    double mx = 0.0, sum = 0.0, it = 1.0 / temp;
    int i;
    for (i = 0; i < k; i++) {
        pi[i] = it * xq_det_log((double)visits[i] + 1e-10);
        if (i == 0 || pi[i] > mx) mx = pi[i];
    }
    for (i = 0; i < k; i++) { pi[i] = xq_det_exp(pi[i] - mx); sum += pi[i]; }
    for (i = 0; i < k; i++) pi[i] = pi[i] / sum;
}

int xq_det_sample(uint64_t seed, uint64_t board_id, uint64_t move_no, const double *pi, int k,
                  double eps, double alpha, double *mixed_out)
{
    double g[XQ_MAX_LEGAL], cdf[XQ_MAX_LEGAL], gs = 0.0, acc = 0.0, ua, ub;
    int i, idx = 0;
    for (i = 0; i < k; i++) { g[i] = det_gamma(seed, board_id, move_no, (uint32_t)i, alpha); gs += g[i]; }
    for (i = 0; i < k; i++) {
        double dir = gs > 0.0 ? g[i] / gs : pi[i];
        double m = (1.0 - eps) * pi[i] + eps * dir;
        if (mixed_out) mixed_out[i] = m;
        acc += m;
        cdf[i] = acc;
    }
    uniform2(seed, board_id, move_no, 0xfffu, 0, &ua, &ub);
    /* np.random.choice: cdf /= cdf[-1]; searchsorted(cdf, u, side="right") */
    for (i = 0; i < k; i++) if (cdf[i] / acc <= ua) idx = i + 1;
    return idx < k ? idx : k - 1;
}
