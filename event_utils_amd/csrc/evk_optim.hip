// The whole optimisation of the linear-flow contrast objective in ONE library call (round 6): the quasi-Newton iteration of
// events_cmax.evk_bfgs -- every pass one evk_cmax_variance_tiled_f32 (value + gradient) or one
// evk_cmax_variance_batch3_tiled_f32 (three step lengths of the line search) -- with the iteration's arithmetic, the LDS
// window of every pass and the result poll in C.  Between two passes of the Python loop lay ~8-10 us of interpreter
// arithmetic and ~7 us of call marshalling (tools/bfgs_passes.py, tools/again_gap.py), a sixth of a pass on 1 M events; here
// the next pass is enqueued as soon as the pinned result slot of the previous one has been read.
// Replaces the loop scipy.optimize.fmin_bfgs runs for the reference (events_cmax.py:313-346); a C host gets the whole of
// optimize_contrast behind one entry point.
//
// The arithmetic follows events_cmax.evk_bfgs operation by operation (same order, float64, pow(., 0.5) for the norms as
// Python's `** 0.5`): both loops visit bit-identical points (tests/test_gpu_hardening.py).
#include "evk_common.h"

#include <cmath>

namespace {

constexpr int WIN_MAX3 = 48;  // LDS window edge cap with three planes (tiled.py: _WIN_MAX[3])

struct Loop {
    // the evaluation calls' arguments
    const float *records;
    const uint32_t *index;
    int64_t n;
    int dom_h, dom_w, tw_log2, th_log2;
    double t_first, t_ref, bounds_w, bounds_h;
    int ch, cw;
    uint32_t iwe_flags;
    double p_scale, p_bound, dt_bound;
    const double *weights;
    int radius;
    uint32_t post_flags;
    void *staging;
    int64_t staging_bytes;
    float *iwe_buf;
    double *out12;
    void *scratch;
    int64_t scratch_bytes;
    float *spill;
    int *parity;
    void *stream;
    // derived
    double span;
    int ntiles;
    int passes = 0;
    bool replan = false;  // a flow the tiled kernels cannot take with this staging buffer: the caller decides
    bool numeric = false;
};

// tiled.py:_iwe_window + the candidate-window test of iwe_plan / _retarget, three planes
static bool window(const Loop &L, double Dx, double Dy, int &S, int &win_w, int &win_h) {
    if (!(std::isfinite(Dx) && std::isfinite(Dy))) return false;
    const int tw = 1 << L.tw_log2, th = 1 << L.th_log2;
    const double s = std::ceil(std::fmax(Dx / (double)(WIN_MAX3 - tw - 4), Dy / (double)(WIN_MAX3 - th - 4)));
    if (!(s <= 64.0)) return false;
    S = s < 1.0 ? 1 : (int)s;
    auto rnd = [](double v) {
        const int r = ((int)v + 3) / 4 * 4;
        return r < WIN_MAX3 ? r : WIN_MAX3;
    };
    win_w = rnd((double)tw + std::ceil(Dx / (double)S) + 4.0);
    win_h = rnd((double)th + std::ceil(Dy / (double)S) + 4.0);
    const double cand = (std::ceil((Dx + win_w) / (double)tw) + 1.0) * (std::ceil((Dy + win_h) / (double)th) + 1.0) * (double)S;
    if (cand > 128.0) return false;
    return evk_iwe_tiled_staging_bytes(L.ntiles, L.n, S, 3, win_w, win_h) <= L.staging_bytes;
}

// f(q) and its gradient: one pass.  objectives.py: f = float32(-var), g = float32(-g_i)
static int eval_fg(Loop &L, const double q[2], double &f, double g[2]) {
    int S, ww, wh;
    if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !window(L, std::fabs(q[0]) * L.span, std::fabs(q[1]) * L.span, S, ww, wh)) {
        L.replan = true;
        return EVK_OK;
    }
    double r[4];
    const int rc = evk_cmax_variance_tiled_f32(L.records, L.index, L.n, L.dom_h, L.dom_w, L.tw_log2, L.th_log2, S, ww, wh,
                                               L.t_first, L.t_ref, q[0], q[1], L.bounds_w, L.bounds_h, L.ch, L.cw,
                                               L.iwe_flags | EVK_IWE_GRADIENT, L.p_scale, L.p_bound, L.dt_bound, L.weights,
                                               L.radius, L.post_flags | EVK_POST_VALUE, L.staging, L.staging_bytes, L.iwe_buf,
                                               L.out12, L.scratch, L.scratch_bytes, L.spill, *L.parity ^ 1, r, L.stream);
    if (rc != EVK_OK) return rc;
    *L.parity ^= 1;
    ++L.passes;
    f = (double)(float)(-r[3]);
    g[0] = (double)(float)(-r[0]);
    g[1] = (double)(float)(-r[1]);
    return EVK_OK;
}

// f at three flows: one pass
static int eval_f3(Loop &L, const double pts[3][2], double fs[3]) {
    double vx[3], vy[3], ax = 0.0, ay = 0.0;
    bool finite = true;
    for (int k = 0; k < 3; ++k) {
        vx[k] = pts[k][0], vy[k] = pts[k][1];
        finite = finite && std::isfinite(vx[k]) && std::isfinite(vy[k]);
        ax = std::fmax(ax, std::fabs(vx[k])), ay = std::fmax(ay, std::fabs(vy[k]));
    }
    int S, ww, wh;
    if (!finite || !window(L, ax * L.span, ay * L.span, S, ww, wh)) {
        L.replan = true;
        return EVK_OK;
    }
    double r[12];
    const int rc = evk_cmax_variance_batch3_tiled_f32(L.records, L.index, L.n, L.dom_h, L.dom_w, L.tw_log2, L.th_log2, S, ww, wh,
                                                      L.t_first, L.t_ref, vx, vy, L.bounds_w, L.bounds_h, L.ch, L.cw,
                                                      L.iwe_flags, L.p_scale, L.p_bound, L.dt_bound, L.weights, L.radius,
                                                      L.staging, L.staging_bytes, L.iwe_buf, L.out12, L.scratch,
                                                      L.scratch_bytes, L.spill, *L.parity ^ 1, r, L.stream);
    if (rc != EVK_OK) return rc;
    *L.parity ^= 1;
    ++L.passes;
    for (int k = 0; k < 3; ++k) fs[k] = (double)(float)(-r[4 * k + 1]);
    return EVK_OK;
}

// An evaluator the host supplies through the C ABI (evk_bfgs2_minimize): the same loop for any two-parameter objective
struct Callbacks {
    evk_bfgs2_fg_fn fg;
    evk_bfgs2_f3_fn f3;
    void *user;
    int passes = 0;
    bool replan = false;
    bool numeric = false;
};
static int eval_fg(Callbacks &C, const double q[2], double &f, double g[2]) {
    if (!C.fg) return EVK_EINVAL;
    const int rc = C.fg(C.user, q, &f, g);
    if (rc == 1) {
        C.replan = true;
        return EVK_OK;
    }
    if (rc == EVK_OK) ++C.passes;
    return rc;
}
static int eval_f3(Callbacks &C, const double pts[3][2], double fs[3]) {
    if (!C.f3) return EVK_EINVAL;
    const int rc = C.f3(C.user, &pts[0][0], fs);
    if (rc == 1) {
        C.replan = true;
        return EVK_OK;
    }
    if (rc == EVK_OK) ++C.passes;
    return rc;
}

// value + gradient the way the run asked for it: analytic, or forward differences with epsilon = 1 from one three-flow pass
// (the reference's default, events_cmax.py:343)
template <typename EVAL>
static int value_grad(EVAL &L, const double q[2], double &f, double g[2]) {
    if (!L.numeric) return eval_fg(L, q, f, g);
    const double pts[3][2] = {{q[0], q[1]}, {q[0] + 1.0, q[1]}, {q[0], q[1] + 1.0}};
    double fs[3];
    const int rc = eval_f3(L, pts, fs);
    if (rc != EVK_OK || L.replan) return rc;
    f = fs[0];
    g[0] = (fs[1] - fs[0]) / (pts[1][0] - q[0]);
    g[1] = (fs[2] - fs[0]) / (pts[2][1] - q[1]);
    return EVK_OK;
}

static inline double dot2(const double a[2], const double b[2]) {
    double s = 0.0;
    s += a[0] * b[0];
    s += a[1] * b[1];
    return s;
}
static inline double norm2(const double a[2]) { return std::pow(dot2(a, a), 0.5); }
static inline void axpy2(double al, const double d[2], const double base[2], double out[2]) {
    out[0] = base[0] + al * d[0];
    out[1] = base[1] + al * d[1];
}

// events_cmax.evk_bfgs, two parameters, on any evaluator
template <typename EVAL>
static int minimise(EVAL &L, const double *x0, const double *opts, double *result, int trace_cap) {
    const double xtol = opts[0], gtol = opts[1], ftol = opts[2];
    const int maxiter = (int)opts[3];
    L.numeric = opts[4] != 0.0;
    const bool unit_first = opts[5] != 0.0;

    int npoints = 0;
    double *trace = result + 6;
    auto push = [&](const double q[2], double fv, const double gv[2]) {
        if (npoints < trace_cap) {
            double *row = trace + 5 * npoints;
            row[0] = q[0], row[1] = q[1], row[2] = fv, row[3] = gv[0], row[4] = gv[1];
        }
        ++npoints;
    };
    auto finish = [&](const double q[2], double fv, int status) {
        result[0] = q[0], result[1] = q[1], result[2] = fv, result[3] = (double)npoints, result[4] = (double)L.passes;
        result[5] = (double)status;
    };
#define EVK_STEP(call)                         \
    do {                                       \
        const int rc_ = (call);                \
        if (rc_ != EVK_OK) return rc_;         \
        if (L.replan) {                        \
            finish(x, f, 1);                   \
            return EVK_OK;                     \
        }                                      \
    } while (0)

    double x[2] = {x0[0], x0[1]}, f = 0.0, g[2] = {0.0, 0.0};
    EVK_STEP(value_grad(L, x, f, g));
    push(x, f, g);
    double Hm[2][2] = {{1.0, 0.0}, {0.0, 1.0}};
    bool have_curvature = false;
    double scale = 1.0 / std::fmax(norm2(g), 1e-12);
    for (int it = 0; it < maxiter; ++it) {
        if (std::fmax(std::fabs(g[0]), std::fabs(g[1])) <= gtol) break;
        double d[2] = {-dot2(Hm[0], g), -dot2(Hm[1], g)};
        double slope = dot2(g, d);
        if (!(slope < 0.0)) {  // not a descent direction: restart from steepest descent
            Hm[0][0] = 1.0, Hm[0][1] = 0.0, Hm[1][0] = 0.0, Hm[1][1] = 1.0;
            have_curvature = false;
            d[0] = -g[0], d[1] = -g[1];
            slope = -dot2(g, g);
        }
        bool has_best = false, has_new = false;
        double best_f = 0.0, best_a = 0.0, a = scale, f_new = 0.0, g_new[2] = {0.0, 0.0};
        int grown = 0;
        const double dn = norm2(d);
        if (unit_first && have_curvature) {
            double q[2], f1, g1[2];
            axpy2(1.0, d, x, q);
            EVK_STEP(value_grad(L, q, f1, g1));
            if (f1 <= f + 1e-4 * slope) {
                has_best = true, best_f = f1, best_a = 1.0;
                has_new = true, f_new = f1, g_new[0] = g1[0], g_new[1] = g1[1];
            } else {
                a = 1.0 / 9.0, grown = 4;
            }
        }
        while (!has_new && a * dn >= 0.5 * xtol) {
            const double alphas[3] = {a / 3.0, a, 3.0 * a};
            double pts[3][2], fs[3];
            for (int k = 0; k < 3; ++k) axpy2(alphas[k], d, x, pts[k]);
            EVK_STEP(eval_f3(L, pts, fs));
            bool any = false;
            double mf = 0.0, ma = 0.0;  // min over the (value, step length) pairs that satisfy the Armijo condition
            for (int k = 0; k < 3; ++k)
                if (fs[k] <= f + 1e-4 * alphas[k] * slope) {
                    if (!any || fs[k] < mf || (fs[k] == mf && alphas[k] < ma)) mf = fs[k], ma = alphas[k];
                    any = true;
                }
            if (any) {
                if (!has_best || mf < best_f) has_best = true, best_f = mf, best_a = ma;
                if (best_a == alphas[2] && grown < 4) {
                    a = 9.0 * a, ++grown;
                    continue;
                }
                break;
            }
            if (has_best) break;
            a /= 27.0;
        }
        if (!has_best) break;
        double x_new[2];
        axpy2(best_a, d, x, x_new);
        if (!has_new) EVK_STEP(value_grad(L, x_new, f_new, g_new));
        const double s[2] = {x_new[0] - x[0], x_new[1] - x[1]}, y[2] = {g_new[0] - g[0], g_new[1] - g[1]};
        const double sy = dot2(y, s);
        if (sy > 1e-12) {  // H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T
            have_curvature = true;
            const double rho = 1.0 / sy;
            double A[2][2], AH[2][2], Hn[2][2];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) A[i][j] = (i == j ? 1.0 : 0.0) - rho * s[i] * y[j];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) AH[i][j] = 0.0 + A[i][0] * Hm[0][j] + A[i][1] * Hm[1][j];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) Hn[i][j] = (0.0 + AH[i][0] * A[j][0] + AH[i][1] * A[j][1]) + rho * s[i] * s[j];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) Hm[i][j] = Hn[i][j];
        }
        const double gain = f - f_new;
        x[0] = x_new[0], x[1] = x_new[1], f = f_new, g[0] = g_new[0], g[1] = g_new[1], scale = 1.0;
        push(x, f, g);
        if (norm2(s) < xtol || gain <= ftol * std::fabs(f)) break;
    }
#undef EVK_STEP
    finish(x, f, 0);
    return EVK_OK;
}

}  // namespace

extern "C" int evk_cmax_bfgs_variance_tiled_f32(const float *records, const uint32_t *bucket_index, int64_t n, int dom_h, int dom_w,
                                                int tw_log2, int th_log2, double t_first, double t_ref, double bounds_w,
                                                double bounds_h, int canvas_h, int canvas_w, uint32_t iwe_flags, double p_scale,
                                                double p_bound, double dt_bound, const double *host_weights, int radius,
                                                uint32_t post_flags, void *staging, int64_t staging_bytes, float *iwe_buf,
                                                double *out12, void *scratch, int64_t scratch_bytes, float *spill_pair,
                                                int *parity, const double *x0, const double *opts, double *result,
                                                int trace_cap, void *stream) {
    if (!x0 || !opts || !result || !parity || !spill_pair || trace_cap < 0 || (iwe_flags & EVK_IWE_GRADIENT)) return EVK_EINVAL;
    Loop L;
    L.records = records, L.index = bucket_index, L.n = n, L.dom_h = dom_h, L.dom_w = dom_w, L.tw_log2 = tw_log2, L.th_log2 = th_log2;
    L.t_first = t_first, L.t_ref = t_ref, L.bounds_w = bounds_w, L.bounds_h = bounds_h, L.ch = canvas_h, L.cw = canvas_w;
    L.iwe_flags = iwe_flags, L.p_scale = p_scale, L.p_bound = p_bound, L.dt_bound = dt_bound, L.weights = host_weights;
    L.radius = radius, L.post_flags = post_flags & ~(EVK_POST_NONE | EVK_POST_VALUE), L.staging = staging, L.staging_bytes = staging_bytes;
    L.iwe_buf = iwe_buf, L.out12 = out12, L.scratch = scratch, L.scratch_bytes = scratch_bytes, L.spill = spill_pair;
    L.parity = parity, L.stream = stream;
    L.span = std::fabs(t_first - t_ref);
    L.ntiles = evk_bucket_num_tiles(dom_h, dom_w, tw_log2, th_log2);
    if (L.ntiles <= 0) return EVK_EINVAL;
    return minimise(L, x0, opts, result, trace_cap);
}

extern "C" int evk_bfgs2_minimize(evk_bfgs2_fg_fn fg, evk_bfgs2_f3_fn f3, void *user, const double *x0, const double *opts,
                                  double *result, int trace_cap) {
    if (!x0 || !opts || !result || trace_cap < 0 || !f3 || (!fg && opts[4] == 0.0)) return EVK_EINVAL;
    Callbacks C;
    C.fg = fg, C.f3 = f3, C.user = user;
    return minimise(C, x0, opts, result, trace_cap);
}
