// Largest singular value (squared) of an image: the "rms" objective of the reference (objectives.py:266-306) forms
// np.linalg.norm(iwe, 2) of a 2-D array, i.e. the SPECTRAL norm, not the Frobenius norm.  sigma_max^2 is the largest eigenvalue
// of the Gram matrix; it comes from a Lanczos iteration on  v -> A^T (A v)  (or A (A^T v): the shorter side of the image) with
// full re-orthogonalisation, all in float64 in ONE workgroup (the images are <= 4 MB and L2-resident; two matrix-vector products
// per step, at most 96 steps), then a bisection on the tridiagonal matrix the iteration leaves.  Round 5: until then the Gram
// matrix and its eigenvalues came from torch (rocBLAS + rocSOLVER) -- the one place a library routine computed on this path.
#include <mutex>

#include "evk_common.h"

namespace evk {

#define SPEC_THREADS 1024
#define SPEC_MAX_STEPS 96
#define SPEC_MAX_RESTARTS 8    // sweeps of <= SPEC_MAX_STEPS steps, each starting from the previous one's Ritz vector
#define SPEC_MAX_DIM 4096      // longest image side (the two work vectors live in LDS)

// block-wide sum of one double per thread; every thread gets the total (two barriers)
__device__ __forceinline__ double spec_block_sum(double v, double *red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();                       // (red may still be read from the previous call)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < SPEC_THREADS / 64; ++k) t += red[k];   // same order in every thread
    return t;
}

// a: (h, w) row-major float32.  n = min(h, w) is the Lanczos dimension, big = max(h, w).  basis: SPEC_MAX_STEPS * n doubles.
__global__ void __launch_bounds__(SPEC_THREADS) k_spectral_norm_sq(const float *__restrict__ a, int h, int w,
                                                                   double *__restrict__ basis, double *__restrict__ out) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool tall = h >= w;                  // tall: vectors have w entries, op(v) = A^T (A v); else h entries, A (A^T v)
    const int n = tall ? w : h, big = tall ? h : w;
    double *v = sm, *wv = sm + n, *u = sm + 2 * n, *coef = u + big, *alpha = coef + SPEC_MAX_STEPS, *beta = alpha + SPEC_MAX_STEPS,
           *red = beta + SPEC_MAX_STEPS;
    // y[r] = sum_c a[r][c] x[c] (rows: one wave per row, coalesced) ; z[c] = sum_r a[r][c] y[r] (columns: one thread per column)
    auto rows_times = [&](const double *x, double *y) {       // y (h) = A x (w)
        for (int r = wave; r < h; r += SPEC_THREADS / 64) {
            double s = 0.0;
            for (int c = lane; c < w; c += 64) s += (double)a[(int64_t)r * w + c] * x[c];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) y[r] = s;
        }
    };
    auto cols_times = [&](const double *y, double *z) {       // z (w) = A^T y (h)
        for (int c = tid; c < w; c += SPEC_THREADS) {
            double s = 0.0;
            for (int r = 0; r < h; ++r) s += (double)a[(int64_t)r * w + c] * y[r];
            z[c] = s;
        }
    };
    // start vector: positive, not constant (an image's dominant singular vector is never orthogonal to it in practice; the
    // re-orthogonalised iteration would recover from rounding noise even then)
    double nrm = 0.0;
    for (int i = tid; i < n; i += SPEC_THREADS) {
        const double x = 1.0 + (double)(((uint32_t)i * 2654435761u >> 20) & 1023u) * (1.0 / 2048.0);
        v[i] = x, nrm += x * x;
    }
    nrm = sqrt(spec_block_sum(nrm, red));
    for (int i = tid; i < n; i += SPEC_THREADS) v[i] /= nrm;
    __syncthreads();
    const int msteps = n < SPEC_MAX_STEPS ? n : SPEC_MAX_STEPS;
    __shared__ double lam_sh;
    __shared__ double tdd[SPEC_MAX_STEPS], tdu[SPEC_MAX_STEPS], tdl[SPEC_MAX_STEPS], tdu2[SPEC_MAX_STEPS];
    __shared__ int tpiv[SPEC_MAX_STEPS];
    // RESTARTS (round 6).  96 steps need not be enough for a slowly converging spectrum, and a Ritz value is a LOWER bound: the
    // iteration used to return it unchecked.  Now the Ritz vector y is formed, the true residual |op(y) - lambda y| is
    // measured (|lambda - an eigenvalue| <= that residual, always), and while it exceeds 1e-10 lambda the iteration starts
    // again FROM y (a thick restart with one vector: every sweep begins where the last one ended), at most SPEC_MAX_RESTARTS
    // sweeps.  One more operator application per sweep; the usual image converges in the first.
    for (int sweep_no = 0; sweep_no < SPEC_MAX_RESTARTS; ++sweep_no) {
    int m = 0;
    double scale = 0.0;                       // largest |alpha| so far: what "zero" is measured against
    for (int j = 0; j < msteps; ++j) {
        for (int i = tid; i < n; i += SPEC_THREADS) basis[(int64_t)j * n + i] = v[i];
        if (tall) { rows_times(v, u); __syncthreads(); cols_times(u, wv); }
        else { cols_times(v, u); __syncthreads(); rows_times(u, wv); }
        __syncthreads();
        double d = 0.0;
        for (int i = tid; i < n; i += SPEC_THREADS) d += wv[i] * v[i];
        const double aj = spec_block_sum(d, red);
        if (tid == 0) alpha[j] = aj;
        scale = fmax(scale, fabs(aj));
        m = j + 1;
        // full re-orthogonalisation against v_0 .. v_j, twice (classical Gram-Schmidt: all coefficients of a sweep at once,
        // one wave per basis vector)
        for (int sweep = 0; sweep < 2; ++sweep) {
            __syncthreads();                  // basis row j is written, wv is complete
            for (int k = wave; k <= j; k += SPEC_THREADS / 64) {
                double s = 0.0;
                for (int i = lane; i < n; i += 64) s += wv[i] * basis[(int64_t)k * n + i];
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
                if (lane == 0) coef[k] = s;
            }
            __syncthreads();
            for (int i = tid; i < n; i += SPEC_THREADS) {
                double s = wv[i];
                for (int k = 0; k <= j; ++k) s -= coef[k] * basis[(int64_t)k * n + i];
                wv[i] = s;
            }
        }
        __syncthreads();
        double b2 = 0.0;
        for (int i = tid; i < n; i += SPEC_THREADS) b2 += wv[i] * wv[i];
        const double bj = sqrt(spec_block_sum(b2, red));
        if (tid == 0) beta[j] = bj;
        if (!(bj > 1e-13 * scale) || !(scale > 0.0)) break;     // an invariant subspace (low-rank images), or the zero image
        for (int i = tid; i < n; i += SPEC_THREADS) v[i] = wv[i] / bj;
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) {
        // largest eigenvalue of the m x m tridiagonal (alpha, beta): bisection on the Sturm count
        double lo = 0.0, hi = 0.0;
        for (int k = 0; k < m; ++k) {
            const double r = fabs(alpha[k]) + (k > 0 ? fabs(beta[k - 1]) : 0.0) + (k + 1 < m ? fabs(beta[k]) : 0.0);
            hi = fmax(hi, r);
        }
        lo = -hi;
        const double hi0 = hi;   // (the Gershgorin bound: the scale of the tridiagonal)
        for (int it = 0; it < 200 && hi - lo > 1e-15 * fmax(fabs(hi), fabs(lo)); ++it) {
            const double x = 0.5 * (lo + hi);
            int below = 0;                    // eigenvalues < x
            double q = 1.0;
            for (int k = 0; k < m; ++k) {
                const double bb = k > 0 ? beta[k - 1] * beta[k - 1] : 0.0;
                q = alpha[k] - x - (k > 0 ? bb / q : 0.0);
                if (q == 0.0) q = 1e-300;
                below += q < 0.0 ? 1 : 0;
            }
            if (below >= m) hi = x; else lo = x;
        }
        const double lam = m ? fmax(0.5 * (lo + hi), 0.0) : 0.0;
        lam_sh = lam;
        out[0] = lam, out[1] = 0.0;
        // its eigenvector of the tridiagonal by INVERSE ITERATION (a tridiagonal LU with partial pivoting, three solves): the
        // three-term recurrence breaks down exactly where the Lanczos iteration does -- an invariant subspace, beta ~ 0 (images
        // of a handful of events) -- and a poor vector would show as a large residual of a VALUE that is exact there
        {
            const double sig = lam + 1e-13 * fmax(hi0, 1e-300);   // (just above the eigenvalue: T - sig I is non-singular)
            for (int k = 0; k < m; ++k) {
                tdd[k] = alpha[k] - sig;
                tdu[k] = tdl[k] = k + 1 < m ? beta[k] : 0.0;
                tdu2[k] = 0.0, tpiv[k] = 0, coef[k] = 1.0 + 0.01 * k;
            }
            const double tiny = 1e-30 * fmax(hi0, 1e-300);
            for (int k = 0; k + 1 < m; ++k) {
                if (fabs(tdd[k]) >= fabs(tdl[k])) {
                    if (tdd[k] == 0.0) tdd[k] = tiny;
                    const double f = tdl[k] / tdd[k];
                    tdl[k] = f, tdd[k + 1] -= f * tdu[k], tpiv[k] = 0;
                } else {
                    const double f = tdd[k] / tdl[k];
                    tdd[k] = tdl[k], tdl[k] = f;
                    const double t2 = tdd[k + 1];
                    tdd[k + 1] = tdu[k] - f * t2;
                    if (k + 2 < m) tdu2[k] = tdu[k + 1], tdu[k + 1] = -f * tdu2[k];
                    tdu[k] = t2, tpiv[k] = 1;
                }
            }
            if (m && tdd[m - 1] == 0.0) tdd[m - 1] = tiny;
            for (int it = 0; it < 3; ++it) {
                for (int k = 0; k + 1 < m; ++k) {
                    if (tpiv[k]) { const double t2 = coef[k]; coef[k] = coef[k + 1], coef[k + 1] = t2; }
                    coef[k + 1] -= tdl[k] * coef[k];
                }
                for (int k = m - 1; k >= 0; --k) {
                    double r = coef[k];
                    if (k + 1 < m) r -= tdu[k] * coef[k + 1];
                    if (k + 2 < m) r -= tdu2[k] * coef[k + 2];
                    coef[k] = r / tdd[k];
                }
                double nn = 0.0, big = 0.0;
                for (int k = 0; k < m; ++k) big = fmax(big, fabs(coef[k]));
                if (!(big > 0.0) || !(big < 1e300)) { for (int k = 0; k < m; ++k) coef[k] = k == 0 ? 1.0 : 0.0; break; }
                for (int k = 0; k < m; ++k) coef[k] /= big, nn += coef[k] * coef[k];
                nn = 1.0 / sqrt(nn);
                for (int k = 0; k < m; ++k) coef[k] *= nn;
            }
        }
    }
    __syncthreads();
    const double lam = lam_sh;
    if (m == 0) break;
    // Ritz vector y = sum_k s_k v_k -> v (normalised), true residual of (lam, y)
    double yn = 0.0;
    for (int i = tid; i < n; i += SPEC_THREADS) {
        double y = 0.0;
        for (int k = 0; k < m; ++k) y += coef[k] * basis[(int64_t)k * n + i];
        v[i] = y, yn += y * y;
    }
    yn = sqrt(spec_block_sum(yn, red));
    if (!(yn > 0.0)) break;
    for (int i = tid; i < n; i += SPEC_THREADS) v[i] /= yn;
    __syncthreads();
    if (tall) { rows_times(v, u); __syncthreads(); cols_times(u, wv); }
    else { cols_times(v, u); __syncthreads(); rows_times(u, wv); }
    __syncthreads();
    double r2 = 0.0, ry = 0.0;
    for (int i = tid; i < n; i += SPEC_THREADS) ry += wv[i] * v[i];
    const double rq = spec_block_sum(ry, red);          // Rayleigh quotient of y: never below lam's sweep, never above the truth
    for (int i = tid; i < n; i += SPEC_THREADS) { const double d = wv[i] - rq * v[i]; r2 += d * d; }
    const double rn = sqrt(spec_block_sum(r2, red));
    if (tid == 0) out[0] = fmax(rq, lam), out[1] = rq > 0.0 ? rn / rq : 0.0;   // the value and its relative error bound
    if (rn <= 1e-10 * rq || !(rq > 0.0)) break;          // converged (or the zero image)
    __syncthreads();
    }
}

}  // namespace evk

using namespace evk;

extern "C" int64_t evk_spectral_scratch_bytes(int h, int w) {
    if (h <= 0 || w <= 0 || h > SPEC_MAX_DIM || w > SPEC_MAX_DIM) return 0;
    return (int64_t)SPEC_MAX_STEPS * (h < w ? h : w) * (int64_t)sizeof(double);
}

extern "C" int evk_spectral_norm_sq_f32(const float *img, int h, int w, double *out, void *scratch, int64_t scratch_bytes,
                                        void *stream) {
    if (!img || !out || !scratch || h <= 0 || w <= 0 || h > SPEC_MAX_DIM || w > SPEC_MAX_DIM) return EVK_EINVAL;
    if (scratch_bytes < evk_spectral_scratch_bytes(h, w)) return EVK_ESCRATCH;
    const int n = h < w ? h : w, big = h < w ? w : h;
    const size_t lds = (size_t)(2 * n + big + 3 * SPEC_MAX_STEPS + SPEC_THREADS / 64) * sizeof(double);
    static std::once_flag once[64];   // per DEVICE: the attribute belongs to the code object loaded there (and thread-safe)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::call_once(once[dev & 63], [] {
        (void)hipFuncSetAttribute((const void *)k_spectral_norm_sq, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    k_spectral_norm_sq<<<1, SPEC_THREADS, lds, (hipStream_t)stream>>>(img, h, w, (double *)scratch, out);
    return launch_status();
}
