/* TEST INFRASTRUCTURE ONLY -- plain C restatement of the reference hot loop (checker, never shipped).
 *
 * Per level: evaluate the moment functions of fine and coarse values in the operation order of
 * numpy.polynomial.legendre.legvander / polynomial.polyvander (call sites mlmc/moments.py:126,197),
 * drop samples with a NaN moment (mlmc/quantity/quantity_estimate.py:6-14), accumulate
 * sum(fine - coarse) and sum((fine - coarse)^2) (quantity_estimate.py:59-65); covariance variant
 * accumulates f_i f_j - c_i c_j and its square (quantity_estimate.py:131-147).
 * Sums use Neumaier compensation: this is the checker, it should be at least as accurate as NumPy's
 * pairwise sums.  Build: gcc -O2 -ffp-contract=off -shared -fPIC (oracle/Makefile).
 * Pinned through tests/test_oracle_golden.py::test_c_oracle_matches_numpy_oracle (NumPy oracle is pinned
 * bit-exact to the imported reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { LEGENDRE = 0, MONOMIAL = 1, FOURIER = 2, IDENTITY = 3 };

typedef struct {
    int kind, size;
    double shift, scale, ref0, ref1;
    int is_log, is_clip;
} basis_t;

static double transform(const basis_t *b, double x) {
    if (b->kind == IDENTITY) return x;
    double v = b->is_log ? log(x) : x;
    double t = (v - b->shift) * b->scale + b->ref0; /* moments.py:69-70 */
    if (b->is_clip && (t < b->ref0 || t > b->ref1)) return NAN; /* moments.py:58-67 */
    return t;
}

/* out[0..R) = moment functions of transformed value t; NaN propagates exactly as in NumPy */
static void eval_terms(const basis_t *b, double t, int R, double *out) {
    if (b->kind == IDENTITY) { out[0] = t; return; }
    if (b->kind == FOURIER) {
        out[0] = 1.0; /* res[:, 0] = 1 (moments.py:156); NaN t gives NaN in the other columns only */
        for (int i = 1; i < R; ++i) {
            int k = (i + 1) / 2;
            out[i] = (i & 1) ? cos(t * k) : sin(t * k);
        }
        if (t != t) out[0] = NAN; /* caller treats the sample as masked: any NaN column drops it */
        return;
    }
    out[0] = t * 0 + 1;
    if (R > 1) out[1] = t;
    for (int i = 2; i < R; ++i) {
        if (b->kind == LEGENDRE)
            out[i] = (out[i - 1] * t * (2 * i - 1) - out[i - 2] * (i - 1)) / i;
        else
            out[i] = out[i - 1] * t;
    }
}

static inline void nsum(double *s, double *c, double v) { /* Neumaier */
    double t = *s + v;
    if (fabs(*s) >= fabs(v)) *c += (*s - t) + v; else *c += (v - t) + *s;
    *s = t;
}

static int has_nan(const double *v, int R) {
    for (int i = 0; i < R; ++i) if (v[i] != v[i]) return 1;
    return 0;
}

/* coarse == NULL at level 0. s, sp: [R] outputs. returns 0 */
int oracle_moments_level(const basis_t *b, const double *fine, const double *coarse, int64_t n,
                         double *s, double *sp, int64_t *n_keep, int64_t *n_rm) {
    int R = b->size;
    double *f = malloc(sizeof(double) * R * 6);
    double *c = f + R, *cs = c + R, *csp = cs + R, *ss = csp + R, *ssp = ss + R;
    memset(cs, 0, sizeof(double) * R * 4);
    int64_t keep = 0, rm = 0;
    for (int64_t k = 0; k < n; ++k) {
        eval_terms(b, transform(b, fine[k]), R, f);
        int bad = has_nan(f, R);
        if (coarse) { eval_terms(b, transform(b, coarse[k]), R, c); bad |= has_nan(c, R); }
        if (bad) { rm++; continue; }
        keep++;
        for (int i = 0; i < R; ++i) {
            double d = coarse ? f[i] - c[i] : f[i];
            nsum(&ss[i], &cs[i], d);
            nsum(&ssp[i], &csp[i], d * d);
        }
    }
    for (int i = 0; i < R; ++i) { s[i] = ss[i] + cs[i]; sp[i] = ssp[i] + csp[i]; }
    *n_keep = keep; *n_rm = rm;
    free(f);
    return 0;
}

/* s, sp: [R*R] outputs, row-major (i, j) */
int oracle_cov_level(const basis_t *b, const double *fine, const double *coarse, int64_t n,
                     double *s, double *sp, int64_t *n_keep, int64_t *n_rm) {
    int R = b->size;
    size_t RR = (size_t)R * R;
    double *f = malloc(sizeof(double) * (2 * R + 4 * RR));
    double *c = f + R, *ss = c + R, *cs = ss + RR, *ssp = cs + RR, *csp = ssp + RR;
    memset(ss, 0, sizeof(double) * 4 * RR);
    int64_t keep = 0, rm = 0;
    for (int64_t k = 0; k < n; ++k) {
        eval_terms(b, transform(b, fine[k]), R, f);
        int bad = has_nan(f, R);
        if (coarse) { eval_terms(b, transform(b, coarse[k]), R, c); bad |= has_nan(c, R); }
        if (bad) { rm++; continue; }
        keep++;
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < R; ++j) {
                double d = f[i] * f[j];
                if (coarse) d -= c[i] * c[j];
                nsum(&ss[i * R + j], &cs[i * R + j], d);
                nsum(&ssp[i * R + j], &csp[i * R + j], d * d);
            }
    }
    for (size_t i = 0; i < RR; ++i) { s[i] = ss[i] + cs[i]; sp[i] = ssp[i] + csp[i]; }
    *n_keep = keep; *n_rm = rm;
    free(f);
    return 0;
}
