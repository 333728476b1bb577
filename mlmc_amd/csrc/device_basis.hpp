// Device-side moment-function generators shared by the moments and covariance kernels.
//
// Reference semantics (mlmc/moments.py): t = (x - shift) * scale + ref0, optionally after log(x)
// (:27-39, :69-70); with safe_eval a value whose t lies outside [ref0, ref1] becomes NaN (:58-67) and
// the whole sample is later dropped by mask_nan_samples (quantity_estimate.py:6-14).  The transform is
// evaluated with two separate roundings (the file is compiled with -ffp-contract=off) so that the
// keep/drop decision is bit-identical to NumPy's; everything after the decision may use FMA.
#pragma once
#include "common.hpp"

namespace mlmc {

// Transformed value and keep flag of one raw sample value.
__device__ __forceinline__ double transform_value(const BasisParams &bp, double x, bool &keep) {
    if (bp.kind == MLMC_IDENTITY) {
        keep = !(x != x);
        return x;
    }
    if (bp.is_log) {
        // The reference decides on t computed with NumPy's log (moments.py:27-39); the device log may differ from it in the
        // last bit, which would flip a sample that sits on the edge of the domain.  The decision is therefore taken on the
        // raw value against thresholds the host has bisected with its own log (x -> t is monotone); the device log only
        // supplies the VALUE of t (1 ulp of it is far inside the 1e-10 value tolerance).
        const double tl = (log(x) - bp.shift) * bp.scale + bp.ref0;
        if (bp.is_clip)
            keep = (x >= bp.x_lo) && (x <= bp.x_hi);                 // NaN -> false
        else
            keep = (x > 0.0) && (x < __builtin_inf()) && !(tl != tl) && !(fabs(tl) == __builtin_inf());
        return tl;
    }
    double t = (x - bp.shift) * bp.scale + bp.ref0;   // two roundings, no contraction
    if (bp.is_clip)
        keep = (t >= bp.ref0) && (t <= bp.ref1);      // NaN -> false
    else
        keep = !(t != t) && !(fabs(t) == __builtin_inf());   // x*0+1 is NaN for NaN/inf (legvander/polyvander v[0])
    return t;
}

// The same for the common configuration (no log, safe_eval clipping), free of run-time switches.
__device__ __forceinline__ double transform_plain(const BasisParams &bp, double x, bool &keep) {
    const double t = (x - bp.shift) * bp.scale + bp.ref0;   // two roundings, no contraction
    keep = (t >= bp.ref0) && (t <= bp.ref1);                // NaN -> false
    return t;
}

// Coefficients 4 g_i, g_i = (i-1)^2 / ((2i-1)(2i-3)), of the scaled monic Legendre recurrence (TermGen below) as a
// compile-time table: in the fully unrolled accumulation kernels every index is a constant, so the values become
// instruction operands (s_mov literals) instead of scalar loads that the wave would have to wait for inside the hot loop.
constexpr int LEGENDRE_MAX_TERMS = 512;
struct LegendreG {
    double v[LEGENDRE_MAX_TERMS];
    constexpr LegendreG() : v() {
        for (int i = 0; i < LEGENDRE_MAX_TERMS; ++i)
            v[i] = i < 2 ? 0.0 : 4.0 * ((double)((long long)(i - 1) * (i - 1)) / (double)((long long)(2 * i - 1) * (2 * i - 3)));
    }
};
__device__ constexpr LegendreG kLegendreG = LegendreG();

// Sequential generator of basis terms 0, 1, 2, ... for one value.  `w` (1 = kept, 0 = masked) is
// folded into the seed so that a masked value yields exactly 0 in every term.
//
// LEGENDRE uses the monic recurrence Q_i = x Q_{i-1} - g_i Q_{i-2}, g_i = (i-1)^2 / ((2i-1)(2i-3)),
// i.e. one multiply and one FMA per term instead of the five operations of
// numpy.polynomial.legendre.legvander (P_i = (P_{i-1} x (2i-1) - P_{i-2} (i-1)) / i), in the scaled form
// q_i = 2^i Q_i:  q_i = (2x) q_{i-1} - (4 g_i) q_{i-2}.  The monic Q_i shrink like 2^-i on [-1, 1] (their squares
// would leave the normal fp64 range near i = 480), the q_i stay O(1) up to the largest accepted size; powers of two
// scale exactly, so every q_i is bit for bit 2^i times the unscaled value.  The true Legendre value is
// P_i = c_i q_i with c_i = (leading coefficient of P_i) / 2^i ~ 1 / sqrt(pi i), applied once to the finished sums.
template <int KIND>
struct TermGen {
    double x, p1, p2, c1, s1;
    __device__ __forceinline__ void init(double x_, double w, const BasisParams &) {
        x = KIND == MLMC_LEGENDRE ? 2.0 * x_ : x_;
        p1 = w;   // term 0
        p2 = 0.0;
        if (KIND == MLMC_FOURIER) {
            sincos(x_, &s1, &c1);
            p1 = w;    // cos(0 t) * w
            p2 = 0.0;  // sin(0 t) * w
        }
    }
    // one step at an index >= 2 known only at run time: the term's coefficient comes from the caller (fetched ahead of the
    // dependent chain), `j` has the parity of the index
    __device__ __forceinline__ double skip(int j, double g) {
        if (KIND == MLMC_LEGENDRE) {
            const double q = __builtin_fma(x, p1, -(g * p2));
            p2 = p1;
            p1 = q;
            return q;
        } else {
            return next(2 + (j & 1));
        }
    }
    // must be called with i = 0, 1, 2, ... in order
    __device__ __forceinline__ double next(int i) {
        if (i == 0) return p1;
        if (KIND == MLMC_LEGENDRE) {
            double q;
            if (i == 1) q = x * p1;
            else q = __builtin_fma(x, p1, -(kLegendreG.v[i] * p2));
            p2 = p1;
            p1 = q;
            return q;
        } else if (KIND == MLMC_MONOMIAL) {
            p1 = p1 * x;
            return p1;
        } else if (KIND == MLMC_FOURIER) {
            if (i & 1) {   // cos(k t), k = (i + 1) / 2: rotate (cos, sin) by t
                double c = __builtin_fma(p1, c1, -(p2 * s1));
                double s = __builtin_fma(p2, c1, p1 * s1);
                p1 = c;
                p2 = s;
                return c;
            }
            return p2;     // sin(k t), k = i / 2
        } else {           // IDENTITY has a single term
            return 0.0;
        }
    }
};

// IDENTITY: the single "term" is the value itself.
template <>
struct TermGen<MLMC_IDENTITY> {
    double v;
    __device__ __forceinline__ void init(double x_, double w, const BasisParams &) { v = x_ * w; }
    __device__ __forceinline__ double next(int) { return v; }
};

// SPLINE: phi_0 = 1, phi_r = B_r(t), r >= 1, of the clamped uniform cubic B-spline basis B_0..B_{nb-1} on
// [ref0, ref1] (nb = size, ns = nb - 3 knot spans).  At most four B-splines are non-zero at a point: init() finds the
// span and evaluates them with the Cox-de Boor recurrence (The NURBS Book, A2.2); next(i) selects.  Stateless in i.
template <>
struct TermGen<MLMC_SPLINE> {
    double n0, n1, n2, n3, w;
    int k;   // B_k .. B_{k+3} are the non-zero ones
    static __device__ __forceinline__ double knot(int j, int ns, double inv_ns) {
        int c = j - 3;
        c = c < 0 ? 0 : (c > ns ? ns : c);
        return (double)c * inv_ns;
    }
    __device__ __forceinline__ void init(double t, double w_, const BasisParams &bp) {
        w = w_;
        const int ns = bp.size - 3;
        const double inv_ns = 1.0 / (double)ns;
        const double u = (t - bp.ref0) / (bp.ref1 - bp.ref0);
        int s = (int)(u * (double)ns);
        s = s < 0 ? 0 : (s > ns - 1 ? ns - 1 : s);
        k = s;
        if (s >= 2 && s <= ns - 3) {
            // interior span: none of the knots t_{s+1} .. t_{s+6} that the recurrence touches is a repeated end knot, the four
            // B-splines are the uniform cubic pieces of the local coordinate v = u ns - s in [0, 1) -- no divisions
            const double v = u * (double)ns - (double)s;
            const double v2 = v * v, v3 = v2 * v, om = 1.0 - v;
            constexpr double sixth = 1.0 / 6.0;
            n0 = (om * om * om * sixth) * w;
            n1 = (__builtin_fma(3.0, v3, __builtin_fma(-6.0, v2, 4.0)) * sixth) * w;
            n2 = (__builtin_fma(-3.0, v3, __builtin_fma(3.0, v2, __builtin_fma(3.0, v, 1.0))) * sixth) * w;
            n3 = (v3 * sixth) * w;
            return;
        }
        const int mu = s + 3;
        double N[4], left[4], right[4];
        N[0] = 1.0;
#pragma unroll
        for (int j = 1; j <= 3; ++j) {
            left[j] = u - knot(mu + 1 - j, ns, inv_ns);
            right[j] = knot(mu + j, ns, inv_ns) - u;
            double saved = 0.0;
#pragma unroll
            for (int r = 0; r < j; ++r) {
                const double temp = N[r] / (right[r + 1] + left[j - r]);
                N[r] = saved + right[r + 1] * temp;
                saved = left[j - r] * temp;
            }
            N[j] = saved;
        }
        n0 = N[0] * w; n1 = N[1] * w; n2 = N[2] * w; n3 = N[3] * w;
    }
    __device__ __forceinline__ double next(int i) {
        if (i == 0) return w;
        const int j = i - k;
        return j == 0 ? n0 : (j == 1 ? n1 : (j == 2 ? n2 : (j == 3 ? n3 : 0.0)));
    }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace mlmc
