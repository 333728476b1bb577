// C-ABI entry points of libmlmc_hip.so (include/mlmc_hip.h): runtime, basis objects, accumulators.
#include <cmath>
#include <chrono>
#include <cstring>
#include <new>

#include "common.hpp"
#include <algorithm>
#include <map>

namespace mlmc {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(const std::string &msg) {
    g_err = msg;
    return 1;
}
Runtime &rt() {
    static Runtime r;
    return r;
}
std::recursive_mutex &api_mutex() {
    static std::recursive_mutex m;
    return m;
}

void bind_thread_to_device() {
    static thread_local int bound = -1;
    const Runtime &r = rt();
    if (r.ready && bound != r.device) {
        if (hipSetDevice(r.device) == hipSuccess) bound = r.device;
    } else if (!r.ready) {
        bound = -1;           // mlmc_shutdown / a later mlmc_init may bind another device
    }
}

int ensure(void **p, size_t *cap, size_t bytes) {
    if (bytes <= *cap && *p) return 0;
    if (*p) {
        // earlier launches on the stream may still use the old buffer
        MLMC_HIP_CHECK(wait_stream(rt().stream));
        MLMC_HIP_CHECK(hipFree(*p));
        *p = nullptr;
        *cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    MLMC_HIP_CHECK(hipMalloc(p, want));
    *cap = want;
    return 0;
}

static int timing_collect(mlmc_accum *a);
int timing_begin(mlmc_accum *a) {
    if (!(rt().flags & 1)) return 0;
    if (a->ev_used >= 8192) {   // bound the pool: fold the finished pairs into the totals
        MLMC_HIP_CHECK(hipEventSynchronize(a->ev[a->ev_used - 1]));
        if (int rc = timing_collect(a)) return rc;
    }
    if (a->ev_used + 2 > a->ev.size()) {
        hipEvent_t e0, e1;
        MLMC_HIP_CHECK(hipEventCreate(&e0));
        MLMC_HIP_CHECK(hipEventCreate(&e1));
        a->ev.push_back(e0);
        a->ev.push_back(e1);
    }
    MLMC_HIP_CHECK(hipEventRecord(a->ev[a->ev_used], rt().stream));
    return 0;
}
int timing_end(mlmc_accum *a) {
    if (!(rt().flags & 1)) return 0;
    MLMC_HIP_CHECK(hipEventRecord(a->ev[a->ev_used + 1], rt().stream));
    a->ev_used += 2;
    return 0;
}
static int timing_collect(mlmc_accum *a) {
    for (size_t i = 0; i + 1 < a->ev_used; i += 2) {
        float ms = 0.f;
        MLMC_HIP_CHECK(hipEventElapsedTime(&ms, a->ev[i], a->ev[i + 1]));
        a->ms_total += ms;
    }
    a->ev_used = 0;
    return 0;
}

// Raw-value keep interval of a log-domain basis (mlmc_hip.h, mlmc_basis_desc): the smallest / largest positive double whose
// t = (log(x) - shift) * scale + ref0 lies in [ref0, ref1], by bisection over the bit patterns of the positive doubles
// (ordered like the values) with the host C library's log.  Nothing kept: (inf, 0).
static void log_keep_interval(const BasisParams &p, double *x_lo, double *x_hi) {
    auto t_of = [&](uint64_t bits) {
        double x;
        std::memcpy(&x, &bits, sizeof x);
        volatile double v = std::log(x);          // volatile: two roundings like NumPy, whatever the host compiler contracts
        volatile double w = (v - p.shift) * p.scale;
        return (double)(w + p.ref0);
    };
    auto as_double = [](uint64_t bits) { double x; std::memcpy(&x, &bits, sizeof x); return x; };
    const uint64_t top = 0x7ff0000000000000ull;   // +inf; bit patterns 1 .. top - 1 are the positive finite doubles
    // first x with t >= ref0 (predicate false ... false true ... true)
    uint64_t lo = 1, hi = top;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (t_of(mid) >= p.ref0) hi = mid; else lo = mid + 1;
    }
    const uint64_t first = lo;
    // last x with t <= ref1 (true ... true false ... false)
    lo = 0; hi = top - 1;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (t_of(mid) <= p.ref1) lo = mid; else hi = mid - 1;
    }
    const uint64_t last = lo;
    if (first >= top || last == 0 || first > last) {
        *x_lo = HUGE_VAL;
        *x_hi = 0.0;
        return;
    }
    *x_lo = as_double(first);
    *x_hi = as_double(last);
}

// Product linearisation phi_i phi_j = sum_k c_ijk phi_k of the polynomial families, k < 2 R - 1 (mlmc_hip.h,
// mlmc_accum_aux_kernel_time).  Legendre: Adams' formula with A(n) = (2n - 1)!! / n!,
//   c_ijk = (2k + 1) / (2s + 1) * A(s - i) A(s - j) A(s - k) / A(s),  2s = i + j + k,  |i - j| <= k <= i + j,  i + j + k even,
// walked along k by ratio recurrences in extended precision (every factor is O(1), no factorial is formed) and rounded once;
// max / min of (i, j) enter, so the table is bit for bit symmetric.  Monomials: t^i t^j = t^(i + j).
bool product_table(int kind, int R, std::vector<double> &out) {
    if (kind != MLMC_LEGENDRE && kind != MLMC_MONOMIAL) return false;
    const int K = 2 * R - 1;
    const size_t RR = (size_t)R * R;
    out.assign((size_t)K * RR, 0.0);
    for (int i = 0; i < R; ++i)
        for (int j = 0; j < R; ++j) {
            const size_t ij = (size_t)i * R + j;
            if (kind == MLMC_MONOMIAL) { out[(size_t)(i + j) * RR + ij] = 1.0; continue; }
            typedef long double ld;
            const int hi = i > j ? i : j, lo = i > j ? j : i;
            const ld diff = (ld)(hi - lo);
            // first term k = |i - j| (s = hi):  (2 diff + 1) / (2 hi + 1) * A(diff) A(lo) / A(hi)
            ld c = (2 * diff + 1) / (2 * (ld)hi + 1);
            for (int t = 1; t <= lo; ++t) {
                const ld tt = (ld)t;
                c = c * ((2 * tt - 1) / tt * (diff + tt) / (2 * (diff + tt) - 1));
            }
            for (int t = 0; t <= lo; ++t) {          // term t: k = diff + 2 t, s = hi + t
                const int k = (hi - lo) + 2 * t;
                out[(size_t)k * RR + ij] = (double)c;
                if (t == lo) break;
                const ld s = (ld)hi + t, kk = (ld)k;
                const ld si = s - hi, sj = s - lo, sk = s - kk;
                ld ratio = ((2 * kk + 5) / (2 * s + 3)) / ((2 * kk + 1) / (2 * s + 1));
                ratio = ratio * ((2 * si + 1) / (si + 1)) * ((2 * sj + 1) / (sj + 1));
                ratio = ratio * (sk / (2 * sk - 1));          // A(sk - 1) / A(sk)
                ratio = ratio * ((s + 1) / (2 * s + 1));      // A(s) / A(s + 1)
                c = c * ratio;
            }
        }
    return true;
}

// (phi_i phi_j)^2 = sum_k c2_ijk phi_k, k < 4 R - 3: the level-0 second moments of the covariance from level sums of moments.
// Legendre: the coefficients are (2k + 1) / 2 * int P_k (P_i P_j)^2, a polynomial of degree <= 8 R - 8 integrated exactly by a
// Gauss-Legendre rule of 4 R points, nodes and weights by Newton's iteration, everything in extended precision and rounded
// once; odd k and k > 2 (i + j) are exact zeros; non-negative, rows sum to one.  Monomials: t^(2 (i + j)).
bool square_product_table(int kind, int R, std::vector<double> &out) {
    if (kind != MLMC_LEGENDRE && kind != MLMC_MONOMIAL) return false;
    const int K = 4 * R - 3;
    const size_t RR = (size_t)R * R;
    out.assign((size_t)K * RR, 0.0);
    if (kind == MLMC_MONOMIAL) {
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < R; ++j) out[(size_t)(2 * (i + j)) * RR + (size_t)i * R + j] = 1.0;
        return true;
    }
    typedef long double ld;
    const int Q = 4 * R;
    const ld pi = 3.14159265358979323846264338327950288L;
    std::vector<ld> x(Q), w(Q), P((size_t)K * Q);
    for (int q = 0; q < Q; ++q) {
        ld t = std::cos(pi * ((ld)q + 0.75L) / ((ld)Q + 0.5L)), dp = 1.0L;
        for (int it = 0; it < 100; ++it) {
            ld p0 = 1.0L, p1 = t;
            for (int k = 2; k <= Q; ++k) {
                const ld p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
                p0 = p1;
                p1 = p2;
            }
            dp = Q * (t * p1 - p0) / (t * t - 1.0L);
            const ld dt = p1 / dp;
            t -= dt;
            if (std::fabs(dt) < 1e-20L) break;
        }
        {   // derivative at the converged node for the weight
            ld p0 = 1.0L, p1 = t;
            for (int k = 2; k <= Q; ++k) {
                const ld p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
                p0 = p1;
                p1 = p2;
            }
            dp = Q * (t * p1 - p0) / (t * t - 1.0L);
        }
        x[q] = t;
        w[q] = 2.0L / ((1.0L - t * t) * dp * dp);
        ld p0 = 1.0L, p1 = t;
        P[(size_t)0 * Q + q] = 1.0L;
        if (K > 1) P[(size_t)1 * Q + q] = t;
        for (int k = 2; k < K; ++k) {
            const ld p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
            p0 = p1;
            p1 = p2;
            P[(size_t)k * Q + q] = p2;
        }
    }
    std::vector<ld> sq(Q);
    for (int i = 0; i < R; ++i)
        for (int j = i; j < R; ++j) {
            for (int q = 0; q < Q; ++q) {
                const ld u = P[(size_t)i * Q + q] * P[(size_t)j * Q + q];
                sq[q] = w[q] * u * u;
            }
            for (int k = 0; k <= 2 * (i + j); k += 2) {
                ld acc = 0.0L;
                const ld *Pk = &P[(size_t)k * Q];
                for (int q = 0; q < Q; ++q) acc += sq[q] * Pk[q];
                const double c = (double)(acc * (2 * k + 1) / 2);
                out[(size_t)k * RR + (size_t)i * R + j] = c;
                out[(size_t)k * RR + (size_t)j * R + i] = c;
            }
        }
    // (P_0 P_j)^2 = P_j^2: these rows are the product table's own (Adams' formula) -- in particular c2_00k = delta_k0 EXACTLY,
    // which the exact sample counts in the P_0 P_0 entries (and vars[0] == 0) rest on, whatever the last bit of the quadrature
    std::vector<double> t1;
    product_table(kind, R, t1);
    for (int j = 0; j < R; ++j)
        for (int k = 0; k < K; ++k) {
            const double c = k < 2 * R - 1 ? t1[(size_t)k * RR + (size_t)j * R + j] : 0.0;
            out[(size_t)k * RR + (size_t)j] = c;
            out[(size_t)k * RR + (size_t)j * R] = c;
        }
    return true;
}

static bool linearize_enabled() {
    const char *e = std::getenv("MLMC_HIP_LINEARIZE");
    return !(e && e[0] == '0');
}

static int lin_min_size() {      // covariances of more moments than this take the linearised mean (tuning aid)
    const char *e = std::getenv("MLMC_HIP_LINEARIZE_MIN");
    return e ? std::atoi(e) : 16;
}

static int need_runtime() {
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    return 0;
}

}  // namespace mlmc

using namespace mlmc;

extern "C" {

int mlmc_abi_version(void) { return MLMC_ABI_VERSION; }
const char *mlmc_last_error(void) { return g_err.c_str(); }

int mlmc_init(int device, int flags) {
    MLMC_API_GUARD;
    Runtime &r = rt();
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail("no HIP device available: libmlmc_hip has no CPU fallback");
    if (device < 0 || device >= count) return fail("mlmc_init: device index out of range");
    if (r.ready && r.device == device) {
        r.flags = flags;
        return 0;
    }
    if (r.ready) return fail("mlmc_init: this process is already bound to another device (one process per GPU)");
    MLMC_HIP_CHECK(hipSetDevice(device));
    MLMC_HIP_CHECK(hipGetDeviceProperties(&r.prop, device));
    if (std::strncmp(r.prop.gcnArchName, "gfx950", 6) != 0)
        return fail(std::string("mlmc_init: kernels are built for gfx950 only, device is ") + r.prop.gcnArchName);
    MLMC_HIP_CHECK(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    r.own_stream = true;
    r.device = device;
    r.flags = flags;
    r.n_cu = r.prop.multiProcessorCount;
    r.ready = true;
    return 0;
}

}  // extern "C"

namespace mlmc {
hipError_t wait_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(4)) break;
    }
    return hipStreamSynchronize(st);   // long waits (covariance passes, uploads) sleep as usual
}
}  // namespace mlmc

extern "C" {

int mlmc_set_stream(void *stream) {
    MLMC_API_GUARD;
    Runtime &r = rt();
    if (!r.ready) return fail("mlmc_init has not been called (no HIP device bound)");
    MLMC_HIP_CHECK(wait_stream(r.stream));
    if (r.own_stream) (void)hipStreamDestroy(r.stream);
    r.stream = (hipStream_t)stream;      // NULL = the legacy default stream
    r.own_stream = false;
    return 0;
}

int mlmc_wait_event(void *hip_event) {
    MLMC_API_GUARD;
    Runtime &r = rt();
    if (!r.ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!hip_event) return fail("mlmc_wait_event: null event");
    MLMC_HIP_CHECK(hipStreamWaitEvent(r.stream, (hipEvent_t)hip_event, 0));
    return 0;
}

int mlmc_synchronize(void) {
    MLMC_API_GUARD;
    Runtime &r = rt();
    if (!r.ready) return fail("mlmc_init has not been called (no HIP device bound)");
    MLMC_HIP_CHECK(wait_stream(r.stream));
    return 0;
}

void mlmc_shutdown(void) {
    MLMC_API_GUARD;
    Runtime &r = rt();
    if (!r.ready) return;
    (void)wait_stream(r.stream);
    if (r.own_stream) (void)hipStreamDestroy(r.stream);
    r.stream = nullptr;
    r.ready = false;
    r.device = -1;
}

int mlmc_device_info(char *name, int name_len, int *n_cu, int *wave_size, int64_t *hbm_bytes) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (name && name_len > 0) {
        std::strncpy(name, rt().prop.name, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (n_cu) *n_cu = rt().prop.multiProcessorCount;
    if (wave_size) *wave_size = rt().prop.warpSize;
    if (hbm_bytes) *hbm_bytes = (int64_t)rt().prop.totalGlobalMem;
    return 0;
}

// ---- basis ---------------------------------------------------------------------------------
int mlmc_basis_create(const mlmc_basis_desc *d, mlmc_basis **out) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!d || !out) return fail("mlmc_basis_create: null argument");
    if (d->size <= 0) return fail("mlmc_basis_create: size must be > 0");   // moments.py:11 assert size > 0
    if (d->kind < MLMC_LEGENDRE || d->kind > MLMC_SPLINE) return fail("mlmc_basis_create: unknown kind");
    if (d->kind == MLMC_SPLINE && d->size < 4) return fail("mlmc_basis_create: cubic spline moments need size >= 4");
    if (d->kind == MLMC_IDENTITY && d->size != 1) return fail("mlmc_basis_create: IDENTITY has size 1");
    if (d->out_size < 0 || (d->out_size > 0 && !d->matrix)) return fail("mlmc_basis_create: matrix missing");
    if (d->kind == MLMC_LEGENDRE && d->size > 512) return fail("mlmc_basis_create: at most 512 Legendre moments");
    mlmc_basis *b = new (std::nothrow) mlmc_basis();
    if (!b) return fail("out of memory");
    b->p.kind = d->kind;
    b->p.size = d->size;
    b->p.shift = d->shift;
    b->p.scale = d->scale;
    b->p.ref0 = d->ref0;
    b->p.ref1 = d->ref1;
    b->p.is_log = d->is_log;
    b->p.is_clip = d->is_clip;
    b->p.x_lo = d->x_lo;
    b->p.x_hi = d->x_hi;
    if (d->is_log && d->is_clip && d->x_lo == 0.0 && d->x_hi == 0.0) log_keep_interval(b->p, &b->p.x_lo, &b->p.x_hi);
    b->out_size = d->out_size;
    const int R = d->size;
    b->scale_c.assign(R, 1.0);
    if (d->kind == MLMC_LEGENDRE) {
        // P_i = c_i q_i with q_i = 2^i x (monic Legendre polynomial) (device_basis.hpp): c_i = c_{i-1} (2i-1) / (2i), c_0 = 1
        long double c = 1.0L;
        for (int i = 0; i < R; ++i) {
            if (i >= 1) c = c * (long double)(2 * i - 1) / (long double)(2 * i);
            b->scale_c[i] = (double)c;
        }
    }
    hipError_t e = hipMalloc(&b->d_scale, sizeof(double) * R);
    if (e == hipSuccess) e = hipMemcpy(b->d_scale, b->scale_c.data(), sizeof(double) * R, hipMemcpyHostToDevice);
    if (e == hipSuccess && d->out_size > 0) {
        b->matrix.assign(d->matrix, d->matrix + (size_t)d->out_size * R);
        e = hipMalloc(&b->d_matrix, sizeof(double) * b->matrix.size());
        if (e == hipSuccess)
            e = hipMemcpy(b->d_matrix, b->matrix.data(), sizeof(double) * b->matrix.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        mlmc_basis_destroy(b);
        return fail(std::string("mlmc_basis_create: ") + hipGetErrorString(e));
    }
    *out = b;
    return 0;
}

void mlmc_basis_destroy(mlmc_basis *b) {
    MLMC_API_GUARD;
    if (!b) return;
    if (b->d_scale) (void)hipFree(b->d_scale);
    if (b->d_matrix) (void)hipFree(b->d_matrix);
    delete b;
}

int mlmc_basis_eval(const mlmc_basis *b, const double *x, int64_t n, int32_t size, double *out, int mem_kind) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!b || (n > 0 && (!x || !out))) return fail("mlmc_basis_eval: null argument");
    const int max_size = b->out_size > 0 ? b->out_size : b->p.size;
    if (size <= 0 || size > max_size) return fail("mlmc_basis_eval: size out of range");
    if (n == 0) return 0;
    hipStream_t st = rt().stream;
    if (mem_kind == MLMC_DEVICE) {
        if (int rc = launch_eval(b, x, n, size, out)) return rc;
        MLMC_HIP_CHECK(wait_stream(st));
        return 0;
    }
    double *d_x = nullptr, *d_o = nullptr;
    hipError_t e0 = hipMalloc(&d_x, sizeof(double) * (size_t)n);
    if (e0 == hipSuccess) e0 = hipMalloc(&d_o, sizeof(double) * (size_t)n * size);
    if (e0 == hipSuccess) e0 = hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st);
    int rc = e0 == hipSuccess ? launch_eval(b, d_x, n, size, d_o) : fail(std::string("mlmc_basis_eval: ") + hipGetErrorString(e0));
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, d_o, sizeof(double) * (size_t)n * size, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = wait_stream(st);
        if (e != hipSuccess) rc = fail(std::string("mlmc_basis_eval copy back: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_x);
    (void)hipFree(d_o);
    return rc;
}

// The coefficient tables of the linearisations live on the device once per (family, size) for the life of the process (4 MB +
// 8 MB at R = 64): accumulators share them.
struct LinTables {
    double *d_prod = nullptr, *d_prod2 = nullptr;
};
static int lin_tables(int kind, int R, bool squares, LinTables **out) {
    static std::map<std::pair<int, int>, LinTables> cache;
    LinTables &t = cache[std::make_pair(kind, R)];
    if (!t.d_prod) {
        std::vector<double> h;
        product_table(kind, R, h);
        MLMC_HIP_CHECK(hipMalloc(&t.d_prod, sizeof(double) * h.size()));
        MLMC_HIP_CHECK(hipMemcpy(t.d_prod, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    }
    if (squares && !t.d_prod2) {
        std::vector<double> h;                  // ~0.2 s of host time at R = 64
        square_product_table(kind, R, h);
        MLMC_HIP_CHECK(hipMalloc(&t.d_prod2, sizeof(double) * h.size()));
        MLMC_HIP_CHECK(hipMemcpy(t.d_prod2, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
    }
    *out = &t;
    return 0;
}

// Inner accumulators of a covariance accumulator that takes (part of) its sums from level sums of moments; called with the
// first chunk that is large enough.
static int ensure_lin(mlmc_accum *a) {
    if (a->lin) return 0;
    const mlmc_basis *b = a->basis;
    LinTables *t = nullptr;
    if (int rc = lin_tables(b->p.kind, a->R, a->lin0_eligible, &t)) return rc;
    mlmc_basis_desc d;
    std::memset(&d, 0, sizeof(d));
    d.kind = b->p.kind;
    d.size = 2 * a->R - 1;
    d.shift = b->p.shift; d.scale = b->p.scale; d.ref0 = b->p.ref0; d.ref1 = b->p.ref1;
    d.is_log = b->p.is_log; d.is_clip = b->p.is_clip;
    d.x_lo = b->p.x_lo; d.x_hi = b->p.x_hi;              // the very thresholds of the caller's basis: the same samples are kept
    int rc = mlmc_basis_create(&d, &a->lin_basis);
    if (!rc) rc = mlmc_accum_create(a->lin_basis, a->n_levels, MLMC_MODE_MOMENTS | MLMC_MODE_MEAN_ONLY, a->n_comp, &a->lin);
    if (!rc) {
        a->lin_basis->p.x_lo = b->p.x_lo;                // (a desc with x_lo == x_hi == 0 would have been bisected anew)
        a->lin_basis->p.x_hi = b->p.x_hi;
        a->lin->host_outputs = false;                    // its totals are read on the device (launch_cov_finalize)
        a->lin_K = d.size;
        a->d_lin_prod = t->d_prod;
    }
    if (!rc && a->lin0_eligible) {
        d.size = 4 * a->R - 3;
        rc = mlmc_basis_create(&d, &a->lin0_basis);
        if (!rc) rc = mlmc_accum_create(a->lin0_basis, a->n_levels, MLMC_MODE_MOMENTS | MLMC_MODE_MEAN_ONLY, a->n_comp, &a->lin0);
        if (!rc) {
            a->lin0_basis->p.x_lo = b->p.x_lo;
            a->lin0_basis->p.x_hi = b->p.x_hi;
            a->lin0->host_outputs = false;
            a->lin0_K = d.size;
            a->d_lin0_prod = t->d_prod2;
        }
    }
    if (rc) {            // no half-built state: the accumulator goes on with all three Gram matrices
        if (a->lin) mlmc_accum_destroy(a->lin);
        if (a->lin_basis) mlmc_basis_destroy(a->lin_basis);
        if (a->lin0) mlmc_accum_destroy(a->lin0);
        if (a->lin0_basis) mlmc_basis_destroy(a->lin0_basis);
        a->lin = a->lin0 = nullptr;
        a->lin_basis = a->lin0_basis = nullptr;
        a->lin_eligible = a->lin0_eligible = false;
    }
    return rc;
}

// ---- accumulators ---------------------------------------------------------------------------
int mlmc_accum_create(const mlmc_basis *b, int32_t n_levels, int32_t mode, int32_t n_comp, mlmc_accum **out) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!b || !out) return fail("mlmc_accum_create: null argument");
    if (n_levels <= 0 || n_comp <= 0) return fail("mlmc_accum_create: n_levels and n_comp must be > 0");
    const bool mean_only = (mode & MLMC_MODE_MEAN_ONLY) != 0;
    mode &= ~MLMC_MODE_MEAN_ONLY;
    if (mode != MLMC_MODE_MOMENTS && mode != MLMC_MODE_COV) return fail("mlmc_accum_create: unknown mode");
    mlmc_accum *a = new (std::nothrow) mlmc_accum();
    if (!a) return fail("out of memory");
    a->basis = b;
    a->n_levels = n_levels;
    a->mode = mode;
    // only the passes that exist solely for the second moments are skipped: the three-Gram covariance pass becomes one
    // Gram matrix, the diff-Gram pass of TransformedMoments disappears; plain moments keep their (free) sum of squares
    // covariance from materialised values: TransformedMoments, and plain bases beyond the 128 moments of the in-register kernels
    const bool spline_band = mode == MLMC_MODE_COV && mean_only && b->out_size == 0 && b->p.kind == MLMC_SPLINE && b->p.size <= SPLINE_BAND_MAX_R;
    a->cov_from_values = mode == MLMC_MODE_COV && (b->out_size > 0 || (b->p.size > 128 && !spline_band));
    a->mean_only = mean_only && ((mode == MLMC_MODE_COV && !a->cov_from_values) || (mode == MLMC_MODE_MOMENTS && b->out_size > 0));
    // plain polynomial moments with 64 < R <= 128: the mean-only form of the term-split kernel covers them in ONE pass
    // (moments.hip, k_moments_accum_split<..., SQ = false>); every other plain size keeps its free sums of squares
    a->mean_only_plain = mean_only && mode == MLMC_MODE_MOMENTS && b->out_size == 0 &&
                         (b->p.kind == MLMC_LEGENDRE || b->p.kind == MLMC_MONOMIAL) && b->p.size > 64 && b->p.size <= 256;
    a->n_comp = n_comp;
    a->R = b->p.size;
    a->Rout = b->out_size > 0 ? b->out_size : b->p.size;
    a->RP = (((mode == MLMC_MODE_COV && b->out_size > 0 ? a->Rout : a->R) + 15) / 16) * 16;   // (plain bases: Rout == R)
    if (mode == MLMC_MODE_MOMENTS) {
        a->K = (int64_t)n_comp * a->Rout;
        a->int_width = 2 * (int64_t)a->R + (b->out_size > 0 ? (int64_t)a->RP * a->RP : 0);
    } else {
        a->K = (int64_t)n_comp * a->Rout * a->Rout;
        a->int_width = 3 * (int64_t)a->RP * a->RP;
    }
    const size_t tot = (size_t)n_levels * n_comp * a->int_width;
    a->state_bytes = sizeof(double) * tot + sizeof(int64_t) * 2 * n_levels;
    a->out_bytes = (sizeof(int64_t) + sizeof(double)) * 2 * n_levels + 2 * sizeof(double) * (size_t)n_levels * a->K;
    if (hipMalloc(&a->d_state, a->state_bytes) != hipSuccess || hipMalloc(&a->d_out, a->out_bytes) != hipSuccess ||
        hipHostMalloc(&a->h_out, a->out_bytes, hipHostMallocDefault) != hipSuccess) {
        mlmc_accum_destroy(a);
        return fail("mlmc_accum_create: device / pinned allocation failed");
    }
    a->d_totals = (double *)a->d_state;
    a->d_counts = (int64_t *)(a->d_totals + tot);
    a->level_flushed.assign(n_levels, 0);
    if (mode == MLMC_MODE_MOMENTS && n_comp == 1 && b->out_size == 0) {
        void *dev_view = nullptr;
        if (hipHostGetDevicePointer(&dev_view, a->h_out, 0) == hipSuccess && dev_view) {
            a->host_outputs = true;
            a->h_out_n = (int64_t *)dev_view;
            a->h_out_s = (double *)(a->h_out_n + 2 * (size_t)n_levels) + 2 * (size_t)n_levels;
            a->h_out_sp = a->h_out_s + (size_t)n_levels * a->K;
        }
    }
    a->d_out_n = (int64_t *)a->d_out;
    a->d_out_nd = (double *)(a->d_out_n + 2 * (size_t)n_levels);
    a->d_out_s = a->d_out_nd + 2 * (size_t)n_levels;
    a->d_out_sp = a->d_out_s + (size_t)n_levels * a->K;
    // covariance WITH variances of 17..128 plain polynomial moments: mean through the product linearisation (mlmc_hip.h) -- the
    // inner accumulators and the coefficient tables come into being with the first chunk large enough to use them (ensure_lin):
    // an accumulator that only ever sees small chunks costs what it cost before
    if (mode == MLMC_MODE_COV && !mean_only && !a->cov_from_values && b->out_size == 0 && b->p.size > lin_min_size() && b->p.size <= 128 &&
        (b->p.kind == MLMC_LEGENDRE || b->p.kind == MLMC_MONOMIAL) && b->p.is_clip &&
        std::fabs(b->p.ref0) <= 1.0 && std::fabs(b->p.ref1) <= 1.0 &&          // (monomials on a wider reference domain: t^(4 R - 4) may overflow)
        linearize_enabled()) {
        // (is_clip: without clipping to the domain a value outside it makes the HIGH extended terms overflow while the low products it
        // is not needed for stay finite, and 0 * inf in the contraction would turn those into NaN; the direct form touches, per
        // entry, only its own factors)
        a->lin_eligible = true;
        // break-even of the auxiliary pass (three more launches per estimate, ~30 us) against the matrix time it saves
        // (~80 ps per sample at 33..64 moments, ~9 ps at 17..32): measured with tools/kbench.py --n
        const char *min_n = std::getenv("MLMC_HIP_LINEARIZE_MIN_N");
        a->lin_min_n = min_n ? std::atoll(min_n) : (a->R > 32 ? 100000 : 1500000);
        // level 0 without the matrix cores (<= 64 moments: the 4 R - 3 extended terms fit the two windows of the mean-only kernel)
        const char *lin0_env = std::getenv("MLMC_HIP_LINEARIZE_LEVEL0");
        a->lin0_eligible = a->R <= 64 && !(lin0_env && lin0_env[0] == '0');
    }
    *out = a;
    return mlmc_accum_reset(a);
}

int mlmc_accum_reset(mlmc_accum *a) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!a) return fail("mlmc_accum_reset: null argument");
    hipStream_t st = rt().stream;
    a->pending.clear();
    std::fill(a->level_flushed.begin(), a->level_flushed.end(), 0);
    MLMC_HIP_CHECK(hipMemsetAsync(a->d_state, 0, a->state_bytes, st));
    a->lin_used = a->lin0_used = false;
    if (a->lin0)
        if (int rc = mlmc_accum_reset(a->lin0)) return rc;
    if (a->lin) return mlmc_accum_reset(a->lin);
    return 0;
}

void mlmc_accum_destroy(mlmc_accum *a) {
    MLMC_API_GUARD;
    if (!a) return;
    if (rt().ready) (void)wait_stream(rt().stream);
    if (a->lin) mlmc_accum_destroy(a->lin);
    if (a->lin_basis) mlmc_basis_destroy(a->lin_basis);
    if (a->lin0) mlmc_accum_destroy(a->lin0);
    if (a->lin0_basis) mlmc_basis_destroy(a->lin0_basis);         // (the coefficient tables are shared: lin_tables)
    void *ptrs[] = {a->d_state, a->d_partials, a->d_pcounts, a->d_stage_f, a->d_stage_c, a->d_mask, a->d_out, a->d_vals_f, a->d_vals_c, a->d_vals_tmp};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (a->h_out) (void)hipHostFree(a->h_out);
    for (hipEvent_t e : a->ev) (void)hipEventDestroy(e);
    delete a;
}

int mlmc_accum_push(mlmc_accum *a, int32_t level, const double *fine, const double *coarse, int64_t n, int mem_kind) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!a) return fail("mlmc_accum_push: null argument");
    if (level < 0 || level >= a->n_levels) return fail("mlmc_accum_push: level out of range");
    if (n < 0) return fail("mlmc_accum_push: negative n");
    if (n == 0) return 0;
    if (!fine) return fail("mlmc_accum_push: fine is NULL");
    if (n > ((int64_t)1 << 40)) return fail("mlmc_accum_push: chunk too large, split it");
    hipStream_t st = rt().stream;
    const size_t bytes = sizeof(double) * (size_t)n * a->n_comp;
    const double *d_f = fine, *d_c = coarse;
    if (mem_kind == MLMC_HOST) {
        // staging buffers are reused across pushes: the previous push's kernels must have consumed them
        if (bytes > a->stage_cap) {
            MLMC_HIP_CHECK(wait_stream(st));
            if (a->d_stage_f) (void)hipFree(a->d_stage_f);
            if (a->d_stage_c) (void)hipFree(a->d_stage_c);
            a->d_stage_f = a->d_stage_c = nullptr;
            a->stage_cap = 0;
            MLMC_HIP_CHECK(hipMalloc(&a->d_stage_f, bytes));
            MLMC_HIP_CHECK(hipMalloc(&a->d_stage_c, bytes));
            a->stage_cap = bytes;
        } else {
            MLMC_HIP_CHECK(wait_stream(st));
        }
        MLMC_HIP_CHECK(hipMemcpyAsync(a->d_stage_f, fine, bytes, hipMemcpyHostToDevice, st));
        d_f = a->d_stage_f;
        if (coarse) {
            MLMC_HIP_CHECK(hipMemcpyAsync(a->d_stage_c, coarse, bytes, hipMemcpyHostToDevice, st));
            d_c = a->d_stage_c;
        }
        // the caller may reuse its buffers as soon as push returns: pageable memory has been staged by the runtime at
        // this point, pinned memory has not -- wait for the copies (the kernels below still run asynchronously)
        MLMC_HIP_CHECK(wait_stream(st));
    } else if (mem_kind != MLMC_DEVICE) {
        return fail("mlmc_accum_push: bad mem_kind");
    }
    const uint8_t *d_mask = nullptr;
    bool count_in_kernel = true;
    if (a->n_comp > 1) {
        if (int rc = ensure((void **)&a->d_mask, &a->mask_cap, (size_t)n)) return rc;
        if (int rc = launch_mask(a, d_f, d_c, n, a->d_mask, a->d_counts + 2 * (int64_t)level)) return rc;
        d_mask = a->d_mask;
        count_in_kernel = false;
    }
    for (int m = 0; m < a->n_comp; ++m) {
        const double *f_m = d_f + (int64_t)m * n;
        const double *c_m = d_c ? d_c + (int64_t)m * n : nullptr;
        const bool count = count_in_kernel && m == 0;
        int rc;
        if (a->mode == MLMC_MODE_MOMENTS) {
            // chunks that stay valid until finalize (device memory, no staging / mask scratch) are gathered into one launch;
            // the components of a vector quantity share the chunk's mask scratch: they are gathered too, and launched
            // together when the push ends (below)
            const bool defer = mem_kind == MLMC_DEVICE || a->n_comp > 1;
            rc = launch_moments_accum(a, level, m, f_m, c_m, d_mask, n, count, defer);
            if (!rc && a->basis->out_size > 0 && !a->mean_only) {
                if (a->R <= 128) {
                    rc = launch_cov_accum(a, level, m, f_m, c_m, d_mask, n, false, 1);
                } else {
                    // difference Gram of more than 128 underlying moments: from materialised (scaled) values, chunk by chunk
                    int64_t chunk = 1 << 22;
                    while (chunk > 4096 && chunk * a->R > ((int64_t)1 << 28)) chunk >>= 1;
                    const size_t need = sizeof(double) * (size_t)(n < chunk ? n : chunk) * a->R;
                    if (need > a->vals_cap) {
                        MLMC_HIP_CHECK(wait_stream(st));
                        if (a->d_vals_f) (void)hipFree(a->d_vals_f);
                        if (a->d_vals_c) (void)hipFree(a->d_vals_c);
                        a->d_vals_f = a->d_vals_c = nullptr;
                        a->vals_cap = 0;
                        MLMC_HIP_CHECK(hipMalloc(&a->d_vals_f, need));
                        MLMC_HIP_CHECK(hipMalloc(&a->d_vals_c, need));
                        a->vals_cap = need;
                    }
                    for (int64_t off = 0; off < n && !rc; off += chunk) {
                        const int64_t m_n = (n - off < chunk) ? n - off : chunk;
                        rc = launch_eval_scaled_base(a->basis, f_m + off, m_n, a->d_vals_f);
                        if (!rc && c_m) rc = launch_eval_scaled_base(a->basis, c_m + off, m_n, a->d_vals_c);
                        if (!rc) rc = launch_cov_from_values(a, level, m, a->d_vals_f, c_m ? a->d_vals_c : nullptr,
                                                             d_mask ? d_mask + off : nullptr, m_n, false, 1);
                    }
                }
            }
        } else if (a->cov_from_values) {
            // covariance of TransformedMoments / of more than 128 moments: materialise the moment values chunk by chunk
            // (eval [+ matrix product]), then the MFMA covariance kernel reads them back, one 64 x 64 output block per launch
            const int R1 = a->Rout;
            // <= 2 GB of values per side (of 288 GB): a chunk of 2^18 samples left every launch four batches per workgroup, and the
            // 64 x 64-block launches of a 140-moment covariance spent their time on ramps and reductions (17 TFLOP/s)
            int64_t chunk = 1 << 22;
            while (chunk > 4096 && chunk * R1 > ((int64_t)1 << 28)) chunk >>= 1;
            const size_t need = sizeof(double) * (size_t)(n < chunk ? n : chunk) * R1;
            if (need > a->vals_cap) {
                MLMC_HIP_CHECK(wait_stream(st));
                if (a->d_vals_f) (void)hipFree(a->d_vals_f);
                if (a->d_vals_c) (void)hipFree(a->d_vals_c);
                a->d_vals_f = a->d_vals_c = nullptr;
                a->vals_cap = 0;
                MLMC_HIP_CHECK(hipMalloc(&a->d_vals_f, need));
                MLMC_HIP_CHECK(hipMalloc(&a->d_vals_c, need));
                a->vals_cap = need;
            }
            if (a->basis->out_size > 0) {      // workspace for the underlying values of one side (no allocation / sync per chunk)
                const size_t need_tmp = sizeof(double) * (size_t)(n < chunk ? n : chunk) * a->R;
                if (need_tmp > a->vals_tmp_cap) {
                    MLMC_HIP_CHECK(wait_stream(st));
                    if (a->d_vals_tmp) (void)hipFree(a->d_vals_tmp);
                    a->d_vals_tmp = nullptr;
                    a->vals_tmp_cap = 0;
                    MLMC_HIP_CHECK(hipMalloc(&a->d_vals_tmp, need_tmp));
                    a->vals_tmp_cap = need_tmp;
                }
            }
            rc = 0;
            for (int64_t off = 0; off < n && !rc; off += chunk) {
                const int64_t m_n = (n - off < chunk) ? n - off : chunk;
                rc = launch_eval(a->basis, f_m + off, m_n, R1, a->d_vals_f, a->d_vals_tmp);
                if (!rc && c_m) rc = launch_eval(a->basis, c_m + off, m_n, R1, a->d_vals_c, a->d_vals_tmp);
                if (!rc) rc = launch_cov_from_values(a, level, m, a->d_vals_f, c_m ? a->d_vals_c : nullptr,
                                                     d_mask ? d_mask + off : nullptr, m_n, count);
            }
        } else {
            // all components of a vector quantity in one launch (grid.y = component; they share the mask)
            // level 0 (one value per sample) of <= 64 moments: mean AND second moments from the level sums of 4 R - 3 moments --
            // no matrix pass; the moments kernel does the counting
            if (a->lin_eligible && !a->lin && n >= a->lin_min_n)
                if (int rcl = ensure_lin(a)) return rcl;
            const bool use_lin0 = a->lin0 && !d_c && n >= a->lin_min_n;
            if (use_lin0) {
                a->lin0_used = true;
                rc = launch_moments_accum(a->lin0, level, m, f_m, nullptr, d_mask, n, count_in_kernel && m == 0,
                                          mem_kind == MLMC_DEVICE || a->n_comp > 1);
                if (rc) return rc;
                continue;
            }
            const bool use_lin = a->lin && n >= a->lin_min_n;
            rc = m == 0 ? launch_cov_accum(a, level, 0, d_f, d_c, d_mask, n, count_in_kernel, a->mean_only ? 2 : (use_lin ? 3 : 0), a->n_comp) : 0;
            // the extended moments of the linearised mean over the same chunk (same mask; the covariance kernel counts);
            // device chunks wait for finalize and go out as ONE launch over all levels; staged host chunks and the components of a
            // vector quantity (shared mask scratch) are launched when the push ends
            if (!rc && use_lin) a->lin_used = true;
            if (!rc && use_lin)
                rc = launch_moments_accum(a->lin, level, m, f_m, c_m, d_mask, n, false, mem_kind == MLMC_DEVICE || a->n_comp > 1);
        }
        if (rc) return rc;
    }
    // staged host data and the mask scratch are reused by the next push: their segments must be launched now
    if (a->mode == MLMC_MODE_MOMENTS && (mem_kind == MLMC_HOST || a->n_comp > 1))
        if (int rc = flush_moments(a)) return rc;
    if (a->lin && (mem_kind == MLMC_HOST || a->n_comp > 1))
        if (int rc = flush_moments(a->lin)) return rc;
    if (a->lin0 && (mem_kind == MLMC_HOST || a->n_comp > 1))
        if (int rc = flush_moments(a->lin0)) return rc;
    return 0;
}

int mlmc_accum_finalize(mlmc_accum *a, int64_t *n, int64_t *n_rm, double *s, double *sp, int mem_kind) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!a || !n || !n_rm || !s || !sp) return fail("mlmc_accum_finalize: null argument");
    hipStream_t st = rt().stream;
    if (a->mode == MLMC_MODE_MOMENTS)
        if (int rcf = flush_moments(a)) return rcf;
    const int L = a->n_levels;
    const size_t nk = (size_t)L * a->K;
    if (a->host_outputs && mem_kind == MLMC_HOST) {
        // the grid reductions already wrote the finished rows of every pushed level into the pinned mirror
        MLMC_HIP_CHECK(wait_stream(st));
        const int64_t *hn = (const int64_t *)a->h_out;
        const double *hs = (const double *)(hn + 2 * (size_t)L) + 2 * (size_t)L;
        const size_t K = (size_t)a->K;
        for (int l = 0; l < L; ++l) {
            if (a->level_flushed[l]) {
                n[l] = hn[l];
                n_rm[l] = hn[L + l];
                std::memcpy(s + l * K, hs + l * K, sizeof(double) * K);
                std::memcpy(sp + l * K, hs + nk + l * K, sizeof(double) * K);
            } else {
                n[l] = n_rm[l] = 0;
                std::fill(s + l * K, s + (l + 1) * K, 0.0);
                std::fill(sp + l * K, sp + (l + 1) * K, 0.0);
            }
        }
        return 0;
    }
    int rc = (a->mode == MLMC_MODE_MOMENTS) ? launch_moments_finalize(a) : launch_cov_finalize(a);
    if (rc) return rc;
    if (mem_kind == MLMC_DEVICE) {
        MLMC_HIP_CHECK(hipMemcpyAsync(n, a->d_out_n, sizeof(int64_t) * L, hipMemcpyDeviceToDevice, st));
        MLMC_HIP_CHECK(hipMemcpyAsync(n_rm, a->d_out_n + L, sizeof(int64_t) * L, hipMemcpyDeviceToDevice, st));
        MLMC_HIP_CHECK(hipMemcpyAsync(s, a->d_out_s, sizeof(double) * nk, hipMemcpyDeviceToDevice, st));
        MLMC_HIP_CHECK(hipMemcpyAsync(sp, a->d_out_sp, sizeof(double) * nk, hipMemcpyDeviceToDevice, st));
        MLMC_HIP_CHECK(wait_stream(st));
    } else {
        // one packed copy into the pinned mirror, then plain host copies into the caller's arrays
        MLMC_HIP_CHECK(hipMemcpyAsync(a->h_out, a->d_out, a->out_bytes, hipMemcpyDeviceToHost, st));
        MLMC_HIP_CHECK(wait_stream(st));
        const int64_t *hn = (const int64_t *)a->h_out;
        const double *hs = (const double *)(hn + 2 * (size_t)L) + 2 * (size_t)L;
        std::memcpy(n, hn, sizeof(int64_t) * L);
        std::memcpy(n_rm, hn + L, sizeof(int64_t) * L);
        std::memcpy(s, hs, sizeof(double) * nk);
        std::memcpy(sp, hs + nk, sizeof(double) * nk);
    }
    return 0;
}

int mlmc_accum_estimate(mlmc_accum *a, int32_t n_chunks, const int32_t *levels, const double *const *fine,
                        const double *const *coarse, const int64_t *n_samples, int mem_kind, int64_t *n, int64_t *n_rm,
                        double *s, double *sp) {
    MLMC_API_GUARD;
    if (!a || n_chunks < 0 || (n_chunks > 0 && (!levels || !fine || !n_samples))) return fail("mlmc_accum_estimate: null argument");
    if (int rc = mlmc_accum_reset(a)) return rc;
    for (int k = 0; k < n_chunks; ++k)
        if (int rc = mlmc_accum_push(a, levels[k], fine[k], coarse ? coarse[k] : nullptr, n_samples[k], mem_kind)) return rc;
    return mlmc_accum_finalize(a, n, n_rm, s, sp, MLMC_HOST);
}

int mlmc_accum_finalize_packed(mlmc_accum *a, double *packed, int mem_kind) {
    MLMC_API_GUARD;
    if (need_runtime()) return 1;
    if (!a || !packed) return fail("mlmc_accum_finalize_packed: null argument");
    hipStream_t st = rt().stream;
    if (a->mode == MLMC_MODE_MOMENTS && a->host_outputs && mem_kind == MLMC_DEVICE && a->R <= MAX_TERMS_PER_PASS &&
        (int)a->pending.size() == a->n_levels &&
        std::none_of(a->level_flushed.begin(), a->level_flushed.end(), [](char f) { return f != 0; })) {
        // the whole estimate is pending as one launch: its grid reduction writes the packed buffer directly
        // (every level is covered, so no finalize kernel and no device-to-device copy are needed)
        a->packed_target = packed;
        const int rcf = flush_moments(a);
        a->packed_target = nullptr;
        return rcf;
    }
    if (a->mode == MLMC_MODE_MOMENTS)
        if (int rcf = flush_moments(a)) return rcf;
    int rc = (a->mode == MLMC_MODE_MOMENTS) ? launch_moments_finalize(a) : launch_cov_finalize(a);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * (2 * (size_t)a->n_levels + 2 * (size_t)a->n_levels * a->K);
    MLMC_HIP_CHECK(hipMemcpyAsync(packed, a->d_out_nd, bytes, mem_kind == MLMC_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    if (mem_kind == MLMC_DEVICE) return 0;   // stream-ordered: the caller's collective on the same stream needs no host sync
    MLMC_HIP_CHECK(wait_stream(st));
    return 0;
}

int mlmc_accum_estimate_packed(mlmc_accum *a, int32_t n_chunks, const int32_t *levels, const double *const *fine,
                               const double *const *coarse, const int64_t *n_samples, int mem_kind, double *packed,
                               int packed_kind) {
    MLMC_API_GUARD;
    if (!a || n_chunks < 0 || (n_chunks > 0 && (!levels || !fine || !n_samples))) return fail("mlmc_accum_estimate_packed: null argument");
    if (int rc = mlmc_accum_reset(a)) return rc;
    for (int k = 0; k < n_chunks; ++k)
        if (int rc = mlmc_accum_push(a, levels[k], fine[k], coarse ? coarse[k] : nullptr, n_samples[k], mem_kind)) return rc;
    return mlmc_accum_finalize_packed(a, packed, packed_kind);
}

int mlmc_accum_kernel_time(mlmc_accum *a, double *ms, int64_t *launches, int64_t *alg_bytes) {
    MLMC_API_GUARD;
    if (!a) return fail("mlmc_accum_kernel_time: null argument");
    if (a->ev_used) {   // the event pairs are read lazily, here: wait for the last one, then add them up
        MLMC_HIP_CHECK(hipEventSynchronize(a->ev[a->ev_used - 1]));
        if (int rc = timing_collect(a)) return rc;
    }
    if (ms) *ms = a->ms_total;
    if (launches) *launches = a->launches;
    if (alg_bytes) *alg_bytes = a->alg_bytes;
    a->ms_total = 0;
    a->launches = 0;
    a->alg_bytes = 0;
    return 0;
}

int mlmc_accum_aux_kernel_time(mlmc_accum *a, double *ms, int64_t *launches, int64_t *alg_bytes) {
    MLMC_API_GUARD;
    if (!a) return fail("mlmc_accum_aux_kernel_time: null argument");
    double t = 0.0, t0 = 0.0;
    int64_t l = 0, l0 = 0, b = 0, b0 = 0;
    if (a->lin)
        if (int rc = mlmc_accum_kernel_time(a->lin, &t, &l, &b)) return rc;
    if (a->lin0)
        if (int rc = mlmc_accum_kernel_time(a->lin0, &t0, &l0, &b0)) return rc;
    if (ms) *ms = t + t0;
    if (launches) *launches = l + l0;
    if (alg_bytes) *alg_bytes = b + b0;
    return 0;
}

int mlmc_linearization_table(int32_t kind, int32_t R, int32_t squares, double *out, int64_t out_len) {
    if (!out) return fail("mlmc_linearization_table: null argument");
    if (kind != MLMC_LEGENDRE && kind != MLMC_MONOMIAL) return fail("mlmc_linearization_table: Legendre or monomial moments");
    if (R < 1 || R > (squares ? 64 : 128)) return fail("mlmc_linearization_table: size out of range");
    const int64_t K = squares ? 4 * (int64_t)R - 3 : 2 * (int64_t)R - 1;
    if (out_len < K * R * R) return fail("mlmc_linearization_table: output too small");
    std::vector<double> t;
    if (squares) square_product_table(kind, R, t);
    else product_table(kind, R, t);
    std::memcpy(out, t.data(), sizeof(double) * t.size());
    return 0;
}

int mlmc_accum_kernel_flops(mlmc_accum *a, int64_t *mfma_flops) {
    MLMC_API_GUARD;
    if (!a) return fail("mlmc_accum_kernel_flops: null argument");
    if (mfma_flops) *mfma_flops = a->mfma_flops;
    a->mfma_flops = 0;
    return 0;
}

}  // extern "C"
