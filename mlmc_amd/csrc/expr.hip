// Quantity expressions on the device (include/mlmc_hip.h, "quantity expressions").
//
// Reference: mlmc/quantity/quantity.py evaluates a tree of NumPy closures chunk by chunk, one temporary [M, n, 2] array
// per node (:117-135, arithmetic :166-246, ufuncs :366-400, masks :250-306, select :137-164).  Here the host lowers the
// tree to a register program (mlmc_amd/quantity/lowering.py) and ONE kernel evaluates it per sample: the stored rows
// are read once (16 B per pair, one 128-bit load), every node is applied in registers, only the result rows are
// written.  HBM-bound: algorithmic bytes = 16 B per referenced stored row + 16 B per result row, per sample pair.
//
// The register file lives in LDS as [reg][side][thread] (a thread touches only its own column: conflict-free), because
// registers are indexed by the program.  The program itself is read through uniform (scalar) loads.
#include "common.hpp"
#include "expr_jit.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace mlmc;

struct mlmc_expr {
    std::vector<mlmc_expr_instr> prog;
    int n_regs = 0, n_in = 0, n_out = 0;
    bool selects = false;
    bool heavy = false;                   // uses a libm-backed operation (k_expr<.., HEAVY = true>)
    std::shared_ptr<mlmc::ExprJit> jit;   // compiled form of the program, shared by all handles with the same instructions
    mlmc_expr_instr *d_prog = nullptr;
    const double **d_rows = nullptr;      // device copy of the row pointer table
    // scratch of selecting programs: uncompacted rows, flags, block offsets
    double *d_tmp_f = nullptr; size_t tmp_f_cap = 0;
    double *d_tmp_c = nullptr; size_t tmp_c_cap = 0;
    uint8_t *d_keep = nullptr; size_t keep_cap = 0;
    int64_t *d_offsets = nullptr; size_t offsets_cap = 0;
    int64_t *h_total = nullptr;           // pinned
    // HIP-event timing of k_expr (mlmc_init flag bit0), read lazily by mlmc_expr_kernel_time
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double ms_total = 0;
    int64_t launches = 0, alg_bytes = 0;
    int64_t jit_launches = 0;             // evaluations that ran the compiled kernel (mlmc_expr_state)
};

static int expr_timing_collect(mlmc_expr *e) {
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        float ms = 0.f;
        MLMC_HIP_CHECK(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
        e->ms_total += ms;
    }
    e->ev_used = 0;
    return 0;
}

namespace mlmc {

constexpr int X_THREADS = 256;
constexpr int COMPACT_CHUNK = 4096;   // samples per block of the compaction kernels

__device__ __forceinline__ double np_remainder(double a, double b) {   // numpy npy_remainder / Python float %
    double m = fmod(a, b);
    if (b == 0.0) return m;            // NaN
    if (m != 0.0) {
        if ((b < 0.0) != (m < 0.0)) m += b;
    } else {
        m = copysign(0.0, b);
    }
    return m;
}
__device__ __forceinline__ double np_maximum(double a, double b) { return (a >= b || a != a) ? a : b; }
__device__ __forceinline__ double np_minimum(double a, double b) { return (a <= b || a != a) ? a : b; }
__device__ __forceinline__ double np_sign(double a) { return a != a ? a : (a > 0.0 ? 1.0 : (a < 0.0 ? -1.0 : 0.0)); }

// Register file in LDS: [reg][side][slot][thread]; a thread works on S samples (slots) per instruction so that the
// decode of an instruction and the latency of its loads are shared by S samples.  The result of the latest value-
// producing instruction also stays in VGPRs (`prev`): an operand flagged MLMC_X_A_PREV / MLMC_X_B_PREV is taken from
// there, and a result that is only ever read that way is not written to LDS at all (MLMC_X_NO_WB) -- expression chains
// then run in registers and LDS carries only the values that outlive their successor.
template <int V>
__device__ __forceinline__ void fetch(double (&o)[V], bool is_imm, bool is_prev, double imm, const double (&prev)[V],
                                      const double *x) {
    if (is_imm) {
#pragma unroll
        for (int s = 0; s < V; ++s) o[s] = imm;
    } else if (is_prev) {
#pragma unroll
        for (int s = 0; s < V; ++s) o[s] = prev[s];
    } else {
#pragma unroll
        for (int s = 0; s < V; ++s) o[s] = x[s * X_THREADS];
    }
}

template <int V, bool HEAVY>
__device__ __forceinline__ void binary(int op, double (&r)[V], const double (&av)[V], const double (&bv)[V]) {
#pragma unroll
    for (int s = 0; s < V; ++s) {
        const double a = av[s], b = bv[s];
        switch (op) {
            case MLMC_X_ADD: r[s] = a + b; break;
            case MLMC_X_SUB: r[s] = a - b; break;
            case MLMC_X_MUL: r[s] = a * b; break;
            case MLMC_X_DIV: r[s] = a / b; break;
            case MLMC_X_MOD: if (HEAVY) r[s] = np_remainder(a, b); else r[s] = 0.0; break;
            case MLMC_X_POW: if (HEAVY) r[s] = pow(a, b); else r[s] = 0.0; break;
            case MLMC_X_MAXIMUM: r[s] = np_maximum(a, b); break;
            case MLMC_X_MINIMUM: r[s] = np_minimum(a, b); break;
            case MLMC_X_FMAX: r[s] = fmax(a, b); break;
            case MLMC_X_FMIN: r[s] = fmin(a, b); break;
            case MLMC_X_ATAN2: if (HEAVY) r[s] = atan2(a, b); else r[s] = 0.0; break;
            case MLMC_X_HYPOT: if (HEAVY) r[s] = hypot(a, b); else r[s] = 0.0; break;
            case MLMC_X_FMOD: if (HEAVY) r[s] = fmod(a, b); else r[s] = 0.0; break;
            case MLMC_X_AND: r[s] = (a != 0.0 && b != 0.0) ? 1.0 : 0.0; break;
            case MLMC_X_OR: r[s] = (a != 0.0 || b != 0.0) ? 1.0 : 0.0; break;
            default: r[s] = ((a != 0.0) != (b != 0.0)) ? 1.0 : 0.0; break;   // XOR
        }
    }
}

template <int V, bool HEAVY>
__device__ __forceinline__ void unary(int op, double (&r)[V], const double (&av)[V]) {
#pragma unroll
    for (int s = 0; s < V; ++s) {
        const double a = av[s];
        switch (op) {
            case MLMC_X_NEG: r[s] = -a; break;
            case MLMC_X_ABS: r[s] = fabs(a); break;
            case MLMC_X_SQRT: r[s] = sqrt(a); break;
            case MLMC_X_SQUARE: r[s] = a * a; break;
            case MLMC_X_RECIP: r[s] = 1.0 / a; break;
            case MLMC_X_EXP: if (HEAVY) r[s] = exp(a); else r[s] = 0.0; break;
            case MLMC_X_EXP2: if (HEAVY) r[s] = exp2(a); else r[s] = 0.0; break;
            case MLMC_X_EXPM1: if (HEAVY) r[s] = expm1(a); else r[s] = 0.0; break;
            case MLMC_X_LOG: if (HEAVY) r[s] = log(a); else r[s] = 0.0; break;
            case MLMC_X_LOG2: if (HEAVY) r[s] = log2(a); else r[s] = 0.0; break;
            case MLMC_X_LOG10: if (HEAVY) r[s] = log10(a); else r[s] = 0.0; break;
            case MLMC_X_LOG1P: if (HEAVY) r[s] = log1p(a); else r[s] = 0.0; break;
            case MLMC_X_SIN: if (HEAVY) r[s] = sin(a); else r[s] = 0.0; break;
            case MLMC_X_COS: if (HEAVY) r[s] = cos(a); else r[s] = 0.0; break;
            case MLMC_X_TAN: if (HEAVY) r[s] = tan(a); else r[s] = 0.0; break;
            case MLMC_X_ASIN: if (HEAVY) r[s] = asin(a); else r[s] = 0.0; break;
            case MLMC_X_ACOS: if (HEAVY) r[s] = acos(a); else r[s] = 0.0; break;
            case MLMC_X_ATAN: if (HEAVY) r[s] = atan(a); else r[s] = 0.0; break;
            case MLMC_X_SINH: if (HEAVY) r[s] = sinh(a); else r[s] = 0.0; break;
            case MLMC_X_COSH: if (HEAVY) r[s] = cosh(a); else r[s] = 0.0; break;
            case MLMC_X_TANH: if (HEAVY) r[s] = tanh(a); else r[s] = 0.0; break;
            case MLMC_X_FLOOR: r[s] = floor(a); break;
            case MLMC_X_CEIL: r[s] = ceil(a); break;
            case MLMC_X_TRUNC: r[s] = trunc(a); break;
            case MLMC_X_RINT: r[s] = rint(a); break;
            case MLMC_X_SIGN: r[s] = np_sign(a); break;
            case MLMC_X_CBRT: if (HEAVY) r[s] = cbrt(a); else r[s] = 0.0; break;
            default: r[s] = (a == 0.0) ? 1.0 : 0.0; break;   // NOT
        }
    }
}

__device__ __forceinline__ bool compare(int op, double a, double b) {
    switch (op) {
        case MLMC_X_LT: return a < b;
        case MLMC_X_LE: return a <= b;
        case MLMC_X_GT: return a > b;
        case MLMC_X_GE: return a >= b;
        case MLMC_X_EQ: return a == b;
        default: return a != b;
    }
}

constexpr int X_TABLE_ROWS = 64;   // row pointers passed by value in the kernel arguments (no table upload)
struct RowTable {
    const double *p[X_TABLE_ROWS];
};

// The program counter is uniform over the grid, so every branch below is a scalar branch.  A block covers
// S * X_THREADS consecutive samples; slot k of thread t is sample base + k * X_THREADS + t (coalesced per slot).
// HEAVY = false leaves out the libm-backed operations (exp, log, trigonometry, pow, fmod): their register demand
// (~240 VGPRs, two waves per SIMD) would otherwise set the occupancy of every program, and this kernel lives on
// memory-level parallelism.
template <bool PAIR, int S, bool HEAVY>
__global__ __launch_bounds__(X_THREADS) void k_expr(const mlmc_expr_instr *__restrict__ prog, int n_instr, RowTable tab,
                                                    const double *const *__restrict__ rows, int64_t n, int64_t ss, int64_t cs,
                                                    int n_regs,
                                                    double *__restrict__ out_f, double *__restrict__ out_c,
                                                    uint8_t *__restrict__ keep_out) {
    extern __shared__ double regs[];   // [n_regs][sides][S][X_THREADS]
    constexpr int SIDES = PAIR ? 2 : 1;
    constexpr int V = SIDES * S;       // values per thread and instruction: index side * S + slot
    constexpr int REG_STRIDE = V * X_THREADS;
    const int64_t i0 = (int64_t)blockIdx.x * (S * X_THREADS) + threadIdx.x;
    double *const mine = regs + threadIdx.x;
    bool keep[S];
#pragma unroll
    for (int k = 0; k < S; ++k) keep[k] = true;
    double prev[V];                    // result of the latest value-producing instruction
#pragma unroll
    for (int s = 0; s < V; ++s) prev[s] = 0.0;
    for (int pc = 0; pc < n_instr; ++pc) {
        const mlmc_expr_instr ins = prog[pc];
        double *const d = mine + (size_t)ins.dst * REG_STRIDE;
        const double *const x = mine + (size_t)ins.a * REG_STRIDE;
        const double *const y = mine + (size_t)ins.b * REG_STRIDE;
        const int op = ins.op & MLMC_X_OP_MASK;
        const bool imm_a = (ins.op & MLMC_X_IMM_A) != 0, imm_b = (ins.op & MLMC_X_IMM_B) != 0;
        const bool prev_a = (ins.op & MLMC_X_A_PREV) != 0, prev_b = (ins.op & MLMC_X_B_PREV) != 0;
        const bool write_back = (ins.op & MLMC_X_NO_WB) == 0;
        if (op == MLMC_X_STORE) {
            double v[V];
            fetch<V>(v, false, prev_a, 0.0, prev, x);
#pragma unroll
            for (int k = 0; k < S; ++k) {
                const int64_t i = i0 + (int64_t)k * X_THREADS;
                if (i < n) {
                    out_f[(int64_t)ins.b * n + i] = v[k];
                    if (PAIR) out_c[(int64_t)ins.b * n + i] = v[S + k];
                }
            }
            continue;
        }
        if (op == MLMC_X_SELECT) {
            double v[V];
            fetch<V>(v, false, prev_a, 0.0, prev, x);
#pragma unroll
            for (int k = 0; k < S; ++k) keep[k] = keep[k] && (v[k] != 0.0);
            continue;
        }
        // value-producing instructions: the result goes to `prev` and, unless flagged, to its LDS register
        if (op == MLMC_X_LOAD) {
            // global address space (global_load, not flat_load) and branch-free: the index is clamped into the row and the value
            // selected afterwards, so the S loads of an instruction are issued back to back
            typedef const __attribute__((address_space(1))) double *gptr;
            typedef double d2 __attribute__((ext_vector_type(2)));
            typedef const __attribute__((address_space(1))) d2 *gptr2;
            const gptr row = (gptr)(rows ? rows[ins.a] : tab.p[ins.a]);
            if (PAIR) {
                double2 v[S];
                if (ss == 2 && cs == 1) {           // interleaved (fine, coarse) pairs: one 128-bit load per sample
#pragma unroll
                    for (int k = 0; k < S; ++k) {   // S independent loads in flight
                        const int64_t i = i0 + (int64_t)k * X_THREADS;
                        const d2 t = ((gptr2)row)[i < n ? i : n - 1];
                        v[k] = make_double2(t.x, t.y);
                    }
                } else {                            // a row of a stored [n][2][M] block: strides 2 M (sample), M (side)
#pragma unroll
                    for (int k = 0; k < S; ++k) {
                        const int64_t i = i0 + (int64_t)k * X_THREADS;
                        const int64_t ic = i < n ? i : n - 1;
                        v[k] = make_double2(row[ic * ss], row[ic * ss + cs]);
                    }
                }
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const bool in = i0 + (int64_t)k * X_THREADS < n;
                    prev[k] = in ? v[k].x : 0.0;
                    prev[(V - S) + k] = in ? v[k].y : 0.0;
                }
            } else {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const int64_t i = i0 + (int64_t)k * X_THREADS;
                    const double v = row[(i < n ? i : n - 1) * ss];
                    prev[k] = i < n ? v : 0.0;
                }
            }
        } else if (op == MLMC_X_CONST) {
#pragma unroll
            for (int s = 0; s < V; ++s) prev[s] = ins.imm;
        } else if (op >= MLMC_X_LT && op <= MLMC_X_NE) {
            double av[V], bv[V];
            fetch<V>(av, imm_a, prev_a, ins.imm, prev, x);
            fetch<V>(bv, imm_b, prev_b, ins.imm, prev, y);
#pragma unroll
            for (int k = 0; k < S; ++k) {
                bool r = compare(op, av[k], bv[k]);
                if (PAIR) r = compare(op, av[(V - S) + k], bv[(V - S) + k]) && r;
                prev[k] = r ? 1.0 : 0.0;
                if (PAIR) prev[(V - S) + k] = prev[k];
            }
        } else if ((op >= MLMC_X_ADD && op <= MLMC_X_FMOD) || op == MLMC_X_AND || op == MLMC_X_OR || op == MLMC_X_XOR) {
            double av[V], bv[V];
            fetch<V>(av, imm_a, prev_a, ins.imm, prev, x);
            fetch<V>(bv, imm_b, prev_b, ins.imm, prev, y);
            binary<V, HEAVY>(op, prev, av, bv);
        } else {
            double av[V];
            fetch<V>(av, false, prev_a, 0.0, prev, x);
            unary<V, HEAVY>(op, prev, av);
        }
        if (write_back) {
#pragma unroll
            for (int s = 0; s < V; ++s) d[s * X_THREADS] = prev[s];
        }
    }
    if (keep_out) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int64_t i = i0 + (int64_t)k * X_THREADS;
            if (i < n) keep_out[i] = keep[k] ? 1 : 0;
        }
    }
}

// ---- order-preserving compaction of the selected samples --------------------------------------------------------
__global__ __launch_bounds__(X_THREADS) void k_keep_counts(const uint8_t *__restrict__ keep, int64_t n, int64_t *__restrict__ counts) {
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * COMPACT_CHUNK;
    int c = 0;
    for (int k = threadIdx.x; k < COMPACT_CHUNK; k += X_THREADS) {
        const int64_t i = base + k;
        c += (i < n && keep[i]) ? 1 : 0;
    }
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the block counts in place (one block); offsets[nblk] = total
__global__ __launch_bounds__(1024) void k_scan_counts(int64_t *__restrict__ counts, int nblk, int64_t *__restrict__ total_out) {
    __shared__ int64_t part[1024];
    const int per = (nblk + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(nblk, lo + per);
    int64_t s = 0;
    for (int k = lo; k < hi; ++k) s += counts[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int k = 0; k < 1024; ++k) { const int64_t v = part[k]; part[k] = run; run += v; }
        counts[nblk] = run;
        *total_out = run;
    }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int k = lo; k < hi; ++k) { const int64_t v = counts[k]; counts[k] = run; run += v; }
}

__global__ __launch_bounds__(X_THREADS) void k_compact(const uint8_t *__restrict__ keep, const int64_t *__restrict__ offsets,
                                                       int nblk, int64_t n, int n_rows, const double *__restrict__ src_f,
                                                       const double *__restrict__ src_c, double *__restrict__ dst_f,
                                                       double *__restrict__ dst_c) {
    __shared__ int wcnt[4];
    const int64_t n_sel = offsets[nblk];
    int64_t run = offsets[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * COMPACT_CHUNK;
    for (int k0 = 0; k0 < COMPACT_CHUNK; k0 += X_THREADS) {
        const int64_t i = base + k0 + threadIdx.x;
        const bool kp = i < n && keep[i];
        const unsigned long long ball = __ballot(kp);
        if (lane == 0) wcnt[wave] = __popcll(ball);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int c = wcnt[w]; if (w < wave) before += c; total += c; }
        if (kp) {
            const int64_t pos = run + before + __popcll(ball & ((1ull << lane) - 1ull));
            for (int r = 0; r < n_rows; ++r) {
                dst_f[(int64_t)r * n_sel + pos] = src_f[(int64_t)r * n + i];
                if (src_c) dst_c[(int64_t)r * n_sel + pos] = src_c[(int64_t)r * n + i];
            }
        }
        run += total;
        __syncthreads();
    }
}

}  // namespace mlmc

extern "C" {

int mlmc_expr_create(const mlmc_expr_instr *prog, int32_t n_instr, int32_t n_regs, int32_t n_in_rows, int32_t n_out_rows,
                     mlmc_expr **out) {
    MLMC_API_GUARD;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!prog || !out) return fail("mlmc_expr_create: null argument");
    if (n_instr < 1 || n_instr > MLMC_EXPR_MAX_INSTR) return fail("mlmc_expr_create: program length out of range");
    if (n_regs < 1 || n_regs > MLMC_EXPR_MAX_REGS) return fail("mlmc_expr_create: register count out of range");
    if (n_in_rows < 1 || n_out_rows < 1) return fail("mlmc_expr_create: a program needs input and output rows");
    bool selects = false, heavy = false, produced = false;
    std::vector<char> written(n_regs, 0), stored(n_out_rows, 0);
    for (int k = 0; k < n_instr; ++k) {   // validate: the kernel trusts every index
        mlmc_expr_instr in = prog[k];
        const bool imm_a = (in.op & MLMC_X_IMM_A) != 0, imm_b = (in.op & MLMC_X_IMM_B) != 0;
        const bool prev_a = (in.op & MLMC_X_A_PREV) != 0, prev_b = (in.op & MLMC_X_B_PREV) != 0;
        const bool no_wb = (in.op & MLMC_X_NO_WB) != 0;
        in.op &= MLMC_X_OP_MASK;
        if (in.op >= MLMC_X_N_OPS) return fail("mlmc_expr_create: unknown opcode");
        const bool two_operands = (in.op >= MLMC_X_ADD && in.op <= MLMC_X_FMOD) || (in.op >= MLMC_X_LT && in.op <= MLMC_X_NE);
        if ((imm_a || imm_b) && (!two_operands || (imm_a && imm_b)))
            return fail("mlmc_expr_create: immediate operands are for arithmetic / comparison instructions, one per instruction");
        const bool uses_a = in.op != MLMC_X_LOAD && in.op != MLMC_X_CONST && !imm_a;
        const bool uses_b = (two_operands || in.op == MLMC_X_AND || in.op == MLMC_X_OR || in.op == MLMC_X_XOR) && !imm_b;
        const bool produces = in.op != MLMC_X_STORE && in.op != MLMC_X_SELECT;
        if ((prev_a && !uses_a) || (prev_b && !uses_b) || ((prev_a || prev_b) && !produced))
            return fail("mlmc_expr_create: a chained operand needs a register operand slot and an earlier result");
        if (no_wb && !produces) return fail("mlmc_expr_create: only value-producing instructions can skip the write-back");
        const bool reads_a = uses_a && !prev_a, reads_b = uses_b && !prev_b;
        const bool writes = produces && !no_wb;
        if (produces) produced = true;
        if (in.op == MLMC_X_LOAD && in.a >= n_in_rows) return fail("mlmc_expr_create: input row out of range");
        if (reads_a && (in.a >= n_regs || !written[in.a])) return fail("mlmc_expr_create: operand a reads an unset register");
        if (reads_b && (in.b >= n_regs || !written[in.b])) return fail("mlmc_expr_create: operand b reads an unset register");
        if (in.op == MLMC_X_STORE) {
            if (in.b >= n_out_rows) return fail("mlmc_expr_create: output row out of range");
            stored[in.b] = 1;
        }
        if (writes) {
            if (in.dst >= n_regs) return fail("mlmc_expr_create: destination register out of range");
            written[in.dst] = 1;
        }
        if (in.op == MLMC_X_SELECT) selects = true;
        switch (in.op) {
            case MLMC_X_MOD: case MLMC_X_POW: case MLMC_X_ATAN2: case MLMC_X_HYPOT: case MLMC_X_FMOD: case MLMC_X_EXP:
            case MLMC_X_EXP2: case MLMC_X_EXPM1: case MLMC_X_LOG: case MLMC_X_LOG2: case MLMC_X_LOG10: case MLMC_X_LOG1P:
            case MLMC_X_SIN: case MLMC_X_COS: case MLMC_X_TAN: case MLMC_X_ASIN: case MLMC_X_ACOS: case MLMC_X_ATAN:
            case MLMC_X_SINH: case MLMC_X_COSH: case MLMC_X_TANH: case MLMC_X_CBRT: heavy = true; break;
            default: break;
        }
    }
    for (int r = 0; r < n_out_rows; ++r)
        if (!stored[r]) return fail("mlmc_expr_create: an output row is never stored");
    mlmc_expr *e = new (std::nothrow) mlmc_expr();
    if (!e) return fail("mlmc_expr_create: out of memory");
    e->prog.assign(prog, prog + n_instr);
    e->n_regs = n_regs;
    e->n_in = n_in_rows;
    e->n_out = n_out_rows;
    e->selects = selects;
    e->heavy = heavy;
    e->jit = expr_jit_lookup(e->prog, n_regs, n_in_rows);
    if (hipMalloc(&e->d_prog, sizeof(mlmc_expr_instr) * n_instr) != hipSuccess ||
        hipMalloc(&e->d_rows, sizeof(double *) * n_in_rows) != hipSuccess ||
        hipHostMalloc(&e->h_total, sizeof(int64_t), hipHostMallocDefault) != hipSuccess) {
        mlmc_expr_destroy(e);
        return fail("mlmc_expr_create: device allocation failed");
    }
    MLMC_HIP_CHECK(hipMemcpy(e->d_prog, prog, sizeof(mlmc_expr_instr) * n_instr, hipMemcpyHostToDevice));
    *out = e;
    return 0;
}

void mlmc_expr_destroy(mlmc_expr *e) {
    MLMC_API_GUARD;
    if (!e) return;
    if (rt().ready) (void)wait_stream(rt().stream);
    void *ptrs[] = {e->d_prog, (void *)e->d_rows, e->d_tmp_f, e->d_tmp_c, e->d_keep, e->d_offsets};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->h_total) (void)hipHostFree(e->h_total);
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    delete e;
}

int mlmc_expr_eval(mlmc_expr *e, const double *const *rows_in, int32_t has_coarse, int64_t n, int64_t sample_stride,
                   int64_t side_stride, double *fine_out, double *coarse_out, int64_t *n_selected) {
    MLMC_API_GUARD;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!e || !rows_in || !fine_out) return fail("mlmc_expr_eval: null argument");
    if (has_coarse && !coarse_out) return fail("mlmc_expr_eval: coarse_out is NULL");
    if (n < 0 || n > ((int64_t)1 << 31)) return fail("mlmc_expr_eval: n out of range, split the chunk");
    if (sample_stride < 1 || (has_coarse && side_stride < 1)) return fail("mlmc_expr_eval: strides must be positive");
    for (int r = 0; r < e->n_in; ++r)
        if (!rows_in[r]) return fail("mlmc_expr_eval: a stored row pointer is NULL");
    if (n_selected) *n_selected = n;
    if (n == 0) return 0;
    hipStream_t st = rt().stream;
    RowTable tab;
    std::memset(&tab, 0, sizeof(tab));
    const double *const *d_rows = nullptr;
    if (e->n_in <= X_TABLE_ROWS) {
        for (int r = 0; r < e->n_in; ++r) tab.p[r] = rows_in[r];
    } else {
        // the table of the previous call may still be read by its kernel: drain the stream, then overwrite it
        MLMC_HIP_CHECK(wait_stream(st));
        MLMC_HIP_CHECK(hipMemcpy((void *)e->d_rows, rows_in, sizeof(double *) * e->n_in, hipMemcpyHostToDevice));
        d_rows = e->d_rows;
    }
    double *tf = fine_out, *tc = coarse_out;
    uint8_t *keep = nullptr;
    int nblk = 0;
    if (e->selects) {
        const size_t bytes = sizeof(double) * (size_t)e->n_out * (size_t)n;
        if (int rc = ensure((void **)&e->d_tmp_f, &e->tmp_f_cap, bytes)) return rc;
        if (has_coarse)
            if (int rc = ensure((void **)&e->d_tmp_c, &e->tmp_c_cap, bytes)) return rc;
        if (int rc = ensure((void **)&e->d_keep, &e->keep_cap, (size_t)n)) return rc;
        nblk = (int)((n + COMPACT_CHUNK - 1) / COMPACT_CHUNK);
        if (int rc = ensure((void **)&e->d_offsets, &e->offsets_cap, sizeof(int64_t) * ((size_t)nblk + 1))) return rc;
        tf = e->d_tmp_f;
        tc = has_coarse ? e->d_tmp_c : nullptr;
        keep = e->d_keep;
    }
    // samples per thread: two (measured best for both kernels: 5.3-6.1 TB/s on copy / arithmetic programs; with four
    // the light kernel loses occupancy and the libm kernel spills), as long as the register file of a block stays
    // within 32 KB of LDS
    const int sides = has_coarse ? 2 : 1;
    // level 0 (8 bytes per sample and row instead of 16): four samples per thread keep as many bytes in flight as two pairs
    const int s_max = (has_coarse || getenv("MLMC_EXPR_L0_S2")) ? 2 : 4;
    int S = 1;
    while (S < s_max && (size_t)e->n_regs * sides * (2 * S) * X_THREADS * sizeof(double) <= 32768) S *= 2;
    if (const char *force = getenv("MLMC_EXPR_SLOTS")) {   // tuning aid
        const int f = atoi(force);
        if ((f == 1 || f == 2 || f == 4) && (size_t)e->n_regs * sides * f * X_THREADS * sizeof(double) <= 65536) S = f;
    }
    const size_t lds = sizeof(double) * (size_t)e->n_regs * sides * S * X_THREADS;
    const unsigned blocks = (unsigned)((n + (int64_t)S * X_THREADS - 1) / ((int64_t)S * X_THREADS));
    const int n_instr = (int)e->prog.size();
    const bool timed = (rt().flags & 1) != 0;
    if (timed) {
        if (e->ev_used >= 8192) {
            MLMC_HIP_CHECK(hipEventSynchronize(e->ev[e->ev_used - 1]));
            if (int rc = expr_timing_collect(e)) return rc;
        }
        if (e->ev_used + 2 > e->ev.size()) {
            hipEvent_t e0, e1;
            MLMC_HIP_CHECK(hipEventCreate(&e0));
            MLMC_HIP_CHECK(hipEventCreate(&e1));
            e->ev.push_back(e0);
            e->ev.push_back(e1);
        }
        MLMC_HIP_CHECK(hipEventRecord(e->ev[e->ev_used], st));
    }
#define MLMC_X_LAUNCH(P, SS, H)                                                                                            \
    hipLaunchKernelGGL((k_expr<P, SS, H>), dim3(blocks), dim3(X_THREADS), lds, st, e->d_prog, n_instr, tab, d_rows, n, sample_stride, side_stride, e->n_regs, \
                       tf, tc, keep)
#define MLMC_X_LAUNCH_S(P, H)                                                                                              \
    do {                                                                                                                   \
        if (S == 4) MLMC_X_LAUNCH(P, 4, H); else if (S == 2) MLMC_X_LAUNCH(P, 2, H); else MLMC_X_LAUNCH(P, 1, H);          \
    } while (0)
    if (e->jit && expr_jit_ready(*e->jit)) {
        // the program's own kernel (expr_jit.hip): registers are locals, no LDS, no decode; same rows bit for bit
        if (int rc = expr_jit_launch(*e->jit, has_coarse != 0, &tab, d_rows, n, sample_stride, side_stride, tf, tc, keep, st)) return rc;
        e->jit_launches += 1;
    } else if (has_coarse) {
        if (e->heavy) MLMC_X_LAUNCH_S(true, true); else MLMC_X_LAUNCH_S(true, false);
    } else {
        if (e->heavy) MLMC_X_LAUNCH_S(false, true); else MLMC_X_LAUNCH_S(false, false);
    }
#undef MLMC_X_LAUNCH_S
#undef MLMC_X_LAUNCH
    MLMC_HIP_CHECK(hipGetLastError());
    if (timed) {
        MLMC_HIP_CHECK(hipEventRecord(e->ev[e->ev_used + 1], st));
        e->ev_used += 2;
    }
    e->launches += 1;
    e->alg_bytes += (int64_t)sizeof(double) * sides * n * (e->n_in + e->n_out);   // every referenced row in, every result row out
    if (!e->selects) return 0;
    hipLaunchKernelGGL(k_keep_counts, dim3(nblk), dim3(X_THREADS), 0, st, keep, n, e->d_offsets);
    hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, e->d_offsets, nblk, e->h_total);
    hipLaunchKernelGGL(k_compact, dim3(nblk), dim3(X_THREADS), 0, st, keep, e->d_offsets, nblk, n, e->n_out, tf, tc, fine_out,
                       has_coarse ? coarse_out : nullptr);
    MLMC_HIP_CHECK(hipGetLastError());
    MLMC_HIP_CHECK(wait_stream(st));
    if (n_selected) *n_selected = *e->h_total;
    return 0;
}

int mlmc_expr_state(mlmc_expr *e, int32_t *state, int64_t *compiled_launches) {
    MLMC_API_GUARD;
    if (!e) return fail("mlmc_expr_state: null argument");
    if (state) *state = e->jit ? e->jit->state : -2;
    if (compiled_launches) *compiled_launches = e->jit_launches;
    return 0;
}

int mlmc_expr_kernel_time(mlmc_expr *e, double *ms, int64_t *launches, int64_t *alg_bytes) {
    MLMC_API_GUARD;
    if (!e) return fail("mlmc_expr_kernel_time: null argument");
    if (e->ev_used) {
        MLMC_HIP_CHECK(hipEventSynchronize(e->ev[e->ev_used - 1]));
        if (int rc = expr_timing_collect(e)) return rc;
    }
    if (ms) *ms = e->ms_total;
    if (launches) *launches = e->launches;
    if (alg_bytes) *alg_bytes = e->alg_bytes;
    e->ms_total = 0;
    e->launches = 0;
    e->alg_bytes = 0;
    return 0;
}

}  // extern "C"
