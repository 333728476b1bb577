// Quantity expressions compiled per program (include/mlmc_hip.h, "quantity expressions").
//
// k_expr (expr.hip) interprets a register program: every instruction costs a scalar fetch, a switch and LDS traffic for the
// register file -- about 100 issue slots per instruction and wave, which kept light trees at 57-61 % of the HBM roofline.
// A program that is evaluated again and again (every estimate of an analysis re-evaluates the same tree over every chunk)
// is therefore turned into HIP source -- one straight-line statement block per instruction, registers as local variables,
// no LDS, no decode -- compiled for gfx950 with hiprtc and launched through the module API.  Same operations in the same
// order with -ffp-contract=off: the rows are bit for bit those of the interpreter (tests compare the two).
//
// Policy: the third evaluation of a program (counted over all handles with the same instructions) compiles it, synchronously
// (20-50 ms; the first compilation of a process ~0.4 s: comgr start-up) -- one-off trees never pay, repeated ones pay once.
// Code objects are cached per program text for the life of the process.  hiprtc is loaded with dlopen: without it, or on any
// compiler error, the interpreter stays in charge (MLMC_EXPR_JIT=0 forces that; MLMC_EXPR_JIT_AFTER=n moves the threshold,
// 0 = compile at the first evaluation).
#include "common.hpp"
#include "expr_jit.hpp"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace mlmc {

namespace {

// ---- hiprtc through dlopen ---------------------------------------------------------------------------------------
struct Rtc {
    typedef struct _hiprtcProgram *Program;
    int (*create)(Program *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(Program, int, const char **) = nullptr;
    int (*log_size)(Program, size_t *) = nullptr;
    int (*log)(Program, char *) = nullptr;
    int (*code_size)(Program, size_t *) = nullptr;
    int (*code)(Program, char *) = nullptr;
    int (*destroy)(Program *) = nullptr;
    bool ok = false;
};

const Rtc &rtc() {
    static Rtc r = [] {
        Rtc x;
        void *h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
        x.create = (decltype(x.create))dlsym(h, "hiprtcCreateProgram");
        x.compile = (decltype(x.compile))dlsym(h, "hiprtcCompileProgram");
        x.log_size = (decltype(x.log_size))dlsym(h, "hiprtcGetProgramLogSize");
        x.log = (decltype(x.log))dlsym(h, "hiprtcGetProgramLog");
        x.code_size = (decltype(x.code_size))dlsym(h, "hiprtcGetCodeSize");
        x.code = (decltype(x.code))dlsym(h, "hiprtcGetCode");
        x.destroy = (decltype(x.destroy))dlsym(h, "hiprtcDestroyProgram");
        x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
        return x;
    }();
    return r;
}

// ---- code generation ------------------------------------------------------------------------------------------------
std::string lit(double v) {   // exact literal
    char buf[64];
    if (v != v) return "__builtin_nan(\"\")";
    if (v == __builtin_inf()) return "__builtin_inf()";
    if (v == -__builtin_inf()) return "(-__builtin_inf())";
    std::snprintf(buf, sizeof(buf), "%a", v);
    return std::string("(") + buf + ")";
}

const char *binary_expr(int op) {
    switch (op) {
        case MLMC_X_ADD: return "a + b";
        case MLMC_X_SUB: return "a - b";
        case MLMC_X_MUL: return "a * b";
        case MLMC_X_DIV: return "a / b";
        case MLMC_X_MOD: return "np_remainder(a, b)";
        case MLMC_X_POW: return "pow(a, b)";
        case MLMC_X_MAXIMUM: return "np_maximum(a, b)";
        case MLMC_X_MINIMUM: return "np_minimum(a, b)";
        case MLMC_X_FMAX: return "fmax(a, b)";
        case MLMC_X_FMIN: return "fmin(a, b)";
        case MLMC_X_ATAN2: return "atan2(a, b)";
        case MLMC_X_HYPOT: return "hypot(a, b)";
        case MLMC_X_FMOD: return "fmod(a, b)";
        case MLMC_X_AND: return "((a != 0.0 && b != 0.0) ? 1.0 : 0.0)";
        case MLMC_X_OR: return "((a != 0.0 || b != 0.0) ? 1.0 : 0.0)";
        case MLMC_X_XOR: return "(((a != 0.0) != (b != 0.0)) ? 1.0 : 0.0)";
        default: return nullptr;
    }
}

const char *unary_expr(int op) {
    switch (op) {
        case MLMC_X_NEG: return "-a";
        case MLMC_X_ABS: return "fabs(a)";
        case MLMC_X_SQRT: return "sqrt(a)";
        case MLMC_X_SQUARE: return "a * a";
        case MLMC_X_RECIP: return "1.0 / a";
        case MLMC_X_EXP: return "exp(a)";
        case MLMC_X_EXP2: return "exp2(a)";
        case MLMC_X_EXPM1: return "expm1(a)";
        case MLMC_X_LOG: return "log(a)";
        case MLMC_X_LOG2: return "log2(a)";
        case MLMC_X_LOG10: return "log10(a)";
        case MLMC_X_LOG1P: return "log1p(a)";
        case MLMC_X_SIN: return "sin(a)";
        case MLMC_X_COS: return "cos(a)";
        case MLMC_X_TAN: return "tan(a)";
        case MLMC_X_ASIN: return "asin(a)";
        case MLMC_X_ACOS: return "acos(a)";
        case MLMC_X_ATAN: return "atan(a)";
        case MLMC_X_SINH: return "sinh(a)";
        case MLMC_X_COSH: return "cosh(a)";
        case MLMC_X_TANH: return "tanh(a)";
        case MLMC_X_FLOOR: return "floor(a)";
        case MLMC_X_CEIL: return "ceil(a)";
        case MLMC_X_TRUNC: return "trunc(a)";
        case MLMC_X_RINT: return "rint(a)";
        case MLMC_X_SIGN: return "np_sign(a)";
        case MLMC_X_CBRT: return "cbrt(a)";
        case MLMC_X_NOT: return "((a == 0.0) ? 1.0 : 0.0)";
        default: return nullptr;
    }
}

const char *compare_op(int op) {
    switch (op) {
        case MLMC_X_LT: return "<";
        case MLMC_X_LE: return "<=";
        case MLMC_X_GT: return ">";
        case MLMC_X_GE: return ">=";
        case MLMC_X_EQ: return "==";
        default: return "!=";
    }
}

// operand `which` of an instruction at value index s: immediate, the chained previous result, or a register
std::string operand(const mlmc_expr_instr &in, bool is_a, const char *idx) {
    const bool imm = (in.op & (is_a ? MLMC_X_IMM_A : MLMC_X_IMM_B)) != 0;
    const bool prev = (in.op & (is_a ? MLMC_X_A_PREV : MLMC_X_B_PREV)) != 0;
    if (imm) return lit(in.imm);
    if (prev) return std::string("prev[") + idx + "]";
    return "r" + std::to_string(is_a ? in.a : in.b) + "[" + idx + "]";
}

const char *PRELUDE = R"(
typedef long long i64;
typedef const __attribute__((address_space(1))) double *gptr;      // global address space: global_load, not flat_load
typedef const __attribute__((address_space(1))) double2 *gptr2;
struct RowTable { const double *p[64]; };
__device__ __forceinline__ double np_remainder(double a, double b) {
    double m = fmod(a, b);
    if (b == 0.0) return m;
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

// PACKED: the stored rows are interleaved (fine, coarse) pairs, one 128-bit load per sample.  Loads are branch-free (index
// clamped to the last sample, value selected afterwards), so the body is one basic block and the scheduler is free to issue
// every load of the program before the first use.
template <bool PAIR, int S, bool PACKED>
__device__ __forceinline__ void body(const RowTable &tab, const double *const *__restrict__ rows, i64 n, i64 ss, i64 cs,
                                     double *__restrict__ out_f, double *__restrict__ out_c, unsigned char *__restrict__ keep_out) {
    constexpr int V = (PAIR ? 2 : 1) * S;
    const i64 i0 = (i64)blockIdx.x * (S * 256) + threadIdx.x;
    i64 ic[S];                          // sample index clamped into the row
    bool in[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const i64 i = i0 + (i64)k * 256;
        in[k] = i < n;
        ic[k] = in[k] ? i : n - 1;
    }
    bool keep[S];
#pragma unroll
    for (int k = 0; k < S; ++k) keep[k] = true;
    double prev[V];
#pragma unroll
    for (int s = 0; s < V; ++s) prev[s] = 0.0;
)";

const char *EPILOGUE = R"(
    if (keep_out) {
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const i64 i = i0 + (i64)k * 256;
            if (i < n) keep_out[i] = keep[k] ? 1 : 0;
        }
    }
}
extern "C" __global__ __launch_bounds__(256) void k_pair(RowTable tab, const double *const *rows, i64 n, i64 ss, i64 cs, double *out_f,
                                                          double *out_c, unsigned char *keep_out) {
    if (ss == 2 && cs == 1) body<true, 2, true>(tab, rows, n, ss, cs, out_f, out_c, keep_out);
    else body<true, 2, false>(tab, rows, n, ss, cs, out_f, out_c, keep_out);
}
extern "C" __global__ __launch_bounds__(256) void k_single(RowTable tab, const double *const *rows, i64 n, i64 ss, i64 cs, double *out_f,
                                                            double *out_c, unsigned char *keep_out) {
    body<false, 4, false>(tab, rows, n, ss, cs, out_f, out_c, keep_out);
}
)";

std::string generate(const std::vector<mlmc_expr_instr> &prog, int n_regs, int n_in) {
    std::string s = PRELUDE;
    // the same rule as mlmc_expr_eval (expr.hip): up to 64 stored rows the pointers are passed by value in `tab`, beyond
    // that `tab` is zero and they come through `rows` -- decided by the program's row count, not by the rows it loads
    const bool use_table = n_in <= 64;
    for (int r = 0; r < n_regs; ++r) s += "    double r" + std::to_string(r) + "[V];\n";
    for (const mlmc_expr_instr &in : prog) {
        const int op = in.op & MLMC_X_OP_MASK;
        const bool wb = (in.op & MLMC_X_NO_WB) == 0;
        const std::string dst = "r" + std::to_string(in.dst);
        if (op == MLMC_X_STORE) {
            s += "    {\n#pragma unroll\n        for (int k = 0; k < S; ++k) {\n            const i64 i = i0 + (i64)k * 256;\n            if (i < n) {\n";
            s += "                out_f[(i64)" + std::to_string(in.b) + " * n + i] = " + operand(in, true, "k") + ";\n";
            s += "                if (PAIR) out_c[(i64)" + std::to_string(in.b) + " * n + i] = " + operand(in, true, "S + k") + ";\n";
            s += "            }\n        }\n    }\n";
            continue;
        }
        if (op == MLMC_X_SELECT) {
            s += "    {\n#pragma unroll\n        for (int k = 0; k < S; ++k) keep[k] = keep[k] && (" + operand(in, true, "k") + " != 0.0);\n    }\n";
            continue;
        }
        s += "    {\n        double t[V];\n";
        if (op == MLMC_X_LOAD) {
            // the host passes the row pointers by value (kernel argument `tab`) up to 64 stored rows, through `rows` beyond
            s += std::string("        const gptr row = (gptr)") + (use_table ? "tab.p[" : "rows[") + std::to_string(in.a) + "];\n";
            s += "        if (PAIR) {\n#pragma unroll\n            for (int k = 0; k < S; ++k) {\n"
                 "                double2 v;\n"
                 "                if (PACKED) v = ((gptr2)row)[ic[k]];\n"
                 "                else v = make_double2(row[ic[k] * ss], row[ic[k] * ss + cs]);\n"
                 "                t[k] = in[k] ? v.x : 0.0;\n                t[(V - S) + k] = in[k] ? v.y : 0.0;\n            }\n        } else {\n#pragma unroll\n"
                 "            for (int k = 0; k < S; ++k) {\n                const double v = row[ic[k] * ss];\n"
                 "                t[k] = in[k] ? v : 0.0;\n            }\n        }\n";
        } else if (op == MLMC_X_CONST) {
            s += "#pragma unroll\n        for (int s = 0; s < V; ++s) t[s] = " + lit(in.imm) + ";\n";
        } else if (op >= MLMC_X_LT && op <= MLMC_X_NE) {
            const std::string c = compare_op(op);
            s += "#pragma unroll\n        for (int k = 0; k < S; ++k) {\n";
            s += "            bool r = " + operand(in, true, "k") + " " + c + " " + operand(in, false, "k") + ";\n";
            s += "            if (PAIR) r = (" + operand(in, true, "(V - S) + k") + " " + c + " " + operand(in, false, "(V - S) + k") + ") && r;\n";
            s += "            t[k] = r ? 1.0 : 0.0;\n            if (PAIR) t[(V - S) + k] = t[k];\n        }\n";
        } else if (const char *be = binary_expr(op)) {
            s += "#pragma unroll\n        for (int s = 0; s < V; ++s) {\n";
            s += "            const double a = " + operand(in, true, "s") + ", b = " + operand(in, false, "s") + ";\n";
            s += std::string("            t[s] = ") + be + ";\n        }\n";
        } else if (const char *ue = unary_expr(op)) {
            s += "#pragma unroll\n        for (int s = 0; s < V; ++s) {\n";
            s += "            const double a = " + operand(in, true, "s") + ";\n";
            s += std::string("            t[s] = ") + ue + ";\n        }\n";
        } else {
            return std::string();   // unknown opcode: no compiled form
        }
        s += "#pragma unroll\n        for (int s = 0; s < V; ++s) prev[s] = t[s];\n";
        if (wb) s += "#pragma unroll\n        for (int s = 0; s < V; ++s) " + dst + "[s] = t[s];\n";
        s += "    }\n";
    }
    s += EPILOGUE;
    return s;
}

// Process-wide cache of compiled forms, keyed by the program text.  Never destroyed (a leaked heap object): its entries unload
// their code objects in their destructors, which must not run after the HIP runtime has shut down at process exit.
std::map<std::string, std::shared_ptr<ExprJit>> &cache() {
    static auto *c = new std::map<std::string, std::shared_ptr<ExprJit>>();
    return *c;
}
// entries; beyond that, forms no live mlmc_expr refers to are dropped (MLMC_EXPR_JIT_CACHE_MAX overrides: tests)
size_t jit_cache_max() {
    const char *v = std::getenv("MLMC_EXPR_JIT_CACHE_MAX");
    const long n = v ? std::atol(v) : 0;
    return n > 0 ? (size_t)n : 512;
}

// evaluations of a program before it is compiled; -1: compiled forms are switched off (read at every call: tests flip it)
int jit_threshold() {
    const char *off = std::getenv("MLMC_EXPR_JIT");
    if (off && std::strcmp(off, "0") == 0) return -1;
    const char *after = std::getenv("MLMC_EXPR_JIT_AFTER");
    return after ? std::atoi(after) : 2;
}

}  // namespace

std::shared_ptr<ExprJit> expr_jit_lookup(const std::vector<mlmc_expr_instr> &prog, int n_regs, int n_in_rows) {
    if (prog.size() > 512) return nullptr;
    std::string key((const char *)prog.data(), prog.size() * sizeof(mlmc_expr_instr));
    key.push_back((char)n_regs);
    key.push_back(n_in_rows <= 64 ? 't' : 'r');    // where the generated code reads the row pointers from
    auto &c = cache();
    auto it = c.find(key);
    if (it != c.end()) return it->second;
    if (c.size() >= jit_cache_max()) {
        // a long-running host that keeps building new trees: forget the forms only the cache still holds (their modules are
        // unloaded by ~ExprJit; forms in use by a live handle stay -- the handle's shared_ptr keeps them alive either way)
        for (auto jt = c.begin(); jt != c.end();)
            jt = jt->second.use_count() == 1 ? c.erase(jt) : std::next(jt);
    }
    auto e = std::make_shared<ExprJit>();
    e->prog = prog;
    e->n_regs = n_regs;
    e->n_in = n_in_rows;
    c.emplace(std::move(key), e);
    return e;
}

// Called under the API lock at every evaluation: counts the use, compiles at the threshold.  true: launch the compiled kernel.
bool expr_jit_ready(ExprJit &j) {
    const int threshold = jit_threshold();
    if (threshold < 0 || j.state < 0) return false;
    if (j.state == 2) return true;
    if (j.uses++ < threshold) return false;
    j.state = -1;                       // whatever fails below: the interpreter keeps the program
    const Rtc &r = rtc();
    if (!r.ok) return false;
    const std::string src = generate(j.prog, j.n_regs, j.n_in);
    if (src.empty()) return false;
    Rtc::Program prog = nullptr;
    if (r.create(&prog, src.c_str(), "mlmc_expr_jit.hip", 0, nullptr, nullptr) != 0) return false;
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    const int rc = r.compile(prog, 4, opts);
    if (rc != 0) {
        if (std::getenv("MLMC_EXPR_JIT_VERBOSE")) {
            size_t ls = 0;
            r.log_size(prog, &ls);
            std::string log(ls + 1, 0);
            r.log(prog, &log[0]);
            std::fprintf(stderr, "mlmc expr jit: compile failed (%d)\n%s\n", rc, log.c_str());
        }
        r.destroy(&prog);
        return false;
    }
    size_t cs = 0;
    r.code_size(prog, &cs);
    j.code.resize(cs);
    r.code(prog, j.code.data());
    r.destroy(&prog);
    if (hipModuleLoadData(&j.module, j.code.data()) != hipSuccess) return false;
    if (hipModuleGetFunction(&j.fn_pair, j.module, "k_pair") != hipSuccess ||
        hipModuleGetFunction(&j.fn_single, j.module, "k_single") != hipSuccess)
        return false;
    j.state = 2;
    return true;
}

int expr_jit_launch(ExprJit &j, bool pair, const void *tab_bytes, const double *const *d_rows, int64_t n, int64_t ss, int64_t cs,
                    double *out_f, double *out_c, uint8_t *keep, hipStream_t st) {
    const int S = pair ? 2 : 4;
    const unsigned blocks = (unsigned)((n + (int64_t)S * 256 - 1) / ((int64_t)S * 256));
    long long n_ = n, ss_ = ss, cs_ = cs;
    void *args[] = {(void *)tab_bytes, (void *)&d_rows, (void *)&n_, (void *)&ss_, (void *)&cs_, (void *)&out_f, (void *)&out_c, (void *)&keep};
    MLMC_HIP_CHECK(hipModuleLaunchKernel(pair ? j.fn_pair : j.fn_single, blocks, 1, 1, 256, 1, 1, 0, st, args, nullptr));
    return 0;
}

}  // namespace mlmc
