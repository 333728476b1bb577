// Per-program compiled form of a quantity expression (expr_jit.hip); used by expr.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <vector>

#include "../../include/mlmc_hip.h"

namespace mlmc {

struct ExprJit {
    std::vector<mlmc_expr_instr> prog;
    int n_regs = 0;
    int n_in = 0;                  // stored rows of the program: <= 64 -> the row pointers arrive by value (RowTable), else through `rows`
    int uses = 0;
    int state = 0;                 // 0: not compiled yet, 2: module loaded, -1: no compiled form (hiprtc missing / error)
    std::vector<char> code;
    hipModule_t module = nullptr;
    hipFunction_t fn_pair = nullptr, fn_single = nullptr;
    ExprJit() = default;
    ExprJit(const ExprJit &) = delete;
    ExprJit &operator=(const ExprJit &) = delete;
    ~ExprJit() {
        if (module) (void)hipModuleUnload(module);      // runs under the API lock (handle destruction / cache trimming)
    }
};

// Entry of the process-wide cache for this program (nullptr: compiled forms are switched off or the program is too long).
std::shared_ptr<ExprJit> expr_jit_lookup(const std::vector<mlmc_expr_instr> &prog, int n_regs, int n_in_rows);
// Count one evaluation; compile when the threshold is reached.  true: the compiled kernel can be launched.
bool expr_jit_ready(ExprJit &j);
// tab_bytes: the 64 row pointers passed by value (RowTable of expr.hip).
int expr_jit_launch(ExprJit &j, bool pair, const void *tab_bytes, const double *const *d_rows, int64_t n, int64_t ss, int64_t cs,
                    double *out_f, double *out_c, uint8_t *keep, hipStream_t st);

}  // namespace mlmc
