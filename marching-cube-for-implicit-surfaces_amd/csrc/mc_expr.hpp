// mc_expr.hpp -- host-side expression compiler (product code, C++).
//
// Replaces the reference's run-time interpreter Evaluator (Source/evaluator.h:24-86,
// Source/evaluator.cpp:15-237).  The reference re-walks the token list with two
// explicit stacks for EVERY sample (evaluator.cpp:53-107, ~97 % of its sweep time).
// Here the same walk is executed ONCE, symbolically, on the host: operands are DAG
// node ids instead of floats, so the resulting DAG performs exactly the float
// operations, in exactly the order, the reference's evaluate_op recursion would
// (right-to-left reduction of equal-precedence chains, unary minus binding
// tighter than ^, one-level precedence look-ahead: evaluator.cpp:22-48).
// The DAG is then emitted as a straight-line HIP device function that hiprtc
// compiles into the classify / emit kernels (mc_kernels.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mc {

enum class TokType : uint8_t { OP, NUM, VAR, BRAC_O, BRAC_C, NEG, FUNC };

// Grammar extensions (off by default: the reference's tokenizer rejects every letter but x, y, z).
// EXT_TRIG: `sin(` / `cos(` (any case) open a bracket whose value is passed through mc_sinf /
// mc_cosf (include/mc_trig.h) when the bracket closes; otherwise they behave like `(`.
enum : unsigned { EXT_TRIG = 1u };

struct Token {
    TokType type;
    char ch;    // operator / bracket / variable letter; 'N' for NEG; 's' / 'c' for FUNC
    float num;  // NUM: strtof(text) (the reference calls stof per evaluation, evaluator.cpp:82)
};

// evaluator.cpp:139-237.  true = the reference's tokenize() accepts the string.
bool tokenize(const std::string& eq, std::vector<Token>& out, unsigned ext = 0);

enum class NodeOp : uint8_t { CONST, VARX, VARY, VARZ, ADD, SUB, MUL, DIV, POW, POWI, NEG, SIN, COS };

struct Node {
    NodeOp op;
    int a = -1, b = -1;  // operand node ids (value = a op b)
    float cval = 0.0f;   // CONST
    int ipow = 0;        // POWI: literal integer exponent, |ipow| <= 16 (power rule P1)
    uint8_t deps = 0;    // bit0 x, bit1 y, bit2 z
};

struct Program {
    std::string equation;     // as given
    std::vector<Node> nodes;  // topological (operands before users), CSE'd, constants folded
    int root = -1;
    bool uses_general_pow = false;  // some ^ is not a literal-integer power: bit parity with the
                                    // reference is then tolerance-only
};

enum class CompileStatus { OK, PARSE, EVAL };

// Symbolic execution of evaluator.cpp:53-107 / :22-48.
CompileStatus compile(const std::string& eq, Program& out, std::string& err, unsigned ext = 0);

// HIP source of `__device__ __forceinline__ float mc_f(float x, float y, float z)`.
std::string emit_hip(const Program& p, const char* fname = "mc_f");

// HIP source of `mc_f_iv(xl,xh,yl,yh,zl,zh, lo, hi)`: an enclosure [lo,hi] of every value mc_f
// COMPUTES for points of the box.  Each IEEE round-to-nearest operation is monotone, so interval
// arithmetic whose endpoints are rounded the same way encloses the computed (not just the real)
// values exactly -- no widening needed.  Only valid for programs that pass finite_on_domain()
// (no NaN / inf can arise); the classify kernel uses it to prove whole rows of cells uniform
// without sampling them.  Returns "" when the program uses an operation it cannot bound.
std::string emit_hip_interval(const Program& p, const char* fname = "mc_f_iv");

// The same enclosure split for the classify walk, whose boxes change only in y from one step to the next:
// `#define MC_IV_NY n`, `mc_f_iv_y(yl, yh, Y)` (the expensive sub-expressions of y alone: sin / cos, divisions, powers
// above 2 -- evaluated once per tile ROW) and `mc_f_iv_rest(xl,xh,yl,yh,zl,zh, Y, lo, hi)` (everything else, from Y).
// Composition of the two == emit_hip_interval's function, operation for operation.  "" when no sub-expression of y alone
// costs at least min_cost vector instructions to bound.
std::string emit_hip_interval_staged(const Program& p, int min_cost = 40);

// f with its expensive one-variable sub-expressions taken out (see mc_expr.cpp): `#define MC_TAB`, `mc_f_ux / uy / uz`
// (the sub-expressions of one variable) and `mc_f_t` (f from coordinates and those values); "" when there is none.
std::string emit_hip_tabulated(const Program& p, int min_cost = 15);

// Diagnostic host interpretation of the DAG (same float ops; see mc_hip.h mc_expr_debug_eval_host).
float eval_host(const Program& p, float x, float y, float z);

// Interval bound: true when |every intermediate value of f| < 1e30 for all |x|,|y|,|z| <= radius,
// using only +, -, *, negation, literal non-negative integer powers and division by non-zero
// constants.  Then no sample can be inf or NaN (no overflow, hence no inf-inf / 0*inf), and the
// kernels may drop their NaN bookkeeping (MC_FINITE).
bool finite_on_domain(const Program& p, double radius);

// Rough count of vector instructions one evaluation of f costs on the device (add / sub / mul 1, IEEE division 10,
// literal integer powers a few double multiplies, sin / cos ~22, general pow ~300).  mc_runtime picks the emit kernel by
// it: below a measured threshold recomputing a vertex for each of its ~6 output copies is cheaper than sharing it.
int vector_op_cost(const Program& p);

// ---- the DAG as a PROGRAM for the interpreter build of the kernels (mc_kernels.hip, MC_INTERP: what an equation's first
// sweeps run on while hiprtc specialises the kernels for it).  Must match McInterpProg there.
#define MC_INTERP_MAXI 120   // operations
#define MC_INTERP_REGS 16    // live values
#define MC_INTERP_CONSTS 64
struct InterpProg {
    uint32_t n, root, cons_op;
    float cons_rhs;
    uint32_t code[MC_INTERP_MAXI];  // op | dst << 8 | a << 16 | b << 24; operand byte: 0..63 register, 64 / 65 / 66 = x / y / z, 128 + k =
                                    // constant k; ops: 0 + 1 - 2 * 3 / 4 pow 5 neg 6 sin 7 cos 8 square 9 literal-integer power (b = n + 32)
    float cval[MC_INTERP_CONSTS];
};
static_assert(sizeof(InterpProg) == 16 + 4 * MC_INTERP_MAXI + 4 * MC_INTERP_CONSTS, "InterpProg layout");
// One word per operation in topological order (the order emit_hip's lines have), values in registers allocated by
// liveness, constants in a table.  false: the program does not fit the tables.
bool interp_program(const Program& p, InterpProg& out);
// The device interpreter's walk, on the host (diagnostic: mc_hip.h mc_expr_debug_interp_host).
float interp_run_host(const InterpProg& P, float x, float y, float z);

// Power rule P1 (shared by constant folding, eval_host and -- as generated code -- the device).
float pow_literal_int(float a, int n);
float pow_general(float a, float b);

}  // namespace mc
