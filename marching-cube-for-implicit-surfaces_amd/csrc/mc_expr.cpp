// mc_expr.cpp -- see mc_expr.hpp.  Compile with -ffp-contract=off.
#include "mc_expr.hpp"

#include "mc_trig.h"
#include <functional>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>

namespace mc {

// ---------------------------------------------------------------- tokenizer
// Character classes: evaluator.h:37-41.
static bool is_op(char c) { return c == '+' || c == '-' || c == '*' || c == '/' || c == '^'; }
static bool is_num(char c) { return (c >= '0' && c <= '9') || c == '.'; }
static bool is_var(char c) { return (c >= 'x' && c <= 'z') || (c >= 'X' && c <= 'Z'); }

bool tokenize(const std::string& eq_in, std::vector<Token>& out, unsigned ext) {
    out.clear();
    if (eq_in.empty()) return false;  // evaluator.cpp:141
    std::string s;
    s.reserve(eq_in.size());
    for (char c : eq_in)
        if (c != ' ') s.push_back(c);  // :147, only U+0020 is stripped

    enum Last { L_NONE, L_OP, L_NUM, L_VAR, L_BO, L_BC } last = L_NONE;
    bool neg = false;
    int depth = 0;
    auto implicit_mul = [&]() { out.push_back({TokType::OP, '*', 0.0f}); };

    for (size_t i = 0; i < s.size(); ++i) {
        const char c = s[i];
        // :162 -- a '-' is a SIGN when it is the first character or the previous
        // character (not token) is '(' or an operator.
        if (c == '-' && (i == 0 || s[i - 1] == '(' || is_op(s[i - 1]))) {
            if (neg) return false;  // :163 "--"
            neg = true;
            out.push_back({TokType::NEG, 'N', 0.0f});
            continue;  // :167 -- `last` deliberately unchanged
        }
        if ((ext & EXT_TRIG) && i + 3 < s.size() && s[i + 3] == '(' &&
            (((c == 's' || c == 'S') && (s[i + 1] == 'i' || s[i + 1] == 'I') && (s[i + 2] == 'n' || s[i + 2] == 'N')) ||
             ((c == 'c' || c == 'C') && (s[i + 1] == 'o' || s[i + 1] == 'O') && (s[i + 2] == 's' || s[i + 2] == 'S')))) {
            // extension E1: a function name fused with its '(' -- placed and counted like a '('
            if (last == L_VAR || last == L_NUM || last == L_BC) implicit_mul();
            out.push_back({TokType::FUNC, (c == 's' || c == 'S') ? 's' : 'c', 0.0f});
            out.push_back({TokType::BRAC_O, '(', 0.0f});
            ++depth;
            last = L_BO;
            i += 3;
        } else if (c == '(') {  // :170-178
            if (last == L_VAR || last == L_NUM || last == L_BC) implicit_mul();
            out.push_back({TokType::BRAC_O, c, 0.0f});
            ++depth;
            last = L_BO;
        } else if (c == ')') {  // :179-185
            if (neg || last == L_BO || last == L_OP) return false;
            if (depth == 0) return false;
            out.push_back({TokType::BRAC_C, c, 0.0f});
            --depth;
            last = L_BC;
        } else if (is_op(c)) {  // :186-191 (a trailing operator IS accepted)
            if (neg || last == L_BO || last == L_OP || last == L_NONE) return false;
            out.push_back({TokType::OP, c, 0.0f});
            last = L_OP;
        } else if (is_num(c)) {  // :192-214
            if (last == L_VAR || last == L_BC) implicit_mul();
            bool dot = (c == '.');
            size_t j = i;
            while (j + 1 < s.size() && is_num(s[j + 1])) {
                if (s[j + 1] == '.') {
                    if (dot) return false;
                    dot = true;
                }
                ++j;
            }
            const std::string lit = s.substr(i, j - i + 1);
            if (lit == ".") return false;  // :210
            out.push_back({TokType::NUM, '0', std::strtof(lit.c_str(), nullptr)});
            i = j;
            last = L_NUM;
        } else if (is_var(c)) {  // :215-223
            if (last == L_VAR || last == L_NUM || last == L_BC) implicit_mul();
            out.push_back({TokType::VAR, c, 0.0f});
            last = L_VAR;
        } else {
            return false;  // :224
        }
        neg = false;  // :227
    }
    return depth == 0;  // :231
}

// ------------------------------------------------------------- power rule P1
float pow_literal_int(float a, int n) {
    if (n == 0) return 1.0f;
    const int m = n < 0 ? -n : n;
    volatile double p = (double)a;
    volatile double r = p;
    for (int i = 2; i <= m; ++i) r = r * p;
    if (n < 0) r = 1.0 / r;
    return (float)r;
}
float pow_general(float a, float b) { return (float)std::pow((double)a, (double)b); }

static bool literal_int_exponent(float b, int& n) {
    if (!(std::fabs(b) <= 16.0f)) return false;
    if (b != std::floor(b)) return false;
    n = (int)b;
    return true;
}

// ------------------------------------------------------------------ builder
namespace {

struct Builder {
    std::vector<Node> nodes;
    std::map<std::tuple<int, int, int, uint32_t, int>, int> cse;
    bool general_pow = false;

    static uint32_t bits(float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    }
    int intern(const Node& n) {
        auto key = std::make_tuple((int)n.op, n.a, n.b, bits(n.cval), n.ipow);
        auto it = cse.find(key);
        if (it != cse.end()) return it->second;
        nodes.push_back(n);
        cse.emplace(key, (int)nodes.size() - 1);
        return (int)nodes.size() - 1;
    }
    int constant(float v) {
        Node n;
        n.op = NodeOp::CONST;
        n.cval = v;
        return intern(n);
    }
    int var(char c) {
        Node n;
        if (c == 'x' || c == 'X') { n.op = NodeOp::VARX; n.deps = 1; }
        else if (c == 'y' || c == 'Y') { n.op = NodeOp::VARY; n.deps = 2; }
        else { n.op = NodeOp::VARZ; n.deps = 4; }
        return intern(n);
    }
    bool is_const(int id) const { return nodes[id].op == NodeOp::CONST; }

    int neg(int a) {
        if (is_const(a)) return constant(-nodes[a].cval);
        Node n;
        n.op = NodeOp::NEG;
        n.a = a;
        n.deps = nodes[a].deps;
        return intern(n);
    }
    int trig(char f, int a) {  // extension E1
        if (is_const(a)) return constant(f == 's' ? mc_sinf(nodes[a].cval) : mc_cosf(nodes[a].cval));
        Node n;
        n.op = f == 's' ? NodeOp::SIN : NodeOp::COS;
        n.a = a;
        n.deps = nodes[a].deps;
        return intern(n);
    }
    // value = a op b  (evaluator.cpp:127-136)
    int binary(char op, int a, int b) {
        if (op == '^') {
            int e;
            if (is_const(b) && literal_int_exponent(nodes[b].cval, e)) {
                if (e == 0) return constant(1.0f);  // powf(x, 0) == 1 for every x, NaN included
                if (e == 1) return a;               // powf(x, 1) == x
                if (is_const(a)) return constant(pow_literal_int(nodes[a].cval, e));
                Node n;
                n.op = NodeOp::POWI;
                n.a = a;
                n.ipow = e;
                n.deps = nodes[a].deps;
                return intern(n);
            }
            if (is_const(a) && is_const(b)) return constant(pow_general(nodes[a].cval, nodes[b].cval));
            general_pow = true;
            Node n;
            n.op = NodeOp::POW;
            n.a = a;
            n.b = b;
            n.deps = nodes[a].deps | nodes[b].deps;
            return intern(n);
        }
        if (is_const(a) && is_const(b)) {
            volatile float x = nodes[a].cval, y = nodes[b].cval, r = 0.0f;
            switch (op) {
            case '+': r = x + y; break;
            case '-': r = x - y; break;
            case '*': r = x * y; break;
            default: r = x / y; break;
            }
            return constant(r);
        }
        Node n;
        n.op = op == '+' ? NodeOp::ADD : op == '-' ? NodeOp::SUB : op == '*' ? NodeOp::MUL : NodeOp::DIV;
        n.a = a;
        n.b = b;
        // IEEE add / mul are commutative bit for bit (NaN payloads aside): canonical order helps CSE
        if ((n.op == NodeOp::ADD || n.op == NodeOp::MUL) && n.a > n.b) std::swap(n.a, n.b);
        n.deps = nodes[a].deps | nodes[b].deps;
        return intern(n);
    }
};

int precedence(char c) {  // evaluator.cpp:111-124
    switch (c) {
    case 'N': return 4;
    case '^': return 3;
    case '/': case '*': return 2;
    case '+': case '-': return 1;
    default: return 0;
    }
}

struct SymStacks {
    std::vector<char> ops;
    std::vector<int> vals;
};

// evaluator.cpp:22-48 on node ids.  false = the reference would read below a stack.
bool evaluate_op(Builder& B, SymStacks& S) {
    if (S.ops.empty()) return false;
    const char op = S.ops.back();
    S.ops.pop_back();
    if (is_op(op)) {
        if (S.vals.empty()) return false;
        const int val1 = S.vals.back();
        S.vals.pop_back();
        if (!S.ops.empty() && precedence(S.ops.back()) > precedence(op))  // :32-37, ONE level
            if (!evaluate_op(B, S)) return false;
        if (S.vals.empty()) return false;
        const int val2 = S.vals.back();
        S.vals.pop_back();
        S.vals.push_back(B.binary(op, val2, val1));  // :39
        return true;
    }
    if (op == 'N') {  // :42-46
        if (S.vals.empty()) return false;
        S.vals.back() = B.neg(S.vals.back());
        return true;
    }
    return false;  // :47 throw
}

}  // namespace

CompileStatus compile(const std::string& eq, Program& out, std::string& err, unsigned ext) {
    std::vector<Token> toks;
    if (!tokenize(eq, toks, ext)) {
        err = "equation rejected by the tokenizer (Evaluator::set_equation would return false)";
        return CompileStatus::PARSE;
    }
    Builder B;
    SymStacks S;
    const char* underflow =
        "the reference's Evaluator::evaluate would read below its operand/operator stack for this "
        "equation (undefined behaviour there); refused";
    for (const Token& t : toks) {  // evaluator.cpp:62-98
        switch (t.type) {
        case TokType::NEG: S.ops.push_back('N'); break;
        case TokType::VAR: S.vals.push_back(B.var(t.ch)); break;
        case TokType::NUM: S.vals.push_back(B.constant(t.num)); break;
        case TokType::BRAC_O: S.ops.push_back('('); break;
        case TokType::BRAC_C:
            for (;;) {
                if (S.ops.empty()) { err = underflow; return CompileStatus::EVAL; }
                if (S.ops.back() == '(') break;
                if (!evaluate_op(B, S)) { err = underflow; return CompileStatus::EVAL; }
            }
            S.ops.pop_back();
            if (!S.ops.empty() && (S.ops.back() == 'S' || S.ops.back() == 'C')) {  // E1: the bracket was a function call
                const char f = S.ops.back() == 'S' ? 's' : 'c';
                S.ops.pop_back();
                if (S.vals.empty()) { err = underflow; return CompileStatus::EVAL; }
                S.vals.back() = B.trig(f, S.vals.back());
            }
            break;
        case TokType::FUNC: S.ops.push_back(t.ch == 's' ? 'S' : 'C'); break;
        case TokType::OP: S.ops.push_back(t.ch); break;
        }
    }
    while (!S.ops.empty())  // :100-102
        if (!evaluate_op(B, S)) { err = underflow; return CompileStatus::EVAL; }
    if (S.vals.empty()) { err = underflow; return CompileStatus::EVAL; }
    const int root = S.vals.back();  // :105 (anything left below it is never read)

    // keep only what the root reaches, preserving topological order
    std::vector<char> live(B.nodes.size(), 0);
    live[root] = 1;
    for (int i = (int)B.nodes.size() - 1; i >= 0; --i)
        if (live[i]) {
            if (B.nodes[i].a >= 0) live[B.nodes[i].a] = 1;
            if (B.nodes[i].b >= 0) live[B.nodes[i].b] = 1;
        }
    std::vector<int> remap(B.nodes.size(), -1);
    out = Program();
    out.equation = eq;
    for (size_t i = 0; i < B.nodes.size(); ++i)
        if (live[i]) {
            Node n = B.nodes[i];
            if (n.a >= 0) n.a = remap[n.a];
            if (n.b >= 0) n.b = remap[n.b];
            remap[i] = (int)out.nodes.size();
            out.nodes.push_back(n);
        }
    out.root = remap[root];
    out.uses_general_pow = false;
    for (const Node& n : out.nodes)
        if (n.op == NodeOp::POW) out.uses_general_pow = true;
    return CompileStatus::OK;
}

// ------------------------------------------------------------------ codegen
namespace {
std::string node_name(const Program& p, int id) {
    const Node& n = p.nodes[id];
    if (n.op == NodeOp::VARX) return "x";
    if (n.op == NodeOp::VARY) return "y";
    if (n.op == NodeOp::VARZ) return "z";
    return "t" + std::to_string(id);
}
// `const float t<i> = ...;` for node i ("" for a variable)
std::string node_line(const Program& p, size_t i) {
    char buf[256];
    const Node& n = p.nodes[i];
    auto name = [&](int id) { return node_name(p, id); };
    std::string rhs;
    switch (n.op) {
    case NodeOp::VARX: case NodeOp::VARY: case NodeOp::VARZ: return std::string();
    case NodeOp::CONST: {
        uint32_t u;
        std::memcpy(&u, &n.cval, 4);
        std::snprintf(buf, sizeof buf, "__uint_as_float(0x%08xu) /* %.9g */", u, (double)n.cval);
        rhs = buf;
        break;
    }
    case NodeOp::ADD: rhs = name(n.a) + " + " + name(n.b); break;
    case NodeOp::SUB: rhs = name(n.a) + " - " + name(n.b); break;
    case NodeOp::MUL: rhs = name(n.a) + " * " + name(n.b); break;
    case NodeOp::DIV: rhs = name(n.a) + " / " + name(n.b); break;
    case NodeOp::NEG: rhs = "-" + name(n.a); break;
    case NodeOp::SIN: rhs = "mc_sinf(" + name(n.a) + ")"; break;
    case NodeOp::COS: rhs = "mc_cosf(" + name(n.a) + ")"; break;
    case NodeOp::POW: rhs = "mc_pow_general(" + name(n.a) + ", " + name(n.b) + ")"; break;
    case NodeOp::POWI:
        if (n.ipow == 2) rhs = name(n.a) + " * " + name(n.a);
        else rhs = "mc_pow_int<" + std::to_string(n.ipow) + ">(" + name(n.a) + ")";
        break;
    }
    return "    const float " + name((int)i) + " = " + rhs + ";\n";
}
// vector instructions one evaluation of node i costs (operands not included)
int node_cost(const Node& n) {
    switch (n.op) {
    case NodeOp::ADD: case NodeOp::SUB: case NodeOp::MUL: case NodeOp::NEG: return 1;
    case NodeOp::DIV: return 10;
    case NodeOp::POWI: return n.ipow == 2 ? 1 : 6 + 4 * (n.ipow < 0 ? -n.ipow : n.ipow) + (n.ipow < 0 ? 14 : 0);
    case NodeOp::POW: return 300;
    case NodeOp::SIN: case NodeOp::COS: return 22;
    default: return 0;
    }
}
}  // namespace

std::string emit_hip(const Program& p, const char* fname) {
    std::string s;
    s += "// generated from: ";
    for (char c : p.equation) s.push_back((c == '\n' || c == '\r' || c == '\\') ? ' ' : c);
    s += std::string("\n__device__ __forceinline__ float ") + fname + "(float x, float y, float z) {\n";
    s += "    (void)x; (void)y; (void)z;\n";
    for (size_t i = 0; i < p.nodes.size(); ++i) s += node_line(p, i);
    s += "    return " + node_name(p, p.root) + ";\n}\n";
    return s;
}

// f with its expensive ONE-VARIABLE sub-expressions taken out: `mc_f_ux(x, U)`, `mc_f_uy`, `mc_f_uz` evaluate the
// maximal sub-expressions that depend on that variable alone and cost at least min_cost vector instructions (sin / cos,
// divisions, higher powers), `mc_f_t(x, y, z, UX, UY, UZ)` is f from the coordinates and those values -- composed, the
// very operations of mc_f in the same order.  On a lattice such a sub-expression has n1 + 1 distinct values per axis
// however many samples the sweep takes, so mc_runtime tabulates them once per (equation, lattice) -- at the lattice
// coordinates and at coordinate +- h for the normals -- and the kernels read instead of evaluating.  "" when f has
// no such sub-expression or is one itself.  Defines MC_TAB, MC_TAB_NX / NY / NZ (sub-expressions per variable, >= 1 as
// array sizes) and MC_TAB_HAS_X / Y / Z.
std::string emit_hip_tabulated(const Program& p, int min_cost) {
    const size_t n = p.nodes.size();
    std::vector<char> used_outside(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const Node& nd = p.nodes[i];
        const bool uni = nd.deps == 1 || nd.deps == 2 || nd.deps == 4;
        for (int o : {nd.a, nd.b})
            if (o >= 0 && !(uni && p.nodes[o].deps == nd.deps)) used_outside[o] = 1;  // a user that depends on more (or on less: never) than the operand
    }
    const Node& rt = p.nodes[p.root];
    if (rt.deps == 0 || rt.deps == 1 || rt.deps == 2 || rt.deps == 4) return std::string();  // f of one variable: nothing to combine
    std::vector<int> hoisted[3];
    std::vector<char> stage[3], in;
    for (auto& st : stage) st.assign(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const Node& nd = p.nodes[i];
        const int v = nd.deps == 1 ? 0 : nd.deps == 2 ? 1 : nd.deps == 4 ? 2 : -1;
        if (v < 0 || !used_outside[i] || nd.op == NodeOp::VARX || nd.op == NodeOp::VARY || nd.op == NodeOp::VARZ) continue;
        // cost of the node's cone (every node once)
        in.assign(n, 0);
        in[i] = 1;
        int c = 0;
        for (size_t k = i + 1; k-- > 0;) {
            if (!in[k]) continue;
            c += node_cost(p.nodes[k]);
            if (p.nodes[k].a >= 0) in[p.nodes[k].a] = 1;
            if (p.nodes[k].b >= 0) in[p.nodes[k].b] = 1;
        }
        if (c < min_cost) continue;
        hoisted[v].push_back((int)i);
        for (size_t k = 0; k < n; ++k) stage[v][k] = stage[v][k] | in[k];
    }
    if (hoisted[0].empty() && hoisted[1].empty() && hoisted[2].empty()) return std::string();
    for (const auto& h : hoisted)
        if (h.size() > 6) return std::string();
    const char* var = "xyz";
    const char* VAR = "XYZ";
    std::string s = "#define MC_TAB 1\n";
    for (int v = 0; v < 3; ++v) {
        s += std::string("#define MC_TAB_N") + VAR[v] + " " + std::to_string(hoisted[v].empty() ? 1 : hoisted[v].size()) + "\n";
        if (!hoisted[v].empty()) s += std::string("#define MC_TAB_HAS_") + VAR[v] + " 1\n";
    }
    for (int v = 0; v < 3; ++v) {
        s += std::string("__device__ __forceinline__ void mc_f_u") + var[v] + "(float " + var[v] + ", float (&U)[MC_TAB_N" + VAR[v] + "]) {\n";
        s += std::string("    (void)") + var[v] + "; U[0] = 0.0f;\n";
        for (size_t i = 0; i < n; ++i)
            if (stage[v][i]) s += node_line(p, i);
        for (size_t k = 0; k < hoisted[v].size(); ++k) s += "    U[" + std::to_string(k) + "] = t" + std::to_string(hoisted[v][k]) + ";\n";
        s += "}\n";
    }
    std::vector<char> need(n, 0), is_h(n, 0);
    for (const auto& h : hoisted)
        for (int i : h) is_h[i] = 1;
    need[p.root] = 1;
    for (size_t i = n; i-- > 0;) {
        if (!need[i] || is_h[i]) continue;
        if (p.nodes[i].a >= 0) need[p.nodes[i].a] = 1;
        if (p.nodes[i].b >= 0) need[p.nodes[i].b] = 1;
    }
    // MC_TAB_SYM: the three axes' lists hold the SAME functions of their variable (gyroid: {sin, cos} each), possibly in
    // another order: mc_f_ux(t, C) then serves any axis -- mc_tab_sym_y / _z put C into that axis' order -- and mc_emit can
    // compute the vertices of all three edge directions in one pass instead of one pass per axis (mc_emit_sym).
    {
        std::function<std::string(int)> canon = [&](int i) -> std::string {
            const Node& nd = p.nodes[i];
            if (nd.op == NodeOp::VARX || nd.op == NodeOp::VARY || nd.op == NodeOp::VARZ) return "v";
            if (nd.op == NodeOp::CONST) {
                uint32_t u;
                memcpy(&u, &nd.cval, 4);
                char b[16];
                snprintf(b, sizeof b, "#%08x", u);
                return b;
            }
            std::string r = "(" + std::to_string((int)nd.op) + ":" + std::to_string(nd.ipow);
            if (nd.a >= 0) r += "," + canon(nd.a);
            if (nd.b >= 0) r += "," + canon(nd.b);
            return r + ")";
        };
        bool sym = !hoisted[0].empty() && hoisted[0].size() == hoisted[1].size() && hoisted[0].size() == hoisted[2].size();
        std::vector<int> perm[3];
        if (sym) {
            std::vector<std::string> c0;
            for (int i : hoisted[0]) c0.push_back(canon(i));
            for (int v = 1; v < 3 && sym; ++v) {
                std::vector<char> used(c0.size(), 0);
                for (int i : hoisted[v]) {
                    const std::string ci = canon(i);
                    int hit = -1;
                    for (size_t j = 0; j < c0.size(); ++j)
                        if (!used[j] && c0[j] == ci) {
                            hit = (int)j;
                            break;
                        }
                    if (hit < 0) {
                        sym = false;
                        break;
                    }
                    used[hit] = 1;
                    perm[v].push_back(hit);
                }
            }
        }
        if (sym) {
            s += "#define MC_TAB_SYM 1\n";
            for (int v = 1; v < 3; ++v) {
                s += std::string("__device__ __forceinline__ void mc_tab_sym_") + var[v] + "(const float (&C)[MC_TAB_NX], float (&U)[MC_TAB_N" + VAR[v] + "]) {\n";
                for (size_t k = 0; k < perm[v].size(); ++k) s += "    U[" + std::to_string(k) + "] = C[" + std::to_string(perm[v][k]) + "];\n";
                s += "}\n";
            }
        }
    }
    s += "__device__ __forceinline__ float mc_f_t(float x, float y, float z, const float (&UX)[MC_TAB_NX], const float (&UY)[MC_TAB_NY], "
         "const float (&UZ)[MC_TAB_NZ]) {\n    (void)x; (void)y; (void)z; (void)UX; (void)UY; (void)UZ;\n";
    for (int v = 0; v < 3; ++v)
        for (size_t k = 0; k < hoisted[v].size(); ++k)
            s += "    const float t" + std::to_string(hoisted[v][k]) + " = U" + VAR[v] + "[" + std::to_string(k) + "];\n";
    for (size_t i = 0; i < n; ++i)
        if (need[i] && !is_h[i]) s += node_line(p, i);
    s += "    return " + node_name(p, p.root) + ";\n}\n";
    return s;
}

// Interval analysis of the DAG over the whole domain box [-radius, radius]^3, in double with a
// relative widening that covers the float roundings.  false as soon as a value could be inf /
// NaN: a bound >= 1e30, a division or negative power whose base interval does not stay clear of
// zero, or a general (non literal-integer) power.
bool finite_on_domain(const Program& p, double radius) {
    const double LIMIT = 1e30, TINY = 1e-30;
    std::vector<double> lo(p.nodes.size(), 0.0), hi(p.nodes.size(), 0.0);
    auto mulrange = [](double al, double ah, double bl, double bh, double& l, double& h) {
        const double c[4] = {al * bl, al * bh, ah * bl, ah * bh};
        l = std::min(std::min(c[0], c[1]), std::min(c[2], c[3]));
        h = std::max(std::max(c[0], c[1]), std::max(c[2], c[3]));
    };
    auto powrange = [](double al, double ah, int n, double& l, double& h) {  // n >= 1
        if (n % 2 == 0) {
            const double m = std::max(std::fabs(al), std::fabs(ah));
            const double k = (al <= 0.0 && ah >= 0.0) ? 0.0 : std::min(std::fabs(al), std::fabs(ah));
            l = std::pow(k, n);
            h = std::pow(m, n);
        } else {
            l = std::pow(al, n);
            h = std::pow(ah, n);
        }
    };
    for (size_t i = 0; i < p.nodes.size(); ++i) {
        const Node& n = p.nodes[i];
        double l, h;
        switch (n.op) {
        case NodeOp::CONST: l = h = (double)n.cval; break;
        case NodeOp::VARX: case NodeOp::VARY: case NodeOp::VARZ: l = -radius; h = radius; break;
        case NodeOp::ADD: l = lo[n.a] + lo[n.b]; h = hi[n.a] + hi[n.b]; break;
        case NodeOp::SUB: l = lo[n.a] - hi[n.b]; h = hi[n.a] - lo[n.b]; break;
        case NodeOp::NEG: l = -hi[n.a]; h = -lo[n.a]; break;
        case NodeOp::MUL:
            if (n.a == n.b) powrange(lo[n.a], hi[n.a], 2, l, h);
            else mulrange(lo[n.a], hi[n.a], lo[n.b], hi[n.b], l, h);
            break;
        case NodeOp::DIV: {
            const double bl = lo[n.b], bh = hi[n.b];
            if (!(bl > TINY || bh < -TINY)) return false;  // divisor may reach 0
            mulrange(lo[n.a], hi[n.a], 1.0 / bh, 1.0 / bl, l, h);
            break;
        }
        case NodeOp::POWI: {
            const int e = n.ipow < 0 ? -n.ipow : n.ipow;
            double pl, ph;
            powrange(lo[n.a], hi[n.a], e, pl, ph);
            if (n.ipow > 0) { l = pl; h = ph; }
            else {
                if (!(pl > TINY || ph < -TINY)) return false;
                l = 1.0 / ph;
                h = 1.0 / pl;
            }
            break;
        }
        case NodeOp::SIN: case NodeOp::COS:
            // mc_trig.h: finite (and within [-1, 1]) exactly when |argument| < 8192
            if (!(lo[n.a] > -8000.0 && hi[n.a] < 8000.0)) return false;
            l = -1.0;
            h = 1.0;
            break;
        default: return false;  // general pow
        }
        if (!(l == l) || !(h == h)) return false;
        // widen: float roundings of the operands / result, and keep clear of the limits
        const double w = 1e-5 * std::max(std::fabs(l), std::fabs(h)) + 1e-35;
        l -= w;
        h += w;
        if (!(l > -LIMIT && h < LIMIT)) return false;
        lo[i] = l;
        hi[i] = h;
    }
    return true;
}

namespace {
// Interval code of ONE node (appended to s): `const float l<i> = ..., h<i> = ...;` from its operands' l / h (variables are
// xl/xh, yl/yh, zl/zh).  false: the node cannot be bounded (general pow).
bool iv_node(const Program& p, size_t i, std::string& s) {
    char buf[512];
    auto L = [&](int id) -> std::string {
        const Node& n = p.nodes[id];
        if (n.op == NodeOp::VARX) return "xl";
        if (n.op == NodeOp::VARY) return "yl";
        if (n.op == NodeOp::VARZ) return "zl";
        std::snprintf(buf, sizeof buf, "l%d", id);
        return buf;
    };
    auto H = [&](int id) -> std::string {
        const Node& n = p.nodes[id];
        if (n.op == NodeOp::VARX) return "xh";
        if (n.op == NodeOp::VARY) return "yh";
        if (n.op == NodeOp::VARZ) return "zh";
        std::snprintf(buf, sizeof buf, "h%d", id);
        return buf;
    };
    auto cst = [&](float v) -> std::string {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        std::snprintf(buf, sizeof buf, "__uint_as_float(0x%08xu)", u);
        return buf;
    };
    const Node& n = p.nodes[i];
    const std::string l = L((int)i), h = H((int)i);
    std::string lo, hi, pre;
    switch (n.op) {
    case NodeOp::VARX: case NodeOp::VARY: case NodeOp::VARZ: return true;
    case NodeOp::CONST: lo = hi = cst(n.cval); break;
    case NodeOp::ADD: lo = L(n.a) + " + " + L(n.b); hi = H(n.a) + " + " + H(n.b); break;
    case NodeOp::SUB: lo = L(n.a) + " - " + H(n.b); hi = H(n.a) + " - " + L(n.b); break;
    case NodeOp::NEG: lo = "-" + H(n.a); hi = "-" + L(n.a); break;
    case NodeOp::MUL: {
        const Node& A = p.nodes[n.a];
        const Node& B = p.nodes[n.b];
        if (n.a == n.b) {  // square: |x| in [mn, mx]
            pre = "    const float q" + std::to_string(i) + "a = __builtin_fabsf(" + L(n.a) + "), q" + std::to_string(i) +
                  "b = __builtin_fabsf(" + H(n.a) + ");\n    const float q" + std::to_string(i) + "m = __builtin_fmaxf(q" +
                  std::to_string(i) + "a, q" + std::to_string(i) + "b), q" + std::to_string(i) + "n = (" + L(n.a) +
                  " <= 0.0f && " + H(n.a) + " >= 0.0f) ? 0.0f : __builtin_fminf(q" + std::to_string(i) + "a, q" +
                  std::to_string(i) + "b);\n";
            lo = "q" + std::to_string(i) + "n * q" + std::to_string(i) + "n";
            hi = "q" + std::to_string(i) + "m * q" + std::to_string(i) + "m";
        } else if (A.op == NodeOp::CONST || B.op == NodeOp::CONST) {
            const bool a_const = A.op == NodeOp::CONST;
            const float c = a_const ? A.cval : B.cval;
            const int v = a_const ? n.b : n.a;
            // keep the operand order of mc_f (a * b) so both round identically
            auto prod = [&](const std::string& x) { return a_const ? cst(c) + " * " + x : x + " * " + cst(c); };
            if (c >= 0.0f) { lo = prod(L(v)); hi = prod(H(v)); }
            else { lo = prod(H(v)); hi = prod(L(v)); }
        } else {
            const std::string t = "p" + std::to_string(i);
            pre = "    const float " + t + "a = " + L(n.a) + " * " + L(n.b) + ", " + t + "b = " + L(n.a) + " * " + H(n.b) + ", " +
                  t + "c = " + H(n.a) + " * " + L(n.b) + ", " + t + "d = " + H(n.a) + " * " + H(n.b) + ";\n";
            lo = "__builtin_fminf(__builtin_fminf(" + t + "a, " + t + "b), __builtin_fminf(" + t + "c, " + t + "d))";
            hi = "__builtin_fmaxf(__builtin_fmaxf(" + t + "a, " + t + "b), __builtin_fmaxf(" + t + "c, " + t + "d))";
        }
        break;
    }
    case NodeOp::DIV: {
        const Node& B = p.nodes[n.b];
        if (B.op == NodeOp::CONST && B.cval != 0.0f) {
            if (B.cval > 0.0f) { lo = L(n.a) + " / " + cst(B.cval); hi = H(n.a) + " / " + cst(B.cval); }
            else { lo = H(n.a) + " / " + cst(B.cval); hi = L(n.a) + " / " + cst(B.cval); }
        } else {
            // finite_on_domain() proved the divisor keeps one sign (and stays away from 0) on the
            // whole domain, so the quotient is monotone in each operand: extremes at the corners
            const std::string t = "d" + std::to_string(i);
            pre = "    const float " + t + "a = " + L(n.a) + " / " + L(n.b) + ", " + t + "b = " + L(n.a) + " / " + H(n.b) + ", " +
                  t + "c = " + H(n.a) + " / " + L(n.b) + ", " + t + "d = " + H(n.a) + " / " + H(n.b) + ";\n";
            lo = "__builtin_fminf(__builtin_fminf(" + t + "a, " + t + "b), __builtin_fminf(" + t + "c, " + t + "d))";
            hi = "__builtin_fmaxf(__builtin_fmaxf(" + t + "a, " + t + "b), __builtin_fmaxf(" + t + "c, " + t + "d))";
        }
        break;
    }
    case NodeOp::POWI: {
        if (n.ipow == 0 || n.ipow == 1) return false;
        std::snprintf(buf, sizeof buf, "mc_pow_int<%d>", n.ipow);
        const std::string pw = buf;
        if (n.ipow < 0) {
            // base keeps one sign and stays away from 0 (finite_on_domain): x^-n decreases in |x|
            if ((-n.ipow) % 2 == 0) {
                pre = "    const float q" + std::to_string(i) + "a = __builtin_fabsf(" + L(n.a) + "), q" + std::to_string(i) +
                      "b = __builtin_fabsf(" + H(n.a) + ");\n";
                lo = pw + "(__builtin_fmaxf(q" + std::to_string(i) + "a, q" + std::to_string(i) + "b))";
                hi = pw + "(__builtin_fminf(q" + std::to_string(i) + "a, q" + std::to_string(i) + "b))";
            } else {
                lo = pw + "(" + H(n.a) + ")";
                hi = pw + "(" + L(n.a) + ")";
            }
        } else if (n.ipow % 2 == 0) {
            pre = "    const float q" + std::to_string(i) + "a = __builtin_fabsf(" + L(n.a) + "), q" + std::to_string(i) +
                  "b = __builtin_fabsf(" + H(n.a) + ");\n    const float q" + std::to_string(i) + "m = __builtin_fmaxf(q" +
                  std::to_string(i) + "a, q" + std::to_string(i) + "b), q" + std::to_string(i) + "n = (" + L(n.a) +
                  " <= 0.0f && " + H(n.a) + " >= 0.0f) ? 0.0f : __builtin_fminf(q" + std::to_string(i) + "a, q" +
                  std::to_string(i) + "b);\n";
            if (n.ipow == 2) {
                lo = "q" + std::to_string(i) + "n * q" + std::to_string(i) + "n";
                hi = "q" + std::to_string(i) + "m * q" + std::to_string(i) + "m";
            } else {
                lo = pw + "(q" + std::to_string(i) + "n)";
                hi = pw + "(q" + std::to_string(i) + "m)";
            }
        } else {  // odd power: monotone increasing, also as computed (product chain of same-sign factors)
            lo = pw + "(" + L(n.a) + ")";
            hi = pw + "(" + H(n.a) + ")";
        }
        break;
    }
    case NodeOp::SIN: case NodeOp::COS:
        // mc_sin_iv / mc_cos_iv (mc_kernels.hip): endpoints when no extremum can lie inside, else +-1
        s += "    float " + l + ", " + h + ";\n    " + (n.op == NodeOp::SIN ? "mc_sin_iv(" : "mc_cos_iv(") + L(n.a) + ", " + H(n.a) +
             ", " + l + ", " + h + ");\n";
        return true;
    default: return false;  // general pow
    }
    s += pre;
    s += "    const float " + l + " = " + lo + ";\n";
    if (n.op == NodeOp::CONST) s += "    const float " + h + " = " + l + ";\n";
    else s += "    const float " + h + " = " + hi + ";\n";
    return true;
}

// vector instructions the interval code of node i costs (operands not included)
int iv_node_cost(const Node& n) {
    switch (n.op) {
    case NodeOp::ADD: case NodeOp::SUB: case NodeOp::NEG: return 2;
    case NodeOp::MUL: return 10;
    case NodeOp::DIV: return 40;
    case NodeOp::POWI: return 8 + 8 * (n.ipow < 0 ? -n.ipow : n.ipow);
    case NodeOp::SIN: case NodeOp::COS: return 60;
    case NodeOp::POW: return 600;
    default: return 0;
    }
}
}  // namespace

std::string emit_hip_interval(const Program& p, const char* fname) {
    std::string s;
    s += std::string("__device__ __forceinline__ void ") + fname + "(float xl, float xh, float yl, float yh, float zl, float zh, float& lo, float& hi) {\n";
    s += "    (void)xl; (void)xh; (void)yl; (void)yh; (void)zl; (void)zh;\n";
    for (size_t i = 0; i < p.nodes.size(); ++i)
        if (!iv_node(p, i, s)) return std::string();
    const Node& r = p.nodes[p.root];
    const char v = r.op == NodeOp::VARX ? 'x' : r.op == NodeOp::VARY ? 'y' : r.op == NodeOp::VARZ ? 'z' : 0;
    if (v) s += std::string("    lo = ") + v + "l;\n    hi = " + v + "h;\n}\n";
    else s += "    lo = l" + std::to_string(p.root) + ";\n    hi = h" + std::to_string(p.root) + ";\n}\n";
    return s;
}

// The same enclosure in two stages, for the classify walk: the sub-expressions that do NOT depend on x, depend on y and are
// expensive in their y part (sin / cos, divisions, higher powers of y -- alone or combined with anything of z: the gyroid's
// sin(y)*cos(z)) are evaluated once per tile row (mc_f_iv_y, lane = row: the tile's z box is the same for every row) and the
// rest per lane box from those values (mc_f_iv_rest).  The walk's x and z boxes do not change from row to row, so the
// compiler hoists what depends on them alone by itself; what depends on y it cannot (the row is a loop variable).  Round 4:
// "y only" became "not x" -- the product sin(y)*cos(z) used to be an interval product per lane and row step.  "" when
// nothing is worth staging.
std::string emit_hip_interval_staged(const Program& p, int min_cost) {
    const size_t n = p.nodes.size();
    auto xfree_y = [&](const Node& nd) { return (nd.deps & 1) == 0 && (nd.deps & 2) != 0; };
    std::vector<char> used_outside(n, 0);   // some user depends on x (or does not depend on y), or the node is the root
    for (size_t i = 0; i < n; ++i) {
        const Node& nd = p.nodes[i];
        if (xfree_y(nd)) continue;
        if (nd.a >= 0) used_outside[nd.a] = 1;
        if (nd.b >= 0) used_outside[nd.b] = 1;
    }
    used_outside[p.root] = 1;
    // a node's whole cone (each node once); its cost: the nodes that depend on y -- what a row step would pay for it
    auto cone = [&](size_t root, std::vector<char>& in) {
        in.assign(n, 0);
        in[root] = 1;
        int c = 0;
        for (size_t i = root + 1; i-- > 0;) {
            if (!in[i]) continue;
            if (p.nodes[i].deps & 2) c += iv_node_cost(p.nodes[i]);
            if (p.nodes[i].a >= 0) in[p.nodes[i].a] = 1;
            if (p.nodes[i].b >= 0) in[p.nodes[i].b] = 1;
        }
        return c;
    };
    std::vector<int> hoisted;
    std::vector<char> in, ystage(n, 0);
    for (size_t i = 0; i < n; ++i) {
        const Node& nd = p.nodes[i];
        if (!xfree_y(nd) || !used_outside[i] || nd.op == NodeOp::VARY) continue;
        if (cone(i, in) < min_cost) continue;
        hoisted.push_back((int)i);
        for (size_t k = 0; k < n; ++k) ystage[k] = ystage[k] | in[k];
    }
    if (hoisted.empty() || hoisted.size() > 8) return std::string();
    std::string s = "#define MC_IV_NY " + std::to_string(2 * hoisted.size()) + "\n";
    s += "__device__ __forceinline__ void mc_f_iv_y(float yl, float yh, float zl, float zh, float (&Y)[MC_IV_NY]) {\n    (void)zl; (void)zh;\n";
    for (size_t i = 0; i < n; ++i)
        if (ystage[i] && !iv_node(p, i, s)) return std::string();
    for (size_t k = 0; k < hoisted.size(); ++k)
        s += "    Y[" + std::to_string(2 * k) + "] = l" + std::to_string(hoisted[k]) + "; Y[" + std::to_string(2 * k + 1) + "] = h" +
             std::to_string(hoisted[k]) + ";\n";
    s += "}\n";
    // the rest: everything the root needs, cut at the hoisted nodes
    std::vector<char> need(n, 0), is_h(n, 0);
    for (int h : hoisted) is_h[h] = 1;
    need[p.root] = 1;
    for (size_t i = n; i-- > 0;) {
        if (!need[i] || is_h[i]) continue;
        if (p.nodes[i].a >= 0) need[p.nodes[i].a] = 1;
        if (p.nodes[i].b >= 0) need[p.nodes[i].b] = 1;
    }
    s += "__device__ __forceinline__ void mc_f_iv_rest(float xl, float xh, float yl, float yh, float zl, float zh, const float (&Y)[MC_IV_NY], "
         "float& lo, float& hi) {\n    (void)xl; (void)xh; (void)yl; (void)yh; (void)zl; (void)zh;\n";
    for (size_t k = 0; k < hoisted.size(); ++k)
        s += "    const float l" + std::to_string(hoisted[k]) + " = Y[" + std::to_string(2 * k) + "], h" + std::to_string(hoisted[k]) + " = Y[" +
             std::to_string(2 * k + 1) + "];\n";
    for (size_t i = 0; i < n; ++i)
        if (need[i] && !is_h[i] && !iv_node(p, i, s)) return std::string();
    const Node& r = p.nodes[p.root];
    const char v = r.op == NodeOp::VARX ? 'x' : r.op == NodeOp::VARY ? 'y' : r.op == NodeOp::VARZ ? 'z' : 0;
    if (v) s += std::string("    lo = ") + v + "l;\n    hi = " + v + "h;\n}\n";
    else s += "    lo = l" + std::to_string(p.root) + ";\n    hi = h" + std::to_string(p.root) + ";\n}\n";
    return s;
}

float eval_host(const Program& p, float x, float y, float z) {
    std::vector<float> v(p.nodes.size());
    for (size_t i = 0; i < p.nodes.size(); ++i) {
        const Node& n = p.nodes[i];
        volatile float a = n.a >= 0 ? v[n.a] : 0.0f, b = n.b >= 0 ? v[n.b] : 0.0f, r = 0.0f;
        switch (n.op) {
        case NodeOp::CONST: r = n.cval; break;
        case NodeOp::VARX: r = x; break;
        case NodeOp::VARY: r = y; break;
        case NodeOp::VARZ: r = z; break;
        case NodeOp::ADD: r = a + b; break;
        case NodeOp::SUB: r = a - b; break;
        case NodeOp::MUL: r = a * b; break;
        case NodeOp::DIV: r = a / b; break;
        case NodeOp::NEG: r = -a; break;
        case NodeOp::POW: r = pow_general(a, b); break;
        case NodeOp::POWI: r = pow_literal_int(a, n.ipow); break;
        case NodeOp::SIN: r = mc_sinf(a); break;
        case NodeOp::COS: r = mc_cosf(a); break;
        }
        v[i] = r;
    }
    return v[p.root];
}

}  // namespace mc

namespace mc {
bool interp_program(const Program& p, InterpProg& out) {
    memset(&out, 0, sizeof out);
    const int n = (int)p.nodes.size();
    if (n == 0 || p.root < 0) return false;
    std::vector<int> last_use((size_t)n, -1), reg((size_t)n, -1);
    last_use[(size_t)p.root] = n + 1;  // the result is never released
    for (int i = n - 1; i >= 0; --i) {  // users come after their operands: one backward pass marks what the root reaches
        if (last_use[(size_t)i] < 0) continue;
        const Node& nd = p.nodes[(size_t)i];
        if (nd.a >= 0 && last_use[(size_t)nd.a] < i) last_use[(size_t)nd.a] = i;
        if (nd.b >= 0 && last_use[(size_t)nd.b] < i) last_use[(size_t)nd.b] = i;
    }
    uint32_t nconst = 0;
    bool free_reg[MC_INTERP_REGS];
    for (bool& f : free_reg) f = true;
    auto operand = [&](int id, uint32_t& byte) -> bool {  // the operand byte of node `id` (mc_kernels.hip: McInterpProg::code)
        const Node& nd = p.nodes[(size_t)id];
        switch (nd.op) {
        case NodeOp::VARX: byte = 64; return true;
        case NodeOp::VARY: byte = 65; return true;
        case NodeOp::VARZ: byte = 66; return true;
        case NodeOp::CONST: {
            uint32_t bits;
            memcpy(&bits, &nd.cval, 4);
            for (uint32_t k = 0; k < nconst; ++k) {
                uint32_t have;
                memcpy(&have, &out.cval[k], 4);
                if (have == bits) {
                    byte = 128 + k;
                    return true;
                }
            }
            if (nconst == MC_INTERP_CONSTS) return false;
            out.cval[nconst] = nd.cval;
            byte = 128 + nconst++;
            return true;
        }
        default:
            if (reg[(size_t)id] < 0) return false;
            byte = (uint32_t)reg[(size_t)id];
            return true;
        }
    };
    for (int i = 0; i < n; ++i) {
        const Node& nd = p.nodes[(size_t)i];
        if (nd.op == NodeOp::CONST || nd.op == NodeOp::VARX || nd.op == NodeOp::VARY || nd.op == NodeOp::VARZ) continue;
        if (last_use[(size_t)i] < 0) continue;  // (a node nothing uses: the builder leaves none, but it would cost a register)
        uint32_t op, a = 0, b = 0;
        switch (nd.op) {
        case NodeOp::ADD: op = 0; break;
        case NodeOp::SUB: op = 1; break;
        case NodeOp::MUL: op = 2; break;
        case NodeOp::DIV: op = 3; break;
        case NodeOp::POW: op = 4; break;
        case NodeOp::NEG: op = 5; break;
        case NodeOp::SIN: op = 6; break;
        case NodeOp::COS: op = 7; break;
        case NodeOp::POWI: op = nd.ipow == 2 ? 8 : 9; break;
        default: return false;
        }
        if (nd.a < 0 || !operand(nd.a, a)) return false;
        if (op <= 4) {
            if (nd.b < 0 || !operand(nd.b, b)) return false;
        } else if (op == 9) {
            if (nd.ipow < -32 || nd.ipow > 32) return false;
            b = (uint32_t)(nd.ipow + 32);
        }
        // operands whose last use this is give their registers back BEFORE the result takes one (the interpreter fetches
        // both operands before it writes)
        if (nd.a >= 0 && last_use[(size_t)nd.a] == i && reg[(size_t)nd.a] >= 0) free_reg[reg[(size_t)nd.a]] = true;
        if (nd.b >= 0 && last_use[(size_t)nd.b] == i && reg[(size_t)nd.b] >= 0) free_reg[reg[(size_t)nd.b]] = true;
        int d = -1;
        for (int k = 0; k < MC_INTERP_REGS; ++k)
            if (free_reg[k]) {
                d = k;
                break;
            }
        if (d < 0 || out.n == MC_INTERP_MAXI) return false;
        free_reg[d] = false;
        reg[(size_t)i] = d;
        out.code[out.n++] = op | ((uint32_t)d << 8) | (a << 16) | (b << 24);
    }
    // the result: the last operation's value, or -- f is a variable or a constant -- an operand byte
    if (out.n == 0) return operand(p.root, out.root);
    out.root = 64;  // (unused when n > 0)
    return reg[(size_t)p.root] >= 0 && (out.code[out.n - 1] >> 8 & 0xFFu) == (uint32_t)reg[(size_t)p.root];
}

// mc_kernels.hip: mc_interp_run, statement for statement
float interp_run_host(const InterpProg& P, float x, float y, float z) {
    float r[MC_INTERP_REGS] = {0.0f};
    auto fetch = [&](uint32_t o) -> float {
        if (o < 64u) return r[o & (MC_INTERP_REGS - 1)];
        if (o < 128u) return o == 64u ? x : o == 65u ? y : z;
        return P.cval[(o - 128u) & (MC_INTERP_CONSTS - 1)];
    };
    volatile float last = fetch(P.root);
    for (uint32_t i = 0; i < P.n; ++i) {
        const uint32_t w = P.code[i];
        const uint32_t op = w & 0xFFu, d = (w >> 8) & 0xFFu, ao = (w >> 16) & 0xFFu, bo = w >> 24;
        volatile float a = fetch(ao), v;
        if (op == 8u) v = a * a;
        else if (op == 9u) v = pow_literal_int(a, (int)bo - 32);
        else if (op == 5u) v = -a;
        else if (op == 6u) v = mc_sinf(a);
        else if (op == 7u) v = mc_cosf(a);
        else {
            volatile float b = fetch(bo);
            v = op == 0u ? a + b : op == 1u ? a - b : op == 2u ? a * b : op == 3u ? a / b : pow_general(a, b);
        }
        r[d & (MC_INTERP_REGS - 1)] = v;
        last = v;
    }
    return last;
}

int vector_op_cost(const Program& p) {
    int cost = 0;
    for (const Node& n : p.nodes) cost += node_cost(n);
    return cost;
}
}  // namespace mc
