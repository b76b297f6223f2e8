// expr.hip — one-pass evaluation of a pointwise expression tree over evaluation-domain matrices, behind
// tkmk_poly_expr_eval (include/tkmk.h).  Device side of the reference's fused expression evaluator
// (packages/backend/libs/src/bivariate_polynomial/mod.rs:227-260 evaluate_fused_with_domain, :311-435 evaluate_on_domain):
// there every node of the tree is one pass over 2^23 elements into a freshly allocated 256 MiB buffer (prove2's p_comb:
// ~15 passes); here every lane walks the whole postfix program for its element, so the leaves
// are read once and only the result is written.
//
// The program is postfix over a small operand stack kept in LDS (one 32-byte slot per lane and stack level; LDS
// indexing is free, a register-array stack would be spilled).  Values live in two forms, tracked at "compile" time on the
// host, so that no conversion passes are needed: PLAIN (as in memory) and MONT (x R).  mont_mul(plain, mont) = plain and
// mont_mul(mont, mont) = mont; leaves are PLAIN, constants and the (w_x^i - 1) row factors MONT; a product of two PLAIN
// values converts one operand first.  The host emits the explicit conversion steps (TO_MONT_*), the kernel just runs them.
#include <string.h>

#include "common.h"

#define EXPR_MAX_INSTR 128
#define EXPR_MAX_LEAVES 16
#define EXPR_MAX_CONSTS 16
#define EXPR_MAX_DEPTH 6

enum {
    X_PUSH_LEAF = 0,   // push leaves[arg][e]
    X_PUSH_CONST = 1,  // push consts[arg]
    X_ADD = 2,         // (second, top) -> second + top
    X_SUB = 3,         // second - top
    X_MULM = 4,        // mont_mul(second, top)
    X_SCALE = 5,       // top = mont_mul(top, consts[arg])
    X_XM1 = 6,         // top = mont_mul(top, w_x^ix - 1)
    X_TOMONT_TOP = 7,
    X_TOMONT_SECOND = 8,
    X_FROMMONT_TOP = 9,
};

// a leaf operand: element (ix, iy) of the domain reads data[((ix - rot_x) & xmask) * y_len + ((iy - rot_y) & ymask)].  A full matrix
// has xmask = x_size - 1, ymask = y_size - 1; a vector broadcast along Y has ymask = 0, y_len = 1 (X-only polynomials), along X
// xmask = 0 (Y-only); a rotation by (rot_x, rot_y) is the evaluation-domain form of p(w_x^-rot_x X, w_y^-rot_y Y)
struct expr_leaf_t {
    const fr_t *data;
    uint32_t xmask, ymask, y_len, rot_x, rot_y;
    uint32_t plain;   // 1: no view, address = linear element index
};
struct expr_args_t {
    uint32_t n_instr;
    uint8_t op[EXPR_MAX_INSTR];
    uint8_t arg[EXPR_MAX_INSTR];
    expr_leaf_t leaf[EXPR_MAX_LEAVES];
    fr_t cst[EXPR_MAX_CONSTS];   // Montgomery
};

__global__ __launch_bounds__(256) void k_expr_eval(expr_args_t a, const fr_t *__restrict__ xm1, uint32_t y_size, uint64_t total,
                                                  fr_t *__restrict__ out) {
    __shared__ uint4 lo[EXPR_MAX_DEPTH][256], hi[EXPR_MAX_DEPTH][256];
    const uint32_t t = threadIdx.x;
    auto put = [&](uint32_t lvl, const fr_t &v) {
        const uint4 *s = reinterpret_cast<const uint4 *>(&v);
        lo[lvl][t] = s[0];
        hi[lvl][t] = s[1];
    };
    auto get = [&](uint32_t lvl) {
        fr_t v;
        uint4 *d = reinterpret_cast<uint4 *>(&v);
        d[0] = lo[lvl][t];
        d[1] = hi[lvl][t];
        return v;
    };
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + t; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t sp = 0;  // number of live stack levels (uniform over the wave: the program is)
        const uint32_t ix = (uint32_t)(e / y_size), iy = (uint32_t)(e - (uint64_t)ix * y_size);
        for (uint32_t pc = 0; pc < a.n_instr; pc++) {
            const uint32_t op = a.op[pc], arg = a.arg[pc];
            switch (op) {
                case X_PUSH_LEAF: {
                    const expr_leaf_t &L = a.leaf[arg];
                    uint64_t at = L.plain ? e : (uint64_t)((ix - L.rot_x) & L.xmask) * L.y_len + ((iy - L.rot_y) & L.ymask);
                    put(sp++, Fr::canon(tk_load(L.data + at)));
                    break;
                }
                case X_PUSH_CONST: put(sp++, a.cst[arg]); break;
                case X_ADD: put(sp - 2, Fr::add(get(sp - 2), get(sp - 1))); sp--; break;
                case X_SUB: put(sp - 2, Fr::sub(get(sp - 2), get(sp - 1))); sp--; break;
                case X_MULM: put(sp - 2, Fr::mul(get(sp - 2), get(sp - 1))); sp--; break;
                case X_SCALE: put(sp - 1, Fr::mul(get(sp - 1), a.cst[arg])); break;
                case X_XM1: put(sp - 1, Fr::mul(get(sp - 1), tk_load(xm1 + ix))); break;
                case X_TOMONT_TOP: put(sp - 1, Fr::to_mont(get(sp - 1))); break;
                case X_TOMONT_SECOND: put(sp - 2, Fr::to_mont(get(sp - 2))); break;
                default: put(sp - 1, Fr::from_mont(get(sp - 1))); break;  // X_FROMMONT_TOP
            }
        }
        tk_store(out + e, get(0));
    }
}

// xm1[i] = w^i - 1 (Montgomery), w = root of unity of order x_size (the evaluation-domain form of "multiply by X - 1":
// mod.rs:372-378, x_minus_one_evals :504-518)
__global__ __launch_bounds__(256) void k_xm1_table(fr_t *__restrict__ out, fr_t w, uint32_t n, uint32_t first) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tk_store(out + i, Fr::sub(Fr::pow_u64(w, (uint64_t)first + i), Fr::one()));
}

// Host-side "compilation": validates the postfix program and inserts the form conversions.
namespace {
enum form_t { PLAIN = 0, MONT = 1 };
struct emitter {
    expr_args_t *a;
    bool ok = true;
    void emit(uint8_t op, uint8_t arg = 0) {
        if (a->n_instr >= EXPR_MAX_INSTR) {
            ok = false;
            return;
        }
        a->op[a->n_instr] = op;
        a->arg[a->n_instr] = arg;
        a->n_instr++;
    }
};
}  // namespace

// x_global / x_first: the rows [x_first, x_first + x_size) of a domain with x_global rows are evaluated (a row slab of a sharded prover;
// x_global = x_size, x_first = 0 for a whole domain): only the (w_x^ix - 1) factors depend on the global row index
static tkmk_error expr_eval_impl(const tkmk_expr_instr *prog, uint32_t n_instr, expr_args_t &a, uint32_t n_leaves, const tkmk_fr *consts,
                                 uint32_t n_consts, uint32_t x_size, uint32_t y_size, tkmk_fr *out_dev, tkmk_stream stream, uint32_t x_global = 0,
                                 uint32_t x_first = 0);

TK_API tkmk_error tkmk_poly_expr_eval(const tkmk_expr_instr *prog, uint32_t n_instr, const tkmk_fr *const *leaves_dev, uint32_t n_leaves,
                                      const tkmk_fr *consts, uint32_t n_consts, uint32_t x_size, uint32_t y_size, tkmk_fr *out_dev,
                                      tkmk_stream stream) {
    if (!prog || !out_dev || (n_leaves && !leaves_dev) || (n_consts && !consts)) return TKMK_ERR_INVALID_POINTER;
    if (!n_instr || !x_size || !y_size || n_leaves > EXPR_MAX_LEAVES || n_consts > EXPR_MAX_CONSTS) return TKMK_ERR_INVALID_ARGUMENT;
    expr_args_t a;
    memset(&a, 0, sizeof a);
    for (uint32_t k = 0; k < n_leaves; k++) {
        if (!leaves_dev[k]) return TKMK_ERR_INVALID_POINTER;
        a.leaf[k].data = (const fr_t *)leaves_dev[k];
        a.leaf[k].plain = 1;
    }
    return expr_eval_impl(prog, n_instr, a, n_leaves, consts, n_consts, x_size, y_size, out_dev, stream);
}

// the same evaluator over leaf VIEWS (include/tkmk.h tkmk_expr_leaf): vectors broadcast along one axis and cyclic rotations of a
// matrix, so that X-only / Y-only polynomials cost a 1-D transform and p(w^-1 X, Y) shares p's evaluations
TK_API tkmk_error tkmk_poly_expr_eval_views(const tkmk_expr_instr *prog, uint32_t n_instr, const tkmk_expr_leaf *leaves, uint32_t n_leaves,
                                            const tkmk_fr *consts, uint32_t n_consts, uint32_t x_size, uint32_t y_size, tkmk_fr *out_dev,
                                            tkmk_stream stream) {
    if (!prog || !out_dev || (n_leaves && !leaves) || (n_consts && !consts)) return TKMK_ERR_INVALID_POINTER;
    if (!n_instr || !x_size || !y_size || n_leaves > EXPR_MAX_LEAVES || n_consts > EXPR_MAX_CONSTS) return TKMK_ERR_INVALID_ARGUMENT;
    if ((x_size & (x_size - 1)) || (y_size & (y_size - 1))) return TKMK_ERR_INVALID_ARGUMENT;   // rotations are taken modulo the sizes
    expr_args_t a;
    memset(&a, 0, sizeof a);
    for (uint32_t k = 0; k < n_leaves; k++) {
        const tkmk_expr_leaf &l = leaves[k];
        if (!l.data) return TKMK_ERR_INVALID_POINTER;
        if ((l.x_len != x_size && l.x_len != 1) || (l.y_len != y_size && l.y_len != 1)) return TKMK_ERR_INVALID_ARGUMENT;
        if (l.rot_x >= l.x_len || l.rot_y >= l.y_len) return TKMK_ERR_INVALID_ARGUMENT;
        a.leaf[k].data = (const fr_t *)l.data;
        a.leaf[k].xmask = l.x_len - 1, a.leaf[k].ymask = l.y_len - 1, a.leaf[k].y_len = l.y_len;
        a.leaf[k].rot_x = l.rot_x, a.leaf[k].rot_y = l.rot_y;
        a.leaf[k].plain = (l.x_len == x_size && l.y_len == y_size && !l.rot_x && !l.rot_y) ? 1u : 0u;
    }
    return expr_eval_impl(prog, n_instr, a, n_leaves, consts, n_consts, x_size, y_size, out_dev, stream);
}

// The evaluator on ONE ROW SLAB of a larger domain (a rank of a sharded prover, include/tkmk_dist.h ROWS layout): the domain has x_global
// rows of which this call evaluates [x_first, x_first + x_rows); leaves are x_rows x y_size slabs (or vectors broadcast along an axis:
// x_len = 1 / y_len = 1, the x_len = x_rows vector being the slab's piece of an X-only polynomial's evaluations).  rot_x must be 0 — a
// rotation along X crosses slabs and is made beforehand (tkmk_dist_rows_rotate); rot_y is local.  x_rows, y_size, x_global powers of two.
TK_API tkmk_error tkmk_poly_expr_eval_views_slab(const tkmk_expr_instr *prog, uint32_t n_instr, const tkmk_expr_leaf *leaves, uint32_t n_leaves,
                                                 const tkmk_fr *consts, uint32_t n_consts, uint32_t x_global, uint32_t x_first, uint32_t x_rows,
                                                 uint32_t y_size, tkmk_fr *out_dev, tkmk_stream stream) {
    if (!prog || !out_dev || (n_leaves && !leaves) || (n_consts && !consts)) return TKMK_ERR_INVALID_POINTER;
    if (!n_instr || !x_rows || !y_size || n_leaves > EXPR_MAX_LEAVES || n_consts > EXPR_MAX_CONSTS) return TKMK_ERR_INVALID_ARGUMENT;
    if ((x_rows & (x_rows - 1)) || (y_size & (y_size - 1)) || (x_global & (x_global - 1)) || x_first % x_rows || (uint64_t)x_first + x_rows > x_global)
        return TKMK_ERR_INVALID_ARGUMENT;
    expr_args_t a;
    memset(&a, 0, sizeof a);
    for (uint32_t k = 0; k < n_leaves; k++) {
        const tkmk_expr_leaf &l = leaves[k];
        if (!l.data) return TKMK_ERR_INVALID_POINTER;
        if ((l.x_len != x_rows && l.x_len != 1) || (l.y_len != y_size && l.y_len != 1) || l.rot_x || l.rot_y >= l.y_len) return TKMK_ERR_INVALID_ARGUMENT;
        a.leaf[k].data = (const fr_t *)l.data;
        a.leaf[k].xmask = l.x_len - 1, a.leaf[k].ymask = l.y_len - 1, a.leaf[k].y_len = l.y_len;
        a.leaf[k].rot_x = 0, a.leaf[k].rot_y = l.rot_y;
        a.leaf[k].plain = (l.x_len == x_rows && l.y_len == y_size && !l.rot_y) ? 1u : 0u;
    }
    return expr_eval_impl(prog, n_instr, a, n_leaves, consts, n_consts, x_rows, y_size, out_dev, stream, x_global, x_first);
}

static tkmk_error expr_eval_impl(const tkmk_expr_instr *prog, uint32_t n_instr, expr_args_t &a, uint32_t n_leaves, const tkmk_fr *consts,
                                 uint32_t n_consts, uint32_t x_size, uint32_t y_size, tkmk_fr *out_dev, tkmk_stream stream, uint32_t x_global,
                                 uint32_t x_first) {
    TK_TRY(tk_require_device());
    tk_stat_add(TK_STAT_POLY_ELEMENTS, (uint64_t)x_size * y_size);
    if (!x_global) x_global = x_size, x_first = 0;
    for (uint32_t k = 0; k < n_consts; k++) {
        fr_t c;
        for (int i = 0; i < 8; i++) c.l[i] = consts[k].limbs[i];
        a.cst[k] = Fr::to_mont(Fr::canon(c));
    }
    emitter E{&a};
    form_t st[EXPR_MAX_DEPTH];
    uint32_t sp = 0;
    bool uses_xm1 = false;
    for (uint32_t pc = 0; pc < n_instr; pc++) {
        const uint32_t op = prog[pc].op, arg = prog[pc].arg;
        switch (op) {
            case TKMK_EXPR_LEAF:
                if (arg >= n_leaves || sp >= EXPR_MAX_DEPTH) return TKMK_ERR_INVALID_ARGUMENT;
                E.emit(X_PUSH_LEAF, (uint8_t)arg);
                st[sp++] = PLAIN;
                break;
            case TKMK_EXPR_CONST:
                if (arg >= n_consts || sp >= EXPR_MAX_DEPTH) return TKMK_ERR_INVALID_ARGUMENT;
                E.emit(X_PUSH_CONST, (uint8_t)arg);
                st[sp++] = MONT;
                break;
            case TKMK_EXPR_ADD:
            case TKMK_EXPR_SUB:
                if (sp < 2) return TKMK_ERR_INVALID_ARGUMENT;
                if (st[sp - 2] != st[sp - 1]) {  // bring the plain operand to Montgomery form
                    E.emit(st[sp - 1] == PLAIN ? X_TOMONT_TOP : X_TOMONT_SECOND);
                    st[sp - 2] = MONT;
                }
                E.emit(op == TKMK_EXPR_ADD ? X_ADD : X_SUB);
                sp--;
                break;
            case TKMK_EXPR_MUL:
                if (sp < 2) return TKMK_ERR_INVALID_ARGUMENT;
                if (st[sp - 2] == PLAIN && st[sp - 1] == PLAIN) {
                    E.emit(X_TOMONT_TOP);
                    st[sp - 1] = MONT;
                }
                E.emit(X_MULM);  // plain * mont = plain, mont * mont = mont
                st[sp - 2] = (st[sp - 2] == MONT && st[sp - 1] == MONT) ? MONT : PLAIN;
                sp--;
                break;
            case TKMK_EXPR_SCALE:
                if (sp < 1 || arg >= n_consts) return TKMK_ERR_INVALID_ARGUMENT;
                E.emit(X_SCALE, (uint8_t)arg);  // form unchanged
                break;
            case TKMK_EXPR_MUL_X_MINUS_ONE:
                if (sp < 1 || (x_global & (x_global - 1))) return TKMK_ERR_INVALID_ARGUMENT;
                E.emit(X_XM1);
                uses_xm1 = true;
                break;
            default: return TKMK_ERR_INVALID_ARGUMENT;
        }
    }
    if (sp != 1) return TKMK_ERR_INVALID_ARGUMENT;  // a well-formed postfix program leaves exactly its value
    if (st[0] == MONT) E.emit(X_FROMMONT_TOP);
    if (!E.ok) return TKMK_ERR_INVALID_ARGUMENT;
    hipStream_t s = tk_stream(stream);
    tk_frame frame(s);
    tk_scratch tx;
    if (uses_xm1) {
        tkmk_fr w;
        TK_TRY(bls12_381_get_root_of_unity(x_global, &w));
        fr_t wf;
        for (int i = 0; i < 8; i++) wf.l[i] = w.limbs[i];
        TK_TRY(tx.alloc((size_t)x_size * sizeof(fr_t), s));
        hipLaunchKernelGGL(k_xm1_table, tk_div_up(x_size, 256), 256, 0, s, tx.as<fr_t>(), Fr::to_mont(wf), x_size, x_first);
    }
    const uint64_t total = (uint64_t)x_size * y_size;
    uint64_t g = (total + 255) / 256;
    if (g > 256 * 8) g = 256 * 8;
    hipLaunchKernelGGL(k_expr_eval, (unsigned)g, 256, 0, s, a, (const fr_t *)tx.p, y_size, total, (fr_t *)out_dev);
    TK_HIP(hipGetLastError());
    return TKMK_SUCCESS;
}
