// ORACLE (test infrastructure, NOT the product path).
// G2 scalar-multiplication STARK and Fq exponentiation STARK: trace generation and AIR.  Restates reference
//   src/starks/curves/g2/add.rs:46-196 (G2AddAux, generate_g2_add, eval_g2_add),
//   src/starks/curves/g2/ext/{mod,mul,add,sub,convert,modulus_zero,is_modulus_zero}.rs (Fq2 limb polynomials),
//   src/starks/curves/g2/scalar_mul_{view,stark,ctl}.rs (textual twins of the G1 files),
//   src/starks/fields/mul.rs:22-57, exp_view.rs:9-48, exp_stark.rs:36-327, exp_ctl.rs:18-75.
#pragma once
#include "sched.hpp"

namespace orc {

// ================================ G2 ==============================================================
namespace g2 {

static const int G2_LEN = 64;
static const int IEMZ_LEN = 2 + 2 * IMZ_LEN;  // IS_EXT_MODULUS_AUX_ZERO_LEN = 194 (ext/is_modulus_zero.rs:19)
static const int EMZ_LEN = 2 * MZ_LEN;        // EXT_MODULUS_AUX_ZERO_LEN = 160 (ext/modulus_zero.rs:20)
static const int ADD_AUX_LEN = 1 + IEMZ_LEN + 1 + 2 * NL + 3 * EMZ_LEN;  // 708 (add.rs:43-44)
// offsets inside G2AddAux (add.rs:48-56)
static const int AUX_IS_X_EQ = 0, AUX_IS_C0_ZERO = 1, AUX_IS_C1_ZERO = 2, AUX_C0_AUX = 3, AUX_C1_AUX = 3 + IMZ_LEN,
                 AUX_IS_X_EQ_FILTER = 1 + IEMZ_LEN, AUX_LAMBDA = 2 + IEMZ_LEN, AUX_LAMBDA_AUX = AUX_LAMBDA + 2 * NL,
                 AUX_X_AUX = AUX_LAMBDA_AUX + EMZ_LEN, AUX_Y_AUX = AUX_X_AUX + EMZ_LEN;
static inline const Layout& layout() {
  static Layout L(G2_LEN, ADD_AUX_LEN);  // W = 1295
  return L;
}

struct Affine {
  Fq2 x, y;
};
struct Input {
  u64 s[4];
  U256 x[4], off[4];  // x.c0, x.c1, y.c0, y.c1
};

static inline void fq2_to_limbs(const Fq2& v, u64* out /*32*/) {
  int64_t l[16];
  u256_to_limbs(fq_to_u256(v.c0), l);
  for (int i = 0; i < 16; i++) out[i] = (u64)l[i];
  u256_to_limbs(fq_to_u256(v.c1), l);
  for (int i = 0; i < 16; i++) out[16 + i] = (u64)l[i];
}
static inline Affine affine_add(const Affine& a, const Affine& b, Fq2* lambda_out) {
  Fq2 lam;
  if (!(a.x == b.x)) {
    lam = fq2_mul(fq2_sub(b.y, a.y), fq2_inv(fq2_sub(b.x, a.x)));  // add.rs:73
  } else {
    if (!(a.y == b.y) || a.y.is_zero()) throw std::runtime_error("generate_g2_add: b == -a (point at infinity)");
    lam = fq2_mul(fq2_mul(fq2_from_u64(3), fq2_mul(a.x, a.x)), fq2_inv(fq2_mul(fq2_from_u64(2), a.y)));  // add.rs:86
  }
  Affine c;
  c.x = fq2_sub(fq2_sub(fq2_mul(lam, lam), a.x), b.x);
  c.y = fq2_sub(fq2_mul(lam, fq2_sub(a.x, c.x)), a.y);
  if (lambda_out) *lambda_out = lam;
  return c;
}

// U256Ext<i64> helpers (ext/*.rs)
struct Ext16 {
  Pol16<int64_t> c0, c1;
};
struct Ext31 {
  Pol31<int64_t> c0, c1;
};
static inline Ext16 ext_from_limbs(const u64* p) {
  Ext16 r;
  for (int i = 0; i < 16; i++) {
    r.c0[i] = (int64_t)p[i];
    r.c1[i] = (int64_t)p[16 + i];
  }
  return r;
}
static inline Ext16 ext_sub(const Ext16& x, const Ext16& y) { return Ext16{pol_subn(x.c0, y.c0), pol_subn(x.c1, y.c1)}; }
static inline Ext16 ext_add(const Ext16& x, const Ext16& y) { return Ext16{pol_addn(x.c0, y.c0), pol_addn(x.c1, y.c1)}; }
static inline Ext31 ext_mul(const Ext16& x, const Ext16& y) {  // mul_uint256ext (ext/mul.rs:14-32)
  Ext31 r;
  r.c0 = pol_subn(pol_mul_wide<int64_t>(x.c0, y.c0), pol_mul_wide<int64_t>(x.c1, y.c1));
  r.c1 = pol_addn(pol_mul_wide<int64_t>(x.c0, y.c1), pol_mul_wide<int64_t>(x.c1, y.c0));
  return r;
}
static inline Ext31 ext_widen(const Ext16& x) { return Ext31{pol_widen(x.c0), pol_widen(x.c1)}; }
static inline Ext31 ext31_sub(const Ext31& x, const Ext31& y) { return Ext31{pol_subn(x.c0, y.c0), pol_subn(x.c1, y.c1)}; }
static inline Ext31 ext31_add(const Ext31& x, const Ext31& y) { return Ext31{pol_addn(x.c0, y.c0), pol_addn(x.c1, y.c1)}; }
static inline Ext31 ext31_scale(const Ext31& x, int64_t c) { return Ext31{pol_scale<int64_t, 31>(x.c0, c), pol_scale<int64_t, 31>(x.c1, c)}; }
static inline void generate_ext_modulus_zero(const Ext31& in, u64* out /*160*/) {  // ext/modulus_zero.rs:38-46
  generate_modulus_zero(in.c0, out);
  generate_modulus_zero(in.c1, out + MZ_LEN);
}

struct Ops {
  typedef Affine Elem;
  void to_limbs(const Affine& p, u64* out) const {
    fq2_to_limbs(p.x, out);
    fq2_to_limbs(p.y, out + 32);
  }
  // generate_g2_add (add.rs:59-130)
  Affine op(const Affine& a, const Affine& b, const u64* al, const u64* bl, u64* cl, u64* aux) const {
    Fq2 lam;
    Affine c = affine_add(a, b, &lam);
    to_limbs(c, cl);
    Ext16 ax = ext_from_limbs(al), ay = ext_from_limbs(al + 32), bx = ext_from_limbs(bl), by = ext_from_limbs(bl + 32),
          cx = ext_from_limbs(cl), cy = ext_from_limbs(cl + 32);
    Ext16 delta_x = ext_sub(bx, ax);
    // generate_is_ext_modulus_zero (ext/is_modulus_zero.rs:30-46)
    u64 z0 = generate_is_modulus_zero(delta_x.c0, aux + AUX_C0_AUX);
    u64 z1 = generate_is_modulus_zero(delta_x.c1, aux + AUX_C1_AUX);
    aux[AUX_IS_C0_ZERO] = z0;
    aux[AUX_IS_C1_ZERO] = z1;
    u64 is_x_eq = z0 * z1;
    aux[AUX_IS_X_EQ] = is_x_eq;
    aux[AUX_IS_X_EQ_FILTER] = is_x_eq;
    u64 lam_l[32];
    fq2_to_limbs(lam, lam_l);
    for (int i = 0; i < 32; i++) aux[AUX_LAMBDA + i] = lam_l[i];
    Ext16 lm = ext_from_limbs(lam_l);
    if (!is_x_eq) {
      Ext31 diff = ext31_sub(ext_mul(lm, delta_x), ext_widen(ext_sub(by, ay)));
      generate_ext_modulus_zero(diff, aux + AUX_LAMBDA_AUX);
    } else {
      Ext31 diff = ext31_sub(ext31_scale(ext_mul(lm, ay), 2), ext31_scale(ext_mul(ax, ax), 3));
      generate_ext_modulus_zero(diff, aux + AUX_LAMBDA_AUX);
    }
    Ext31 sum_x = ext_widen(ext_add(ext_add(ax, bx), cx));
    generate_ext_modulus_zero(ext31_sub(ext_mul(lm, lm), sum_x), aux + AUX_X_AUX);
    Ext31 cyay = ext_widen(ext_add(cy, ay));
    generate_ext_modulus_zero(ext31_add(ext_mul(lm, ext_sub(cx, ax)), cyay), aux + AUX_Y_AUX);
    return c;
  }
};

static inline Affine load_point(const U256* w) {
  return Affine{Fq2{fq_from_u256(w[0]), fq_from_u256(w[1])}, Fq2{fq_from_u256(w[2]), fq_from_u256(w[3])}};
}
static inline std::vector<std::vector<u64>> generate_trace(const std::vector<Input>& inputs, size_t min_rows,
                                                           std::vector<Affine>* outputs) {
  const Layout& L = layout();
  if (outputs) outputs->resize(inputs.size());
  Ops ops;
  return generate_trace_generic(L, inputs.size(), min_rows, [&](size_t k, u64* rows) {
    Affine out = generate_one_set(L, ops, inputs[k].s, load_point(inputs[k].x), load_point(inputs[k].off), k, rows);
    if (outputs) (*outputs)[k] = out;
  });
}
static inline Affine scalar_mul_offset(const Input& in) {
  Affine dbl = load_point(in.x), sum = load_point(in.off);
  for (int i = 0; i < 256; i++) {
    if ((in.s[i / 64] >> (i % 64)) & 1) sum = affine_add(sum, dbl, nullptr);
    if (i < 255) dbl = affine_add(dbl, dbl, nullptr);
  }
  return sum;
}
// g2_generate_ctl_values (g2/scalar_mul_ctl.rs:57-80)
static inline std::vector<std::vector<std::vector<u64>>> generate_ctl_values(const std::vector<Input>& inputs,
                                                                               const std::vector<Affine>& outputs) {
  std::vector<std::vector<std::vector<u64>>> e(2);
  Ops ops;
  for (size_t k = 0; k < inputs.size(); k++) {
    std::vector<u64> in(64 + 64 + 16 + 1), out(65);
    ops.to_limbs(load_point(inputs[k].x), in.data());
    ops.to_limbs(load_point(inputs[k].off), in.data() + 64);
    for (int i = 0; i < 16; i++) in[128 + i] = (inputs[k].s[i / 4] >> (16 * (i % 4))) & 0xFFFF;
    in[144] = k;
    ops.to_limbs(outputs[k], out.data());
    out[64] = k;
    e[0].push_back(in);
    e[1].push_back(out);
  }
  return e;
}

// ---- AIR ----
template <class T> struct ExtT16 { Pol16<T> c0, c1; };
template <class T> struct ExtT31 { Pol31<T> c0, c1; };
template <class T> static ExtT16<T> ldext(const T* p) {
  ExtT16<T> r;
  for (int i = 0; i < 16; i++) {
    r.c0[i] = p[i];
    r.c1[i] = p[16 + i];
  }
  return r;
}
template <class T> static ExtT16<T> esub(const ExtT16<T>& x, const ExtT16<T>& y) { return {pol_subn(x.c0, y.c0), pol_subn(x.c1, y.c1)}; }
template <class T> static ExtT16<T> eadd(const ExtT16<T>& x, const ExtT16<T>& y) { return {pol_addn(x.c0, y.c0), pol_addn(x.c1, y.c1)}; }
template <class T> static ExtT31<T> emul(const ExtT16<T>& x, const ExtT16<T>& y) {
  return {pol_subn(pol_mul_wide<T>(x.c0, y.c0), pol_mul_wide<T>(x.c1, y.c1)),
          pol_addn(pol_mul_wide<T>(x.c0, y.c1), pol_mul_wide<T>(x.c1, y.c0))};
}
template <class T> static ExtT31<T> ewiden(const ExtT16<T>& x) { return {pol_widen(x.c0), pol_widen(x.c1)}; }
template <class T> static ExtT31<T> e31sub(const ExtT31<T>& x, const ExtT31<T>& y) { return {pol_subn(x.c0, y.c0), pol_subn(x.c1, y.c1)}; }
template <class T> static ExtT31<T> e31add(const ExtT31<T>& x, const ExtT31<T>& y) { return {pol_addn(x.c0, y.c0), pol_addn(x.c1, y.c1)}; }
template <class T> static ExtT31<T> e31scale(const ExtT31<T>& x, T c) { return {pol_scale<T, 31>(x.c0, c), pol_scale<T, 31>(x.c1, c)}; }
template <class T> static void eval_ext_modulus_zero(Consumer<T>& cc, T filter, const ExtT31<T>& in, const T* aux) {
  eval_modulus_zero<T>(cc, filter, in.c0, aux);           // ext/modulus_zero.rs:48-58
  eval_modulus_zero<T>(cc, filter, in.c1, aux + MZ_LEN);
}

// eval_g2_add (add.rs:132-196)
template <class T> static void eval_g2_add(Consumer<T>& cc, T filter, const T* a, const T* b, const T* c, const T* aux) {
  ExtT16<T> ax = ldext(a), ay = ldext(a + 32), bx = ldext(b), by = ldext(b + 32), cx = ldext(c), cy = ldext(c + 32);
  ExtT16<T> lambda = ldext(aux + AUX_LAMBDA);
  ExtT16<T> delta_x = esub(bx, ax);
  // eval_is_ext_modulus_zero (ext/is_modulus_zero.rs:48-75)
  cc.constraint(filter * (aux[AUX_IS_C0_ZERO] * aux[AUX_IS_C1_ZERO] - aux[AUX_IS_X_EQ]));
  eval_is_modulus_zero<T>(cc, filter, delta_x.c0, aux[AUX_IS_C0_ZERO], aux + AUX_C0_AUX);
  eval_is_modulus_zero<T>(cc, filter, delta_x.c1, aux[AUX_IS_C1_ZERO], aux + AUX_C1_AUX);
  T is_x_eq_filter = aux[AUX_IS_X_EQ_FILTER];
  cc.constraint(filter * aux[AUX_IS_X_EQ] - is_x_eq_filter);
  T is_not_eq_filter = filter - is_x_eq_filter;
  eval_ext_modulus_zero<T>(cc, is_not_eq_filter, e31sub(emul(lambda, delta_x), ewiden(esub(by, ay))), aux + AUX_LAMBDA_AUX);
  ExtT31<T> three_x_sq = e31scale(emul(ax, ax), tconst<T>(3));
  ExtT31<T> two_lambda_y = e31scale(emul(lambda, ay), tconst<T>(2));
  eval_ext_modulus_zero<T>(cc, is_x_eq_filter, e31sub(two_lambda_y, three_x_sq), aux + AUX_LAMBDA_AUX);
  eval_eq_n<T>(cc, is_x_eq_filter, a + 32, b + 32, 32);
  ExtT31<T> sum_x = ewiden(eadd(eadd(ax, bx), cx));
  eval_ext_modulus_zero<T>(cc, filter, e31sub(emul(lambda, lambda), sum_x), aux + AUX_X_AUX);
  eval_ext_modulus_zero<T>(cc, filter, e31add(emul(lambda, esub(cx, ax)), ewiden(eadd(cy, ay))), aux + AUX_Y_AUX);
}
template <class T> static void eval_constraints(const T* local, const T* next, Consumer<T>& cc) {
  const Layout& L = layout();
  eval_g2_add<T>(cc, local[L.FILTER], local + L.A, local + L.B, local + L.C, local + L.AUX);
  eval_schedule<T>(L, local, next, cc, false);
}
static inline StarkDef stark_def() {
  StarkDef d;
  d.name = "g2_scalar_mul";
  fill_stark_def(d, layout(), true);
  d.eval_base = [](const F* l, const F* n, Consumer<F>& cc) { eval_constraints<F>(l, n, cc); };
  d.eval_ext = [](const F2* l, const F2* n, Consumer<F2>& cc) { eval_constraints<F2>(l, n, cc); };
  return d;
}
}  // namespace g2

// ================================ Fq exp ==========================================================
namespace fqexp {

static inline const Layout& layout() {
  static Layout L(NL, MZ_LEN);  // W = 427
  return L;
}
struct Input {
  u64 s[4];
  U256 x;
};
struct Ops {
  typedef Fq Elem;
  void to_limbs(const Fq& v, u64* out) const {
    int64_t l[16];
    u256_to_limbs(fq_to_u256(v), l);
    for (int i = 0; i < 16; i++) out[i] = (u64)l[i];
  }
  // generate_fq_mul (fields/mul.rs:22-40)
  Fq op(const Fq& a, const Fq& b, const u64* al, const u64* bl, u64* cl, u64* aux) const {
    Fq c = fq_mul(a, b);
    to_limbs(c, cl);
    Pol16<int64_t> A, B, C;
    for (int i = 0; i < 16; i++) {
      A[i] = (int64_t)al[i];
      B[i] = (int64_t)bl[i];
      C[i] = (int64_t)cl[i];
    }
    generate_modulus_zero(pol_subn(pol_mul_wide<int64_t>(A, B), pol_widen(C)), aux);
    return c;
  }
};
static inline std::vector<std::vector<u64>> generate_trace(const std::vector<Input>& inputs, size_t min_rows, std::vector<Fq>* outputs) {
  const Layout& L = layout();
  if (outputs) outputs->resize(inputs.size());
  Ops ops;
  return generate_trace_generic(L, inputs.size(), min_rows, [&](size_t k, u64* rows) {
    // first row: square = x, a = 1 (exp_stark.rs:114-125)
    Fq out = generate_one_set(L, ops, inputs[k].s, fq_from_u256(inputs[k].x), fq_one(), k, rows);
    if (outputs) (*outputs)[k] = out;
  });
}
static inline Fq pow_s(const Input& in) {
  Fq sq = fq_from_u256(in.x), prod = fq_one();
  for (int i = 0; i < 256; i++) {
    if ((in.s[i / 64] >> (i % 64)) & 1) prod = fq_mul(prod, sq);
    sq = fq_mul(sq, sq);
  }
  return prod;
}
// fq_generate_ctl_values (exp_ctl.rs:54-75)
static inline std::vector<std::vector<std::vector<u64>>> generate_ctl_values(const std::vector<Input>& inputs,
                                                                               const std::vector<Fq>& outputs) {
  std::vector<std::vector<std::vector<u64>>> e(2);
  Ops ops;
  for (size_t k = 0; k < inputs.size(); k++) {
    std::vector<u64> in(33), out(17);
    ops.to_limbs(fq_from_u256(inputs[k].x), in.data());
    for (int i = 0; i < 16; i++) in[16 + i] = (inputs[k].s[i / 4] >> (16 * (i % 4))) & 0xFFFF;
    in[32] = k;
    ops.to_limbs(outputs[k], out.data());
    out[16] = k;
    e[0].push_back(in);
    e[1].push_back(out);
  }
  return e;
}
// eval_fq_mul (mul.rs:43-57) + schedule
template <class T> static void eval_constraints(const T* local, const T* next, Consumer<T>& cc) {
  const Layout& L = layout();
  Pol16<T> a, b, c;
  for (int i = 0; i < 16; i++) {
    a[i] = local[L.A + i];
    b[i] = local[L.B + i];
    c[i] = local[L.C + i];
  }
  eval_modulus_zero<T>(cc, local[L.FILTER], pol_subn(pol_mul_wide<T>(a, b), pol_widen(c)), local + L.AUX);
  eval_schedule<T>(L, local, next, cc, true);
}
static inline StarkDef stark_def() {
  StarkDef d;
  d.name = "fq_exp";
  fill_stark_def(d, layout(), false);
  d.eval_base = [](const F* l, const F* n, Consumer<F>& cc) { eval_constraints<F>(l, n, cc); };
  d.eval_ext = [](const F2* l, const F2* n, Consumer<F2>& cc) { eval_constraints<F2>(l, n, cc); };
  return d;
}
}  // namespace fqexp

}  // namespace orc
