// Host-only evaluation of the three AIRs at the opening point, over the quadratic extension: the vanishing check of the native
// verifier (bn254s_verify / bn254s_verify_host) without a GPU.
//
// This is a second, independent statement of the constraints inside the library - written from the reference's
// eval_packed_generic functions in their own shape (a constraint consumer and one function per Rust function), not from the
// quotient kernels, which re-associate the same constraints for the GPU (quotient_g1.hip, quotient_g2fq.hip).  The two must
// agree on every proof; tests/test_gpu_verify.py checks that they (and the oracle's verifier) accept and reject the same
// proofs with the same messages.
//   consumer            starky ConstraintConsumer: acc_j <- acc_j * alpha_j + c
//   eval_modulus_zero   src/starks/modular/modulus_zero.rs:163-198
//   eval_is_modulus_zero src/starks/modular/is_modulus_zero.rs:69-84;  ext: curves/g2/ext/is_modulus_zero.rs:48-75
//   eval_g1_add         src/starks/curves/g1/add.rs:125-185;            eval_g2_add: curves/g2/add.rs:132-196
//   eval_fq_mul         src/starks/fields/mul.rs:43-57
//   eval_round_flags    src/starks/common/round_flags.rs:46-81
//   schedule            curves/g1/scalar_mul_stark.rs:226-339, curves/g2/scalar_mul_stark.rs:239-339, fields/exp_stark.rs:225-327
//   lookups, CTL        starky lookup / cross_table_lookup checks
#pragma once
#include <array>
#include <vector>
#include "aux.h"
#include "gl_dev.h"
#include "layout.h"

namespace host_air {

struct E {  // element of F2 = F[X]/(X^2 - 7) with value semantics and operators (host only)
  gl2 v;
  E() : v(gl2_make(0, 0)) {}
  E(gl2 x) : v(x) {}
  explicit E(u64 c) : v(gl2_make(c % GL_P, 0)) {}
};
inline E operator+(E a, E b) { return E(gl2_add(a.v, b.v)); }
inline E operator-(E a, E b) { return E(gl2_sub(a.v, b.v)); }
inline E operator*(E a, E b) { return E(gl2_mul(a.v, b.v)); }
inline E& operator+=(E& a, E b) { return a = a + b; }
inline E& operator-=(E& a, E b) { return a = a - b; }
inline E operator*(u64 c, E a) { return E(gl2_mul_base(a.v, c % GL_P)); }
static const E ONE = E((u64)1), ZERO = E((u64)0);

struct Consumer {
  E alpha[2], acc[2], z_last, l_first, l_last;
  long count = 0;
  void constraint(E c) {
    for (int j = 0; j < 2; j++) acc[j] = acc[j] * alpha[j] + c;
    count++;
  }
  void constraint_transition(E c) { constraint(c * z_last); }
  void constraint_first_row(E c) { constraint(c * l_first); }
  void constraint_last_row(E c) { constraint(c * l_last); }
};

static constexpr int NL = 16;  // N_LIMBS
typedef std::array<E, NL> U256;
typedef std::array<E, 2 * NL - 1> Wide;  // 31 coefficients
static const u64 BN254_P_LIMBS[NL] = {64839, 55420, 35862, 15392, 51853, 26737, 27281, 38785,
                                      22621, 33153, 17846, 47184, 41001, 57649, 20082, 12388};  // p, 16-bit limbs, little-endian

struct Row {  // one opened row: value(col) as an extension element
  const u64* w;
  E operator()(int col) const { return E(gl2_make(w[2 * col], w[2 * col + 1])); }
  U256 u256(int col) const {
    U256 r;
    for (int i = 0; i < NL; i++) r[i] = (*this)(col + i);
    return r;
  }
};

inline Wide pol_mul_wide(const U256& a, const U256& b) {
  Wide r;
  for (int i = 0; i < NL; i++)
    for (int j = 0; j < NL; j++) r[i + j] += a[i] * b[j];
  return r;
}
inline Wide widen(const U256& a) {
  Wide r;
  for (int i = 0; i < NL; i++) r[i] = a[i];
  return r;
}
template <size_t N>
inline std::array<E, N> operator+(std::array<E, N> a, const std::array<E, N>& b) {
  for (size_t i = 0; i < N; i++) a[i] += b[i];
  return a;
}
template <size_t N>
inline std::array<E, N> operator-(std::array<E, N> a, const std::array<E, N>& b) {
  for (size_t i = 0; i < N; i++) a[i] -= b[i];
  return a;
}
template <size_t N>
inline std::array<E, N> scale(u64 c, std::array<E, N> a) {
  for (size_t i = 0; i < N; i++) a[i] = c * a[i];
  return a;
}

// ModulusZeroAux: is_quot_positive, quot_abs[17], aux_input_lo[31], aux_input_hi[31] at columns aux .. aux + 79
inline void eval_modulus_zero(Consumer& yc, E filter, const Wide& input, const Row& r, int aux) {
  const E pos = r(aux);
  yc.constraint(filter * (pos * pos - pos));
  const E sign = 2 * pos - ONE;
  std::array<E, 2 * NL> constr;  // q(x) * m(x)
  for (int i = 0; i <= NL; i++) {
    const E q = sign * r(aux + 1 + i);
    for (int j = 0; j < NL; j++) constr[i + j] += BN254_P_LIMBS[j] * q;
  }
  const u64 base = 1ull << 16, offset = 1ull << 29;
  std::array<E, 2 * NL> s;  // aux polynomial (coefficient 31 is zero)
  for (int i = 0; i < 2 * NL - 1; i++) s[i] = r(aux + 18 + i) - E(offset) + base * r(aux + 49 + i);
  // + (x - base) * s(x)
  constr[0] -= base * s[0];
  for (int d = 1; d < 2 * NL; d++) constr[d] += s[d - 1] - base * s[d];
  for (int i = 0; i < 2 * NL - 1; i++) constr[i] -= input[i];
  for (int i = 0; i < 2 * NL; i++) yc.constraint(filter * constr[i]);
}

// IsModulusZeroAux: inv[16], ModulusZeroAux[80] at columns aux .. aux + 95
inline void eval_is_modulus_zero(Consumer& yc, E filter, const U256& input, E is_zero, const Row& r, int aux) {
  Wide diff = pol_mul_wide(input, r.u256(aux));
  diff[0] += is_zero - ONE;
  eval_modulus_zero(yc, filter, diff, r, aux + NL);
  for (int i = 0; i < NL; i++) yc.constraint(filter * (is_zero * input[i]));
}

inline void eval_eq(Consumer& yc, E filter, const Row& ra, int a, const Row& rb, int b, int n) {
  for (int i = 0; i < n; i++) yc.constraint(filter * (ra(a + i) - rb(b + i)));
}

// G1AddAux at aux: is_x_eq, IsModulusZeroAux[96], is_x_eq_filter, lambda[16], lambda_aux[80], x_aux[80], y_aux[80]
inline void eval_g1_add(Consumer& yc, E filter, const Row& r, int a, int b, int c, int aux) {
  const U256 ax = r.u256(a), ay = r.u256(a + NL), bx = r.u256(b), by = r.u256(b + NL), cx = r.u256(c), cy = r.u256(c + NL);
  const U256 delta_x = bx - ax;
  const E is_x_eq = r(aux);
  eval_is_modulus_zero(yc, filter, delta_x, is_x_eq, r, aux + 1);
  const E is_x_eq_filter = r(aux + 97);
  yc.constraint(filter * is_x_eq - is_x_eq_filter);
  const E is_not_eq_filter = filter - is_x_eq_filter;
  const U256 lambda = r.u256(aux + 98);
  const int lambda_aux = aux + 114, x_aux = aux + 194, y_aux = aux + 274;
  eval_modulus_zero(yc, is_not_eq_filter, pol_mul_wide(lambda, delta_x) - widen(by - ay), r, lambda_aux);
  eval_modulus_zero(yc, is_x_eq_filter, scale(2, pol_mul_wide(lambda, ay)) - scale(3, pol_mul_wide(ax, ax)), r, lambda_aux);
  eval_eq(yc, is_x_eq_filter, r, a + NL, r, b + NL, NL);
  eval_modulus_zero(yc, filter, pol_mul_wide(lambda, lambda) - widen(ax + bx + cx), r, x_aux);
  eval_modulus_zero(yc, filter, pol_mul_wide(lambda, cx - ax) + widen(cy + ay), r, y_aux);
}

struct U256Ext {
  U256 c0, c1;
};
struct WideExt {
  Wide c0, c1;
};
inline WideExt mul_ext(const U256Ext& x, const U256Ext& y) {  // (x0 + x1 u)(y0 + y1 u), u^2 = -1  (g2/ext/mul.rs:14-32)
  return {pol_mul_wide(x.c0, y.c0) - pol_mul_wide(x.c1, y.c1), pol_mul_wide(x.c0, y.c1) + pol_mul_wide(x.c1, y.c0)};
}
inline U256Ext ext_at(const Row& r, int col) { return {r.u256(col), r.u256(col + NL)}; }

// G2AddAux at aux (layout.h): is_x_eq, is_c0_zero, is_c1_zero, c0_aux[96], c1_aux[96], is_x_eq_filter, lambda[32],
// lambda_aux[160], x_aux[160], y_aux[160]
inline void eval_g2_add(Consumer& yc, E filter, const Row& r, int a, int b, int c, int aux) {
  const U256Ext ax = ext_at(r, a), ay = ext_at(r, a + 2 * NL), bx = ext_at(r, b), by = ext_at(r, b + 2 * NL), cx = ext_at(r, c),
                cy = ext_at(r, c + 2 * NL);
  const U256Ext delta_x = {bx.c0 - ax.c0, bx.c1 - ax.c1};
  const E is_x_eq = r(aux + G2_AUX_IS_X_EQ), z0 = r(aux + G2_AUX_IS_C0_ZERO), z1 = r(aux + G2_AUX_IS_C1_ZERO);
  yc.constraint(filter * (z0 * z1 - is_x_eq));
  eval_is_modulus_zero(yc, filter, delta_x.c0, z0, r, aux + G2_AUX_C0_AUX);
  eval_is_modulus_zero(yc, filter, delta_x.c1, z1, r, aux + G2_AUX_C1_AUX);
  const E is_x_eq_filter = r(aux + G2_AUX_IS_X_EQ_FILTER);
  yc.constraint(filter * is_x_eq - is_x_eq_filter);
  const E is_not_eq_filter = filter - is_x_eq_filter;
  const U256Ext lambda = ext_at(r, aux + G2_AUX_LAMBDA);
  auto ext_modulus_zero = [&](E f, const WideExt& d, int at) {
    eval_modulus_zero(yc, f, d.c0, r, at);
    eval_modulus_zero(yc, f, d.c1, r, at + 80);
  };
  {
    WideExt d = mul_ext(lambda, delta_x);
    d.c0 = d.c0 - widen(by.c0 - ay.c0);
    d.c1 = d.c1 - widen(by.c1 - ay.c1);
    ext_modulus_zero(is_not_eq_filter, d, aux + G2_AUX_LAMBDA_AUX);
  }
  {
    WideExt ly = mul_ext(lambda, ay), xx = mul_ext(ax, ax);
    WideExt d = {scale(2, ly.c0) - scale(3, xx.c0), scale(2, ly.c1) - scale(3, xx.c1)};
    ext_modulus_zero(is_x_eq_filter, d, aux + G2_AUX_LAMBDA_AUX);
  }
  eval_eq(yc, is_x_eq_filter, r, a + 2 * NL, r, b + 2 * NL, 2 * NL);
  {
    WideExt d = mul_ext(lambda, lambda);
    d.c0 = d.c0 - widen(ax.c0 + bx.c0 + cx.c0);
    d.c1 = d.c1 - widen(ax.c1 + bx.c1 + cx.c1);
    ext_modulus_zero(filter, d, aux + G2_AUX_X_AUX);
  }
  {
    WideExt d = mul_ext(lambda, {cx.c0 - ax.c0, cx.c1 - ax.c1});
    d.c0 = d.c0 + widen(cy.c0 + ay.c0);
    d.c1 = d.c1 + widen(cy.c1 + ay.c1);
    ext_modulus_zero(filter, d, aux + G2_AUX_Y_AUX);
  }
}

inline void eval_fq_mul(Consumer& yc, E filter, const Row& r, int a, int b, int c, int aux) {
  eval_modulus_zero(yc, filter, pol_mul_wide(r.u256(a), r.u256(b)) - widen(r.u256(c)), r, aux);
}

// RoundFlags at flags: is_first_round, is_last_round, counter, inv_counter, inv_counter_prime
inline void eval_round_flags(Consumer& yc, u64 period, E filter, const Row& r, int flags, E next_counter) {
  const E is_first = r(flags), is_last = r(flags + 1), counter = r(flags + 2), inv = r(flags + 3), inv_prime = r(flags + 4);
  const E not_filter = ONE - filter;
  yc.constraint(not_filter * is_first);
  yc.constraint(not_filter * is_last);
  yc.constraint(filter * (counter * inv - (ONE - is_first)));
  yc.constraint(filter * counter * is_first);
  const E counter_prime = counter - E(period - 1);
  yc.constraint(filter * (counter_prime * inv_prime - (ONE - is_last)));
  yc.constraint(filter * counter_prime * is_last);
  yc.constraint(filter * (ONE - is_last) * (next_counter - counter - ONE));
  yc.constraint(filter * is_last * next_counter);
}

// The double-and-add / square-and-multiply schedule shared by the three STARKs, after the operation's own constraints.
// Naming follows the G1 file; for Fq-exp: double = square, sum = product, is_adding = is_mul, idnl = is_sq_not_last and the
// extra "first round: a = 1" block.
template <class L>
inline void eval_schedule(Consumer& yc, const Row& local, const Row& next, bool fq_first_round_a_is_one) {
  const int PL = L::PL;
  const E filter = local(L::FILTER);
  const E is_first = local(L::FLAGS), is_last = local(L::FLAGS + 1), next_is_last = next(L::FLAGS + 1);
  const E is_not_last_round = filter - is_last;
  const E is_next_not_last_round = next(L::FILTER) - next_is_last;
  const E is_adding = local(L::IS_ADDING), idnl = local(L::IDNL);
  yc.constraint(is_first * (is_adding - ONE));
  eval_eq(yc, is_first, local, L::DOUBLE, local, L::B, PL);
  const E bit0 = local(L::BITS);
  eval_eq(yc, bit0 * is_first, local, L::SUM, local, L::C, PL);
  eval_eq(yc, (ONE - bit0) * is_first, local, L::SUM, local, L::A, PL);
  if (fq_first_round_a_is_one)
    for (int i = 0; i < PL; i++) yc.constraint(is_first * (local(L::A + i) - (i == 0 ? ONE : ZERO)));
  // doubling step -> addition step
  eval_eq(yc, idnl, next, L::A, local, L::SUM, PL);
  eval_eq(yc, idnl, next, L::B, local, L::DOUBLE, PL);
  const E nbit0 = next(L::BITS);
  eval_eq(yc, nbit0 * idnl, next, L::SUM, next, L::C, PL);
  eval_eq(yc, (ONE - nbit0) * idnl, next, L::SUM, next, L::A, PL);
  eval_eq(yc, idnl, next, L::DOUBLE, local, L::DOUBLE, PL);
  yc.constraint(idnl * (next(L::IS_ADDING) - ONE));
  yc.constraint(idnl * next(L::IDNL));
  for (int i = 0; i < 256; i++) yc.constraint(idnl * (next(L::BITS + i) - local(L::BITS + (i + 1) % 256)));
  // addition step -> doubling step
  eval_eq(yc, is_adding, next, L::A, local, L::DOUBLE, PL);
  eval_eq(yc, is_adding, next, L::B, local, L::DOUBLE, PL);
  eval_eq(yc, is_adding, next, L::SUM, local, L::SUM, PL);
  eval_eq(yc, is_adding, next, L::DOUBLE, next, L::C, PL);
  yc.constraint(is_adding * next(L::IS_ADDING));
  yc.constraint(is_adding * (next(L::IDNL) - is_next_not_last_round));
  for (int i = 0; i < 256; i++) yc.constraint(is_adding * (next(L::BITS + i) - local(L::BITS + i)));
  eval_round_flags(yc, 512, filter, local, L::FLAGS, next(L::FLAGS + 2));
  yc.constraint(is_not_last_round * (next(L::TIMESTAMP) - local(L::TIMESTAMP)));
  yc.constraint(is_not_last_round * (next(L::FILTER) - filter));
  const E diff = next(L::RANGE) - local(L::RANGE);
  yc.constraint_transition(diff * diff - diff);
  yc.constraint_last_row(local(L::RANGE) - E((u64)65535));
}

// Lookup (LogUp) and cross-table-lookup checks on the auxiliary polynomials: aux = [ch 0: h_0 .. h_{m-1}, Z][ch 1: ...]
// [Z of (ctl 0, ch 0), (ctl 0, ch 1), (ctl 1, ch 0), (ctl 1, ch 1)]
inline void eval_lookups_and_ctls(Consumer& yc, const StarkShape& sh, const Row& local, const Row& next, const Row& aux,
                                  const Row& aux_next, const u64 betas[2], const u64 gammas[2]) {
  const int m = sh.n_helpers(), n = sh.n_rc();
  for (int ch = 0; ch < 2; ch++) {
    const E x = E(betas[ch]);
    E sum_h = ZERO;
    for (int k = 0; k < m; k++) {
      const E h = aux(ch * (m + 1) + k), g0 = x + local(sh.rc_begin + 2 * k);
      if (2 * k + 1 < n) {
        const E g1 = x + local(sh.rc_begin + 2 * k + 1);
        yc.constraint(h * g0 * g1 - g0 - g1);
      } else {
        yc.constraint(h * g0 - ONE);
      }
      sum_h += h;
    }
    const E z = aux(ch * (m + 1) + m), z_next = aux_next(ch * (m + 1) + m), t = x + local(sh.table_col);
    yc.constraint_first_row(z);
    yc.constraint((z_next - z) * t - sum_h * t + local(sh.freq_col));
  }
  for (int ctl = 0; ctl < sh.n_ctl; ctl++) {
    const E filter = local(sh.ctl.filter_col[ctl]);
    for (int ch = 0; ch < 2; ch++) {
      E comb = ZERO;  // sum_i v_i beta^i + gamma
      for (int i = sh.ctl.ncols[ctl] - 1; i >= 0; i--) {
        E val = ZERO;
        for (int b = sh.ctl.col_bits[ctl][i] - 1; b >= 0; b--) val = 2 * val + local(sh.ctl.col_start[ctl][i] + b);
        comb = comb * E(betas[ch]) + val;
      }
      comb += E(gammas[ch]);
      const int zc = 2 * (m + 1) + 2 * ctl + ch;
      const E z = aux(zc), z_next = aux_next(zc);
      yc.constraint_last_row(comb * z - filter);
      yc.constraint_transition(comb * (z - z_next) - filter);
    }
  }
}

// sum_e c_e alpha_j^(K-1-e) for both challenges at the opened rows; false (with a message) on an internal count mismatch.
inline bool vanishing_on_host(int kind, const StarkShape& sh, const u64* local_w, const u64* next_w, const u64* aux_w,
                              const u64* aux_next_w, const u64 alphas[2], const u64 betas[2], const u64 gammas[2], gl2 z_last,
                              gl2 l_first, gl2 l_last, gl2 out[2], std::string& err) {
  Consumer yc;
  for (int j = 0; j < 2; j++) yc.alpha[j] = E(alphas[j]);
  yc.z_last = E(z_last);
  yc.l_first = E(l_first);
  yc.l_last = E(l_last);
  const Row local{local_w}, next{next_w}, aux{aux_w}, aux_next{aux_next_w};
  if (kind == KIND_G1) {
    eval_g1_add(yc, local(G1L::FILTER), local, G1L::A, G1L::B, G1L::C, G1L::AUX);
    eval_schedule<G1L>(yc, local, next, false);
  } else if (kind == KIND_G2) {
    eval_g2_add(yc, local(G2L::FILTER), local, G2L::A, G2L::B, G2L::C, G2L::AUX);
    eval_schedule<G2L>(yc, local, next, false);
  } else {
    eval_fq_mul(yc, local(FQL::FILTER), local, FQL::A, FQL::B, FQL::C, FQL::AUX);
    eval_schedule<FQL>(yc, local, next, true);
  }
  if (yc.count != sh.n_constraints) {
    err = "internal: host AIR emitted " + std::to_string(yc.count) + " constraints, expected " + std::to_string(sh.n_constraints);
    return false;
  }
  eval_lookups_and_ctls(yc, sh, local, next, aux, aux_next, betas, gammas);
  if (yc.count != sh.n_total_constraints()) {
    err = "internal: host AIR emitted " + std::to_string(yc.count) + " constraints in total";
    return false;
  }
  out[0] = yc.acc[0].v;
  out[1] = yc.acc[1].v;
  return true;
}

}  // namespace host_air
