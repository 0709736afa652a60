// ORACLE (test infrastructure, NOT the product path).
// Limb-polynomial arithmetic and the "input == 0 (mod p)" argument: restates reference
// src/starks/modular/pol_utils.rs:74-363, modular/modulus_zero.rs:66-198,
// modular/is_modulus_zero.rs:27-84 and common/round_flags.rs:11-81, common/eq.rs:25-51.
#pragma once
#include "bn254.hpp"
#include "stark.hpp"
#include <array>

namespace orc {

static const int NL = 16;            // N_LIMBS (starks/mod.rs:13)
static const int MZ_LEN = 5 * NL;    // MODULUS_AUX_ZERO_LEN (modulus_zero.rs:60)
static const int IMZ_LEN = NL + MZ_LEN;  // IS_MODULUS_AUX_ZERO_LEN (is_modulus_zero.rs:25)
static const int64_t AUX_COEFF_ABS_MAX = 1LL << 29;

// ---- generic limb polynomials (pol_utils.rs) ------------------------------------------------------
template <class T> using Pol16 = std::array<T, 16>;
template <class T> using Pol31 = std::array<T, 31>;
template <class T> using Pol32 = std::array<T, 32>;

template <class T> static inline T tzero();
template <> inline int64_t tzero<int64_t>() { return 0; }
template <> inline F tzero<F>() { return F(0); }
template <> inline F2 tzero<F2>() { return F2(0); }
template <class T> static inline T tconst(u64 v);
template <> inline int64_t tconst<int64_t>(u64 v) { return (int64_t)v; }
template <> inline F tconst<F>(u64 v) { return F::from_u64(v); }
template <> inline F2 tconst<F2>(u64 v) { return F2::from_u64(v); }

template <class T> static Pol31<T> pol_zero31() { Pol31<T> r; r.fill(tzero<T>()); return r; }
template <class T> static Pol31<T> pol_mul_wide(const Pol16<T>& a, const Pol16<T>& b) {  // pol_utils.rs:207-218
  Pol31<T> r = pol_zero31<T>();
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) r[i + j] = r[i + j] + a[i] * b[j];
  return r;
}
template <class T> static Pol31<T> pol_widen(const Pol16<T>& a) {  // pol_add/pol_sub style zero extension
  Pol31<T> r = pol_zero31<T>();
  for (int i = 0; i < 16; i++) r[i] = a[i];
  return r;
}
template <class T, size_t N> static std::array<T, N> pol_addn(const std::array<T, N>& a, const std::array<T, N>& b) {
  std::array<T, N> r;
  for (size_t i = 0; i < N; i++) r[i] = a[i] + b[i];
  return r;
}
template <class T, size_t N> static std::array<T, N> pol_subn(const std::array<T, N>& a, const std::array<T, N>& b) {
  std::array<T, N> r;
  for (size_t i = 0; i < N; i++) r[i] = a[i] - b[i];
  return r;
}
template <class T, size_t N> static std::array<T, N> pol_scale(const std::array<T, N>& a, T c) {
  std::array<T, N> r;
  for (size_t i = 0; i < N; i++) r[i] = c * a[i];
  return r;
}

static inline const int64_t* modulus_limbs_i64() {
  static int64_t m[16];
  static bool init = false;
  if (!init) {
    u256_to_limbs(BN_P, m);
    init = true;
  }
  return m;
}

// ---- trace generation side --------------------------------------------------------------------------
// generate_modulus_zero (modulus_zero.rs:77-123): writes the 80 values
// [is_quot_positive, quot_abs(17), aux_lo(31), aux_hi(31)] as canonical field elements.
static inline void generate_modulus_zero(const Pol31<int64_t>& input, u64* out /*80*/) {
  BigS in = bigs_from_columns(input.data(), 31);
  BigS quot;
  if (!in.is_zero()) {
    std::vector<uint32_t> q, r;
    if (in.mag.size() < 2) {
      r = in.mag;
    } else {
      mag_divmod(in.mag, bn_p_mag(), q, r);
    }
    if (!r.empty()) throw std::runtime_error("generate_modulus_zero: input not divisible by modulus");
    quot.mag = q;
    quot.neg = in.neg && !q.empty();
  }
  u64 is_quot_positive = (!quot.is_zero() && !quot.neg) ? 1 : 0;
  int64_t quot_limbs[17], quot_abs[17];
  bigs_to_columns(quot, quot_limbs, 17);
  BigS qa = quot;
  qa.neg = false;
  bigs_to_columns(qa, quot_abs, 17);
  const int64_t* mod = modulus_limbs_i64();
  int64_t constr[32];
  for (int i = 0; i < 31; i++) constr[i] = input[i];
  constr[31] = 0;
  for (int i = 0; i < 17; i++)
    for (int j = 0; j < 16; j++) constr[i + j] -= quot_limbs[i] * mod[j];  // pol_mul_wide2 + pol_sub_assign
  // pol_remove_root_2exp::<16> (pol_utils.rs:339-363)
  int64_t aux[32];
  aux[0] = -(constr[0] >> 16);
  for (int d = 1; d < 31; d++) aux[d] = (aux[d - 1] - constr[d]) >> 16;
  aux[31] = 0;
  for (int d = 0; d < 32; d++) {
    aux[d] += AUX_COEFF_ABS_MAX;
    if (aux[d] < 0 || aux[d] > 2 * AUX_COEFF_ABS_MAX) throw std::runtime_error("generate_modulus_zero: aux coefficient out of range");
  }
  out[0] = is_quot_positive;
  for (int i = 0; i < 17; i++) out[1 + i] = (u64)quot_abs[i];
  for (int i = 0; i < 31; i++) out[18 + i] = (u64)(uint16_t)aux[i];
  for (int i = 0; i < 31; i++) out[49 + i] = (u64)(uint16_t)(aux[i] >> 16);
}

// generate_is_modulus_zero (is_modulus_zero.rs:36-66): returns is_zero; writes inv(16) + mz aux(80).
static inline u64 generate_is_modulus_zero(const Pol16<int64_t>& input, u64* out /*96*/) {
  BigS in = bigs_from_columns(input.data(), 16);
  U256 red = bigs_mod_p(in);
  U256 inv = {{0, 0, 0, 0}};
  if (!red.is_zero()) inv = fq_to_u256(fq_inv(fq_from_u256(red)));
  u64 is_zero = inv.is_zero() ? 1 : 0;
  Pol16<int64_t> inv_l;
  u256_to_limbs(inv, inv_l.data());
  Pol31<int64_t> diff = pol_mul_wide<int64_t>(input, inv_l);
  diff[0] += (int64_t)is_zero - 1;
  for (int i = 0; i < 16; i++) out[i] = (u64)inv_l[i];
  generate_modulus_zero(diff, out + 16);
  return is_zero;
}

// ---- constraint side ------------------------------------------------------------------------------
template <class T> static Pol16<T> modulus_T() {
  Pol16<T> m;
  const int64_t* ml = modulus_limbs_i64();
  for (int i = 0; i < 16; i++) m[i] = tconst<T>((u64)ml[i]);
  return m;
}

// eval_modulus_zero (modulus_zero.rs:163-198); aux points at the 80-value block.
template <class T> static void eval_modulus_zero(Consumer<T>& cc, T filter, const Pol31<T>& input, const T* aux) {
  T iqp = aux[0];
  cc.constraint(filter * (iqp * iqp - iqp));
  T quot_sign = tconst<T>(2) * iqp - tconst<T>(1);
  Pol16<T> mod = modulus_T<T>();
  T constr[32];
  for (int i = 0; i < 32; i++) constr[i] = tzero<T>();
  for (int i = 0; i < 17; i++) {
    T q = quot_sign * aux[1 + i];
    for (int j = 0; j < 16; j++) constr[i + j] = constr[i + j] + q * mod[j];
  }
  T base = tconst<T>(1ULL << 16), offset = tconst<T>((u64)AUX_COEFF_ABS_MAX);
  T ap[32];
  for (int i = 0; i < 31; i++) ap[i] = (aux[18 + i] - offset) + base * aux[49 + i];
  ap[31] = tzero<T>();
  // pol_adjoin_root(aux_poly, base): res[0] = -base*a[0]; res[d] = a[d-1] - base*a[d]
  constr[0] = constr[0] + (tzero<T>() - base * ap[0]);
  for (int d = 1; d < 32; d++) constr[d] = constr[d] + (ap[d - 1] - base * ap[d]);
  for (int i = 0; i < 31; i++) constr[i] = constr[i] - input[i];
  for (int i = 0; i < 32; i++) cc.constraint(filter * constr[i]);
}

// eval_is_modulus_zero (is_modulus_zero.rs:69-84); aux points at inv(16) + mz(80).
template <class T> static void eval_is_modulus_zero(Consumer<T>& cc, T filter, const Pol16<T>& input, T is_zero, const T* aux) {
  Pol16<T> inv;
  for (int i = 0; i < 16; i++) inv[i] = aux[i];
  Pol31<T> diff = pol_mul_wide<T>(input, inv);
  diff[0] = diff[0] + (is_zero - tconst<T>(1));
  eval_modulus_zero<T>(cc, filter, diff, aux + 16);
  for (int i = 0; i < 16; i++) cc.constraint(filter * (input[i] * is_zero));
}

// EvalEq (eq.rs:25-51)
template <class T> static void eval_eq(Consumer<T>& cc, T filter, T a, T b) { cc.constraint(filter * (a - b)); }
template <class T> static void eval_eq_n(Consumer<T>& cc, T filter, const T* a, const T* b, int n) {
  for (int i = 0; i < n; i++) eval_eq<T>(cc, filter, a[i], b[i]);
}

// generate_round_flags (round_flags.rs:21-44): [is_first, is_last, counter, inv_counter, inv_counter_prime]
static inline void generate_round_flags(size_t row_index, size_t period, u64 out[5]) {
  u64 counter = row_index % period;
  u64 counter_prime = gl_sub(counter, period - 1);
  out[0] = counter == 0;
  out[1] = counter_prime == 0;
  out[2] = counter;
  out[3] = counter == 0 ? 0 : gl_inv(counter);
  out[4] = counter_prime == 0 ? 0 : gl_inv(counter_prime);
}
// eval_round_flags (round_flags.rs:46-81); rf = the 5 round-flag values of the local row.
template <class T> static void eval_round_flags(Consumer<T>& cc, u64 period, T filter, const T* rf, T next_counter) {
  T one = tconst<T>(1);
  T is_first = rf[0], is_last = rf[1], counter = rf[2], inv_counter = rf[3], inv_counter_prime = rf[4];
  T not_filter = one - filter;
  cc.constraint(not_filter * is_first);
  cc.constraint(not_filter * is_last);
  T is_first_minus_one = one - is_first;
  cc.constraint(filter * (counter * inv_counter - is_first_minus_one));
  cc.constraint(filter * counter * is_first);
  T counter_prime = counter - tconst<T>(period - 1);
  T is_last_minus_one = one - is_last;
  cc.constraint(filter * (counter_prime * inv_counter_prime - is_last_minus_one));
  cc.constraint(filter * counter_prime * is_last);
  T is_not_last = one - is_last;
  cc.constraint(filter * is_not_last * (next_counter - counter - one));
  cc.constraint(filter * is_last * next_counter);
}

}  // namespace orc
