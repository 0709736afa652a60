// ORACLE (test infrastructure, NOT the product path).
// G1 scalar-multiplication STARK: trace generation and AIR.  Restates reference
// src/starks/curves/g1/add.rs:36-185 (G1AddAux, generate_g1_add, eval_g1_add),
// src/starks/curves/g1/scalar_mul_view.rs:10-49 (column layout),
// src/starks/curves/g1/scalar_mul_stark.rs:37-339,493-500 (inputs, generate_trace, eval_packed_generic,
// lookups) and src/starks/curves/g1/scalar_mul_ctl.rs:20-80 (CTL tables and extra looking values).
#pragma once
#include "modular.hpp"

namespace orc {
namespace g1 {

// Column map (scalar_mul_view.rs:34-49 `#[repr(C)]` order; pinned by row_position_correctness :97-117)
static const int G1_LEN = 32;
static const int N_BITS = 256, PERIOD = 512;
static const int ADD_AUX_LEN = 1 + IMZ_LEN + 1 + NL + 3 * MZ_LEN;  // 354 (add.rs:33-34)
static const int COL_DOUBLE = 0, COL_SUM = 32, COL_A = 64, COL_B = 96, COL_C = 128, COL_AUX = 160;
static const int AUX_IS_X_EQ = 0, AUX_IS_X_EQ_AUX = 1, AUX_IS_X_EQ_FILTER = 1 + IMZ_LEN, AUX_LAMBDA = 2 + IMZ_LEN,
                 AUX_LAMBDA_AUX = 2 + IMZ_LEN + NL, AUX_X_AUX = AUX_LAMBDA_AUX + MZ_LEN, AUX_Y_AUX = AUX_X_AUX + MZ_LEN;
static const int COL_BITS = COL_AUX + ADD_AUX_LEN;  // 514
static const int COL_FLAGS = COL_BITS + N_BITS;     // 770: is_first, is_last, counter, inv_counter, inv_counter_prime
static const int COL_TIMESTAMP = COL_FLAGS + 5, COL_IS_ADDING = COL_TIMESTAMP + 1, COL_IDNL = COL_TIMESTAMP + 2,
                 COL_FILTER = COL_TIMESTAMP + 3, COL_FREQ = COL_TIMESTAMP + 4, COL_RANGE = COL_TIMESTAMP + 5;
static const int W = COL_RANGE + 1;  // 781
static const int RC_BEGIN = 2 * G1_LEN, RC_END = 5 * G1_LEN + ADD_AUX_LEN;  // 64..514

struct Affine {
  Fq x, y;
};
struct Input {       // G1ScalarMulInput (scalar_mul_stark.rs:37-41) in ABI wire form
  u64 s[4];          // 256-bit scalar, little-endian words, NOT reduced mod r
  U256 x[2], off[2];  // canonical affine coordinates
};

static inline void point_to_limbs(const Affine& p, u64* out /*32*/) {
  int64_t l[16];
  u256_to_limbs(fq_to_u256(p.x), l);
  for (int i = 0; i < 16; i++) out[i] = (u64)l[i];
  u256_to_limbs(fq_to_u256(p.y), l);
  for (int i = 0; i < 16; i++) out[16 + i] = (u64)l[i];
}

// c = a + b for affine non-infinity points with b != -a (what `a_ark + b_ark` yields at add.rs:56).
static inline Affine affine_add(const Affine& a, const Affine& b, Fq* lambda_out) {
  Fq lam;
  if (!(a.x == b.x)) {
    lam = fq_mul(fq_sub(b.y, a.y), fq_inv(fq_sub(b.x, a.x)));  // add.rs:66
  } else {
    if (!(a.y == b.y) || a.y.is_zero()) throw std::runtime_error("generate_g1_add: b == -a (point at infinity)");
    Fq three = fq_from_u64(3), two = fq_from_u64(2);
    lam = fq_mul(fq_mul(three, fq_mul(a.x, a.x)), fq_inv(fq_mul(two, a.y)));  // add.rs:80
  }
  Affine c;
  c.x = fq_sub(fq_sub(fq_mul(lam, lam), a.x), b.x);
  c.y = fq_sub(fq_mul(lam, fq_sub(a.x, c.x)), a.y);
  if (lambda_out) *lambda_out = lam;
  return c;
}

static inline Pol16<int64_t> limbs_i64(const u64* p) {
  Pol16<int64_t> r;
  for (int i = 0; i < 16; i++) r[i] = (int64_t)p[i];
  return r;
}

// generate_g1_add (add.rs:52-122): a, b given as 32 limbs; writes c (32 limbs) and the 354 aux values.
static inline Affine generate_g1_add(const Affine& a, const Affine& b, const u64* al, const u64* bl, u64* cl, u64* aux) {
  Fq lam;
  Affine c = affine_add(a, b, &lam);
  point_to_limbs(c, cl);
  Pol16<int64_t> ax = limbs_i64(al), ay = limbs_i64(al + 16), bx = limbs_i64(bl), by = limbs_i64(bl + 16), cx = limbs_i64(cl),
                 cy = limbs_i64(cl + 16);
  Pol16<int64_t> delta_x = pol_subn(bx, ax);
  u64 is_x_eq = generate_is_modulus_zero(delta_x, aux + AUX_IS_X_EQ_AUX);
  aux[AUX_IS_X_EQ] = is_x_eq;
  aux[AUX_IS_X_EQ_FILTER] = is_x_eq;  // add.rs:108-110
  int64_t lam_l64[16];
  u256_to_limbs(fq_to_u256(lam), lam_l64);
  Pol16<int64_t> lam_l;
  for (int i = 0; i < 16; i++) {
    lam_l[i] = lam_l64[i];
    aux[AUX_LAMBDA + i] = (u64)lam_l64[i];
  }
  if (!is_x_eq) {
    // diff = lambda*(b.x - a.x) - (b.y - a.y)   (add.rs:70-74)
    Pol31<int64_t> delta_y = pol_widen(pol_subn(by, ay));
    Pol31<int64_t> diff = pol_subn(pol_mul_wide<int64_t>(lam_l, delta_x), delta_y);
    generate_modulus_zero(diff, aux + AUX_LAMBDA_AUX);
  } else {
    // diff = 2*a.y*lambda - 3*a.x^2   (add.rs:84-90)
    Pol31<int64_t> three_x_sq = pol_scale<int64_t, 31>(pol_mul_wide<int64_t>(ax, ax), 3);
    Pol31<int64_t> two_lambda_y = pol_scale<int64_t, 31>(pol_mul_wide<int64_t>(lam_l, ay), 2);
    generate_modulus_zero(pol_subn(two_lambda_y, three_x_sq), aux + AUX_LAMBDA_AUX);
  }
  // diff = lambda^2 - (a.x + b.x + c.x)   (add.rs:94-99)
  Pol31<int64_t> sum_x = pol_widen(pol_addn(pol_addn(ax, bx), cx));
  generate_modulus_zero(pol_subn(pol_mul_wide<int64_t>(lam_l, lam_l), sum_x), aux + AUX_X_AUX);
  // diff = lambda*(c.x - a.x) + c.y + a.y   (add.rs:101-106)
  Pol31<int64_t> cyay = pol_widen(pol_addn(cy, ay));
  generate_modulus_zero(pol_addn(pol_mul_wide<int64_t>(lam_l, pol_subn(cx, ax)), cyay), aux + AUX_Y_AUX);
  return c;
}

// generate_one_set (scalar_mul_stark.rs:92-213): fills rows [row0, row0+512) of a row-major buffer.
// Returns the final running sum (== s*x + offset, asserted by the reference at :105-108).
static inline Affine generate_one_set(const Input& in, u64 timestamp, u64* rows /* 512 x W, zeroed */) {
  Affine x{fq_from_u256(in.x[0]), fq_from_u256(in.x[1])}, off{fq_from_u256(in.off[0]), fq_from_u256(in.off[1])};
  Affine dbl = x, sum = off;
  u64 bits[256];
  for (int i = 0; i < 256; i++) bits[i] = (in.s[i / 64] >> (i % 64)) & 1;
  for (int r = 0; r < PERIOD; r++) {
    u64* row = rows + (size_t)r * W;
    bool adding = (r % 2) == 0;
    Affine a, b;
    if (adding) {
      if (r > 0) {  // rotate bits left (scalar_mul_stark.rs:163-167)
        u64 b0 = bits[0];
        for (int i = 0; i < 255; i++) bits[i] = bits[i + 1];
        bits[255] = b0;
      }
      a = sum;
      b = dbl;
    } else {
      a = dbl;
      b = dbl;
    }
    point_to_limbs(a, row + COL_A);
    point_to_limbs(b, row + COL_B);
    Affine c = generate_g1_add(a, b, row + COL_A, row + COL_B, row + COL_C, row + COL_AUX);
    if (adding) {
      if (bits[0]) sum = c;
    } else {
      dbl = c;
    }
    point_to_limbs(dbl, row + COL_DOUBLE);
    point_to_limbs(sum, row + COL_SUM);
    for (int i = 0; i < 256; i++) row[COL_BITS + i] = bits[i];
    generate_round_flags(r, PERIOD, row + COL_FLAGS);
    row[COL_TIMESTAMP] = timestamp;
    row[COL_IS_ADDING] = adding ? 1 : 0;
    row[COL_IDNL] = adding ? 0 : (1 - row[COL_FLAGS + 1]);
    row[COL_FILTER] = 1;
  }
  return sum;
}

// generate_trace (scalar_mul_stark.rs:55-87): column-major output, trace[c][row].
static inline std::vector<std::vector<u64>> generate_trace(const std::vector<Input>& inputs, size_t min_rows,
                                                           std::vector<Affine>* outputs = nullptr) {
  size_t num_rows = std::max(min_rows, inputs.size() * (size_t)PERIOD);
  size_t p2 = 1;
  while (p2 < num_rows) p2 <<= 1;
  num_rows = p2;
  std::vector<std::vector<u64>> trace(W, std::vector<u64>(num_rows, 0));
  if (outputs) outputs->resize(inputs.size());
  std::string err;
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t k = 0; k < inputs.size(); k++) {
    std::vector<u64> rows((size_t)PERIOD * W, 0);
    try {
      Affine out = generate_one_set(inputs[k], k, rows.data());
      if (outputs) (*outputs)[k] = out;
    } catch (std::exception& e) {
#pragma omp critical
      err = e.what();
    }
    for (int r = 0; r < PERIOD; r++)
      for (int c = 0; c < W; c++) trace[c][k * PERIOD + r] = rows[(size_t)r * W + c];
  }
  if (!err.empty()) throw std::runtime_error(err);
  // generate_range_checks (:71-87)
  const size_t range_max = 1 << 16;
  for (size_t i = 0; i < num_rows; i++) trace[COL_RANGE][i] = i < range_max ? i : range_max - 1;
  std::vector<u64>& freq = trace[COL_FREQ];
  for (int c = RC_BEGIN; c < RC_END; c++)
    for (size_t i = 0; i < num_rows; i++) {
      u64 x = trace[c][i];
      if (x >= range_max) throw std::runtime_error("range check value out of range");
      if (x >= num_rows) throw std::runtime_error("index out of bounds: fewer than 2^16 rows (reference rows[x][FREQ_COL], scalar_mul_stark.rs:84)");
      freq[x] += 1;
    }
  return trace;
}

// s*x + offset via the same double-and-add (used for CTL values; reference uses ark mul_bigint, :105-106).
static inline Affine scalar_mul_offset(const Input& in) {
  Affine dbl{fq_from_u256(in.x[0]), fq_from_u256(in.x[1])}, sum{fq_from_u256(in.off[0]), fq_from_u256(in.off[1])};
  for (int i = 0; i < 256; i++) {
    if ((in.s[i / 64] >> (i % 64)) & 1) sum = affine_add(sum, dbl, nullptr);
    if (i < 255) dbl = affine_add(dbl, dbl, nullptr);
  }
  return sum;
}

// g1_generate_ctl_values (scalar_mul_ctl.rs:57-80): [0] inputs (x|offset|s limbs|timestamp), [1] outputs.
static inline std::vector<std::vector<std::vector<u64>>> generate_ctl_values(const std::vector<Input>& inputs,
                                                                               const std::vector<Affine>& outputs) {
  std::vector<std::vector<std::vector<u64>>> e(2);
  for (size_t k = 0; k < inputs.size(); k++) {
    std::vector<u64> in(81), out(33);
    Affine x{fq_from_u256(inputs[k].x[0]), fq_from_u256(inputs[k].x[1])},
        off{fq_from_u256(inputs[k].off[0]), fq_from_u256(inputs[k].off[1])};
    point_to_limbs(x, in.data());
    point_to_limbs(off, in.data() + 32);
    for (int i = 0; i < 16; i++) in[64 + i] = (inputs[k].s[i / 4] >> (16 * (i % 4))) & 0xFFFF;
    in[80] = k;
    point_to_limbs(outputs[k], out.data());
    out[32] = k;
    e[0].push_back(in);
    e[1].push_back(out);
  }
  return e;
}

// ---- AIR -----------------------------------------------------------------------------------------------
template <class T> static Pol16<T> ld16(const T* p) {
  Pol16<T> r;
  for (int i = 0; i < 16; i++) r[i] = p[i];
  return r;
}

// eval_g1_add (add.rs:125-185)
template <class T> static void eval_g1_add(Consumer<T>& cc, T filter, const T* a, const T* b, const T* c, const T* aux) {
  Pol16<T> ax = ld16(a), ay = ld16(a + 16), bx = ld16(b), by = ld16(b + 16), cx = ld16(c), cy = ld16(c + 16);
  Pol16<T> lambda = ld16(aux + AUX_LAMBDA);
  Pol16<T> delta_x = pol_subn(bx, ax);
  eval_is_modulus_zero<T>(cc, filter, delta_x, aux[AUX_IS_X_EQ], aux + AUX_IS_X_EQ_AUX);
  T is_x_eq_filter = aux[AUX_IS_X_EQ_FILTER];
  cc.constraint(filter * aux[AUX_IS_X_EQ] - is_x_eq_filter);
  T is_not_eq_filter = filter - is_x_eq_filter;
  // a.x != b.x
  Pol31<T> lambda_delta_x = pol_mul_wide<T>(lambda, delta_x);
  Pol31<T> delta_y = pol_widen(pol_subn(by, ay));
  eval_modulus_zero<T>(cc, is_not_eq_filter, pol_subn(lambda_delta_x, delta_y), aux + AUX_LAMBDA_AUX);
  // a.x == b.x
  Pol31<T> three_x_sq = pol_scale<T, 31>(pol_mul_wide<T>(ax, ax), tconst<T>(3));
  Pol31<T> two_lambda_y = pol_scale<T, 31>(pol_mul_wide<T>(lambda, ay), tconst<T>(2));
  eval_modulus_zero<T>(cc, is_x_eq_filter, pol_subn(two_lambda_y, three_x_sq), aux + AUX_LAMBDA_AUX);
  eval_eq_n<T>(cc, is_x_eq_filter, a + 16, b + 16, 16);
  // lambda^2 - (a.x + b.x + c.x)
  Pol31<T> sum_x = pol_widen(pol_addn(pol_addn(ax, bx), cx));
  eval_modulus_zero<T>(cc, filter, pol_subn(pol_mul_wide<T>(lambda, lambda), sum_x), aux + AUX_X_AUX);
  // lambda*(c.x - a.x) + c.y + a.y
  Pol31<T> cyay = pol_widen(pol_addn(cy, ay));
  eval_modulus_zero<T>(cc, filter, pol_addn(pol_mul_wide<T>(lambda, pol_subn(cx, ax)), cyay), aux + AUX_Y_AUX);
}

// eval_packed_generic (scalar_mul_stark.rs:226-339)
template <class T> static void eval_constraints(const T* local, const T* next, Consumer<T>& cc) {
  T one = tconst<T>(1), zero = tzero<T>();
  T l_filter = local[COL_FILTER], n_filter = next[COL_FILTER];
  T is_not_last_round = l_filter - local[COL_FLAGS + 1];
  T is_next_not_last_round = n_filter - next[COL_FLAGS + 1];
  eval_g1_add<T>(cc, l_filter, local + COL_A, local + COL_B, local + COL_C, local + COL_AUX);
  T is_first = local[COL_FLAGS + 0];
  eval_eq<T>(cc, is_first, local[COL_IS_ADDING], one);
  eval_eq_n<T>(cc, is_first, local + COL_DOUBLE, local + COL_B, 32);
  T bit0 = local[COL_BITS];
  eval_eq_n<T>(cc, bit0 * is_first, local + COL_SUM, local + COL_C, 32);
  eval_eq_n<T>(cc, (one - bit0) * is_first, local + COL_SUM, local + COL_A, 32);
  // doubling_step -> addition_step
  T idnl = local[COL_IDNL];
  eval_eq_n<T>(cc, idnl, next + COL_A, local + COL_SUM, 32);
  eval_eq_n<T>(cc, idnl, next + COL_B, local + COL_DOUBLE, 32);
  eval_eq_n<T>(cc, next[COL_BITS] * idnl, next + COL_SUM, next + COL_C, 32);
  eval_eq_n<T>(cc, (one - next[COL_BITS]) * idnl, next + COL_SUM, next + COL_A, 32);
  eval_eq_n<T>(cc, idnl, next + COL_DOUBLE, local + COL_DOUBLE, 32);
  eval_eq<T>(cc, idnl, next[COL_IS_ADDING], one);
  eval_eq<T>(cc, idnl, next[COL_IDNL], zero);
  for (int i = 0; i < N_BITS; i++) eval_eq<T>(cc, idnl, next[COL_BITS + i], local[COL_BITS + (i + 1) % N_BITS]);
  // addition_step -> doubling_step
  T is_adding = local[COL_IS_ADDING];
  eval_eq_n<T>(cc, is_adding, next + COL_A, local + COL_DOUBLE, 32);
  eval_eq_n<T>(cc, is_adding, next + COL_B, local + COL_DOUBLE, 32);
  eval_eq_n<T>(cc, is_adding, next + COL_SUM, local + COL_SUM, 32);
  eval_eq_n<T>(cc, is_adding, next + COL_DOUBLE, next + COL_C, 32);
  eval_eq<T>(cc, is_adding, next[COL_IS_ADDING], zero);
  eval_eq<T>(cc, is_adding, next[COL_IDNL], is_next_not_last_round);
  for (int i = 0; i < N_BITS; i++) eval_eq<T>(cc, is_adding, next[COL_BITS + i], local[COL_BITS + i]);
  eval_round_flags<T>(cc, PERIOD, l_filter, local + COL_FLAGS, next[COL_FLAGS + 2]);
  eval_eq<T>(cc, is_not_last_round, next[COL_TIMESTAMP], local[COL_TIMESTAMP]);
  eval_eq<T>(cc, is_not_last_round, next[COL_FILTER], local[COL_FILTER]);
  T diff = next[COL_RANGE] - local[COL_RANGE];
  cc.constraint_transition(diff * diff - diff);
  cc.constraint_last_row(local[COL_RANGE] - tconst<T>((1 << 16) - 1));
}

// Stark::lookups (:493-500) + g1_scalar_mul_ctl (scalar_mul_ctl.rs:20-55)
static inline StarkDef stark_def() {
  StarkDef d;
  d.name = "g1_scalar_mul";
  d.W = W;
  d.lookup_begin = RC_BEGIN;
  d.lookup_end = RC_END;
  d.table_col = COL_RANGE;
  d.freq_col = COL_FREQ;
  CtlDef in, out;
  for (int i = 0; i < 32; i++) in.cols.push_back(LinComb::single(COL_B + i));
  for (int i = 0; i < 32; i++) in.cols.push_back(LinComb::single(COL_A + i));
  for (int k = 0; k < 16; k++) in.cols.push_back(LinComb::le_bits(COL_BITS + 16 * k, 16));
  in.cols.push_back(LinComb::single(COL_TIMESTAMP));
  in.filter_col = COL_FLAGS + 0;
  for (int i = 0; i < 32; i++) out.cols.push_back(LinComb::single(COL_SUM + i));
  out.cols.push_back(LinComb::single(COL_TIMESTAMP));
  out.filter_col = COL_FLAGS + 1;
  d.ctls = {in, out};
  d.eval_base = [](const F* l, const F* n, Consumer<F>& cc) { eval_constraints<F>(l, n, cc); };
  d.eval_ext = [](const F2* l, const F2* n, Consumer<F2>& cc) { eval_constraints<F2>(l, n, cc); };
  return d;
}

}  // namespace g1
}  // namespace orc
