// ORACLE (test infrastructure, NOT the product path).
// The 512-row double-and-add / square-and-multiply schedule shared by the three STARKs.  Restates the
// common structure of reference src/starks/curves/g1/scalar_mul_stark.rs:92-339 (G1), its textual twin
// src/starks/curves/g2/scalar_mul_stark.rs (G2) and src/starks/fields/exp_stark.rs:92-327 (Fq exp), and the
// `#[repr(C)]` views scalar_mul_view.rs:34-49 / exp_view.rs:31-48:
//   [double|square (PL)] [sum|product (PL)] [a (PL)] [b (PL)] [c (PL)] [op aux (AUXL)] [bits 256]
//   [round_flags 5] [timestamp, is_adding|is_mul, is_doubling_not_last|is_sq_not_last, filter, frequency, range_counter]
#pragma once
#include "modular.hpp"

namespace orc {

struct Layout {
  int PL, AUXL;
  int DOUBLE, SUM, A, B, C, AUX, BITS, FLAGS, TIMESTAMP, IS_ADDING, IDNL, FILTER, FREQ, RANGE, W, RC_BEGIN, RC_END;
  Layout(int pl, int auxl) : PL(pl), AUXL(auxl) {
    DOUBLE = 0;
    SUM = pl;
    A = 2 * pl;
    B = 3 * pl;
    C = 4 * pl;
    AUX = 5 * pl;
    BITS = AUX + auxl;
    FLAGS = BITS + 256;
    TIMESTAMP = FLAGS + 5;
    IS_ADDING = TIMESTAMP + 1;
    IDNL = TIMESTAMP + 2;
    FILTER = TIMESTAMP + 3;
    FREQ = TIMESTAMP + 4;
    RANGE = TIMESTAMP + 5;
    W = RANGE + 1;
    RC_BEGIN = 2 * pl;
    RC_END = 5 * pl + auxl;
  }
};

// One instance = 512 rows.  Ops supplies the group / field operation:
//   typename Elem; void to_limbs(const Elem&, u64* out /*PL*/);
//   Elem op(const Elem& a, const Elem& b, const u64* al, const u64* bl, u64* cl, u64* aux);  // c = a (+|*) b
template <class Ops>
static typename Ops::Elem generate_one_set(const Layout& L, const Ops& ops, const u64 s[4], typename Ops::Elem dbl,
                                           typename Ops::Elem sum, u64 timestamp, u64* rows /* 512 x W, zeroed */) {
  typedef typename Ops::Elem Elem;
  u64 bits[256];
  for (int i = 0; i < 256; i++) bits[i] = (s[i / 64] >> (i % 64)) & 1;
  for (int r = 0; r < 512; r++) {
    u64* row = rows + (size_t)r * L.W;
    bool adding = (r % 2) == 0;
    Elem a, b;
    if (adding) {
      if (r > 0) {  // rotate bits left
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
    ops.to_limbs(a, row + L.A);
    ops.to_limbs(b, row + L.B);
    Elem c = ops.op(a, b, row + L.A, row + L.B, row + L.C, row + L.AUX);
    if (adding) {
      if (bits[0]) sum = c;
    } else {
      dbl = c;
    }
    ops.to_limbs(dbl, row + L.DOUBLE);
    ops.to_limbs(sum, row + L.SUM);
    for (int i = 0; i < 256; i++) row[L.BITS + i] = bits[i];
    generate_round_flags(r, 512, row + L.FLAGS);
    row[L.TIMESTAMP] = timestamp;
    row[L.IS_ADDING] = adding ? 1 : 0;
    row[L.IDNL] = adding ? 0 : (1 - row[L.FLAGS + 1]);
    row[L.FILTER] = 1;
  }
  return sum;
}

// generate_trace + generate_range_checks: `one_set(k, rows)` fills the 512 rows of instance k.
template <class OneSet>
static std::vector<std::vector<u64>> generate_trace_generic(const Layout& L, size_t n, size_t min_rows, OneSet one_set) {
  size_t num_rows = std::max(min_rows, n * (size_t)512), p2 = 1;
  while (p2 < num_rows) p2 <<= 1;
  num_rows = p2;
  std::vector<std::vector<u64>> trace(L.W, std::vector<u64>(num_rows, 0));
  std::string err;
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t k = 0; k < n; k++) {
    std::vector<u64> rows((size_t)512 * L.W, 0);
    try {
      one_set(k, rows.data());
    } catch (std::exception& e) {
#pragma omp critical
      err = e.what();
    }
    for (int r = 0; r < 512; r++)
      for (int c = 0; c < L.W; c++) trace[c][k * 512 + r] = rows[(size_t)r * L.W + c];
  }
  if (!err.empty()) throw std::runtime_error(err);
  const size_t range_max = 1 << 16;
  for (size_t i = 0; i < num_rows; i++) trace[L.RANGE][i] = i < range_max ? i : range_max - 1;
  std::vector<u64>& freq = trace[L.FREQ];
  for (int c = L.RC_BEGIN; c < L.RC_END; c++)
    for (size_t i = 0; i < num_rows; i++) {
      u64 x = trace[c][i];
      if (x >= range_max) throw std::runtime_error("range check value out of range");
      if (x >= num_rows) throw std::runtime_error("index out of bounds: fewer than 2^16 rows (reference rows[x][FREQ_COL])");
      freq[x] += 1;
    }
  return trace;
}

// The schedule part of eval_packed_generic (everything after the add/mul check).  `first_a_is_one`
// inserts the Fq-exp constraint "first round, a = 1" (exp_stark.rs:250-256).
template <class T>
static void eval_schedule(const Layout& L, const T* local, const T* next, Consumer<T>& cc, bool first_a_is_one) {
  const int PL = L.PL;
  T one = tconst<T>(1), zero = tzero<T>();
  T l_filter = local[L.FILTER], n_filter = next[L.FILTER];
  T is_not_last_round = l_filter - local[L.FLAGS + 1];
  T is_next_not_last_round = n_filter - next[L.FLAGS + 1];
  T is_first = local[L.FLAGS + 0];
  eval_eq<T>(cc, is_first, local[L.IS_ADDING], one);
  eval_eq_n<T>(cc, is_first, local + L.DOUBLE, local + L.B, PL);
  T bit0 = local[L.BITS];
  eval_eq_n<T>(cc, bit0 * is_first, local + L.SUM, local + L.C, PL);
  eval_eq_n<T>(cc, (one - bit0) * is_first, local + L.SUM, local + L.A, PL);
  if (first_a_is_one) {
    for (int i = 0; i < PL; i++) eval_eq<T>(cc, is_first, local[L.A + i], i == 0 ? one : zero);
  }
  T idnl = local[L.IDNL];
  eval_eq_n<T>(cc, idnl, next + L.A, local + L.SUM, PL);
  eval_eq_n<T>(cc, idnl, next + L.B, local + L.DOUBLE, PL);
  eval_eq_n<T>(cc, next[L.BITS] * idnl, next + L.SUM, next + L.C, PL);
  eval_eq_n<T>(cc, (one - next[L.BITS]) * idnl, next + L.SUM, next + L.A, PL);
  eval_eq_n<T>(cc, idnl, next + L.DOUBLE, local + L.DOUBLE, PL);
  eval_eq<T>(cc, idnl, next[L.IS_ADDING], one);
  eval_eq<T>(cc, idnl, next[L.IDNL], zero);
  for (int i = 0; i < 256; i++) eval_eq<T>(cc, idnl, next[L.BITS + i], local[L.BITS + (i + 1) % 256]);
  T is_adding = local[L.IS_ADDING];
  eval_eq_n<T>(cc, is_adding, next + L.A, local + L.DOUBLE, PL);
  eval_eq_n<T>(cc, is_adding, next + L.B, local + L.DOUBLE, PL);
  eval_eq_n<T>(cc, is_adding, next + L.SUM, local + L.SUM, PL);
  eval_eq_n<T>(cc, is_adding, next + L.DOUBLE, next + L.C, PL);
  eval_eq<T>(cc, is_adding, next[L.IS_ADDING], zero);
  eval_eq<T>(cc, is_adding, next[L.IDNL], is_next_not_last_round);
  for (int i = 0; i < 256; i++) eval_eq<T>(cc, is_adding, next[L.BITS + i], local[L.BITS + i]);
  eval_round_flags<T>(cc, 512, l_filter, local + L.FLAGS, next[L.FLAGS + 2]);
  eval_eq<T>(cc, is_not_last_round, next[L.TIMESTAMP], local[L.TIMESTAMP]);
  eval_eq<T>(cc, is_not_last_round, next[L.FILTER], local[L.FILTER]);
  T diff = next[L.RANGE] - local[L.RANGE];
  cc.constraint_transition(diff * diff - diff);
  cc.constraint_last_row(local[L.RANGE] - tconst<T>((1 << 16) - 1));
}

// Stark::lookups + the two looked CTL tables (scalar_mul_ctl.rs:20-55 / exp_ctl.rs:18-52):
// input = b | [a] | 16 x le_bits(bits) | timestamp (filter is_first_round); output = sum | timestamp.
static inline void fill_stark_def(StarkDef& d, const Layout& L, bool input_has_a) {
  d.W = L.W;
  d.lookup_begin = L.RC_BEGIN;
  d.lookup_end = L.RC_END;
  d.table_col = L.RANGE;
  d.freq_col = L.FREQ;
  CtlDef in, out;
  for (int i = 0; i < L.PL; i++) in.cols.push_back(LinComb::single(L.B + i));
  if (input_has_a)
    for (int i = 0; i < L.PL; i++) in.cols.push_back(LinComb::single(L.A + i));
  for (int k = 0; k < 16; k++) in.cols.push_back(LinComb::le_bits(L.BITS + 16 * k, 16));
  in.cols.push_back(LinComb::single(L.TIMESTAMP));
  in.filter_col = L.FLAGS + 0;
  for (int i = 0; i < L.PL; i++) out.cols.push_back(LinComb::single(L.SUM + i));
  out.cols.push_back(LinComb::single(L.TIMESTAMP));
  out.filter_col = L.FLAGS + 1;
  d.ctls = {in, out};
}

}  // namespace orc
