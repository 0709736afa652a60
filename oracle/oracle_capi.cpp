// ORACLE (test infrastructure, NOT the product path): C entry points for tests/, smoke() and the
// cpu_baseline leg of bench.py (loaded with ctypes as oracle/liboracle.so).  Nothing in
// plonky2_bn254_amd/ may link or load this library.
#include "g1_stark.hpp"
#include "g2_fq_stark.hpp"
#include <chrono>
#include <cstdio>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {
double now_sec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace orc

using namespace orc;

static thread_local char g_err[512];
static int fail(const std::exception& e) {
  snprintf(g_err, sizeof(g_err), "%s", e.what());
  return -1;
}

static std::vector<g1::Input> unpack_g1(const u64* scalars, const u64* x, const u64* off, size_t n) {
  std::vector<g1::Input> in(n);
  for (size_t k = 0; k < n; k++) {
    memcpy(in[k].s, scalars + 4 * k, 32);
    memcpy(in[k].x[0].l, x + 8 * k, 32);
    memcpy(in[k].x[1].l, x + 8 * k + 4, 32);
    memcpy(in[k].off[0].l, off + 8 * k, 32);
    memcpy(in[k].off[1].l, off + 8 * k + 4, 32);
  }
  return in;
}

extern "C" {

const char* orc_last_error() { return g_err; }
int orc_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
// The restatement stops scaling at a few dozen threads: callers that time it (bench.py's cpu_baseline) pick the thread count
// and report it.
void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// ---- Goldilocks / Poseidon / Merkle ------------------------------------------------------------------
u64 orc_gl_mul(u64 a, u64 b) { return gl_mul(a, b); }
u64 orc_gl_inv(u64 a) { return gl_inv(a); }
void orc_poseidon_permute(u64* state12) { poseidon_permute(state12); }
void orc_poseidon_permute_fast(u64* state12) { poseidon_permute_fast(state12); }
void orc_set_fast_poseidon(int on) { g_fast_poseidon = on != 0; }
// n chained permutations on one state (timing of a single core; tools/oracle_speed.py)
void orc_permute_chain(u64* state12, size_t n, int fast) {
  for (size_t i = 0; i < n; i++) {
    if (fast) poseidon_permute_fast(state12);
    else poseidon_permute(state12);
  }
}
void orc_hash_or_noop(const u64* x, size_t n, u64* out4) {
  Digest d = hash_or_noop(x, n);
  memcpy(out4, d.e, 32);
}
void orc_two_to_one(const u64* l, const u64* r, u64* out4) {
  Digest a, b;
  memcpy(a.e, l, 32);
  memcpy(b.e, r, 32);
  Digest d = two_to_one(a, b);
  memcpy(out4, d.e, 32);
}
// Merkle cap over `n_leaves` leaves (row-major, leaf_len each), in the given order.
void orc_merkle_cap(const u64* leaves, size_t n_leaves, size_t leaf_len, int cap_height, u64* out_cap) {
  std::vector<Digest> d(n_leaves);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n_leaves; i++) d[i] = hash_or_noop(leaves + i * leaf_len, leaf_len);
  MerkleTree t;
  t.build(std::move(d), cap_height);
  for (size_t i = 0; i < t.cap().size(); i++) memcpy(out_cap + 4 * i, t.cap()[i].e, 32);
}
// Challenger replay: observe `n` elements then draw `m` challenges.
void orc_challenger_observe_get(const u64* obs, size_t n, u64* out, size_t m) {
  Challenger ch;
  ch.observe_elements(obs, n);
  for (size_t i = 0; i < m; i++) out[i] = ch.get_challenge();
}

// ---- NTT ---------------------------------------------------------------------------------------------------
// mode 0: fft, 1: ifft, 2: coset_fft(shift), 3: coset_ifft(shift); in place on n base-field elements.
void orc_ntt(u64* data, size_t n, int mode, u64 shift) {
  std::vector<F> v(n);
  for (size_t i = 0; i < n; i++) v[i] = F(data[i]);
  if (mode == 0) fft(v);
  else if (mode == 1) ifft(v);
  else if (mode == 2) coset_fft(v, shift);
  else coset_ifft(v, shift);
  for (size_t i = 0; i < n; i++) data[i] = v[i].v;
}
// PolynomialBatch::from_values on column-major values[C][N]: writes coeffs[C][N], lde[C][2N] in
// bit-reversed (Merkle leaf) order and the 16-digest cap.
void orc_commit_values(const u64* values, size_t C, size_t N, u64* coeffs, u64* lde_bitrev, u64* cap) {
  std::vector<std::vector<u64>> cols(C);
  for (size_t c = 0; c < C; c++) cols[c].assign(values + c * N, values + (c + 1) * N);
  Batch b;
  batch_from_values(b, cols, 1, 4);
  size_t M = 2 * N;
  unsigned lb = log2_strict(M);
  for (size_t c = 0; c < C; c++) {
    if (coeffs) memcpy(coeffs + c * N, b.coeffs[c].data(), 8 * N);
    if (lde_bitrev)
      for (size_t j = 0; j < M; j++) lde_bitrev[c * M + j] = b.lde[c][reverse_bits(j, lb)];
  }
  for (size_t i = 0; i < b.tree.cap().size(); i++) memcpy(cap + 4 * i, b.tree.cap()[i].e, 32);
}

// ---- BN254 ---------------------------------------------------------------------------------------------------
// canonical (non-Montgomery) little-endian 4x64 in/out
void orc_fq_mul(const u64* a, const u64* b, u64* out) {
  U256 x, y;
  memcpy(x.l, a, 32);
  memcpy(y.l, b, 32);
  U256 r = fq_to_u256(fq_mul(fq_from_u256(x), fq_from_u256(y)));
  memcpy(out, r.l, 32);
}
void orc_fq_inv(const u64* a, u64* out) {
  U256 x;
  memcpy(x.l, a, 32);
  U256 r = fq_to_u256(fq_inv(fq_from_u256(x)));
  memcpy(out, r.l, 32);
}
int orc_g1_add(const u64* a8, const u64* b8, u64* out8) {
  try {
    U256 ax, ay, bx, by;
    memcpy(ax.l, a8, 32);
    memcpy(ay.l, a8 + 4, 32);
    memcpy(bx.l, b8, 32);
    memcpy(by.l, b8 + 4, 32);
    g1::Affine a{fq_from_u256(ax), fq_from_u256(ay)}, b{fq_from_u256(bx), fq_from_u256(by)};
    g1::Affine c = g1::affine_add(a, b, nullptr);
    U256 cx = fq_to_u256(c.x), cy = fq_to_u256(c.y);
    memcpy(out8, cx.l, 32);
    memcpy(out8 + 4, cy.l, 32);
    return 0;
  } catch (std::exception& e) {
    return fail(e);
  }
}
// generate_modulus_zero on 31 signed coefficients -> 80 field values. Returns 0 / -1.
int orc_generate_modulus_zero(const int64_t* input31, u64* out80) {
  try {
    Pol31<int64_t> in;
    for (int i = 0; i < 31; i++) in[i] = input31[i];
    generate_modulus_zero(in, out80);
    return 0;
  } catch (std::exception& e) {
    return fail(e);
  }
}

// ---- G1 scalar-mul STARK ---------------------------------------------------------------------------------
int orc_g1_width() { return g1::W; }
size_t orc_g1_num_rows(size_t n, int min_rows_log2) {
  size_t r = std::max((size_t)1 << min_rows_log2, n * 512), p = 1;
  while (p < r) p <<= 1;
  return p;
}
// trace_out: column-major [W][rows]; outputs8: n x 8 canonical coordinates of s*x+offset (may be null).
int orc_g1_generate_trace(const u64* scalars, const u64* x, const u64* off, size_t n, int min_rows_log2, u64* trace_out,
                          u64* outputs8) {
  try {
    auto in = unpack_g1(scalars, x, off, n);
    std::vector<g1::Affine> outs;
    auto tr = g1::generate_trace(in, (size_t)1 << min_rows_log2, &outs);
    size_t rows = tr[0].size();
    for (int c = 0; c < g1::W; c++) memcpy(trace_out + (size_t)c * rows, tr[c].data(), 8 * rows);
    if (outputs8)
      for (size_t k = 0; k < n; k++) {
        U256 ox = fq_to_u256(outs[k].x), oy = fq_to_u256(outs[k].y);
        memcpy(outputs8 + 8 * k, ox.l, 32);
        memcpy(outputs8 + 8 * k + 4, oy.l, 32);
      }
    return 0;
  } catch (std::exception& e) {
    return fail(e);
  }
}

// Evaluate the 1111 AIR constraints on one (local, next) row pair; returns the number emitted and the
// two alpha-accumulators (tests: all-zero on consecutive trace rows).
int orc_g1_eval_constraints(const u64* local, const u64* next, const u64* alphas2, u64 z_last, u64 lfirst, u64 llast,
                            u64* accs2) {
  std::vector<F> l(g1::W), nx(g1::W);
  for (int i = 0; i < g1::W; i++) {
    l[i] = F(local[i]);
    nx[i] = F(next[i]);
  }
  Consumer<F> cc({F(alphas2[0]), F(alphas2[1])}, F(z_last), F(lfirst), F(llast));
  g1::eval_constraints<F>(l.data(), nx.data(), cc);
  accs2[0] = cc.accs[0].v;
  accs2[1] = cc.accs[1].v;
  return (int)cc.count;
}

size_t orc_g1_proof_len(int degree_bits) {
  StarkDef d = g1::stark_def();
  StarkConfig cfg;
  int A = 2 * (d.num_helpers() + 1) + 4, nq = 4, capn = 16;
  std::vector<int> ar = cfg.fri_arities(degree_bits);
  int lde_bits = degree_bits + 1;
  size_t len = 3 * capn * 4 + 2 * (2 * d.W + 2 * A) + 4 + 2 * nq + ar.size() * capn * 4;
  size_t per_q = (d.W + A + nq) + 3 * 4 * (lde_bits - 4);
  int bits = lde_bits, sum = 0;
  for (int a : ar) {
    bits -= a;
    sum += a;
    per_q += 2 * (1 << a) + 4 * (bits - 4);
  }
  len += 84 * per_q + 2 * ((size_t)1 << (degree_bits - sum)) + 1 + 12;
  return len;
}

// Full prove (trace generation + starky prove). proof_out must hold orc_g1_proof_len(degree_bits) words.
// timings (optional, 8 doubles): trace_gen, trace_commit, aux, aux_commit, quotient, quotient_commit, openings, fri.
int orc_g1_prove(const u64* scalars, const u64* x, const u64* off, size_t n, int min_rows_log2, u64* proof_out,
                 size_t proof_cap, u64* outputs8, double* timings) {
  try {
    auto in = unpack_g1(scalars, x, off, n);
    std::vector<g1::Affine> outs;
    double t0 = now_sec();
    auto tr = g1::generate_trace(in, (size_t)1 << min_rows_log2, &outs);
    double t1 = now_sec();
    StarkDef d = g1::stark_def();
    StarkConfig cfg;
    ProveTimings tm;
    Proof pr = prove(d, cfg, tr, &tm);
    std::vector<u64> flat = pr.serialize();
    if (flat.size() > proof_cap) throw std::runtime_error("proof buffer too small");
    memcpy(proof_out, flat.data(), 8 * flat.size());
    if (outputs8)
      for (size_t k = 0; k < n; k++) {
        U256 ox = fq_to_u256(outs[k].x), oy = fq_to_u256(outs[k].y);
        memcpy(outputs8 + 8 * k, ox.l, 32);
        memcpy(outputs8 + 8 * k + 4, oy.l, 32);
      }
    if (timings) {
      timings[0] = t1 - t0;
      timings[1] = tm.trace_commit;
      timings[2] = tm.aux;
      timings[3] = tm.aux_commit;
      timings[4] = tm.quotient;
      timings[5] = tm.quotient_commit;
      timings[6] = tm.openings;
      timings[7] = tm.fri;
    }
    return (int)flat.size();
  } catch (std::exception& e) {
    return fail(e);
  }
}

// Native verify (common/verifier.rs:32-98) of a serialized proof against the inputs' CTL values.
// Returns 0 if accepted, 1 if rejected (reason in orc_last_error), -1 on malformed input.
int orc_g1_verify(const u64* proof, size_t proof_len, int degree_bits, const u64* scalars, const u64* x, const u64* off,
                  size_t n) {
  try {
    StarkDef d = g1::stark_def();
    StarkConfig cfg;
    Proof pr = deserialize_proof(d, cfg, degree_bits, proof, proof_len);
    auto in = unpack_g1(scalars, x, off, n);
    std::vector<g1::Affine> outs(n);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < n; k++) outs[k] = g1::scalar_mul_offset(in[k]);
    auto ctl = g1::generate_ctl_values(in, outs);
    std::string r = verify(d, cfg, pr, ctl);
    if (!r.empty()) {
      snprintf(g_err, sizeof(g_err), "%s", r.c_str());
      return 1;
    }
    return 0;
  } catch (std::exception& e) {
    return fail(e);
  }
}


// ---- generic entry points: kind 0 = G1 scalar mul, 1 = G2 scalar mul, 2 = Fq exponentiation ---------------
// inputs: scalars n x 4; x n x {8,16,4}; off n x {8,16} (ignored for kind 2); outputs n x {8,16,4}.
static int point_words(int kind) { return kind == 0 ? 8 : kind == 1 ? 16 : 4; }
static StarkDef def_of(int kind) { return kind == 0 ? g1::stark_def() : kind == 1 ? g2::stark_def() : fqexp::stark_def(); }

struct GenOut {
  std::vector<std::vector<u64>> trace;
  std::vector<u64> outputs;                            // n x point_words
  std::vector<std::vector<std::vector<u64>>> ctl;      // extra looking values
};
static void put_u256(std::vector<u64>& v, const U256& x) { v.insert(v.end(), x.l, x.l + 4); }

static GenOut run_kind(int kind, const u64* scalars, const u64* x, const u64* off, size_t n, int min_rows_log2, bool want_trace) {
  GenOut o;
  size_t min_rows = (size_t)1 << min_rows_log2;
  if (kind == 0) {
    auto in = unpack_g1(scalars, x, off, n);
    std::vector<g1::Affine> outs(n);
    if (want_trace) o.trace = g1::generate_trace(in, min_rows, &outs);
    else {
#pragma omp parallel for schedule(dynamic, 1)
      for (size_t k = 0; k < n; k++) outs[k] = g1::scalar_mul_offset(in[k]);
    }
    for (auto& p : outs) { put_u256(o.outputs, fq_to_u256(p.x)); put_u256(o.outputs, fq_to_u256(p.y)); }
    o.ctl = g1::generate_ctl_values(in, outs);
  } else if (kind == 1) {
    std::vector<g2::Input> in(n);
    for (size_t k = 0; k < n; k++) {
      memcpy(in[k].s, scalars + 4 * k, 32);
      for (int j = 0; j < 4; j++) {
        memcpy(in[k].x[j].l, x + 16 * k + 4 * j, 32);
        memcpy(in[k].off[j].l, off + 16 * k + 4 * j, 32);
      }
    }
    std::vector<g2::Affine> outs(n);
    if (want_trace) o.trace = g2::generate_trace(in, min_rows, &outs);
    else {
#pragma omp parallel for schedule(dynamic, 1)
      for (size_t k = 0; k < n; k++) outs[k] = g2::scalar_mul_offset(in[k]);
    }
    for (auto& p : outs) {
      put_u256(o.outputs, fq_to_u256(p.x.c0)); put_u256(o.outputs, fq_to_u256(p.x.c1));
      put_u256(o.outputs, fq_to_u256(p.y.c0)); put_u256(o.outputs, fq_to_u256(p.y.c1));
    }
    o.ctl = g2::generate_ctl_values(in, outs);
  } else {
    std::vector<fqexp::Input> in(n);
    for (size_t k = 0; k < n; k++) {
      memcpy(in[k].s, scalars + 4 * k, 32);
      memcpy(in[k].x.l, x + 4 * k, 32);
    }
    std::vector<Fq> outs(n);
    if (want_trace) o.trace = fqexp::generate_trace(in, min_rows, &outs);
    else {
#pragma omp parallel for schedule(dynamic, 1)
      for (size_t k = 0; k < n; k++) outs[k] = fqexp::pow_s(in[k]);
    }
    for (auto& p : outs) put_u256(o.outputs, fq_to_u256(p));
    o.ctl = fqexp::generate_ctl_values(in, outs);
  }
  return o;
}

int orc_stark_width(int kind) { return def_of(kind).W; }
size_t orc_proof_len(int kind, int degree_bits) {
  StarkDef d = def_of(kind);
  StarkConfig cfg;
  int A = 2 * (d.num_helpers() + 1) + 4, nq = 4, capn = 16;
  std::vector<int> ar = cfg.fri_arities(degree_bits);
  int lde_bits = degree_bits + 1;
  size_t len = 3 * capn * 4 + 2 * (2 * d.W + 2 * A) + 4 + 2 * nq + ar.size() * capn * 4;
  size_t per_q = (d.W + A + nq) + 3 * 4 * (lde_bits - 4);
  int bits = lde_bits, sum = 0;
  for (int a : ar) {
    bits -= a;
    sum += a;
    per_q += 2 * (1 << a) + 4 * (bits - 4);
  }
  len += 84 * per_q + 2 * ((size_t)1 << (degree_bits - sum)) + 1 + 12;
  return len;
}
int orc_generate_trace(int kind, const u64* scalars, const u64* x, const u64* off, size_t n, int min_rows_log2, u64* trace_out,
                       u64* outputs) {
  try {
    GenOut g = run_kind(kind, scalars, x, off, n, min_rows_log2, true);
    size_t rows = g.trace[0].size();
    for (size_t c = 0; c < g.trace.size(); c++) memcpy(trace_out + c * rows, g.trace[c].data(), 8 * rows);
    if (outputs) memcpy(outputs, g.outputs.data(), 8 * g.outputs.size());
    return 0;
  } catch (std::exception& e) {
    return fail(e);
  }
}
int orc_prove(int kind, const u64* scalars, const u64* x, const u64* off, size_t n, int min_rows_log2, u64* proof_out,
              size_t proof_cap, u64* outputs, double* timings) {
  try {
    double t0 = now_sec();
    GenOut g = run_kind(kind, scalars, x, off, n, min_rows_log2, true);
    double t1 = now_sec();
    StarkDef d = def_of(kind);
    StarkConfig cfg;
    ProveTimings tm;
    Proof pr = prove(d, cfg, g.trace, &tm);
    std::vector<u64> flat = pr.serialize();
    if (flat.size() > proof_cap) throw std::runtime_error("proof buffer too small");
    memcpy(proof_out, flat.data(), 8 * flat.size());
    if (outputs) memcpy(outputs, g.outputs.data(), 8 * g.outputs.size());
    if (timings) {
      double t[8] = {t1 - t0, tm.trace_commit, tm.aux, tm.aux_commit, tm.quotient, tm.quotient_commit, tm.openings, tm.fri};
      memcpy(timings, t, sizeof(t));
    }
    return (int)flat.size();
  } catch (std::exception& e) {
    return fail(e);
  }
}
int orc_verify(int kind, const u64* proof, size_t proof_len, int degree_bits, const u64* scalars, const u64* x, const u64* off,
               size_t n) {
  try {
    StarkDef d = def_of(kind);
    StarkConfig cfg;
    Proof pr = deserialize_proof(d, cfg, degree_bits, proof, proof_len);
    GenOut g = run_kind(kind, scalars, x, off, n, 16, false);
    std::string r = verify(d, cfg, pr, g.ctl);
    if (!r.empty()) {
      snprintf(g_err, sizeof(g_err), "%s", r.c_str());
      return 1;
    }
    return 0;
  } catch (std::exception& e) {
    return fail(e);
  }
}
// constraints of one (local, next) pair: returns the number emitted, accs2 = the two alpha accumulators
int orc_eval_constraints(int kind, const u64* local, const u64* next, const u64* alphas2, u64 z_last, u64 lfirst, u64 llast,
                         u64* accs2) {
  StarkDef d = def_of(kind);
  std::vector<F> l(d.W), nx(d.W);
  for (int i = 0; i < d.W; i++) {
    l[i] = F(local[i]);
    nx[i] = F(next[i]);
  }
  Consumer<F> cc({F(alphas2[0]), F(alphas2[1])}, F(z_last), F(lfirst), F(llast));
  d.eval_base(l.data(), nx.data(), cc);
  accs2[0] = cc.accs[0].v;
  accs2[1] = cc.accs[1].v;
  return (int)cc.count;
}

}  // extern "C"
