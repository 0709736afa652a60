// ORACLE (test infrastructure, NOT the product path).
// CPU restatement of the starky 0.4.0 / plonky2 0.2.2 prover and native verifier as they are driven
// by reference src/starks/common/prover.rs:18-72 (`prove`) and src/starks/common/verifier.rs:32-98
// (`verify`): PolynomialBatch::from_values, Challenger transcript, get_ctl_data (looked tables only),
// LogUp helper columns, compute_quotient_polys, StarkOpeningSet, PolynomialBatch::prove_openings (FRI).
// The starky/plonky2 sources are NOT in /root/reference (git dependency, Cargo.lock:567-571,801-804);
// the conventions follow SURVEY.md App. A.  parity unpinned: validated by the restated verifier,
// algebraic identities and the Poseidon KATs only (DESIGN.md "Oracle").
#pragma once
#include "gl.hpp"
#include "hash.hpp"
#include "ntt.hpp"
#include <functional>
#include <string>
#include <stdexcept>
#include <cstdio>

namespace orc {

// ---- STARK description -------------------------------------------------------------------------
struct LinComb {  // starky `Column`: sum coef_k * local[col_k]  (no constant / next-row terms needed here)
  std::vector<std::pair<int, u64>> terms;
  static LinComb single(int c) { return LinComb{{{c, 1}}}; }
  static LinComb le_bits(int begin, int n) {
    LinComb l;
    for (int i = 0; i < n; i++) l.terms.push_back({begin + i, 1ULL << i});
    return l;
  }
  template <class T>
  T eval(const T* row) const {
    T acc = T::from_u64(0);
    for (auto& t : terms) acc += row[t.first] * T::from_u64(t.second);
    return acc;
  }
};
struct CtlDef {  // one CrossTableLookup with an empty looking list and this looked table
  std::vector<LinComb> cols;
  int filter_col;
};

// ConstraintConsumer<P>
template <class T>
struct Consumer {
  std::vector<T> alphas, accs;
  T z_last, lagrange_first, lagrange_last;
  size_t count = 0;
  Consumer(const std::vector<T>& al, T zl, T lf, T ll) : alphas(al), accs(al.size()), z_last(zl), lagrange_first(lf), lagrange_last(ll) {}
  void constraint(T c) {
    for (size_t j = 0; j < alphas.size(); j++) accs[j] = accs[j] * alphas[j] + c;
    count++;
  }
  void constraint_transition(T c) { constraint(c * z_last); }
  void constraint_first_row(T c) { constraint(c * lagrange_first); }
  void constraint_last_row(T c) { constraint(c * lagrange_last); }
};

struct StarkDef {
  std::string name;
  int W = 0;
  int lookup_begin = 0, lookup_end = 0;  // Lookup.columns = singles(lookup_begin..lookup_end)
  int table_col = 0, freq_col = 0;
  std::vector<CtlDef> ctls;
  std::function<void(const F*, const F*, Consumer<F>&)> eval_base;
  std::function<void(const F2*, const F2*, Consumer<F2>&)> eval_ext;
  int num_lookup_cols() const { return lookup_end - lookup_begin; }
  int num_helpers() const { return (num_lookup_cols() + 1) / 2; }  // per challenge, without Z
};

struct StarkConfig {  // StarkConfig::standard_fast_config (reference generators/g1/stark_proof.rs:152)
  int num_challenges = 2, rate_bits = 1, cap_height = 4, pow_bits = 16, arity_bits = 4, final_poly_bits = 5,
      num_queries = 84;
  std::vector<int> fri_arities(int degree_bits) const {  // ConstantArityBits(4,5), SURVEY.md A.9
    std::vector<int> a;
    int d = degree_bits;
    while (d > final_poly_bits && d + rate_bits - arity_bits >= cap_height) {
      a.push_back(arity_bits);
      d -= arity_bits;
    }
    return a;
  }
};

// ---- Proof -------------------------------------------------------------------------------------
struct FriQueryStep {
  std::vector<F2> evals;
  std::vector<Digest> path;
};
struct FriInitialOpening {
  std::vector<u64> leaf;
  std::vector<Digest> path;
};
struct FriQueryRound {
  std::vector<FriInitialOpening> initial;  // trace, aux, quotient
  std::vector<FriQueryStep> steps;
};
struct Proof {
  std::vector<Digest> trace_cap, aux_cap, quotient_cap;
  std::vector<F2> local_values, next_values, aux_polys, aux_polys_next, quotient_polys;
  std::vector<u64> ctl_zs_first;
  std::vector<std::vector<Digest>> commit_caps;
  std::vector<FriQueryRound> queries;
  std::vector<F2> final_poly;
  u64 pow_witness = 0;
  u64 init_challenger_state[12] = {0};
  int degree_bits = 0;

  // Canonical flat u64 layout shared with the HIP build (include/bn254_stark.h, "proof layout").
  std::vector<u64> serialize() const {
    std::vector<u64> o;
    auto put_cap = [&](const std::vector<Digest>& c) {
      for (auto& d : c) o.insert(o.end(), d.e, d.e + 4);
    };
    auto put_ext = [&](const std::vector<F2>& v) {
      for (auto& x : v) {
        o.push_back(x.c0);
        o.push_back(x.c1);
      }
    };
    put_cap(trace_cap);
    put_cap(aux_cap);
    put_cap(quotient_cap);
    put_ext(local_values);
    put_ext(next_values);
    put_ext(aux_polys);
    put_ext(aux_polys_next);
    o.insert(o.end(), ctl_zs_first.begin(), ctl_zs_first.end());
    put_ext(quotient_polys);
    for (auto& c : commit_caps) put_cap(c);
    for (auto& q : queries) {
      for (auto& io : q.initial) {
        o.insert(o.end(), io.leaf.begin(), io.leaf.end());
        put_cap(io.path);
      }
      for (auto& s : q.steps) {
        put_ext(s.evals);
        put_cap(s.path);
      }
    }
    put_ext(final_poly);
    o.push_back(pow_witness);
    o.insert(o.end(), init_challenger_state, init_challenger_state + 12);
    return o;
  }
};

// Inverse of serialize(): shapes follow from (def, cfg, degree_bits).
static inline Proof deserialize_proof(const StarkDef& def, const StarkConfig& cfg, int degree_bits, const u64* p, size_t len) {
  Proof pr;
  pr.degree_bits = degree_bits;
  size_t pos = 0;
  auto need = [&](size_t n) {
    if (pos + n > len) throw std::runtime_error("proof too short");
  };
  auto get_cap = [&](size_t n) {
    std::vector<Digest> c(n);
    need(4 * n);
    for (auto& d : c) {
      memcpy(d.e, p + pos, 32);
      pos += 4;
    }
    return c;
  };
  auto get_ext = [&](size_t n) {
    std::vector<F2> v(n);
    need(2 * n);
    for (auto& x : v) {
      x = F2(p[pos], p[pos + 1]);
      pos += 2;
    }
    return v;
  };
  int capn = 1 << cfg.cap_height;
  int A = 2 * (def.num_helpers() + 1) + cfg.num_challenges * (int)def.ctls.size();
  int nq = 2 * cfg.num_challenges;
  pr.trace_cap = get_cap(capn);
  pr.aux_cap = get_cap(capn);
  pr.quotient_cap = get_cap(capn);
  pr.local_values = get_ext(def.W);
  pr.next_values = get_ext(def.W);
  pr.aux_polys = get_ext(A);
  pr.aux_polys_next = get_ext(A);
  size_t nz = cfg.num_challenges * def.ctls.size();
  need(nz);
  pr.ctl_zs_first.assign(p + pos, p + pos + nz);
  pos += nz;
  pr.quotient_polys = get_ext(nq);
  std::vector<int> ar = cfg.fri_arities(degree_bits);
  for (size_t i = 0; i < ar.size(); i++) pr.commit_caps.push_back(get_cap(capn));
  int lde_bits = degree_bits + cfg.rate_bits;
  int widths[3] = {def.W, A, nq};
  for (int q = 0; q < cfg.num_queries; q++) {
    FriQueryRound r;
    for (int t = 0; t < 3; t++) {
      FriInitialOpening io;
      need(widths[t]);
      io.leaf.assign(p + pos, p + pos + widths[t]);
      pos += widths[t];
      io.path = get_cap(lde_bits - cfg.cap_height);
      r.initial.push_back(io);
    }
    int bits = lde_bits;
    for (size_t i = 0; i < ar.size(); i++) {
      FriQueryStep s;
      s.evals = get_ext(1 << ar[i]);
      bits -= ar[i];
      s.path = get_cap(bits - cfg.cap_height);
      r.steps.push_back(s);
    }
    pr.queries.push_back(r);
  }
  int final_len = 1 << (degree_bits - [&] { int s = 0; for (int a : ar) s += a; return s; }());
  pr.final_poly = get_ext(final_len);
  need(13);
  pr.pow_witness = p[pos++];
  memcpy(pr.init_challenger_state, p + pos, 96);
  pos += 12;
  if (pos != len) throw std::runtime_error("proof length mismatch");
  return pr;
}

// ---- Polynomial batch (PolynomialBatch::from_values / from_coeffs) ----------------
struct Batch {
  int degree_bits = 0, rate_bits = 1;
  std::vector<std::vector<u64>> coeffs;  // [C][N]
  std::vector<std::vector<u64>> lde;     // [C][2N], natural order: lde[c][i] = P_c(shift * w_2N^i)
  MerkleTree tree;                       // leaves in bit-reversed order
  size_t ncols() const { return coeffs.size(); }
  size_t lde_size() const { return size_t(1) << (degree_bits + rate_bits); }
  // leaf j (tree order) = natural row reverse_bits(j)
  std::vector<u64> leaf(size_t j) const {
    size_t i = reverse_bits(j, degree_bits + rate_bits);
    std::vector<u64> r(ncols());
    for (size_t c = 0; c < ncols(); c++) r[c] = lde[c][i];
    return r;
  }
};

static inline void batch_from_coeffs(Batch& b, std::vector<std::vector<u64>>&& coeffs, int rate_bits, int cap_height) {
  b.coeffs = std::move(coeffs);
  size_t C = b.coeffs.size(), N = b.coeffs[0].size();
  b.degree_bits = log2_strict(N);
  b.rate_bits = rate_bits;
  size_t M = N << rate_bits;
  b.lde.assign(C, std::vector<u64>());
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t c = 0; c < C; c++) {
    std::vector<F> v(M);
    for (size_t i = 0; i < N; i++) v[i] = F(b.coeffs[c][i]);
    coset_fft(v, GL_GENERATOR);
    b.lde[c].resize(M);
    for (size_t i = 0; i < M; i++) b.lde[c][i] = v[i].v;
  }
  unsigned lb = b.degree_bits + rate_bits;
  std::vector<Digest> digests(M);
  const size_t BLK = 64;  // gather 64 consecutive natural rows at a time (cache-friendly transpose)
#pragma omp parallel
  {
    std::vector<u64> rows(BLK * C);
#pragma omp for schedule(static)
    for (size_t i0 = 0; i0 < M; i0 += BLK) {
      for (size_t c = 0; c < C; c++)
        for (size_t k = 0; k < BLK; k++) rows[k * C + c] = b.lde[c][i0 + k];
      for (size_t k = 0; k < BLK; k++) digests[reverse_bits(i0 + k, lb)] = hash_or_noop(rows.data() + k * C, C);
    }
  }
  b.tree.build(std::move(digests), cap_height);
}
static inline void batch_from_values(Batch& b, const std::vector<std::vector<u64>>& values, int rate_bits, int cap_height) {
  size_t C = values.size();
  std::vector<std::vector<u64>> coeffs(C);
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t c = 0; c < C; c++) {
    std::vector<F> v(values[c].size());
    for (size_t i = 0; i < v.size(); i++) v[i] = F(values[c][i]);
    ifft(v);
    coeffs[c].resize(v.size());
    for (size_t i = 0; i < v.size(); i++) coeffs[c][i] = v[i].v;
  }
  batch_from_coeffs(b, std::move(coeffs), rate_bits, cap_height);
}

// ---- auxiliary columns ---------------------------------------------------------------------------
// starky `lookup_helper_columns` for one challenge: helpers then Z.
static inline std::vector<std::vector<u64>> lookup_helper_columns(const StarkDef& def, const std::vector<std::vector<u64>>& trace,
                                                                   u64 challenge) {
  size_t N = trace[0].size();
  int n = def.num_lookup_cols(), m = def.num_helpers();
  std::vector<std::vector<u64>> out(m + 1, std::vector<u64>(N));
#pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < m; k++) {
    int c0 = def.lookup_begin + 2 * k, c1 = c0 + 1;
    bool two = (2 * k + 1) < n;
    std::vector<u64> d0(N), d1;
    for (size_t i = 0; i < N; i++) d0[i] = gl_add(trace[c0][i], challenge);
    std::vector<u64> i0 = gl_batch_inv(d0);
    if (two) {
      d1.resize(N);
      for (size_t i = 0; i < N; i++) d1[i] = gl_add(trace[c1][i], challenge);
      std::vector<u64> i1 = gl_batch_inv(d1);
      for (size_t i = 0; i < N; i++) out[k][i] = gl_add(i0[i], i1[i]);
    } else {
      out[k] = i0;
    }
  }
  std::vector<u64> tbl(N);
  for (size_t i = 0; i < N; i++) tbl[i] = gl_add(challenge, trace[def.table_col][i]);
  std::vector<u64> tinv = gl_batch_inv(tbl);
  // Z_(i+1) = Z_i + sum_k h_k(i) - freq(i) / (x + table(i)).  The row sums are formed column by column over blocks of rows
  // (the helper columns are column-major: walking them row by row touches m cache lines per row), then one serial prefix sum.
  std::vector<u64> rowsum(N, 0);
  const size_t BLK = 2048;
#pragma omp parallel for schedule(static)
  for (size_t i0 = 0; i0 < N; i0 += BLK) {
    const size_t i1 = i0 + BLK < N ? i0 + BLK : N;
    for (int k = 0; k < m; k++)
      for (size_t i = i0; i < i1; i++) rowsum[i] = gl_add(rowsum[i], out[k][i]);
    for (size_t i = i0; i < i1; i++) rowsum[i] = gl_sub(rowsum[i], gl_mul(trace[def.freq_col][i], tinv[i]));
  }
  std::vector<u64>& z = out[m];
  z[0] = 0;
  for (size_t i = 0; i + 1 < N; i++) z[i + 1] = gl_add(z[i], rowsum[i]);
  return out;
}

// GrandProductChallenge::combine: sum v_i beta^i + gamma
template <class T>
static T combine(const std::vector<T>& v, T beta, T gamma) {
  T acc = T::from_u64(0);
  for (size_t i = v.size(); i-- > 0;) acc = acc * beta + v[i];
  return acc + gamma;
}

// starky `partial_sums` for a looked table with a single column set.
static inline std::vector<u64> ctl_z_column(const CtlDef& ctl, const std::vector<std::vector<u64>>& trace, u64 beta, u64 gamma) {
  size_t N = trace[0].size(), W = trace.size();
  std::vector<u64> h(N, 0);
  std::vector<F> row(W);
  for (size_t i = 0; i < N; i++) {
    u64 f = trace[ctl.filter_col][i];
    if (f == 1) {
      for (size_t c = 0; c < W; c++) row[c] = F(trace[c][i]);
      std::vector<F> ev;
      for (auto& lc : ctl.cols) ev.push_back(lc.eval<F>(row.data()));
      h[i] = gl_inv(combine<F>(ev, F(beta), F(gamma)).v);
    } else if (f != 0) {
      throw std::runtime_error("Non-binary filter?");
    }
  }
  std::vector<u64> z(N);
  z[N - 1] = h[N - 1];
  for (size_t i = N - 1; i-- > 0;) z[i] = gl_add(z[i + 1], h[i]);
  return z;
}

// ---- vanishing polynomial (starky eval_vanishing_poly: stark constraints, lookups, CTLs) --------
template <class T>
struct AuxVars {
  const T* local;  // auxiliary polys at the local point (A values)
  const T* next;
};
template <class T>
static void eval_vanishing(const StarkDef& def, const StarkConfig& cfg, const T* local, const T* next, const AuxVars<T>& aux,
                           const std::vector<u64>& ctl_betas, const std::vector<u64>& ctl_gammas, Consumer<T>& cc,
                           const std::function<void(const T*, const T*, Consumer<T>&)>& eval) {
  eval(local, next, cc);
  // eval_packed_lookups_generic: lookup challenges = CTL betas.
  int m = def.num_helpers(), n = def.num_lookup_cols();
  int start = 0;
  for (int ch = 0; ch < cfg.num_challenges; ch++) {
    T x = T::from_u64(ctl_betas[ch]);
    for (int k = 0; k < m; k++) {
      T h = aux.local[start + k];
      T f0 = local[def.lookup_begin + 2 * k] + x;
      if (2 * k + 1 < n) {
        T f1 = local[def.lookup_begin + 2 * k + 1] + x;
        cc.constraint(f1 * f0 * h - f1 - f0);
      } else {
        cc.constraint(f0 * h - T::from_u64(1));
      }
    }
    T z = aux.local[start + m], nz = aux.next[start + m];
    T twc = local[def.table_col] + x;
    T hs = T::from_u64(0);
    for (int k = 0; k < m; k++) hs += aux.local[start + k];
    T y = hs * twc - local[def.freq_col];
    cc.constraint_first_row(z);
    cc.constraint((nz - z) * twc - y);
    start += m + 1;
  }
  // eval_cross_table_lookup_checks: one Z per (ctl, challenge), no helper columns.
  int zi = start;
  for (size_t k = 0; k < def.ctls.size(); k++)
    for (int ch = 0; ch < cfg.num_challenges; ch++) {
      const CtlDef& ctl = def.ctls[k];
      std::vector<T> ev;
      for (auto& lc : ctl.cols) ev.push_back(lc.eval<T>(local));
      T comb = combine<T>(ev, T::from_u64(ctl_betas[ch]), T::from_u64(ctl_gammas[ch]));
      T f0 = local[ctl.filter_col];
      T lz = aux.local[zi], nz = aux.next[zi];
      cc.constraint_last_row(comb * lz - f0);
      cc.constraint_transition(comb * (lz - nz) - f0);
      zi++;
    }
}

// ---- FRI helpers -----------------------------------------------------------------------------------
static inline F2 eval_poly_base_at_ext(const std::vector<u64>& coeffs, F2 z) {
  F2 acc(0);
  for (size_t i = coeffs.size(); i-- > 0;) acc = acc * z + F2(coeffs[i]);
  return acc;
}
static inline F2 eval_poly_ext(const std::vector<F2>& coeffs, F2 z) {
  F2 acc(0);
  for (size_t i = coeffs.size(); i-- > 0;) acc = acc * z + coeffs[i];
  return acc;
}

struct FriBatchInfo {
  F2 point;
  std::vector<std::pair<int, int>> polys;  // (oracle, index)
};
// Stark::fri_instance with 0 CTL helper columns.
static inline std::vector<FriBatchInfo> fri_instance(const StarkDef& def, const StarkConfig& cfg, F2 zeta, u64 g) {
  int A = 2 * (def.num_helpers() + 1) + cfg.num_challenges * (int)def.ctls.size();
  int nq = 2 * cfg.num_challenges;
  int num_lookup = 2 * (def.num_helpers() + 1);
  std::vector<FriBatchInfo> b(3);
  b[0].point = zeta;
  for (int i = 0; i < def.W; i++) b[0].polys.push_back({0, i});
  for (int i = 0; i < A; i++) b[0].polys.push_back({1, i});
  for (int i = 0; i < nq; i++) b[0].polys.push_back({2, i});
  b[1].point = zeta.scalar_mul(g);
  for (int i = 0; i < def.W; i++) b[1].polys.push_back({0, i});
  for (int i = 0; i < A; i++) b[1].polys.push_back({1, i});
  b[2].point = F2(1);
  for (int i = num_lookup; i < A; i++) b[2].polys.push_back({1, i});
  return b;
}

// ---- prove ---------------------------------------------------------------------------------------
struct ProveTimings {
  double trace_commit = 0, aux = 0, aux_commit = 0, quotient = 0, quotient_commit = 0, openings = 0, fri = 0;
};
double now_sec();

static inline Proof prove(const StarkDef& def, const StarkConfig& cfg, const std::vector<std::vector<u64>>& trace,
                          ProveTimings* tm = nullptr) {
  Proof pr;
  size_t N = trace[0].size();
  int degree_bits = log2_strict(N);
  pr.degree_bits = degree_bits;
  size_t M = N << cfg.rate_bits;
  int lde_bits = degree_bits + cfg.rate_bits;
  assert((int)trace.size() == def.W);
  double t0 = now_sec();

  // prover.rs:31-44
  Batch tb;
  batch_from_values(tb, trace, cfg.rate_bits, cfg.cap_height);
  Challenger ch;
  pr.trace_cap = tb.tree.cap();
  ch.observe_cap(pr.trace_cap);
  double t1 = now_sec();
  // get_ctl_data: challenges then Z columns (prover.rs:46-52)
  std::vector<u64> betas(cfg.num_challenges), gammas(cfg.num_challenges);
  for (int i = 0; i < cfg.num_challenges; i++) {
    betas[i] = ch.get_challenge();
    gammas[i] = ch.get_challenge();
  }
  std::vector<std::vector<u64>> ctl_zs;
  for (auto& ctl : def.ctls)
    for (int i = 0; i < cfg.num_challenges; i++) ctl_zs.push_back(ctl_z_column(ctl, trace, betas[i], gammas[i]));
  ch.compact(pr.init_challenger_state);  // prover.rs:54

  // prove_with_commitment: lookup challenges = CTL betas
  std::vector<std::vector<u64>> aux_cols;
  for (int i = 0; i < cfg.num_challenges; i++) {
    auto cols = lookup_helper_columns(def, trace, betas[i]);
    for (auto& c : cols) aux_cols.push_back(std::move(c));
  }
  for (auto& z : ctl_zs) aux_cols.push_back(std::move(z));
  int A = (int)aux_cols.size();
  double t2 = now_sec();
  Batch ab;
  batch_from_values(ab, aux_cols, cfg.rate_bits, cfg.cap_height);
  pr.aux_cap = ab.tree.cap();
  ch.observe_cap(pr.aux_cap);
  double t3 = now_sec();
  std::vector<u64> alphas(cfg.num_challenges);
  for (auto& a : alphas) a = ch.get_challenge();

  // compute_quotient_polys
  std::vector<std::vector<F>> qvals(cfg.num_challenges, std::vector<F>(M));
  {
    u64 g = gl_root_of_unity(degree_bits);
    u64 last = gl_inv(g);
    // Lagrange selectors on the coset via ifft / lde / coset_fft, like PolynomialValues::lde_onto_coset
    auto selector = [&](size_t idx) {
      std::vector<F> v(N);
      v[idx] = F(1);
      ifft(v);
      v.resize(M);
      coset_fft(v, GL_GENERATOR);
      return v;
    };
    std::vector<F> lfirst = selector(0), llast = selector(N - 1);
    u64 w = gl_root_of_unity(lde_bits);
    std::vector<u64> xs(M);
    xs[0] = GL_GENERATOR;
    for (size_t i = 1; i < M; i++) xs[i] = gl_mul(xs[i - 1], w);
    // Z_H on coset: shift^N * (w_2N^i)^N - 1, period 2^rate_bits
    u64 gpn = gl_pow(GL_GENERATOR, N);
    std::vector<u64> zh_inv(1 << cfg.rate_bits);
    for (size_t i = 0; i < zh_inv.size(); i++) zh_inv[i] = gl_inv(gl_sub(gl_mul(gpn, gl_pow(gl_root_of_unity(cfg.rate_bits), i)), 1));
    size_t next_step = size_t(1) << cfg.rate_bits;  // quotient_degree_bits == rate_bits == 1
    std::vector<F> alphaF;
    for (u64 a : alphas) alphaF.push_back(F(a));
#pragma omp parallel
    {
      std::vector<F> loc(def.W), nxt(def.W), al(A), an(A);
#pragma omp for schedule(static)
      for (size_t i = 0; i < M; i++) {
        size_t in = (i + next_step) % M;
        for (int c = 0; c < def.W; c++) {
          loc[c] = F(tb.lde[c][i]);
          nxt[c] = F(tb.lde[c][in]);
        }
        for (int c = 0; c < A; c++) {
          al[c] = F(ab.lde[c][i]);
          an[c] = F(ab.lde[c][in]);
        }
        Consumer<F> cc(alphaF, F(gl_sub(xs[i], last)), lfirst[i], llast[i]);
        AuxVars<F> av{al.data(), an.data()};
        eval_vanishing<F>(def, cfg, loc.data(), nxt.data(), av, betas, gammas, cc, def.eval_base);
        for (int j = 0; j < cfg.num_challenges; j++) qvals[j][i] = cc.accs[j] * F(zh_inv[i % zh_inv.size()]);
      }
    }
  }
  std::vector<std::vector<u64>> qchunks;
  for (int j = 0; j < cfg.num_challenges; j++) {
    coset_ifft(qvals[j], GL_GENERATOR);
    for (size_t k = 0; k < (M / N); k++) {
      std::vector<u64> chunk(N);
      for (size_t i = 0; i < N; i++) chunk[i] = qvals[j][k * N + i].v;
      qchunks.push_back(std::move(chunk));
    }
  }
  double t4 = now_sec();
  Batch qb;
  batch_from_coeffs(qb, std::move(qchunks), cfg.rate_bits, cfg.cap_height);
  pr.quotient_cap = qb.tree.cap();
  ch.observe_cap(pr.quotient_cap);
  double t5 = now_sec();

  F2 zeta = ch.get_ext_challenge();
  u64 g = gl_root_of_unity(degree_bits);
  if (zeta.exp_power_of_2(degree_bits) == F2(1)) throw std::runtime_error("Opening point is in the subgroup.");
  F2 zeta_next = zeta.scalar_mul(g);
  auto eval_all = [&](const Batch& b, F2 z) {
    std::vector<F2> r(b.ncols());
#pragma omp parallel for schedule(dynamic, 8)
    for (size_t c = 0; c < b.ncols(); c++) r[c] = eval_poly_base_at_ext(b.coeffs[c], z);
    return r;
  };
  pr.local_values = eval_all(tb, zeta);
  pr.next_values = eval_all(tb, zeta_next);
  pr.aux_polys = eval_all(ab, zeta);
  pr.aux_polys_next = eval_all(ab, zeta_next);
  pr.quotient_polys = eval_all(qb, zeta);
  int num_lookup = 2 * (def.num_helpers() + 1);
  for (int i = num_lookup; i < A; i++) {
    u64 s = 0;  // poly evaluated at 1 = sum of coefficients
    for (u64 c : ab.coeffs[i]) s = gl_add(s, c);
    pr.ctl_zs_first.push_back(s);
  }
  // observe_openings: zeta batch, zeta_next batch, ctl_zs_first (lifted)
  for (auto& x : pr.local_values) ch.observe_ext(x);
  for (auto& x : pr.aux_polys) ch.observe_ext(x);
  for (auto& x : pr.quotient_polys) ch.observe_ext(x);
  for (auto& x : pr.next_values) ch.observe_ext(x);
  for (auto& x : pr.aux_polys_next) ch.observe_ext(x);
  for (u64 x : pr.ctl_zs_first) ch.observe_ext(F2(x));
  double t6 = now_sec();

  // PolynomialBatch::prove_openings
  const Batch* oracles[3] = {&tb, &ab, &qb};
  F2 alpha = ch.get_ext_challenge();
  std::vector<F2> final_poly(N, F2(0));
  for (const FriBatchInfo& bi : fri_instance(def, cfg, zeta, g)) {
    // composition = sum alpha^j f_j
    std::vector<F2> comp(N, F2(0));
    std::vector<F2> apow(bi.polys.size());
    F2 ap(1);
    for (size_t j = 0; j < apow.size(); j++) {
      apow[j] = ap;
      ap = ap * alpha;
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < N; i++) {
      F2 acc(0);
      for (size_t j = 0; j < bi.polys.size(); j++) acc += apow[j].scalar_mul(oracles[bi.polys[j].first]->coeffs[bi.polys[j].second][i]);
      comp[i] = acc;
    }
    // divide_by_linear: quotient of (P(X) - P(z)) / (X - z), padded back to N coefficients
    std::vector<F2> quo(N, F2(0));
    F2 acc(0);
    for (size_t i = N; i-- > 0;) {
      acc = acc * bi.point + comp[i];
      if (i > 0) quo[i - 1] = acc;
    }
    // alpha.shift_poly(final_poly): multiply what is there by alpha^(#polys of this batch)
    for (size_t i = 0; i < N; i++) final_poly[i] = final_poly[i] * ap + quo[i];
  }
  std::vector<F2> coeffs = final_poly;
  coeffs.resize(M, F2(0));
  std::vector<F2> values = coeffs;
  coset_fft(values, GL_GENERATOR);

  // fri_committed_trees
  std::vector<int> arities = cfg.fri_arities(degree_bits);
  std::vector<MerkleTree> trees;
  std::vector<std::vector<F2>> layer_values;  // bit-reversed values per layer (for query openings)
  u64 shift = GL_GENERATOR;
  for (int ab_ : arities) {
    size_t arity = size_t(1) << ab_;
    size_t n = values.size();
    unsigned lb = log2_strict(n);
    std::vector<F2> rev(n);
    for (size_t i = 0; i < n; i++) rev[reverse_bits(i, lb)] = values[i];
    std::vector<Digest> digests(n / arity);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n / arity; i++) {
      std::vector<u64> flat(2 * arity);
      for (size_t k = 0; k < arity; k++) {
        flat[2 * k] = rev[i * arity + k].c0;
        flat[2 * k + 1] = rev[i * arity + k].c1;
      }
      digests[i] = hash_or_noop(flat.data(), flat.size());
    }
    MerkleTree t;
    t.build(std::move(digests), cfg.cap_height);
    ch.observe_cap(t.cap());
    pr.commit_caps.push_back(t.cap());
    trees.push_back(std::move(t));
    layer_values.push_back(std::move(rev));
    F2 beta = ch.get_ext_challenge();
    std::vector<F2> folded(coeffs.size() / arity);
    for (size_t i = 0; i < folded.size(); i++) {
      F2 acc(0);
      for (size_t k = arity; k-- > 0;) acc = acc * beta + coeffs[i * arity + k];
      folded[i] = acc;
    }
    coeffs = std::move(folded);
    shift = gl_pow(shift, arity);
    values = coeffs;
    coset_fft(values, shift);
  }
  coeffs.resize(coeffs.size() >> cfg.rate_bits);
  pr.final_poly = coeffs;
  for (auto& x : coeffs) ch.observe_ext(x);

  // fri_proof_of_work: smallest witness (upstream uses rayon find_any)
  {
    u64 st[12];
    memcpy(st, ch.state, sizeof(st));
    size_t pos = ch.in_buf.size();
    for (size_t i = 0; i < pos; i++) st[i] = ch.in_buf[i];
    u64 found = ~0ULL;
    for (u64 base = 0; found == ~0ULL; base += (1 << 16)) {
#pragma omp parallel for schedule(static)
      for (u64 c = base; c < base + (1 << 16); c++) {
        u64 s2[12];
        memcpy(s2, st, sizeof(s2));
        s2[pos] = c;
        poseidon_hash_permute(s2);
        if ((s2[7] >> (64 - cfg.pow_bits)) == 0) {  // >= pow_bits leading zeros
#pragma omp critical
          if (c < found) found = c;
        }
      }
    }
    pr.pow_witness = found;
    ch.observe_element(found);
    u64 resp = ch.get_challenge();
    if (resp >> (64 - cfg.pow_bits)) throw std::runtime_error("pow self-check failed");
  }

  // fri_prover_query_rounds
  for (int q = 0; q < cfg.num_queries; q++) {
    size_t x_index = ch.get_challenge() % M;
    FriQueryRound r;
    for (int t = 0; t < 3; t++) {
      FriInitialOpening io;
      io.leaf = oracles[t]->leaf(x_index);
      io.path = oracles[t]->tree.prove(x_index);
      r.initial.push_back(std::move(io));
    }
    size_t xi = x_index;
    for (size_t l = 0; l < arities.size(); l++) {
      size_t arity = size_t(1) << arities[l];
      size_t ci = xi >> arities[l];
      FriQueryStep s;
      s.evals.assign(layer_values[l].begin() + ci * arity, layer_values[l].begin() + (ci + 1) * arity);
      s.path = trees[l].prove(ci);
      r.steps.push_back(std::move(s));
      xi = ci;
    }
    pr.queries.push_back(std::move(r));
  }
  double t7 = now_sec();
  if (tm) {
    tm->trace_commit = t1 - t0;
    tm->aux = t2 - t1;
    tm->aux_commit = t3 - t2;
    tm->quotient = t4 - t3;
    tm->quotient_commit = t5 - t4;
    tm->openings = t6 - t5;
    tm->fri = t7 - t6;
  }
  (void)lde_bits;
  return pr;
}

// ---- verify (reference common/verifier.rs:32-98 + starky verify_stark_proof_with_challenges +
//      plonky2 verify_fri_proof).  Returns "" on success, else the failure reason. -----------------
static inline std::string verify(const StarkDef& def, const StarkConfig& cfg, const Proof& pr,
                                 const std::vector<std::vector<std::vector<u64>>>& extra_looking_values) {
  int degree_bits = pr.degree_bits;
  size_t N = size_t(1) << degree_bits, M = N << cfg.rate_bits;
  int lde_bits = degree_bits + cfg.rate_bits;
  int A = 2 * (def.num_helpers() + 1) + cfg.num_challenges * (int)def.ctls.size();
  int num_lookup = 2 * (def.num_helpers() + 1);
  int nq = 2 * cfg.num_challenges;
  if ((int)pr.local_values.size() != def.W || (int)pr.next_values.size() != def.W || (int)pr.aux_polys.size() != A ||
      (int)pr.aux_polys_next.size() != A || (int)pr.quotient_polys.size() != nq ||
      pr.ctl_zs_first.size() != cfg.num_challenges * def.ctls.size())
    return "bad proof shape";

  Challenger ch;
  ch.observe_cap(pr.trace_cap);
  std::vector<u64> betas(cfg.num_challenges), gammas(cfg.num_challenges);
  for (int i = 0; i < cfg.num_challenges; i++) {
    betas[i] = ch.get_challenge();
    gammas[i] = ch.get_challenge();
  }
  u64 st[12];
  ch.compact(st);
  if (memcmp(st, pr.init_challenger_state, sizeof(st))) return "init_challenger_state mismatch";
  ch.observe_cap(pr.aux_cap);
  std::vector<F2> alphas;
  for (int i = 0; i < cfg.num_challenges; i++) alphas.push_back(F2(ch.get_challenge()));
  ch.observe_cap(pr.quotient_cap);
  F2 zeta = ch.get_ext_challenge();
  for (auto& x : pr.local_values) ch.observe_ext(x);
  for (auto& x : pr.aux_polys) ch.observe_ext(x);
  for (auto& x : pr.quotient_polys) ch.observe_ext(x);
  for (auto& x : pr.next_values) ch.observe_ext(x);
  for (auto& x : pr.aux_polys_next) ch.observe_ext(x);
  for (u64 x : pr.ctl_zs_first) ch.observe_ext(F2(x));
  // fri_challenges
  F2 fri_alpha = ch.get_ext_challenge();
  std::vector<int> arities = cfg.fri_arities(degree_bits);
  if (pr.commit_caps.size() != arities.size()) return "bad number of FRI layers";
  std::vector<F2> fri_betas;
  for (auto& cap : pr.commit_caps) {
    ch.observe_cap(cap);
    fri_betas.push_back(ch.get_ext_challenge());
  }
  for (auto& x : pr.final_poly) ch.observe_ext(x);
  ch.observe_element(pr.pow_witness);
  u64 pow_response = ch.get_challenge();
  std::vector<size_t> query_indices;
  for (int q = 0; q < cfg.num_queries; q++) query_indices.push_back(ch.get_challenge() % M);

  // vanishing polynomial at zeta
  u64 g = gl_root_of_unity(degree_bits);
  F2 zeta_pow = zeta.exp_power_of_2(degree_bits);
  F2 z_h = zeta_pow - F2(1);
  F2 nF((u64)N % GL_P);
  F2 l0 = z_h * (nF * (zeta - F2(1))).inv();
  F2 llast = z_h * (nF * (zeta.scalar_mul(g) - F2(1))).inv();
  Consumer<F2> cc(alphas, zeta - F2(gl_inv(g)), l0, llast);
  AuxVars<F2> av{pr.aux_polys.data(), pr.aux_polys_next.data()};
  eval_vanishing<F2>(def, cfg, pr.local_values.data(), pr.next_values.data(), av, betas, gammas, cc, def.eval_ext);
  for (int j = 0; j < cfg.num_challenges; j++) {
    // t(zeta) = t_0 + t_1 zeta^N
    F2 t = pr.quotient_polys[2 * j] + pr.quotient_polys[2 * j + 1] * zeta_pow;
    if (cc.accs[j] != z_h * t) return "Mismatch between evaluation and opening of quotient polynomial";
  }

  // verify_fri_proof
  if (pow_response >> (64 - cfg.pow_bits)) return "Invalid proof of work witness";
  if ((int)pr.queries.size() != cfg.num_queries) return "Number of query rounds does not match config";
  int sum_ar = 0;
  for (int a : arities) sum_ar += a;
  if (pr.final_poly.size() != (size_t(1) << (degree_bits - sum_ar))) return "bad final poly length";
  std::vector<FriBatchInfo> inst = fri_instance(def, cfg, zeta, g);
  // PrecomputedReducedOpenings
  std::vector<std::vector<F2>> opening_batches(3);
  for (auto& x : pr.local_values) opening_batches[0].push_back(x);
  for (auto& x : pr.aux_polys) opening_batches[0].push_back(x);
  for (auto& x : pr.quotient_polys) opening_batches[0].push_back(x);
  for (auto& x : pr.next_values) opening_batches[1].push_back(x);
  for (auto& x : pr.aux_polys_next) opening_batches[1].push_back(x);
  for (u64 x : pr.ctl_zs_first) opening_batches[2].push_back(F2(x));
  std::vector<F2> reduced(3);
  for (int b = 0; b < 3; b++) reduced[b] = eval_poly_ext(opening_batches[b], fri_alpha);  // sum v_j alpha^j
  const std::vector<Digest>* caps[3] = {&pr.trace_cap, &pr.aux_cap, &pr.quotient_cap};
  int widths[3] = {def.W, A, nq};
  for (int q = 0; q < cfg.num_queries; q++) {
    size_t x_index = query_indices[q];
    const FriQueryRound& r = pr.queries[q];
    if (r.initial.size() != 3 || r.steps.size() != arities.size()) return "bad query round shape";
    for (int t = 0; t < 3; t++) {
      if ((int)r.initial[t].leaf.size() != widths[t]) return "bad initial leaf width";
      if (!merkle_verify(r.initial[t].leaf.data(), r.initial[t].leaf.size(), x_index, *caps[t], r.initial[t].path))
        return "Invalid Merkle proof (initial tree)";
    }
    u64 subgroup_x = gl_mul(GL_GENERATOR, gl_pow(gl_root_of_unity(lde_bits), reverse_bits(x_index, lde_bits)));
    // fri_combine_initial
    F2 sum(0);
    for (int b = 0; b < 3; b++) {
      F2 acc(0);
      const auto& polys = inst[b].polys;
      for (size_t j = polys.size(); j-- > 0;) acc = acc * fri_alpha + F2(r.initial[polys[j].first].leaf[polys[j].second]);
      F2 num = acc - reduced[b];
      F2 den = F2(subgroup_x) - inst[b].point;
      sum = sum * fri_alpha.pow(polys.size()) + num * den.inv();
    }
    F2 old_eval = sum;
    size_t xi = x_index;
    for (size_t l = 0; l < arities.size(); l++) {
      size_t arity = size_t(1) << arities[l];
      const std::vector<F2>& evals = r.steps[l].evals;
      if (evals.size() != arity) return "bad evals length";
      size_t coset_index = xi >> arities[l], within = xi & (arity - 1);
      if (evals[within] != old_eval) return "FRI consistency check failed";
      // compute_evaluation: interpolate {(x g^i, P(x g^i))} at beta
      u64 gg = gl_root_of_unity(arities[l]);
      std::vector<F2> ev(arity);
      for (size_t i = 0; i < arity; i++) ev[reverse_bits(i, arities[l])] = evals[i];
      size_t rev_within = reverse_bits(within, arities[l]);
      u64 coset_start = gl_mul(subgroup_x, gl_pow(gg, arity - rev_within));
      std::vector<u64> pts(arity);
      pts[0] = coset_start;
      for (size_t i = 1; i < arity; i++) pts[i] = gl_mul(pts[i - 1], gg);
      F2 res(0);
      for (size_t i = 0; i < arity; i++) {  // plain Lagrange interpolation
        F2 num(1);
        u64 den = 1;
        for (size_t k = 0; k < arity; k++)
          if (k != i) {
            num = num * (fri_betas[l] - F2(pts[k]));
            den = gl_mul(den, gl_sub(pts[i], pts[k]));
          }
        res += ev[i] * num.scalar_mul(gl_inv(den));
      }
      old_eval = res;
      std::vector<u64> flat(2 * arity);
      for (size_t k = 0; k < arity; k++) {
        flat[2 * k] = evals[k].c0;
        flat[2 * k + 1] = evals[k].c1;
      }
      if (!merkle_verify(flat.data(), flat.size(), coset_index, pr.commit_caps[l], r.steps[l].path))
        return "Invalid Merkle proof (FRI layer)";
      for (int k = 0; k < arities[l]; k++) subgroup_x = gl_mul(subgroup_x, subgroup_x);
      xi = coset_index;
    }
    if (eval_poly_ext(pr.final_poly, F2(subgroup_x)) != old_eval) return "Final polynomial evaluation is invalid";
  }

  // CTL check (verify_cross_table_lookups with only extra looking values; ctl_values.rs:28-47)
  size_t zi = 0;
  for (size_t k = 0; k < def.ctls.size(); k++)
    for (int c = 0; c < cfg.num_challenges; c++) {
      u64 sum = 0;
      for (auto& v : extra_looking_values[k]) {
        std::vector<F> fv;
        for (u64 x : v) fv.push_back(F(x));
        sum = gl_add(sum, gl_inv(combine<F>(fv, F(betas[c]), F(gammas[c])).v));
      }
      if (sum != pr.ctl_zs_first[zi]) return "CTL sum mismatch";
      zi++;
    }
  (void)num_lookup;
  return "";
}

}  // namespace orc
