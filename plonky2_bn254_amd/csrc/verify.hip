// Native verification of a proof: bn254s_verify.
//
// Replaces the reference's `verify` (src/starks/common/verifier.rs:32-98: challenges from the transcript, starky's
// verify_stark_proof_with_challenges, verify_cross_table_lookups with only extra looking values, ctl_values.rs:28-47) and
// plonky2's verify_fri_proof, on the flat word layout of include/bn254_stark.h.
//
// The transcript, the FRI checks (Merkle paths, initial combination, arity-16 folds, final polynomial) and the CTL sums
// run on the host.  The vanishing check needs sum_e alpha^(K-1-e) c_e(local(zeta), next(zeta)) * selector_e with the
// openings in the quadratic extension.  The AIR exists once in this library, as the quotient kernels over the base
// field, so it is evaluated there: every constraint has degree <= 3 in the opened values and base-field coefficients
// (alpha, beta, gamma are base-field challenges), hence along the line v(t) = v0 + t v1 through the opening v0 + v1 X the
// weighted sum is a cubic in t.  The kernels evaluate it at t = 0..4 on a tiny synthetic "LDE" (the fifth point checks
// the degree bound), once per selector class (plain / transition / first row / last row, chosen through custom point
// tables), the cubic is interpolated and X^2 = 7 substituted.  Prover and verifier therefore share one statement of the
// AIR; the independent restatement lives in oracle/ and is what the tests compare against.
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"
#include "prover.h"
#include "quotient.h"
#include "transcript.h"
#include "verify_air_host.h"

namespace {

struct View {  // offsets into the word layout
  int W, A, NQ = 4, NCTLZ = 4, L, log_n, log_m2, cap_h, P;
  std::vector<int> layer_path;
  size_t caps, local, next, aux, aux_next, ctlz, quot, fri_caps, queries, wpq, final_poly, final_len, pow, init, total;
};

bool make_view(int kind, const bn254s_params& P, int degree_bits, View& v) {
  StarkShape sh = shape_for(kind);
  v.W = sh.W;
  v.A = sh.n_aux();
  v.log_n = degree_bits;
  v.log_m2 = degree_bits + P.rate_bits;
  v.cap_h = P.cap_height;
  v.P = v.log_m2 - v.cap_h;
  std::vector<int> ar = fri_arities(P, degree_bits);
  v.L = (int)ar.size();
  size_t o = 0;
  v.caps = o; o += 3 * 64;
  v.local = o; o += 2 * (size_t)v.W;
  v.next = o; o += 2 * (size_t)v.W;
  v.aux = o; o += 2 * (size_t)v.A;
  v.aux_next = o; o += 2 * (size_t)v.A;
  v.ctlz = o; o += v.NCTLZ;
  v.quot = o; o += 2 * (size_t)v.NQ;
  v.fri_caps = o; o += (size_t)v.L * 64;
  v.queries = o;
  v.wpq = (size_t)(v.W + v.A + v.NQ) + 3 * (size_t)v.P * 4;
  int bits = v.log_m2, sum = 0;
  v.layer_path.clear();
  for (int l = 0; l < v.L; l++) {
    bits -= ar[l];
    sum += ar[l];
    int pl = bits - v.cap_h;
    if (pl < 0) return false;
    v.layer_path.push_back(pl);
    v.wpq += 2 * ((size_t)1 << ar[l]) + (size_t)pl * 4;
  }
  o += v.wpq * P.num_queries;
  v.final_len = (size_t)1 << (degree_bits - sum);
  v.final_poly = o; o += 2 * v.final_len;
  v.pow = o; o += 1;
  v.init = o; o += 12;
  v.total = o;
  return true;
}

// ---- host hashing (PoseidonHash::hash_or_noop / two_to_one, MerkleProof verification) ------------------------------
void hash_or_noop(const u64* in, size_t n, u64 out[4]) {
  if (n <= 4) {
    for (size_t i = 0; i < 4; i++) out[i] = i < n ? in[i] : 0;
    return;
  }
  u64 st[12] = {0};
  for (size_t i = 0; i < n; i += 8) {
    size_t m = n - i < 8 ? n - i : 8;
    for (size_t k = 0; k < m; k++) st[k] = in[i + k];
    poseidon_permute(st);
  }
  memcpy(out, st, 32);
}
bool merkle_ok(const u64* leaf, size_t leaf_len, size_t index, const u64* cap, const u64* path, int path_len) {
  u64 cur[4];
  hash_or_noop(leaf, leaf_len, cur);
  for (int i = 0; i < path_len; i++) {
    u64 nxt[4];
    if (index & 1) poseidon_two_to_one(path + 4 * i, cur, nxt);
    else poseidon_two_to_one(cur, path + 4 * i, nxt);
    memcpy(cur, nxt, 32);
    index >>= 1;
  }
  return memcmp(cur, cap + 4 * index, 32) == 0;
}

inline gl2 ext(const u64* w) { return gl2_make(w[0], w[1]); }
inline gl2 base(u64 x) { return gl2_make(x, 0); }
u32 rev_bits(u32 x, int bits) { return bits ? bitrev32(x, bits) : 0; }

// ---- vanishing sum through the quotient kernels (see the header comment) -------------------------------------------
constexpr unsigned VLOG = 7;  // synthetic domain: N = 128, 2N = 256 points = one workgroup
constexpr int VT = 5, VS = 4, VPTS = VT * VS;

#define VCHK(call)                                             \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) {                                    \
      err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return BN254S_E_HIP;                                     \
    }                                                          \
  } while (0)

int vanishing_on_gpu(bn254s_ctx* c, int kind, const StarkShape& sh, const u64* words, const View& v, const u64 alphas[2],
                     const u64 betas[2], const u64 gammas[2], gl2 z_last, gl2 l_first, gl2 l_last, gl2 out[2], std::string& err) {
  const size_t M2 = (size_t)2 << VLOG, N = (size_t)1 << VLOG;
  const int W = v.W, A = v.A, K = sh.n_total_constraints();
  hipStream_t st = c->stream;
  // synthetic LDE: point j = 5 s + t holds local(t) = v0 + t v1; its "next" position holds next(t)
  std::vector<u64> tl((size_t)W * M2, 0), al((size_t)A * M2, 0), px(M2, 0), pf(M2, 0), pl(M2, 0);
  size_t jn[VPTS];
  for (int j = 0; j < VPTS; j++) {
    u32 k = bitrev32((u32)j, VLOG);
    jn[j] = bitrev32((k + 1) & (u32)(N - 1), VLOG);
    if (jn[j] < (size_t)VPTS) {
      err = "internal: synthetic next position collides";
      return BN254S_E_INTERNAL;
    }
  }
  auto fill = [&](std::vector<u64>& buf, int ncols, size_t off_local, size_t off_next) {
    for (int col = 0; col < ncols; col++) {
      const u64 *lv = words + off_local + 2 * (size_t)col, *nv = words + off_next + 2 * (size_t)col;
      for (int j = 0; j < VPTS; j++) {
        u64 t = (u64)(j % VT);
        buf[(size_t)col * M2 + j] = gl_add(lv[0], gl_mul(t, lv[1]));
        buf[(size_t)col * M2 + jn[j]] = gl_add(nv[0], gl_mul(t, nv[1]));
      }
    }
  };
  fill(tl, W, v.local, v.next);
  fill(al, A, v.aux, v.aux_next);
  for (int t = 0; t < VT; t++) {  // selector classes: s = 1 transition (x - w^-1 with w^-1 := 0), 2 first row, 3 last row
    px[VT * 1 + t] = 1;
    pf[VT * 2 + t] = 1;
    pl[VT * 3 + t] = 1;
  }
  std::vector<u64> hW, hmzt;
  const int* mz_e0 = nullptr;
  int nblk = kind == KIND_G1 ? g1_quotient_mz_blocks(&mz_e0) : kind == KIND_G2 ? g2_quotient_mz_blocks(&mz_e0) : fq_quotient_mz_blocks(&mz_e0);
  quotient_host_tables(K, alphas, mz_e0, nblk, hW, hmzt);

  u64* d_tl = c->words("vf_tl", tl.size());
  u64* d_al = c->words("vf_al", al.size());
  u64* d_pt = c->words("vf_pt", 3 * M2);
  u64* d_W = c->words("vf_W", hW.size());
  u64* d_mzt = c->words("vf_mzt", hmzt.size());
  u64* d_out = c->words("vf_out", 4 * N);
  u64* d_part = c->words("vf_part", (size_t)QUOTIENT_MAX_PARTS * 2 * M2);
  if (!d_tl || !d_al || !d_pt || !d_W || !d_mzt || !d_out || !d_part) {
    err = c->BufPool::err;
    return BN254S_E_OOM;
  }
  VCHK(hipMemcpyAsync(d_tl, tl.data(), tl.size() * 8, hipMemcpyHostToDevice, st));
  VCHK(hipMemcpyAsync(d_al, al.data(), al.size() * 8, hipMemcpyHostToDevice, st));
  VCHK(hipMemcpyAsync(d_pt, px.data(), M2 * 8, hipMemcpyHostToDevice, st));
  VCHK(hipMemcpyAsync(d_pt + M2, pf.data(), M2 * 8, hipMemcpyHostToDevice, st));
  VCHK(hipMemcpyAsync(d_pt + 2 * M2, pl.data(), M2 * 8, hipMemcpyHostToDevice, st));
  VCHK(hipMemcpyAsync(d_W, hW.data(), hW.size() * 8, hipMemcpyHostToDevice, st));
  VCHK(hipMemcpyAsync(d_mzt, hmzt.data(), hmzt.size() * 8, hipMemcpyHostToDevice, st));
  QPointTables pt;
  pt.x = d_pt;
  pt.lfirst = d_pt + M2;
  pt.llast = d_pt + 2 * M2;
  QArgs QA;
  quotient_fill_args(QA, sh, d_tl, d_al, d_W, d_mzt, pt, betas, gammas, VLOG, d_out, d_part);
  QA.zh_inv[0] = QA.zh_inv[1] = 1;  // no division by Z_H
  QA.w_inv = 0;                     // transition selector = the x table itself
  if (kind == KIND_G1) g1_quotient_launch(QA, sh, st);
  else if (kind == KIND_G2) g2_quotient_launch(QA, sh, st);
  else fq_quotient_launch(QA, sh, st);
  std::vector<u64> ho(4 * N);
  VCHK(hipMemcpyAsync(ho.data(), d_out, ho.size() * 8, hipMemcpyDeviceToHost, st));
  VCHK(hipStreamSynchronize(st));
  VCHK(hipGetLastError());
  const u64 inv2 = gl_inv(2), inv6 = gl_inv(6);
  const gl2 sel[VS] = {gl2_make(1, 0), z_last, l_first, l_last};
  for (int a = 0; a < 2; a++) {
    gl2 tot = gl2_make(0, 0);
    u64 plain[VT];
    for (int s = 0; s < VS; s++) {
      u64 y[VT];
      for (int t = 0; t < VT; t++) {
        int j = VT * s + t;
        u64 e = ho[(size_t)(a * 2 + 0) * N + bitrev32((u32)j, VLOG)];  // coset 0, natural index
        if (s == 0) plain[t] = e;
        y[t] = s == 0 ? e : gl_sub(e, plain[t]);
      }
      // Newton form on t = 0..3: y0 + d1 t + d2 t(t-1)/2 + d3 t(t-1)(t-2)/6
      u64 d1 = gl_sub(y[1], y[0]);
      u64 d2 = gl_add(gl_sub(y[2], gl_dbl(y[1])), y[0]);
      u64 d3 = gl_sub(gl_add(gl_sub(y[3], gl_mul(3, y[2])), gl_mul(3, y[1])), y[0]);
      u64 p4 = gl_add(gl_add(y[0], gl_mul(4, d1)), gl_add(gl_mul(6, d2), gl_mul(4, d3)));
      if (p4 != y[4]) {
        err = "internal: a constraint class is not cubic in the openings";
        return BN254S_E_INTERNAL;
      }
      // t -> X with X^2 = 7: t(t-1) = 7 - X, t(t-1)(t-2) = 9X - 21
      u64 h2 = gl_mul(d2, inv2), h3 = gl_mul(d3, inv6);
      gl2 g = gl2_make(gl_sub(gl_add(y[0], gl_mul(7, h2)), gl_mul(21, h3)), gl_add(gl_sub(d1, h2), gl_mul(9, h3)));
      tot = gl2_add(tot, gl2_mul(sel[s], g));
    }
    out[a] = tot;
  }
  return BN254S_OK;
}

// value columns of the two cross-table lookups for instance k (scalar_mul_ctl.rs:57-80, g2 twin, exp_ctl.rs:54-75)
void limbs16(const u64* words4, u64* out) {
  for (int i = 0; i < 16; i++) out[i] = (words4[i >> 2] >> (16 * (i & 3))) & 0xFFFF;
}
void ctl_rows(int kind, const u64* scalars, const u64* x, const u64* off, const u64* outputs, size_t k, std::vector<u64>& in,
              std::vector<u64>& out) {
  const int PW = point_words(kind), PL = 4 * PW;  // 16-bit limbs per point
  in.assign((kind == KIND_FQ ? PL : 2 * PL) + 17, 0);
  out.assign(PL + 1, 0);
  size_t m = 0;
  for (int w = 0; w < PW; w += 4, m += 16) limbs16(x + PW * k + w, &in[m]);
  if (kind != KIND_FQ)
    for (int w = 0; w < PW; w += 4, m += 16) limbs16(off + PW * k + w, &in[m]);
  limbs16(scalars + 4 * k, &in[m]);
  m += 16;
  in[m] = k;
  m = 0;
  for (int w = 0; w < PW; w += 4, m += 16) limbs16(outputs + PW * k + w, &out[m]);
  out[m] = k;
}

int verify_impl(bn254s_ctx* c, int kind, const bn254s_params& P, int degree_bits, const u64* w, size_t n_words, const u64* scalars,
                const u64* x, const u64* off, const u64* outputs, size_t n, std::string& err) {
  View v;
  if (!make_view(kind, P, degree_bits, v) || v.total != n_words) {
    err = "bad proof shape";
    return BN254S_E_VERIFY;
  }
  const StarkShape sh = shape_for(kind);
  const size_t N = (size_t)1 << degree_bits, M2 = (size_t)1 << v.log_m2;
  const int W = v.W, A = v.A, NQ = v.NQ;
  for (size_t i = 0; i < n_words; i++)
    if (w[i] >= GL_P && i != v.pow) {
      err = "non-canonical field element";
      return BN254S_E_VERIFY;
    }
  // ---- challenges (prover.rs:40-54 / verifier.rs:47-74 + starky get_challenges) ---------------------------------
  Challenger ch;
  ch.observe_n(w + v.caps, 64);
  u64 betas[2], gammas[2], alphas[2];
  for (int i = 0; i < 2; i++) {
    betas[i] = ch.challenge();
    gammas[i] = ch.challenge();
  }
  u64 st12[12];
  ch.compact(st12);
  if (memcmp(st12, w + v.init, sizeof(st12))) {
    err = "init_challenger_state mismatch";
    return BN254S_E_VERIFY;
  }
  ch.observe_n(w + v.caps + 64, 64);
  for (int i = 0; i < 2; i++) alphas[i] = ch.challenge();
  ch.observe_n(w + v.caps + 128, 64);
  gl2 zeta = ch.challenge_ext();
  ch.observe_n(w + v.local, 2 * (size_t)W);
  ch.observe_n(w + v.aux, 2 * (size_t)A);
  ch.observe_n(w + v.quot, 2 * (size_t)NQ);
  ch.observe_n(w + v.next, 2 * (size_t)W);
  ch.observe_n(w + v.aux_next, 2 * (size_t)A);
  for (int i = 0; i < v.NCTLZ; i++) {
    ch.observe(w[v.ctlz + i]);
    ch.observe(0);
  }
  gl2 fri_alpha = ch.challenge_ext();
  const std::vector<int> arities = fri_arities(P, degree_bits);
  std::vector<gl2> fri_betas;
  for (int l = 0; l < v.L; l++) {
    ch.observe_n(w + v.fri_caps + 64 * (size_t)l, 64);
    fri_betas.push_back(ch.challenge_ext());
  }
  ch.observe_n(w + v.final_poly, 2 * v.final_len);
  ch.observe(w[v.pow]);
  u64 pow_response = ch.challenge();
  std::vector<size_t> qidx(P.num_queries);
  for (auto& q : qidx) q = (size_t)(ch.challenge() % M2);

  // ---- vanishing polynomial at zeta --------------------------------------------------------------------------------
  const u64 g = gl_root_of_unity(degree_bits);
  gl2 zeta_pow = zeta;
  for (int i = 0; i < degree_bits; i++) zeta_pow = gl2_mul(zeta_pow, zeta_pow);
  const gl2 one = gl2_make(1, 0);
  gl2 z_h = gl2_sub(zeta_pow, one);
  if (gl2_eq(z_h, gl2_make(0, 0))) {
    err = "Opening point is in the subgroup.";
    return BN254S_E_VERIFY;
  }
  const u64 nF = (u64)N % GL_P;
  gl2 l_first = gl2_mul(z_h, gl2_inv(gl2_mul_base(gl2_sub(zeta, one), nF)));
  gl2 l_last = gl2_mul(z_h, gl2_inv(gl2_mul_base(gl2_sub(gl2_mul_base(zeta, g), one), nF)));
  gl2 z_last = gl2_sub(zeta, base(gl_inv(g)));
  gl2 van[2];
  if (c) {  // batched hosts with a context: the constraint sum through the quotient kernels
    int rc = vanishing_on_gpu(c, kind, sh, w, v, alphas, betas, gammas, z_last, l_first, l_last, van, err);
    if (rc != BN254S_OK) return rc;
  } else {  // no context: the independent host statement of the AIR (verify_air_host.h), no GPU involved
    if (!host_air::vanishing_on_host(kind, sh, w + v.local, w + v.next, w + v.aux, w + v.aux_next, alphas, betas, gammas, z_last,
                                     l_first, l_last, van, err))
      return BN254S_E_INTERNAL;
  }
  for (int j = 0; j < 2; j++) {
    gl2 t = gl2_add(ext(w + v.quot + 4 * j), gl2_mul(ext(w + v.quot + 4 * j + 2), zeta_pow));
    if (!gl2_eq(van[j], gl2_mul(z_h, t))) {
      err = "Mismatch between evaluation and opening of quotient polynomial";
      return BN254S_E_VERIFY;
    }
  }

  // ---- verify_fri_proof -----------------------------------------------------------------------------------------------
  if (pow_response >> (64 - P.pow_bits)) {
    err = "Invalid proof of work witness";
    return BN254S_E_VERIFY;
  }
  // batches of fri_instance: zeta (trace | aux | quotient), g zeta (trace | aux), 1 (CTL Z columns of the aux oracle)
  const int num_lookup = sh.n_lookup_cols();
  struct PolyRef {
    int oracle, col;
  };
  std::vector<PolyRef> batch[3];
  std::vector<gl2> opened[3];
  for (int p = 0; p < W; p++) { batch[0].push_back({0, p}); opened[0].push_back(ext(w + v.local + 2 * (size_t)p)); }
  for (int p = 0; p < A; p++) { batch[0].push_back({1, p}); opened[0].push_back(ext(w + v.aux + 2 * (size_t)p)); }
  for (int p = 0; p < NQ; p++) { batch[0].push_back({2, p}); opened[0].push_back(ext(w + v.quot + 2 * (size_t)p)); }
  for (int p = 0; p < W; p++) { batch[1].push_back({0, p}); opened[1].push_back(ext(w + v.next + 2 * (size_t)p)); }
  for (int p = 0; p < A; p++) { batch[1].push_back({1, p}); opened[1].push_back(ext(w + v.aux_next + 2 * (size_t)p)); }
  for (int p = 0; p < v.NCTLZ; p++) { batch[2].push_back({1, num_lookup + p}); opened[2].push_back(base(w[v.ctlz + p])); }
  const gl2 points[3] = {zeta, gl2_mul_base(zeta, g), one};
  gl2 reduced[3], alpha_len[3];
  for (int b = 0; b < 3; b++) {
    gl2 acc = gl2_make(0, 0);
    for (size_t j = opened[b].size(); j-- > 0;) acc = gl2_add(gl2_mul(acc, fri_alpha), opened[b][j]);
    reduced[b] = acc;
    alpha_len[b] = gl2_pow(fri_alpha, batch[b].size());
  }
  const int widths[3] = {W, A, NQ};
  const u64 w_lde = gl_root_of_unity(v.log_m2);
  for (uint32_t q = 0; q < P.num_queries; q++) {
    const u64* r = w + v.queries + v.wpq * q;
    const size_t x_index = qidx[q];
    const u64* leaf[3];
    for (int t = 0; t < 3; t++) {
      leaf[t] = r;
      const u64* path = r + widths[t];
      if (!merkle_ok(leaf[t], widths[t], x_index, w + v.caps + 64 * (size_t)t, path, v.P)) {
        err = "Invalid Merkle proof (initial tree)";
        return BN254S_E_VERIFY;
      }
      r = path + 4 * (size_t)v.P;
    }
    u64 subgroup_x = gl_mul(GL_GEN, gl_pow(w_lde, rev_bits((u32)x_index, v.log_m2)));
    gl2 sum = gl2_make(0, 0);  // fri_combine_initial
    for (int b = 0; b < 3; b++) {
      gl2 acc = gl2_make(0, 0);
      for (size_t j = batch[b].size(); j-- > 0;)
        acc = gl2_add(gl2_mul(acc, fri_alpha), base(leaf[batch[b][j].oracle][batch[b][j].col]));
      gl2 num = gl2_sub(acc, reduced[b]);
      gl2 den = gl2_sub(base(subgroup_x), points[b]);
      sum = gl2_add(gl2_mul(sum, alpha_len[b]), gl2_mul(num, gl2_inv(den)));
    }
    gl2 old_eval = sum;
    size_t xi = x_index;
    for (int l = 0; l < v.L; l++) {
      const int ab = arities[l];
      const size_t arity = (size_t)1 << ab;
      const u64* evals = r;
      const u64* path = r + 2 * arity;
      r = path + 4 * (size_t)v.layer_path[l];
      const size_t coset_index = xi >> ab, within = xi & (arity - 1);
      if (!gl2_eq(ext(evals + 2 * within), old_eval)) {
        err = "FRI consistency check failed";
        return BN254S_E_VERIFY;
      }
      // compute_evaluation: interpolate {(x gg^i, P(x gg^i))} at beta
      const u64 gg = gl_root_of_unity(ab);
      std::vector<gl2> ev(arity);
      for (size_t i = 0; i < arity; i++) ev[rev_bits((u32)i, ab)] = ext(evals + 2 * i);
      const size_t rev_within = rev_bits((u32)within, ab);
      std::vector<u64> pts(arity);
      pts[0] = gl_mul(subgroup_x, gl_pow(gg, arity - rev_within));
      for (size_t i = 1; i < arity; i++) pts[i] = gl_mul(pts[i - 1], gg);
      gl2 res = gl2_make(0, 0);
      for (size_t i = 0; i < arity; i++) {
        gl2 num = one;
        u64 den = 1;
        for (size_t k = 0; k < arity; k++)
          if (k != i) {
            num = gl2_mul(num, gl2_sub(fri_betas[l], base(pts[k])));
            den = gl_mul(den, gl_sub(pts[i], pts[k]));
          }
        res = gl2_add(res, gl2_mul(ev[i], gl2_mul_base(num, gl_inv(den))));
      }
      old_eval = res;
      if (!merkle_ok(evals, 2 * arity, coset_index, w + v.fri_caps + 64 * (size_t)l, path, v.layer_path[l])) {
        err = "Invalid Merkle proof (FRI layer)";
        return BN254S_E_VERIFY;
      }
      for (int k = 0; k < ab; k++) subgroup_x = gl_mul(subgroup_x, subgroup_x);
      xi = coset_index;
    }
    gl2 fin = gl2_make(0, 0);
    for (size_t i = v.final_len; i-- > 0;) fin = gl2_add(gl2_mul_base(fin, subgroup_x), ext(w + v.final_poly + 2 * i));
    if (!gl2_eq(fin, old_eval)) {
      err = "Final polynomial evaluation is invalid";
      return BN254S_E_VERIFY;
    }
  }

  // ---- cross-table lookups against the claimed inputs / outputs (ctl_values.rs:28-47) -------------------------------
  u64 sums[2][2] = {{0, 0}, {0, 0}};
  std::vector<u64> in, out;
  for (size_t k = 0; k < n; k++) {
    ctl_rows(kind, scalars, x, off, outputs, k, in, out);
    const std::vector<u64>* rows[2] = {&in, &out};
    for (int t = 0; t < 2; t++)
      for (int cidx = 0; cidx < 2; cidx++) {
        u64 acc = 0;
        for (size_t m = rows[t]->size(); m-- > 0;) acc = gl_add(gl_mul(acc, betas[cidx]), (*rows[t])[m]);
        acc = gl_add(acc, gammas[cidx]);
        if (acc == 0) {
          err = "CTL denominator vanishes";
          return BN254S_E_VERIFY;
        }
        sums[t][cidx] = gl_add(sums[t][cidx], gl_inv(acc));
      }
  }
  for (int t = 0; t < 2; t++)
    for (int cidx = 0; cidx < 2; cidx++)
      if (sums[t][cidx] != w[v.ctlz + 2 * t + cidx]) {
        err = "CTL sum mismatch";
        return BN254S_E_VERIFY;
      }
  return BN254S_OK;
}

}  // namespace

extern "C" int bn254s_ctl_values(int kind, const uint64_t* scalars, const uint64_t* x, const uint64_t* off, const uint64_t* outputs,
                                 size_t n, uint64_t* in_rows, uint64_t* out_rows) {
  if (kind < 0 || kind > 2 || !scalars || !x || (kind != KIND_FQ && !off) || !outputs || !in_rows || !out_rows)
    return BN254S_E_INVALID_ARG;
  std::vector<u64> in, out;
  for (size_t k = 0; k < n; k++) {
    ctl_rows(kind, scalars, x, off, outputs, k, in, out);
    memcpy(in_rows + k * in.size(), in.data(), in.size() * 8);
    memcpy(out_rows + k * out.size(), out.data(), out.size() * 8);
  }
  return BN254S_OK;
}

extern "C" int bn254s_verify_host(int kind, const bn254s_params* params, uint32_t degree_bits, const uint64_t* words, size_t n_words,
                                  const uint64_t* scalars, const uint64_t* x, const uint64_t* off, const uint64_t* outputs, size_t n,
                                  char* err_buf, size_t err_cap) {
  if (err_buf && err_cap) err_buf[0] = 0;
  if (kind < 0 || kind > 2 || !params || params->struct_size != sizeof(bn254s_params) || !words || !scalars || !x ||
      (kind != KIND_FQ && !off) || !outputs || n == 0 || degree_bits < 7 || degree_bits > 30)
    return BN254S_E_INVALID_ARG;
  std::string err;
  int rc;
  if (params->num_challenges != 2 || params->rate_bits != 1) {
    err = "only num_challenges = 2, rate_bits = 1 are supported";
    rc = BN254S_E_UNSUPPORTED;
  } else {
    rc = verify_impl(nullptr, kind, *params, (int)degree_bits, words, n_words, scalars, x, off, outputs, n, err);
  }
  if (rc != BN254S_OK && err_buf && err_cap) {
    strncpy(err_buf, err.c_str(), err_cap - 1);
    err_buf[err_cap - 1] = 0;
  }
  return rc;
}

extern "C" int bn254s_verify(bn254s_ctx* c, int kind, const bn254s_params* params, uint32_t degree_bits, const uint64_t* words,
                             size_t n_words, const uint64_t* scalars, const uint64_t* x, const uint64_t* off, const uint64_t* outputs,
                             size_t n) {
  if (!c || kind < 0 || kind > 2 || !params || params->struct_size != sizeof(bn254s_params) || !words || !scalars || !x ||
      (kind != KIND_FQ && !off) || !outputs || n == 0 || degree_bits < 7 || degree_bits > 30)
    return BN254S_E_INVALID_ARG;
  if (params->num_challenges != 2 || params->rate_bits != 1) {
    c->set_err("only num_challenges = 2, rate_bits = 1 are supported");
    return BN254S_E_UNSUPPORTED;
  }
  if (hipSetDevice(c->device) != hipSuccess) return BN254S_E_HIP;
  std::string err;
  int rc = verify_impl(c, kind, *params, (int)degree_bits, words, n_words, scalars, x, off, outputs, n, err);
  if (rc != BN254S_OK) c->set_err(err);
  return rc;
}
