// ORACLE (test infrastructure, NOT the product path).
// Poseidon-Goldilocks permutation (width 12, 8 full + 22 partial rounds, x^7), the plonky2 0.2.2
// hashing conventions (hash_no_pad / hash_or_noop / two_to_one), Merkle tree with cap, and the
// duplex-sponge Challenger.  Restates the un-vendored `plonky2::hash::{poseidon,hashing,merkle_tree}`
// and `plonky2::iop::challenger` used at reference src/starks/common/prover.rs:31-44,54
// (SURVEY.md App. A.3/A.4).  Permutation pinned by the KATs in tests/golden/poseidon_kat.json;
// the sponge/tree conventions are "parity unpinned" (see DESIGN.md).
#pragma once
#include "gl.hpp"
#include <array>
#include <cstring>

namespace orc {

static const u64 POSEIDON_RC[360] = {
#include "poseidon_constants.inc"
};
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG0 = 8;

static inline u64 sbox7(u64 x) {
  u64 x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x2, x);
  return gl_mul(x3, x4);
}
static inline void mds_layer(u64 s[12]) {
  // out[r] = sum_i s[(i+r)%12]*CIRC[i] + (r==0)*8*s[0]: the constants are below 2^6, so the thirteen 64 x 6-bit products of
  // a row sum to less than 2^74 in a 128-bit accumulator; one reduction per row.
  u64 t[24], out[12];
  for (int i = 0; i < 12; i++) t[i] = t[i + 12] = s[i];
  for (int r = 0; r < 12; r++) {
    u128 acc = r == 0 ? (u128)t[0] * MDS_DIAG0 : 0;
    for (int i = 0; i < 12; i++) acc += (u128)t[i + r] * MDS_CIRC[i];
    out[r] = gl_reduce128(acc);
  }
  for (int i = 0; i < 12; i++) s[i] = out[i];
}
static inline void poseidon_permute(u64 s[12]) {
  for (int rnd = 0; rnd < 30; rnd++) {
    for (int i = 0; i < 12; i++) s[i] = gl_add(s[i], POSEIDON_RC[12 * rnd + i]);
    if (rnd < 4 || rnd >= 26) {
      for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
    } else {
      s[0] = sbox7(s[0]);
    }
    mds_layer(s);
  }
}

// The same permutation with the 22 partial rounds in their sparse form (one S-box, a 12-term dot product and eleven
// multiply-adds per round instead of a dense 12 x 12 layer): algebraically identical, ~3x faster - the oracle hashes 20.7 M
// times per G1 proof, and bench.py's cpu_baseline times it.  Tables: tools/derive_poseidon_host_fast.py (derivation there;
// it checks them against the textbook form on the KATs).  tests/test_oracle_golden.py pins this function against
// poseidon_permute (above, the restatement proper) on the KATs, edge values and random states.
#include "poseidon_fast_tables.inc"
static inline u64 dot_reduce(u128 acc, u64 carries) {  // acc + carries 2^128 (carries <= 12); 2^128 = -2^32 mod p
  return gl_sub(gl_reduce128(acc), carries << 32);
}
static inline void full_round_fast(u64 s[12], const u64* rc) {
  for (int i = 0; i < 12; i++) s[i] = sbox7(gl_add(s[i], rc[i]));
  mds_layer(s);
}
static inline void poseidon_permute_fast(u64 s[12]) {
  for (int rnd = 0; rnd < 4; rnd++) full_round_fast(s, POSEIDON_RC + 12 * rnd);
  for (int t = 0; t < 22; t++) {
    const u64 x0 = sbox7(gl_add(s[0], PHF_K[t]));
    const u64 *w = PHF_W + 11 * t, *u = PHF_U + 11 * t;
    // new0 = M00 x0 + sum_i w_i s_(i+1): twelve 128-bit products summed in 192 bits (carry counted separately)
    u128 acc = (u128)x0 * PHF_M00[t];
    u64 carries = 0;
    for (int i = 0; i < 11; i++) {
      const u128 p = (u128)w[i] * s[i + 1];
      acc += p;
      carries += acc < p;
    }
    for (int i = 0; i < 11; i++) s[i + 1] = gl_reduce128((u128)u[i] * x0 + s[i + 1]);
    s[0] = dot_reduce(acc, carries);
  }
  u64 y[11];
  for (int r = 0; r < 11; r++) {
    u128 acc = 0;
    u64 carries = 0;
    for (int i = 0; i < 11; i++) {
      const u128 p = (u128)PHF_DENSE[11 * r + i] * s[i + 1];
      acc += p;
      carries += acc < p;
    }
    y[r] = dot_reduce(acc, carries);
  }
  for (int r = 0; r < 11; r++) s[r + 1] = y[r];
  full_round_fast(s, PHF_RC26M);
  for (int rnd = 27; rnd < 30; rnd++) full_round_fast(s, POSEIDON_RC + 12 * rnd);
}
// What the hashing below calls (orc_set_fast_poseidon(0) switches the whole oracle back to the textbook form).
static bool g_fast_poseidon = true;
static inline void poseidon_hash_permute(u64 s[12]) {
  if (g_fast_poseidon) poseidon_permute_fast(s);
  else poseidon_permute(s);
}

struct Digest {
  u64 e[4];
  bool operator==(const Digest& o) const { return !memcmp(e, o.e, sizeof(e)); }
};

// hash_no_pad: overwrite-mode sponge, rate 8.
static inline Digest hash_no_pad(const u64* x, size_t n) {
  u64 st[12] = {0};
  for (size_t off = 0; off < n; off += 8) {
    size_t len = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < len; i++) st[i] = x[off + i];
    poseidon_hash_permute(st);
  }
  Digest d;
  memcpy(d.e, st, sizeof(d.e));
  return d;
}
static inline Digest hash_or_noop(const u64* x, size_t n) {
  if (n <= 4) {
    Digest d = {{0, 0, 0, 0}};
    for (size_t i = 0; i < n; i++) d.e[i] = x[i];
    return d;
  }
  return hash_no_pad(x, n);
}
static inline Digest two_to_one(const Digest& l, const Digest& r) {
  u64 st[12] = {l.e[0], l.e[1], l.e[2], l.e[3], r.e[0], r.e[1], r.e[2], r.e[3], 0, 0, 0, 0};
  poseidon_hash_permute(st);
  Digest d;
  memcpy(d.e, st, sizeof(d.e));
  return d;
}

// Merkle tree over 2^k leaf digests with a cap at height `cap_height` (MerkleTree::new).
// layers[0] = leaf digests, layers[j] has 2^(k-j) nodes; cap = layers[k - cap_height].
struct MerkleTree {
  unsigned log_leaves = 0, cap_height = 0;
  std::vector<std::vector<Digest>> layers;
  const std::vector<Digest>& cap() const { return layers[log_leaves - cap_height]; }
  void build(std::vector<Digest>&& leaf_digests, unsigned cap_h) {
    log_leaves = log2_strict(leaf_digests.size());
    cap_height = cap_h;
    assert(cap_h <= log_leaves);
    layers.clear();
    layers.push_back(std::move(leaf_digests));
    for (unsigned j = 0; j < log_leaves - cap_h; j++) {
      const std::vector<Digest>& prev = layers.back();
      std::vector<Digest> next(prev.size() / 2);
#pragma omp parallel for schedule(static) if (next.size() >= 1024)
      for (size_t i = 0; i < next.size(); i++) next[i] = two_to_one(prev[2 * i], prev[2 * i + 1]);
      layers.push_back(std::move(next));
    }
  }
  // Siblings bottom-up (MerkleTree::prove).
  std::vector<Digest> prove(size_t idx) const {
    std::vector<Digest> p;
    for (unsigned j = 0; j < log_leaves - cap_height; j++) {
      p.push_back(layers[j][idx ^ 1]);
      idx >>= 1;
    }
    return p;
  }
};

static inline bool merkle_verify(const u64* leaf, size_t leaf_len, size_t idx, const std::vector<Digest>& cap,
                                 const std::vector<Digest>& proof) {
  Digest cur = hash_or_noop(leaf, leaf_len);
  for (const Digest& sib : proof) {
    cur = (idx & 1) ? two_to_one(sib, cur) : two_to_one(cur, sib);
    idx >>= 1;
  }
  return idx < cap.size() && cur == cap[idx];
}

// Challenger<F, PoseidonHash>.
struct Challenger {
  u64 state[12];
  std::vector<u64> in_buf, out_buf;
  Challenger() { memset(state, 0, sizeof(state)); }
  void duplexing() {
    assert(in_buf.size() <= 8);
    for (size_t i = 0; i < in_buf.size(); i++) state[i] = in_buf[i];
    in_buf.clear();
    poseidon_hash_permute(state);
    out_buf.assign(state, state + 8);
  }
  void observe_element(u64 e) {
    out_buf.clear();
    in_buf.push_back(e);
    if (in_buf.size() == 8) duplexing();
  }
  void observe_elements(const u64* e, size_t n) {
    for (size_t i = 0; i < n; i++) observe_element(e[i]);
  }
  void observe_hash(const Digest& d) { observe_elements(d.e, 4); }
  void observe_cap(const std::vector<Digest>& cap) {
    for (const Digest& d : cap) observe_hash(d);
  }
  void observe_ext(const F2& x) {
    observe_element(x.c0);
    observe_element(x.c1);
  }
  u64 get_challenge() {
    if (!in_buf.empty() || out_buf.empty()) duplexing();
    u64 r = out_buf.back();
    out_buf.pop_back();
    return r;
  }
  F2 get_ext_challenge() {
    u64 a = get_challenge();
    u64 b = get_challenge();
    return F2(a, b);
  }
  // compact(): flush inputs, drop outputs, return the 12-lane state (prover.rs:54).
  void compact(u64 out_state[12]) {
    if (!in_buf.empty()) duplexing();
    out_buf.clear();
    memcpy(out_state, state, sizeof(state));
  }
};

}  // namespace orc
