// Host-side Fiat-Shamir transcript: plonky2 `Challenger<F, PoseidonHash>` (duplex sponge, width 12,
// rate 8, overwrite mode) as driven by reference src/starks/common/prover.rs:40-44,54 and by starky's
// prove_with_commitment / FRI prover.  Sequential by nature (a few hundred dependent permutations per
// proof), so it runs on the host between kernel launches; everything it consumes is a few KB.
#pragma once
#include <vector>
#include <cstring>
#include "poseidon_dev.h"

struct Challenger {
  u64 state[12];
  u64 in_buf[8];
  int in_len = 0;
  u64 out_buf[8];
  int out_len = 0;
  Challenger() { memset(state, 0, sizeof(state)); }
  void duplexing() {
    for (int i = 0; i < in_len; i++) state[i] = in_buf[i];
    in_len = 0;
    poseidon_permute(state);
    memcpy(out_buf, state, sizeof(out_buf));
    out_len = 8;
  }
  void observe(u64 e) {
    out_len = 0;
    in_buf[in_len++] = e;
    if (in_len == 8) duplexing();
  }
  void observe_n(const u64* e, size_t n) {
    for (size_t i = 0; i < n; i++) observe(e[i]);
  }
  u64 challenge() {
    if (in_len != 0 || out_len == 0) duplexing();
    return out_buf[--out_len];
  }
  gl2 challenge_ext() {
    u64 a = challenge();
    u64 b = challenge();
    return gl2_make(a, b);
  }
  void compact(u64 out_state[12]) {
    if (in_len != 0) duplexing();
    out_len = 0;
    memcpy(out_state, state, sizeof(state));
  }
};
