/*
 * prove_g1.c - the C ABI of include/bn254_stark.h from plain C (no Python, no C++): proves a few G1 scalar multiplications
 * on GPU 0, verifies the proof with bn254s_verify and prints the per-stage GPU milliseconds.
 *
 * This is the call sequence the Rust shim of INTEGRATION.md makes from G1StarkProofGenerator::run_once
 * (reference src/generators/g1/stark_proof.rs:136-179).
 *
 *   gcc -std=c99 -Iinclude examples/prove_g1.c -Lplonky2_bn254_amd -lbn254stark -Wl,-rpath,$PWD/plonky2_bn254_amd -o prove_g1
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bn254_stark.h"

/* s * G + G for the generator G = (1, 2): scalars 1..n, x = offset = G */
int main(void) {
  enum { N = 4 };
  uint64_t scalars[N * 4], x[N * 8], off[N * 8];
  memset(scalars, 0, sizeof scalars);
  memset(x, 0, sizeof x);
  for (int i = 0; i < N; i++) {
    scalars[4 * i] = (uint64_t)(i + 2);
    x[8 * i] = 1;     /* x coordinate, little-endian words */
    x[8 * i + 4] = 2; /* y coordinate */
  }
  memcpy(off, x, sizeof x);

  bn254s_ctx* ctx = NULL;
  int rc = bn254s_ctx_create(0, &ctx);
  if (rc != BN254S_OK) {
    fprintf(stderr, "bn254s_ctx_create: %d (is a GPU visible?)\n", rc);
    return 2;
  }
  bn254s_params params;
  bn254s_params_default(&params);
  bn254s_proof* proof = NULL;
  rc = bn254s_prove_g1(ctx, &params, scalars, x, off, N, &proof);
  if (rc != BN254S_OK) {
    fprintf(stderr, "bn254s_prove_g1: %d (%s)\n", rc, bn254s_last_error(ctx));
    return 1;
  }
  const uint64_t *words, *outs;
  size_t n_words, n_outs, n_stages;
  const float* ms;
  bn254s_proof_words(proof, &words, &n_words);
  bn254s_proof_outputs(proof, &outs, &n_outs);
  bn254s_proof_stage_ms(proof, &ms, &n_stages);
  printf("proof: %zu words, degree_bits %d; (2G + G).x = %016llx...\n", n_words, bn254s_proof_degree_bits(proof),
         (unsigned long long)outs[3]);
  for (size_t i = 0; i < n_stages; i++) printf("  %-16s %7.2f ms\n", bn254s_stage_name(i), ms[i]);
  rc = bn254s_verify(ctx, 0, &params, (uint32_t)bn254s_proof_degree_bits(proof), words, n_words, scalars, x, off, outs, N);
  printf("bn254s_verify: %d%s%s\n", rc, rc ? " " : "", rc ? bn254s_last_error(ctx) : "");
  /* a flipped bit must be rejected */
  uint64_t* bad = (uint64_t*)malloc(n_words * 8);
  memcpy(bad, words, n_words * 8);
  bad[300] ^= 1;
  int rc2 = bn254s_verify(ctx, 0, &params, (uint32_t)bn254s_proof_degree_bits(proof), bad, n_words, scalars, x, off, outs, N);
  printf("corrupted proof: %d (%s)\n", rc2, bn254s_last_error(ctx));
  free(bad);
  bn254s_proof_free(proof);
  bn254s_ctx_destroy(ctx);
  return (rc == BN254S_OK && rc2 == BN254S_E_VERIFY) ? 0 : 1;
}
