/*
 * bn254_stark.h - C ABI of the MI355X-native prover for the BN254 scalar-multiplication STARKs.
 *
 * Drop-in boundary: one call replaces the body of the reference's witness generator
 *   G1StarkProofGenerator::run_once   src/generators/g1/stark_proof.rs:136-179
 * namely lines :143-163 (outputs = s*x+offset, generate_trace, starks::common::prover::prove).  The caller
 * (a Rust shim) keeps get_witness / set_witness / verify / set_stark_proof_target.
 * G2 (src/generators/g2/stark_proof.rs:136-179) and Fq-exp (src/generators/fq/stark_proof.rs:135-178)
 * entry points have the same shape.
 *
 * Wire formats (all little-endian u64 words, canonical = reduced, non-Montgomery):
 *   scalar  : 4 words, any 256-bit value (NOT reduced modulo the group order; common/utils.rs:21-25)
 *   Fq      : 4 words, value < p
 *   G1 point: 8 words = x, y (affine, never infinity; src/curves/g1.rs:170-174)
 *   G2 point: 16 words = x.c0, x.c1, y.c0, y.c1
 *   Goldilocks element: 1 word < 2^64 - 2^32 + 1; extension element: 2 words (c0, c1)
 *   Poseidon digest: 4 words
 *
 * Proof layout returned by bn254s_proof_words() - field order of starky's StarkProofWithMetadata
 * (reference src/starks/common/prover.rs:66-71), W = trace width, A = auxiliary polys, L = FRI layers,
 * P = lde_bits - cap_height Merkle path length of the initial trees:
 *   trace_cap[16*4] aux_cap[16*4] quotient_cap[16*4]
 *   openings: local_values[W*2] next_values[W*2] auxiliary_polys[A*2] auxiliary_polys_next[A*2]
 *             ctl_zs_first[4] quotient_polys[4*2]
 *   commit_phase_merkle_caps[L][16*4]
 *   query_round_proofs[84]: for oracle in (trace, aux, quotient): leaf[width] path[P*4];
 *                           for layer l: evals[16*2] path[P_l*4]
 *   final_poly[len*2]  pow_witness[1]  init_challenger_state[12]
 *
 * Errors: every function returns 0 on success or a negative BN254S_E_* code; the reference panics
 * (unwrap at stark_proof.rs:163,172), the Rust shim maps non-zero to panic!.
 * Threading: one call at a time per context (one host thread enters the context at a time); any number of contexts (one per
 * GPU / per host thread).  A batch opened with bn254s_prove_batch_begin stays in flight after _begin returns: until its _end
 * the context may still be entered, one call at a time, and every proving entry point (bn254s_prove_g1 / _g2 / _fq_exp /
 * _batch* / bn254s_map_to_g2) queues behind the open batches on the same worker pool and runs on a free slot (stream +
 * workspace) of its own, so it can never share device state with a proof of the open batch; bn254s_verify, _commit_values,
 * _generate_trace and the _bench_* calls use the context's own stream and buffers and are independent of open batches.
 */
#ifndef BN254_STARK_H
#define BN254_STARK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN254S_ABI_VERSION 1

enum {
  BN254S_OK = 0,
  BN254S_E_INVALID_ARG = -1,
  BN254S_E_HIP = -2,            /* HIP runtime / launch failure, see bn254s_last_error */
  BN254S_E_OOM = -3,
  BN254S_E_INVALID_POINT = -4,  /* a + (-a) met during the double-and-add chain (generate_g1_add, add.rs:49-51) */
  BN254S_E_UNSUPPORTED = -5,    /* shape not implemented by this build */
  BN254S_E_TRANSCRIPT = -6,     /* opening point inside the subgroup (starky "Opening point is in the subgroup") */
  BN254S_E_INTERNAL = -7,       /* a device-side self check failed (the reference's assert!s in modulus_zero.rs:82,99-103) */
  BN254S_E_VERIFY = -8          /* bn254s_verify: the proof was rejected; bn254s_last_error holds the reference's error text */
};

typedef struct bn254s_ctx bn254s_ctx;
typedef struct bn254s_proof bn254s_proof;

/* StarkConfig::standard_fast_config() + min_rows (stark_proof.rs:152-154). POD, versioned by struct_size. */
typedef struct bn254s_params {
  uint32_t struct_size;
  uint32_t security_bits;   /* 100 */
  uint32_t num_challenges;  /* 2 */
  uint32_t rate_bits;       /* 1 */
  uint32_t cap_height;      /* 4 */
  uint32_t pow_bits;        /* 16 */
  uint32_t arity_bits;      /* 4  (ConstantArityBits(4, 5)) */
  uint32_t final_poly_bits; /* 5 */
  uint32_t num_queries;     /* 84 */
  uint32_t min_rows_log2;   /* 16 */
} bn254s_params;

void bn254s_params_default(bn254s_params* p);
int bn254s_abi_version(void);

/* One context per GPU: owns the HIP stream(s), twiddle tables and the pooled device workspace.
 * SURVEY.md section 8(b) sketched `bn254s_ctx_create(device_ids, n)` and `device_mask` / `batch_mode` fields in
 * bn254s_params; this ABI keeps the parameters a pure restatement of StarkConfig and expresses both ideas as calls instead:
 * one context per device (this function, any number of times) + bn254s_prove_batch_multi(ctxs, n_ctx, ...) for "these
 * devices", and bn254s_prove_g1 (one proof for all jobs, what Bn254Hook::constrain needs) vs bn254s_prove_batch (independent
 * per_proof-sized proofs) for the batch mode. */
int bn254s_ctx_create(int device_id, bn254s_ctx** out);
void bn254s_ctx_destroy(bn254s_ctx* ctx);
const char* bn254s_last_error(const bn254s_ctx* ctx);
/* Gives the device memory of every idle slot back to the driver (a slot = stream + workspace of one proof in flight; the
 * workspaces are grow-only otherwise: after a 2^23-row proof slot 0 keeps ~245 GB).  Slots that are proving right now are left
 * alone.  The next proof on a trimmed slot allocates its workspace again (a few ms). */
int bn254s_ctx_trim(bn254s_ctx* ctx);

/* Prove n G1 scalar multiplications s_i * x_i + offset_i in ONE STARK (timestamps 0..n-1), like
 * G1ScalarMulStark::generate_trace + prove (scalar_mul_stark.rs:55-69, common/prover.rs:18-72).
 * rows = max(2^min_rows_log2, 512 n) rounded up to a power of two; 2^16 .. 2^23 rows (n <= 16384) are supported
 * (a G1 proof of 2^22 rows keeps about 200 GB resident, one of 2^23 rows about 245 GB: its workspace is laid out to fit;
 * a G2 proof of 2^23 rows, whose two LDEs alone would be 296 GB, runs in a streaming workspace that keeps only coefficients
 * resident and recomputes LDE rows where they are needed: 8.0 s, ~250 GB),
 * i.e. one proof can cover all calls of a circuit exactly as Bn254Hook::constrain batches them (hook.rs:63-71). */
int bn254s_prove_g1(bn254s_ctx* ctx, const bn254s_params* params, const uint64_t* scalars /* n x 4 */,
                    const uint64_t* x /* n x 8 */, const uint64_t* offset /* n x 8 */, size_t n, bn254s_proof** out);

/* Throughput entry point: n_total jobs are cut into ceil(n_total / per_proof) independent proofs
 * (per_proof = 128 gives the reference test shape, 2^16 rows) that are pipelined on the GPU.
 * proofs_out must have room for that many pointers. */
int bn254s_prove_g1_batch(bn254s_ctx* ctx, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                          const uint64_t* offset, size_t n_total, size_t per_proof, bn254s_proof** proofs_out);

/* Generic form of the above: kind 0 = G1, 1 = G2 (points 16 words), 2 = Fq exp (x 4 words, offset NULL). */
int bn254s_prove_batch(bn254s_ctx* ctx, int kind, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                       const uint64_t* offset, size_t n_total, size_t per_proof, bn254s_proof** proofs_out);

/* The same call in two halves, for a caller that keeps the GPU fed: _begin queues the proofs of the batch on the context's
 * worker threads and returns; _end waits for them and returns what bn254s_prove_batch would have (on an error every proof of
 * the batch is freed and its slot in proofs_out is NULL).  Batches are served in the order of their _begin calls, up to twelve
 * proofs in flight in total, so the first proofs of the next batch run while the last ones of the current batch finish (the
 * reference's callers do the same with rayon over independent circuits).  scalars / x / offset / proofs_out must stay valid
 * until _end; every handle must be passed to _end exactly once, before bn254s_ctx_destroy.  bn254s_prove_batch is
 * _begin followed by _end, and the single-proof entry points are a batch of one proof: all of them may be called while
 * handles are open (see "Threading" at the top). */
typedef struct bn254s_batch bn254s_batch;
int bn254s_prove_batch_begin(bn254s_ctx* ctx, int kind, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                             const uint64_t* offset, size_t n_total, size_t per_proof, bn254s_proof** proofs_out,
                             bn254s_batch** handle);
int bn254s_prove_batch_end(bn254s_batch* handle);

/* Several GPUs from one process: proof i is proven by ctxs[i mod n_ctx] (one context per GPU); no inter-GPU traffic.
 * Same arguments and results as bn254s_prove_batch otherwise. */
int bn254s_prove_batch_multi(bn254s_ctx** ctxs, size_t n_ctx, int kind, const bn254s_params* params, const uint64_t* scalars,
                             const uint64_t* x, const uint64_t* offset, size_t n_total, size_t per_proof, bn254s_proof** proofs);

/* Same for G2 (points n x 16 words: x.c0, x.c1, y.c0, y.c1): src/generators/g2/stark_proof.rs:136-179. */
int bn254s_prove_g2(bn254s_ctx* ctx, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                    const uint64_t* offset, size_t n, bn254s_proof** out);
/* Fq exponentiation x_i ^ s_i (FqExpInput { s, x }, src/starks/fields/exp_stark.rs:36-39): no offset. */
int bn254s_prove_fq_exp(bn254s_ctx* ctx, const bn254s_params* params, const uint64_t* scalars /* n x 4 */,
                        const uint64_t* x /* n x 4 */, size_t n, bn254s_proof** out);

/* Proof accessors.  Pointers stay valid until bn254s_proof_free. */
int bn254s_proof_words(const bn254s_proof* p, const uint64_t** data, size_t* len);
/* Per-field accessor: the words of one field of StarkProofWithMetadata (reference common/prover.rs:66-71), i.e. what
 * set_stark_proof_target (generators/g1/stark_proof.rs:173) walks field by field.  Extension elements are two words
 * (c0, c1), digests four, a cap 16 digests; BN254S_SEC_QUERY_ROUNDS is the 84 query rounds back to back in the layout
 * documented at the top of this file (per round: three initial-tree openings, then one opening per FRI layer). */
enum {
  BN254S_SEC_TRACE_CAP = 0,
  BN254S_SEC_AUX_CAP = 1,
  BN254S_SEC_QUOTIENT_CAP = 2,
  BN254S_SEC_LOCAL_VALUES = 3,
  BN254S_SEC_NEXT_VALUES = 4,
  BN254S_SEC_AUX_POLYS = 5,
  BN254S_SEC_AUX_POLYS_NEXT = 6,
  BN254S_SEC_CTL_ZS_FIRST = 7,
  BN254S_SEC_QUOTIENT_POLYS = 8,
  BN254S_SEC_FRI_CAPS = 9,
  BN254S_SEC_QUERY_ROUNDS = 10,
  BN254S_SEC_FINAL_POLY = 11,
  BN254S_SEC_POW_WITNESS = 12,
  BN254S_SEC_INIT_CHALLENGER_STATE = 13,
  BN254S_SEC_COUNT = 14
};
int bn254s_proof_section(const bn254s_proof* p, int id, const uint64_t** data, size_t* len);
int bn254s_proof_degree_bits(const bn254s_proof* p);
/* n x (8 | 16 | 4) words: the outputs s*x+offset (what run_once writes with set_witness at :147-149). */
int bn254s_proof_outputs(const bn254s_proof* p, const uint64_t** data, size_t* len);
/* Per-stage GPU milliseconds of the call that produced the proof (names via bn254s_stage_name). */
int bn254s_proof_stage_ms(const bn254s_proof* p, const float** ms, size_t* n_stages);
const char* bn254s_stage_name(size_t stage);
size_t bn254s_proof_serialize(const bn254s_proof* p, uint8_t* buf, size_t cap); /* LE bytes of the word layout */
void bn254s_proof_free(bn254s_proof* p);

/* Native verification of a proof in the word layout above: the reference's `verify` (src/starks/common/verifier.rs:32-98:
 * challenges, starky's verify_stark_proof_with_challenges, plonky2's verify_fri_proof) plus the cross-table-lookup check
 * against the claimed inputs and outputs (common/ctl_values.rs:28-47 with the rows of scalar_mul_ctl.rs:57-80 /
 * g2 twin / exp_ctl.rs:54-75; timestamps are 0..n-1).  kind 0 = G1, 1 = G2, 2 = Fq exp (offset NULL); outputs = the
 * n x (8 | 16 | 4) words of bn254s_proof_outputs.  Returns BN254S_OK, or BN254S_E_VERIFY with the reason in
 * bn254s_last_error.  With a context the constraint sum at zeta is evaluated by the quotient kernels (csrc/verify.hip); see
 * bn254s_verify_host for the GPU-free form. */
int bn254s_verify(bn254s_ctx* ctx, int kind, const bn254s_params* params, uint32_t degree_bits, const uint64_t* words,
                  size_t n_words, const uint64_t* scalars, const uint64_t* x, const uint64_t* offset, const uint64_t* outputs,
                  size_t n);
/* The same verifier without a context and without a GPU: the constraint sum at zeta comes from an independent host statement
 * of the three AIRs over the quadratic extension (csrc/verify_air_host.h, written from the reference's eval_packed_generic
 * functions, not from the quotient kernels), everything else (transcript, FRI, Merkle paths, CTL sums) is host code in both.
 * A few milliseconds per proof on one core.  On rejection the reference verifier's error text is copied to err_buf
 * (may be NULL). */
int bn254s_verify_host(int kind, const bn254s_params* params, uint32_t degree_bits, const uint64_t* words, size_t n_words,
                       const uint64_t* scalars, const uint64_t* x, const uint64_t* offset, const uint64_t* outputs, size_t n,
                       char* err_buf, size_t err_cap);

/* The extra looking values of the two cross-table lookups (reference g1_generate_ctl_values, scalar_mul_ctl.rs:57-80; G2 twin;
 * fq_generate_ctl_values, exp_ctl.rs:54-75), i.e. what run_once hands to set_ctl_values_target (stark_proof.rs:174-178):
 * per instance one input row  [x limbs | offset limbs (not for Fq) | 16 scalar limbs | timestamp]  of 81 / 145 / 33 words and
 * one output row [output limbs | timestamp] of 33 / 65 / 17 words, all 16-bit little-endian limbs.  Host only, no context. */
int bn254s_ctl_values(int kind, const uint64_t* scalars, const uint64_t* x, const uint64_t* offset, const uint64_t* outputs,
                      size_t n, uint64_t* in_rows, uint64_t* out_rows);

/* map_to_g2 of n Fq2 elements u (8 words: c0, c1) - reference src/utils/hash_to_g2.rs:113-148 and its circuit :150-207, the
 * pipeline of BASELINE config 5: the Shallue-van de Woestijne candidates and the signed square root are computed on the device,
 * the two Legendre symbols per input are proven as Fq exponentiations ((p-1)/2, norm(g(x_i))), the cofactor is cleared by a
 * proven G2 scalar multiplication cofactor * (x, y) + offset, and output - offset is returned.
 * offsets: n non-infinity G2 points (what set_random_g2 supplies), 16 words each.  out_points: n x 16 words.
 * fq_jobs (may be NULL): 2n x 8 words (scalar | x) of the Legendre jobs; g2_jobs (may be NULL): n x 20 words (scalar | point)
 * of the cofactor-clearing jobs - the claimed inputs bn254s_verify needs.  fq_proofs: ceil(2n / 128) proofs, g2_proofs:
 * ceil(n / 128) proofs, each to be released with bn254s_proof_free. */
int bn254s_map_to_g2(bn254s_ctx* ctx, const bn254s_params* params, const uint64_t* u, const uint64_t* offsets, size_t n,
                     uint64_t* out_points, uint64_t* fq_jobs, uint64_t* g2_jobs, bn254s_proof** fq_proofs,
                     bn254s_proof** g2_proofs);

/* hash_to_fq2 (src/utils/hash_to_g2.rs:76-87): Poseidon challenger over `len` Goldilocks elements -> u in Fq2 (8 words), the
 * input of bn254s_map_to_g2; together they are the reference's hash_to_g2.  Host only, no context. */
int bn254s_hash_to_fq2(const uint64_t* input, size_t len, uint64_t* out /* 8 */);
/* n inputs of `len` elements each at once, on the device (inputs[n][len] -> out[n][8]); same values as n calls of the above. */
int bn254s_hash_to_fq2_batch(bn254s_ctx* ctx, const uint64_t* inputs, size_t n, size_t len, uint64_t* out);

/* ---- kernel-level entry points (parity tests and bench.py's roofline leg) ------------------------------ */
/* PolynomialBatch::from_values on host column-major values[C][2^16]: outputs (any may be NULL)
 * coeffs[C][N], lde[C][2N] in Merkle-leaf (bit-reversed) order, cap[16*4]. */
int bn254s_commit_values(bn254s_ctx* ctx, const uint64_t* values, size_t ncols, uint64_t* coeffs, uint64_t* lde,
                         uint64_t* cap);
/* Times `iters` runs of the NTT/LDE stage (iNTT + both coset NTTs) on ncols resident columns of 2^16
 * synthetic values; returns average milliseconds per run through *ms (HIP events on the kernels' stream). */
int bn254s_bench_ntt(bn254s_ctx* ctx, size_t ncols, int iters, float* ms);
/* The same, and the shader clock (MHz) the GPU held while the stage ran: mean over the timed iterations and the slowest ~10 us
 * interval (one extra wave samples the core-clock counter against the constant 100 MHz counter). */
int bn254s_bench_ntt_clock(bn254s_ctx* ctx, size_t ncols, int iters, float* ms, float* mhz, float* mhz_min);
/* Issue cost of the half-rate vector instruction class (64-bit adds / shifts / compares, v_mad_u64_u32, carry instructions) with
 * eight waves per SIMD: nanoseconds per wave-instruction and SIMD, and the shader clock held during the measurement. */
int bn254s_bench_issue(bn254s_ctx* ctx, float* ns_per_issue, float* mhz);
/* PMC calibration: `iters` plain copies of `words` u64 with 8-byte-per-lane loads/stores (known traffic). */
int bn254s_bench_copy(bn254s_ctx* ctx, size_t words, int iters);
/* Times the Merkle leaf-hash kernel alone: ncols columns x 2^log_leaves rows of synthetic data, ms per run. */
int bn254s_bench_leafhash(bn254s_ctx* ctx, size_t ncols, int log_leaves, int iters, float* ms);
/* Poseidon permutation of `n` 12-word states in place (host buffer). */
int bn254s_poseidon_permute(bn254s_ctx* ctx, uint64_t* states, size_t n);
/* debug: the hand-written Goldilocks sequences of the NTT kernels (csrc/gl_asm.h) on n operand pairs; out[n][17] =
 * a+b, a-b, b-a, a*b, a*2^{12,24,32,36,48,60,64,1,31}, a*2^-{12,24,1,31} (all mod p) */
int bn254s_selftest_field(bn254s_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out);
/* Debug: BN254 Fq inversion as trace generation uses it (ark-ff `inverse()` at add.rs:66,80): x[n][4] canonical little-endian
 * words -> out[n][8] = x^-1 mod p twice, by the divstep inversion of the product path and by Fermat's little theorem. */
int bn254s_selftest_fq_inv(bn254s_ctx* ctx, const uint64_t* x, size_t n, uint64_t* out);
/* Trace generation only: column-major trace[W][rows] copied to the host buffer.
 * kind: 0 = G1 scalar mul (W 781), 1 = G2 scalar mul (W 1295), 2 = Fq exp (W 427; offset ignored). */
int bn254s_generate_trace(bn254s_ctx* ctx, int kind, const uint64_t* scalars, const uint64_t* x, const uint64_t* offset,
                          size_t n, uint32_t min_rows_log2, uint64_t* trace_out, uint64_t* outputs);
int bn254s_g1_generate_trace(bn254s_ctx* ctx, const uint64_t* scalars, const uint64_t* x, const uint64_t* offset,
                             size_t n, uint32_t min_rows_log2, uint64_t* trace_out, uint64_t* outputs);

#ifdef __cplusplus
}
#endif
#endif /* BN254_STARK_H */
