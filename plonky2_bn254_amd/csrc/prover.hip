// Proving entry points (placeholder until the full pipeline lands in this file).
#include "ctx.h"
extern "C" {
int bn254s_prove_g1(bn254s_ctx* c, const bn254s_params*, const uint64_t*, const uint64_t*, const uint64_t*, size_t, bn254s_proof**) { if (c) c->err = "not implemented"; return BN254S_E_UNSUPPORTED; }
int bn254s_prove_g1_batch(bn254s_ctx* c, const bn254s_params*, const uint64_t*, const uint64_t*, const uint64_t*, size_t, size_t, bn254s_proof**) { if (c) c->err = "not implemented"; return BN254S_E_UNSUPPORTED; }
int bn254s_prove_g2(bn254s_ctx* c, const bn254s_params*, const uint64_t*, const uint64_t*, const uint64_t*, size_t, bn254s_proof**) { if (c) c->err = "not implemented"; return BN254S_E_UNSUPPORTED; }
int bn254s_prove_fq_exp(bn254s_ctx* c, const bn254s_params*, const uint64_t*, const uint64_t*, const uint64_t*, size_t, bn254s_proof**) { if (c) c->err = "not implemented"; return BN254S_E_UNSUPPORTED; }
int bn254s_proof_words(const bn254s_proof*, const uint64_t**, size_t*) { return BN254S_E_UNSUPPORTED; }
int bn254s_proof_degree_bits(const bn254s_proof*) { return 0; }
int bn254s_proof_outputs(const bn254s_proof*, const uint64_t**, size_t*) { return BN254S_E_UNSUPPORTED; }
int bn254s_proof_stage_ms(const bn254s_proof*, const float**, size_t*) { return BN254S_E_UNSUPPORTED; }
const char* bn254s_stage_name(size_t) { return ""; }
size_t bn254s_proof_serialize(const bn254s_proof*, uint8_t*, size_t) { return 0; }
void bn254s_proof_free(bn254s_proof*) {}
int bn254s_g1_generate_trace(bn254s_ctx* c, const uint64_t*, const uint64_t*, const uint64_t*, size_t, uint32_t, uint64_t*, uint64_t*) { if (c) c->err = "not implemented"; return BN254S_E_UNSUPPORTED; }
}
