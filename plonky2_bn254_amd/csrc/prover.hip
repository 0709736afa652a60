// Proving entry points (placeholder until the full pipeline lands in this file).
#include "ctx.h"
#include "trace_g1.h"
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
}

static size_t rows_for(size_t n, uint32_t min_rows_log2) {
  size_t r = std::max((size_t)1 << min_rows_log2, n * 512), p = 1;
  while (p < r) p <<= 1;
  return p;
}

extern "C" int bn254s_g1_generate_trace(bn254s_ctx* c, const uint64_t* scalars, const uint64_t* x, const uint64_t* off,
                                        size_t n, uint32_t min_rows_log2, uint64_t* trace_out, uint64_t* outputs) {
  if (!c || !scalars || !x || !off || n == 0 || !trace_out) return BN254S_E_INVALID_ARG;
  if (min_rows_log2 < 16) {  // the range-check table needs all 2^16 values (scalar_mul_stark.rs:71-87)
    c->err = "min_rows_log2 must be >= 16";
    return BN254S_E_INVALID_ARG;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  size_t N = rows_for(n, min_rows_log2);
  u64* d_in = c->words("g1.in", n * 20);
  u64* d_trace = c->words("g1.trace", (size_t)G1_W * N);
  u64* d_scr = c->words("g1.scratch", g1_trace_scratch_words(n));
  u64* d_out = c->words("g1.out", n * 8 + 8);
  if (!d_in || !d_trace || !d_scr || !d_out) return BN254S_E_OOM;
  int* d_err = (int*)(d_out + n * 8);
  HIP_TRY(c, hipMemsetAsync(d_err, 0, 4, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_in, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_in + 4 * n, x, n * 64, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_in + 12 * n, off, n * 64, hipMemcpyHostToDevice, c->stream));
  if (g1_generate_trace_device(d_in, d_in + 4 * n, d_in + 12 * n, n, d_trace, N, d_scr, d_out, d_err, c->stream)) {
    c->err = "trace generation launch failed";
    return BN254S_E_HIP;
  }
  int h_err = 0;
  HIP_TRY(c, hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(trace_out, d_trace, (size_t)G1_W * N * 8, hipMemcpyDeviceToHost, c->stream));
  if (outputs) HIP_TRY(c, hipMemcpyAsync(outputs, d_out, n * 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (h_err) {
    c->err = "trace generation reported device error " + std::to_string(h_err);
    return h_err;
  }
  return BN254S_OK;
}
