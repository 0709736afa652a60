// map_to_g2 on the device: the front-end of BASELINE config 5.
//
// Replaces the witness arithmetic of the reference's `map_to_g2` / `map_to_g2_circuit` (src/utils/hash_to_g2.rs:113-148,
// 150-207: Shallue-van de Woestijne map with Z = 1, RFC 9380 6.6.1) around its two STARK job kinds:
//   k_m2g_candidates: x1, x2, x3 and the 2n Legendre jobs  norm(g(x_i))^((p-1)/2)   (is_square, fields/fq2.rs:235-240)
//   [2n fq_exp proofs]
//   k_m2g_select:     x = first candidate with a square g(x), y = sqrt(g(x)) with sgn(y) = sgn(u)  (hash_to_g2.rs:128-145)
//   [n G2 proofs of cofactor * (x, y) + offset]
//   k_m2g_finish:     output - offset                                                (hash_to_g2.rs:200-205)
// Square roots follow ark-ff 0.4 (Fq: a^((p+1)/4); Fq2: the "complex method" of QuadExtField::sqrt), like the Python
// front-end tools/map_to_g2_ref.py, which is the parity reference of tests/test_map_to_g2.py.
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"
#include "fq_dev.h"
#include "transcript.h"
#include "../../include/bn254_stark.h"
#include "map_to_g2_constants.inc"

namespace {

struct M2GConsts {
  u64 b2[8], gz[8], nz2[8], tv4[8], tv6[8], leg[4], sq[4], two_inv[4], cof[4];
};
M2GConsts host_consts() {
  M2GConsts c;
  memcpy(c.b2, M2G_B2, 64);
  memcpy(c.gz, M2G_GZ, 64);
  memcpy(c.nz2, M2G_NEG_Z_BY_2, 64);
  memcpy(c.tv4, M2G_TV4, 64);
  memcpy(c.tv6, M2G_TV6, 64);
  memcpy(c.leg, M2G_LEGENDRE_EXP, 32);
  memcpy(c.sq, M2G_SQRT_EXP, 32);
  memcpy(c.two_inv, M2G_TWO_INV, 32);
  memcpy(c.cof, M2G_COFACTOR, 32);
  return c;
}

__device__ __forceinline__ fq2 f2_from(const u64* w) {
  fq2 r;
  r.c0 = fq_from_canonical(w);
  r.c1 = fq_from_canonical(w + 4);
  return r;
}
__device__ __forceinline__ void f2_store_canonical(u64* w, const fq2& a) {
  const fqw c0 = fq_to_canonical(a.c0), c1 = fq_to_canonical(a.c1);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    w[i] = c0.l[i];
    w[4 + i] = c1.l[i];
  }
}
__device__ __forceinline__ fq2 f2_neg(const fq2& a) {
  fq2 r;
  r.c0 = fq_neg(a.c0);
  r.c1 = fq_neg(a.c1);
  return r;
}
__device__ __noinline__ fq fq_pow_words(const fq& a, const u64* e) {  // a^e, e < 2^256
  fq r = fq_one();
  for (int i = 255; i >= 0; i--) {
    r = fq_sqr(r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = fq_mul(r, a);
  }
  return r;
}
__device__ __forceinline__ fq2 f2_inv(const fq2& a) { return fq2_inv_from_norm_inv(a, fq_inv(fq2_norm(a))); }
__device__ __forceinline__ fq2 g_rhs(const fq2& x, const fq2& b2) { return fq2_add(fq2_mul(fq2_sqr(x), x), b2); }
__device__ __forceinline__ bool fq_is_square(const fq& a, const M2GConsts& C) {
  return fq_is_zero(a) || fq_eq(fq_pow_words(a, C.leg), fq_one());
}
// src/fields/sgn.rs:20-27
__device__ __forceinline__ bool f2_sgn(const fq2& a) {
  const fqw c0 = fq_to_canonical(a.c0), c1 = fq_to_canonical(a.c1);
  const bool zero0 = (c0.l[0] | c0.l[1] | c0.l[2] | c0.l[3]) == 0;
  return (c0.l[0] & 1) || (zero0 && (c1.l[0] & 1));
}
// QuadExtField::sqrt of ark-ff 0.4 for a square a; *ok = false if the candidate does not square back
__device__ __noinline__ fq2 f2_sqrt(const fq2& a, const M2GConsts& C, bool* ok) {
  fq2 r;
  if (fq_is_zero(a.c1)) {
    if (fq_is_square(a.c0, C)) {
      r.c0 = fq_pow_words(a.c0, C.sq);
      r.c1 = fq_zero();
    } else {
      r.c0 = fq_zero();
      r.c1 = fq_pow_words(fq_neg(a.c0), C.sq);
    }
  } else {
    const fq two_inv = fq_from_canonical(C.two_inv);
    const fq alpha = fq_pow_words(fq2_norm(a), C.sq);
    fq delta = fq_mul(fq_add(alpha, a.c0), two_inv);
    if (!fq_is_square(delta, C)) delta = fq_sub(delta, alpha);
    r.c0 = fq_pow_words(delta, C.sq);
    r.c1 = fq_mul(fq_mul(a.c1, two_inv), fq_inv(r.c0));
  }
  *ok = fq2_eq(fq2_sqr(r), a);
  return r;
}

// cand: 3 x 8 words per input (x1, x2, x3 canonical); fq_s / fq_x: the 2n fq_exp jobs
__global__ __launch_bounds__(64) void k_m2g_candidates(const u64* __restrict__ u, size_t n, M2GConsts C, u64* __restrict__ cand,
                                                       u64* __restrict__ fq_s, u64* __restrict__ fq_x) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const fq2 uu = f2_from(u + 8 * k), gz = f2_from(C.gz), b2 = f2_from(C.b2);
  fq2 tv1 = fq2_mul(fq2_sqr(uu), gz);
  const fq2 tv2 = fq2_add(fq2_one(), tv1);
  tv1 = fq2_sub(fq2_one(), tv1);
  const fq2 tv3 = f2_inv(fq2_mul(tv1, tv2));
  const fq2 tv5 = fq2_mul(fq2_mul(fq2_mul(uu, tv1), tv3), f2_from(C.tv4));
  const fq2 nz2 = f2_from(C.nz2);
  fq2 x[3];
  x[0] = fq2_sub(nz2, tv5);
  x[1] = fq2_add(nz2, tv5);
  const fq2 t = fq2_mul(fq2_sqr(tv2), tv3);
  x[2] = fq2_add(fq2_one(), fq2_mul(f2_from(C.tv6), fq2_sqr(t)));
#pragma unroll 1
  for (int i = 0; i < 3; i++) f2_store_canonical(cand + (3 * k + i) * 8, x[i]);
#pragma unroll 1
  for (int i = 0; i < 2; i++) {
    const fqw nrm = fq_to_canonical(fq2_norm(g_rhs(x[i], b2)));
    for (int w = 0; w < 4; w++) {
      fq_x[(2 * k + i) * 4 + w] = nrm.l[w];
      fq_s[(2 * k + i) * 4 + w] = C.leg[w];
    }
  }
}

// legendre: the 2n fq_exp outputs (canonical words); g2_s / g2_x: the n cofactor-clearing jobs
__global__ __launch_bounds__(64) void k_m2g_select(const u64* __restrict__ u, const u64* __restrict__ cand,
                                                   const u64* __restrict__ legendre, size_t n, M2GConsts C, u64* __restrict__ g2_s,
                                                   u64* __restrict__ g2_x, int* __restrict__ err) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  auto is_one = [&](const u64* w) { return w[0] == 1 && (w[1] | w[2] | w[3]) == 0; };
  const int pick = is_one(legendre + (2 * k) * 4) ? 0 : is_one(legendre + (2 * k + 1) * 4) ? 1 : 2;
  const fq2 x = f2_from(cand + (3 * k + pick) * 8);
  bool ok;
  fq2 y = f2_sqrt(g_rhs(x, f2_from(C.b2)), C, &ok);
  if (!ok) atomicCAS(err, 0, BN254S_E_INTERNAL);  // g(x) is not a square: inconsistent Legendre results
  if (f2_sgn(f2_from(u + 8 * k)) != f2_sgn(y)) y = f2_neg(y);
  for (int w = 0; w < 8; w++) g2_x[16 * k + w] = cand[(3 * k + pick) * 8 + w];
  f2_store_canonical(g2_x + 16 * k + 8, y);
  for (int w = 0; w < 4; w++) g2_s[4 * k + w] = C.cof[w];
}

// out = o - off (affine; o != +-off for a random offset: reported otherwise)
__global__ __launch_bounds__(64) void k_m2g_finish(const u64* __restrict__ o, const u64* __restrict__ off, size_t n, u64* __restrict__ out,
                                                   int* __restrict__ err) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const fq2 x1 = f2_from(o + 16 * k), y1 = f2_from(o + 16 * k + 8);
  const fq2 x2 = f2_from(off + 16 * k), y2 = f2_neg(f2_from(off + 16 * k + 8));
  const fq2 dx = fq2_sub(x2, x1);
  if (fq2_is_zero(dx)) {
    atomicCAS(err, 0, BN254S_E_INVALID_POINT);
    return;
  }
  const fq2 lam = fq2_mul(fq2_sub(y2, y1), f2_inv(dx));
  const fq2 x3 = fq2_sub(fq2_sub(fq2_sqr(lam), x1), x2);
  const fq2 y3 = fq2_sub(fq2_mul(lam, fq2_sub(x1, x3)), y1);
  f2_store_canonical(out + 16 * k, x3);
  f2_store_canonical(out + 16 * k + 8, y3);
}

// hash_to_fq2 for n inputs of `len` Goldilocks elements each, one lane per input (hash_to_g2.rs:76-87): the challenger absorbs the
// input in chunks of eight (overwrite mode; a last partial chunk, or an empty input, is permuted when the first challenge is
// asked for), then 2 x 16 challenges are popped - eight per permutation, lane 7 first - and their low 32 bits are the
// little-endian limbs of two 512-bit integers, each reduced modulo p: lo + hi 2^256 with both halves taken into Montgomery form.
__device__ static constexpr u32 M2G_TWO256_MONT[FQ_NL] = {0x3b5ae02, 0x3ab61e1, 0xaa4ba8, 0x3e9cdb2, 0x2f6ebb, 0x4553ac, 0x116483a, 0x468428, 0x236e920, 0x33056};  // 2^256 R mod p
__global__ __launch_bounds__(64) void k_hash_to_fq2(const u64* __restrict__ in, size_t n, size_t len, u64* __restrict__ out) {
#if defined(__HIP_DEVICE_COMPILE__)
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u64* x = in + k * len;
  u64 st[12];
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = 0;
  size_t pos = 0;
  bool have_out = false;
  for (; len - pos >= 8; pos += 8) {
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = x[pos + i];
    poseidon_permute(st);
    have_out = true;
  }
  const int rem = (int)(len - pos);
  if (rem > 0 || !have_out) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (i < rem) st[i] = x[pos + i];
    poseidon_permute(st);
  }
  fq r2, c256;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) {
    r2.l[i] = FQ_R2[i];
    c256.l[i] = M2G_TWO256_MONT[i];
  }
#pragma unroll 1
  for (int coord = 0; coord < 2; coord++) {
    u64 w[8];  // the 512-bit integer as eight 64-bit words: limb 2j | limb 2j+1 << 32
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (coord + half > 0) poseidon_permute(st);  // output buffer used up (8 challenges per permutation)
#pragma unroll
      for (int j = 0; j < 4; j++) w[4 * half + j] = (u64)(u32)st[7 - 2 * j] | ((u64)(u32)st[6 - 2 * j] << 32);
    }
    const fq lo = fq_mul(fq_unpack(w), r2), hi = fq_mul(fq_unpack(w + 4), r2);  // (values below 2^256 < 6 p: loose operands)
    const fqw c = fq_to_canonical(fq_add(lo, fq_mul(hi, c256)));
#pragma unroll
    for (int i = 0; i < 4; i++) out[8 * k + 4 * coord + i] = c.l[i];
  }
#endif
}

#define MCHK(call)                                                \
  do {                                                            \
    hipError_t e_ = (call);                                       \
    if (e_ != hipSuccess) {                                       \
      c->set_err(std::string(#call) + ": " + hipGetErrorString(e_)); \
      return BN254S_E_HIP;                                        \
    }                                                             \
  } while (0)

}  // namespace

// hash_to_fq2 (src/utils/hash_to_g2.rs:76-87): Challenger::observe_elements(input); each coordinate = the low 32 bits of 16
// challenges as little-endian limbs of a 512-bit integer, reduced modulo p (f_slice_to_biguint, :226-240, then `.into()`).
extern "C" int bn254s_hash_to_fq2(const uint64_t* input, size_t len, uint64_t* out) {
  if ((!input && len) || !out) return BN254S_E_INVALID_ARG;
  static const u64 PW[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  Challenger ch;
  ch.observe_n(input, len);
  for (int coord = 0; coord < 2; coord++) {
    u32 limbs[16];
    for (int i = 0; i < 16; i++) limbs[i] = (u32)ch.challenge();
    u64 r[4] = {0, 0, 0, 0};  // r = value mod p, one bit at a time from the top (r < p < 2^254: the doubling cannot overflow)
    for (int bit = 511; bit >= 0; bit--) {
      u64 carry = (limbs[bit >> 5] >> (bit & 31)) & 1;
      for (int w = 0; w < 4; w++) {
        u64 nc = r[w] >> 63;
        r[w] = (r[w] << 1) | carry;
        carry = nc;
      }
      bool ge = true;
      for (int w = 3; w >= 0; w--) {
        if (r[w] != PW[w]) {
          ge = r[w] > PW[w];
          break;
        }
      }
      if (ge) {
        u64 borrow = 0;
        for (int w = 0; w < 4; w++) {
          unsigned __int128 d = (unsigned __int128)r[w] - PW[w] - borrow;
          r[w] = (u64)d;
          borrow = (u64)(d >> 64) & 1;
        }
      }
    }
    memcpy(out + 4 * coord, r, 32);
  }
  return BN254S_OK;
}

// The same for n inputs at once on the device (inputs[n][len] -> out[n][8]): the step before bn254s_map_to_g2 when a circuit hashes
// many messages to G2.
extern "C" int bn254s_hash_to_fq2_batch(bn254s_ctx* c, const uint64_t* inputs, size_t n, size_t len, uint64_t* out) {
  if (!c || !out || (!inputs && len) || n == 0) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  u64* d_in = c->words("h2f.in", n * len + 1);
  u64* d_out = c->words("h2f.out", n * 8);
  if (!d_in || !d_out) return BN254S_E_OOM;
  if (len) HIP_TRY(c, hipMemcpyAsync(d_in, inputs, n * len * 8, hipMemcpyHostToDevice, c->stream));
  k_hash_to_fq2<<<(unsigned)((n + 63) / 64), 64, 0, c->stream>>>(d_in, n, len, d_out);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, d_out, n * 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BN254S_OK;
}

extern "C" int bn254s_map_to_g2(bn254s_ctx* c, const bn254s_params* params, const uint64_t* u, const uint64_t* offsets, size_t n,
                                uint64_t* out_points, uint64_t* fq_jobs, uint64_t* g2_jobs, bn254s_proof** fq_proofs,
                                bn254s_proof** g2_proofs) {
  if (!c || !params || !u || !offsets || !out_points || !fq_proofs || !g2_proofs || n == 0) return BN254S_E_INVALID_ARG;
  MCHK(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  const M2GConsts C = host_consts();
  const unsigned grid = (unsigned)((n + 63) / 64);
  u64* d = c->words("m2g", n * (8 + 24 + 8 + 8 + 8 + 4 + 16 + 16 + 16 + 16) + 16);
  if (!d) return BN254S_E_OOM;
  u64* d_u = d;
  u64* d_cand = d_u + 8 * n;
  u64* d_fq_s = d_cand + 24 * n;
  u64* d_fq_x = d_fq_s + 8 * n;
  u64* d_leg = d_fq_x + 8 * n;
  u64* d_g2_s = d_leg + 8 * n;
  u64* d_g2_x = d_g2_s + 4 * n;
  u64* d_off = d_g2_x + 16 * n;
  u64* d_o = d_off + 16 * n;
  u64* d_out = d_o + 16 * n;
  int* d_err = (int*)(d_out + 16 * n);
  MCHK(hipMemsetAsync(d_err, 0, 8, st));
  MCHK(hipMemcpyAsync(d_u, u, n * 64, hipMemcpyHostToDevice, st));
  k_m2g_candidates<<<grid, 64, 0, st>>>(d_u, n, C, d_cand, d_fq_s, d_fq_x);
  std::vector<u64> fs(8 * n), fx(8 * n);
  MCHK(hipMemcpyAsync(fs.data(), d_fq_s, fs.size() * 8, hipMemcpyDeviceToHost, st));
  MCHK(hipMemcpyAsync(fx.data(), d_fq_x, fx.size() * 8, hipMemcpyDeviceToHost, st));
  MCHK(hipStreamSynchronize(st));
  // 2n Legendre exponentiations in 128-instance proofs
  int rc = bn254s_prove_batch(c, 2, params, fs.data(), fx.data(), nullptr, 2 * n, 128, fq_proofs);
  if (rc != BN254S_OK) return rc;
  const size_t n_fq = (2 * n + 127) / 128, n_g2 = (n + 127) / 128;
  std::vector<u64> leg(8 * n);
  for (size_t i = 0, pos = 0; i < n_fq; i++) {
    const uint64_t* o;
    size_t len;
    bn254s_proof_outputs(fq_proofs[i], &o, &len);
    memcpy(leg.data() + pos, o, len * 8);
    pos += len;
  }
  MCHK(hipMemcpyAsync(d_leg, leg.data(), leg.size() * 8, hipMemcpyHostToDevice, st));
  k_m2g_select<<<grid, 64, 0, st>>>(d_u, d_cand, d_leg, n, C, d_g2_s, d_g2_x, d_err);
  std::vector<u64> gs(4 * n), gx(16 * n);
  int h_err = 0;
  MCHK(hipMemcpyAsync(gs.data(), d_g2_s, gs.size() * 8, hipMemcpyDeviceToHost, st));
  MCHK(hipMemcpyAsync(gx.data(), d_g2_x, gx.size() * 8, hipMemcpyDeviceToHost, st));
  MCHK(hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
  MCHK(hipStreamSynchronize(st));
  if (h_err) {
    c->set_err("map_to_g2: square root self-check failed");
    for (size_t i = 0; i < n_fq; i++) bn254s_proof_free(fq_proofs[i]);
    return h_err;
  }
  // n cofactor-clearing scalar multiplications
  rc = bn254s_prove_batch(c, 1, params, gs.data(), gx.data(), offsets, n, 128, g2_proofs);
  if (rc != BN254S_OK) {
    for (size_t i = 0; i < n_fq; i++) bn254s_proof_free(fq_proofs[i]);
    return rc;
  }
  std::vector<u64> outs(16 * n);
  for (size_t i = 0, pos = 0; i < n_g2; i++) {
    const uint64_t* o;
    size_t len;
    bn254s_proof_outputs(g2_proofs[i], &o, &len);
    memcpy(outs.data() + pos, o, len * 8);
    pos += len;
  }
  MCHK(hipMemcpyAsync(d_o, outs.data(), outs.size() * 8, hipMemcpyHostToDevice, st));
  MCHK(hipMemcpyAsync(d_off, offsets, n * 128, hipMemcpyHostToDevice, st));
  k_m2g_finish<<<grid, 64, 0, st>>>(d_o, d_off, n, d_out, d_err);
  MCHK(hipMemcpyAsync(out_points, d_out, n * 128, hipMemcpyDeviceToHost, st));
  MCHK(hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
  MCHK(hipStreamSynchronize(st));
  if (fq_jobs) {  // the job arrays (what bn254s_verify needs as claimed inputs): 2n x (scalar 4 | x 4), n x (scalar 4 | x 16)
    for (size_t k = 0; k < 2 * n; k++) {
      memcpy(fq_jobs + 8 * k, fs.data() + 4 * k, 32);
      memcpy(fq_jobs + 8 * k + 4, fx.data() + 4 * k, 32);
    }
  }
  if (g2_jobs) {
    for (size_t k = 0; k < n; k++) {
      memcpy(g2_jobs + 20 * k, gs.data() + 4 * k, 32);
      memcpy(g2_jobs + 20 * k + 4, gx.data() + 16 * k, 128);
    }
  }
  if (h_err) {
    c->set_err("map_to_g2: output equals +-offset");
    return h_err;
  }
  return BN254S_OK;
}
