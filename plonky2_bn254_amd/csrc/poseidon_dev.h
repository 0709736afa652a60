// Poseidon-Goldilocks permutation (width 12, 4+22+4 rounds, x^7) and the plonky2 hashing conventions
// hash_no_pad / hash_or_noop / two_to_one (PoseidonHash of PoseidonGoldilocksConfig, the config the
// reference proves with: src/starks/curves/g1/scalar_mul_stark.rs:549, generators/g1/stark_proof.rs:152).
// Round constants: tools/derive_poseidon_constants.py (ChaCha8 seed-0 derivation, KAT-checked).
#pragma once
#include "gl_dev.h"

static const u64 POSEIDON_RC_HOST[360] = {
#include "poseidon_constants.inc"
};
static __constant__ u64 POSEIDON_RC_DEV[360] = {
#include "poseidon_constants.inc"
};
// partial rounds: the constants of lanes 1..11 pushed through the linear layer (tools/derive_poseidon_constants.py)
static __constant__ u64 POSEIDON_FOLD_DEV[30 * 24] = {
#include "poseidon_partial_fold.inc"
};

GL_HD u64 poseidon_rc(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return POSEIDON_RC_DEV[i];
#else
  return POSEIDON_RC_HOST[i];
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// Inside the permutation the GPU keeps any representative below 2^64 (no ">= p" test after a product, a sum or an MDS fold):
// products, the 32-bit halves of the MDS and gl_add_lazy accept such inputs, and the twelve lanes are canonicalised once at
// the end of the permutation.
__device__ __forceinline__ u64 gl_mul_lazy(u64 a, u64 b) {
  unsigned __int128 x = (unsigned __int128)a * b;
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  t0 = (lo < hi_hi) ? t0 - GL_EPS : t0;
  u64 t1 = hi_lo * GL_EPS;
  u64 r = t0 + t1;
  return (r < t1) ? r + GL_EPS : r;
}
__device__ __forceinline__ u64 poseidon_sbox_lazy(u64 x) {
  u64 x2 = gl_mul_lazy(x, x), x4 = gl_mul_lazy(x2, x2), x3 = gl_mul_lazy(x2, x);
  return gl_mul_lazy(x3, x4);
}
// a + b, a any representative, b canonical; result any representative
__device__ __forceinline__ u64 gl_add_lazy(u64 a, u64 b) {
  u64 t = a + b;
  return t < a ? t + GL_EPS : t;
}
#endif

GL_HD u64 poseidon_sbox(u64 x) {
  u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x2, x);
  return gl_mul(x3, x4);
}

// MDS: out[r] = sum_i s[(i+r)%12]*C[i] + (r==0)*8*s[0], C = {17,15,41,16,2,28,13,13,39,18,34,20}.
// Lanes are split in 32-bit halves so that the small-constant products accumulate without carries
// (each half-sum < 2^41); one reduction per output lane:
//   al + ah 2^32 = al + ah_lo 2^32 + ah_hi 2^64 == al + ah_hi (2^32 - 1) + ah_lo 2^32   (mod p),
// the first two terms stay below 2^42, and the last addition is canonicalised like gl_add (wrap and ">= p" are both "+ EPS").
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950: every term is one v_mad_u64_u32 (32 x 32 + 64).  The constants 2, 8 and 16 are handed over in SGPRs the optimiser
// cannot see through, otherwise it turns those products into 64-bit shifts that cost two extra moves each.
// FOLD: the accumulators of output r start from k[2r], k[2r+1] (folded constants of a partial round) instead of 0.
template <bool FOLD>
__device__ __forceinline__ void poseidon_mds(u64 s[12], const u64* __restrict__ k) {
  u32 c2, c8, c16;
  asm("s_mov_b32 %0, 2" : "=s"(c2));
  asm("s_mov_b32 %0, 8" : "=s"(c8));
  asm("s_mov_b32 %0, 16" : "=s"(c16));
  const u32 C[12] = {17, 15, 41, c16, c2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
  u64 out[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  // four output lanes (eight independent accumulator chains) at a time: a single chain of dependent mads would leave the
  // SIMD waiting on the multiplier's latency when only two waves are resident (2^17 leaves)
#pragma unroll
  for (int r0 = 0; r0 < 12; r0 += 4) {
    u64 al[4], ah[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      al[q] = FOLD ? k[2 * (r0 + q)] : 0;
      ah[q] = FOLD ? k[2 * (r0 + q) + 1] : 0;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        al[q] += (u64)lo[(i + r0 + q) % 12] * C[i];
        ah[q] += (u64)hi[(i + r0 + q) % 12] * C[i];
      }
    }
    if (r0 == 0) {
      al[0] += (u64)lo[0] * c8;
      ah[0] += (u64)hi[0] * c8;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      u64 t = al[q] + (u64)(u32)(ah[q] >> 32) * 0xFFFFFFFFull;
      u64 x = ah[q] << 32;
      u64 sum = t + x;
      out[r0 + q] = (sum < x) ? sum + GL_EPS : sum;  // any representative; poseidon_permute canonicalises at the end
    }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = out[i];
}
#else
// host (the Fiat-Shamir transcript: ~700 dependent permutations per proof between kernel launches): one 64 x 8-bit product
// per term into a 128-bit sum, one reduction per output lane (3.1 us per permutation; the 32-bit-halves form with rolled
// loops took 7.2 us, i.e. 5 ms of host time on every proof's critical path)
inline void poseidon_mds(u64 s[12]) {
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 t[24], out[12];
  for (int i = 0; i < 12; i++) t[i] = t[i + 12] = s[i];
#pragma unroll
  for (int r = 0; r < 12; r++) {
    unsigned __int128 acc = r == 0 ? (unsigned __int128)t[0] * 8 : 0;
#pragma unroll
    for (int i = 0; i < 12; i++) acc += (unsigned __int128)t[i + r] * C[i];
    out[r] = gl_reduce128((u64)acc, (u64)(acc >> 64));
  }
  for (int i = 0; i < 12; i++) s[i] = out[i];
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// The compiler's version of the permutation (exact for every input; the fall-back of poseidon_permute below).
__device__ __forceinline__ void poseidon_permute_plain(u64 s[12]) {
#pragma unroll 1
  for (int half = 0; half < 2; half++) {
#pragma unroll 1
    for (int rnd = 26 * half; rnd < 26 * half + 4; rnd++) {  // 4 full rounds
#pragma unroll
      for (int i = 0; i < 12; i++) s[i] = poseidon_sbox_lazy(gl_add_lazy(s[i], POSEIDON_RC_DEV[12 * rnd + i]));
      poseidon_mds<false>(s, nullptr);
    }
    if (half == 0) {
#pragma unroll 1
      for (int rnd = 4; rnd < 26; rnd++) {  // 22 partial rounds
        s[0] = poseidon_sbox_lazy(gl_add_lazy(s[0], POSEIDON_RC_DEV[12 * rnd]));
        poseidon_mds<true>(s, POSEIDON_FOLD_DEV + 24 * rnd);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = s[i] >= GL_P ? s[i] - GL_P : s[i];
}

// Hand-scheduled permutation (tools/gen_poseidon_asm.py -> poseidon_asm.inc): 12.7 k vector instructions instead of the
// compiler's 24.4 k (the 22 partial rounds run as seven blocks of three rounds with ONE linear layer each, plus one single
// round); every intermediate is "some representative below 2^64".  The fast code's short forms do not cover a few digit patterns; running min / max registers detect them and the
// statement then repeats the permutation with its exact code (about one wave-permutation in 100), so the result is exact for
// every input.  The statement owns v26..v126 and the scalar registers listed in POSEIDON_ASM_CLOBBERS; callers should keep
// little else alive across it.
// BN254S_POSEIDON_PLAIN (compile time) selects the compiler's code everywhere (A/B measurements, tools/ubench).
#include "poseidon_asm.inc"
#endif
static __constant__ __attribute__((aligned(64))) u32 POSEIDON_INIT_DEV[31 * 48] = {
#include "poseidon_init.inc"
};
// records of the merged partial-round blocks (three rounds per linear layer, tools/gen_poseidon_asm.py: block_tables)
static __constant__ __attribute__((aligned(64))) u32 POSEIDON_BLK_DEV[7 * 15 * 16] = {
#include "poseidon_blocks.inc"
};

#if !defined(__HIP_DEVICE_COMPILE__)
// Host permutation (the Fiat-Shamir transcript and the host verifier: ~620 dependent permutations per G1 proof on the
// proof's critical path).  The 22 partial rounds run in their sparse form - one S-box, one 12-term dot product and eleven
// multiply-adds per round instead of a dense 12 x 12 layer - followed by one dense 11 x 11 layer; tables and derivation:
// tools/derive_poseidon_host_fast.py.  Same field elements as the textbook rounds (tests/test_oracle_poseidon.py,
// tests/test_gpu_kernels.py compare it with the oracle's and the device's permutation).
#include "poseidon_host_fast.inc"
struct PoseidonDot {  // sum of up to 16 products of 64-bit values, reduced once
  unsigned __int128 lo = 0, hi = 0;
  inline void mad(u64 a, u64 b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    lo += (u64)p;
    hi += (u64)(p >> 64);
  }
  inline u64 reduce() const {  // lo + 2^64 hi with 2^64 = EPS: below 2^101
    const unsigned __int128 t = lo + hi * GL_EPS;
    return gl_reduce128((u64)t, (u64)(t >> 64));
  }
};
// x + c for the S-box input: any representative below 2^64 (gl_mul is exact for those)
inline u64 poseidon_add_rc(u64 x, u64 c) {
  const u64 t = x + c;
  return t < x ? t + GL_EPS : t;
}
inline void poseidon_full_round_host(u64 s[12], const u64* rc) {
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = poseidon_sbox(poseidon_add_rc(s[i], rc[i]));
  poseidon_mds(s);
}
inline void poseidon_permute_host(u64 s[12]) {
  for (int rnd = 0; rnd < 4; rnd++) poseidon_full_round_host(s, POSEIDON_RC_HOST + 12 * rnd);
  for (int t = 0; t < 22; t++) {
    const u64 x0 = poseidon_sbox(poseidon_add_rc(s[0], PHF_K[t]));
    const u64 *w = PHF_W + 11 * t, *u = PHF_U + 11 * t;
    PoseidonDot d;
    d.mad(x0, PHF_M00[t]);
#pragma unroll
    for (int i = 0; i < 11; i++) d.mad(w[i], s[i + 1]);
#pragma unroll
    for (int i = 0; i < 11; i++) {
      const unsigned __int128 p = (unsigned __int128)u[i] * x0 + s[i + 1];
      s[i + 1] = gl_reduce128((u64)p, (u64)(p >> 64));
    }
    s[0] = d.reduce();
  }
  u64 y[11];
  for (int r = 0; r < 11; r++) {
    PoseidonDot d;
#pragma unroll
    for (int i = 0; i < 11; i++) d.mad(PHF_DENSE[11 * r + i], s[i + 1]);
    y[r] = d.reduce();
  }
  for (int r = 0; r < 11; r++) s[r + 1] = y[r];
  poseidon_full_round_host(s, PHF_RC26M);
  for (int rnd = 27; rnd < 30; rnd++) poseidon_full_round_host(s, POSEIDON_RC_HOST + 12 * rnd);
}
#endif

// In place, canonical result.
GL_HD void poseidon_permute(u64 s[12]) {
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(BN254S_POSEIDON_PLAIN)
  poseidon_permute_plain(s);
#else
  u64 x[12];
#pragma unroll
  for (int i = 0; i < 12; i++) x[i] = gl_add_lazy(s[i], POSEIDON_RC_DEV[i]);
  asm(POSEIDON_ASM_PERMUTE
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]), [x6] "+v"(x[6]),
        [x7] "+v"(x[7]), [x8] "+v"(x[8]), [x9] "+v"(x[9]), [x10] "+v"(x[10]), [x11] "+v"(x[11])
      : [tab] "s"(POSEIDON_INIT_DEV), [blk] "s"(POSEIDON_BLK_DEV)
      : POSEIDON_ASM_CLOBBERS);
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = x[i] >= GL_P ? x[i] - GL_P : x[i];
#endif
#else
  poseidon_permute_host(s);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// Cooperative permutation for the latency-bound places (upper Merkle levels, FRI layer trees): 16 adjacent lanes work on ONE
// state, lane l < 12 holding element l (lanes 12..15 idle).  A round is one S-box per lane instead of twelve in sequence and
// one MDS row per lane (the other eleven elements come through ds_bpermute), so a permutation takes about a fifth of the time
// of the one-lane form - at five times the instruction count, which is why leaf hashing of the big commitments stays
// one-lane.  Same field elements as poseidon_permute.  All 64 lanes of the wave must call it.
__device__ __forceinline__ u64 poseidon_permute_coop(u64 s, int l) {
  const int lane = (int)(threadIdx.x & 63);
  const int base = lane & ~15;
  u32 c2, c8, c16;
  asm("s_mov_b32 %0, 2" : "=s"(c2));
  asm("s_mov_b32 %0, 8" : "=s"(c8));
  asm("s_mov_b32 %0, 16" : "=s"(c16));
  const u32 C[12] = {17, 15, 41, c16, c2, 28, 13, 13, 39, 18, 34, 20};
  const int lc = l < 12 ? l : 0;  // idle lanes compute on lane 0's constants; their results are never read
  // the lane's constants of the next round are fetched while the current round runs (they sit on the critical path otherwise)
  u64 rc = POSEIDON_RC_DEV[lc], k_lo = POSEIDON_FOLD_DEV[2 * lc], k_hi = POSEIDON_FOLD_DEV[2 * lc + 1];
#pragma unroll 1
  for (int rnd = 0; rnd < 30; rnd++) {
    const int nr = rnd < 29 ? rnd + 1 : 29;
    const u64 rc_n = POSEIDON_RC_DEV[12 * nr + lc], k_lo_n = POSEIDON_FOLD_DEV[24 * nr + 2 * lc],
              k_hi_n = POSEIDON_FOLD_DEV[24 * nr + 2 * lc + 1];
    const bool full = rnd < 4 || rnd >= 26;
    if (full || l == 0) s = poseidon_sbox_lazy(gl_add_lazy(s, rc));  // partial rounds: lane 0 only, the rest is in k_lo / k_hi
    u64 al = k_lo, ah = k_hi;                                         // zero in the full rounds
    const u32 slo = (u32)s, shi = (u32)(s >> 32);
#pragma unroll
    for (int k = 0; k < 12; k++) {
      int src = lc + k;
      src = base + (src >= 12 ? src - 12 : src);
      const u32 vlo = (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)slo);
      const u32 vhi = (u32)__builtin_amdgcn_ds_bpermute(src << 2, (int)shi);
      al += (u64)vlo * C[k];
      ah += (u64)vhi * C[k];
    }
    if (l == 0) {
      al += (u64)slo * c8;
      ah += (u64)shi * c8;
    }
    const u64 t = al + (u64)(u32)(ah >> 32) * 0xFFFFFFFFull;
    const u64 x = ah << 32;
    const u64 sum = t + x;
    s = (sum < x) ? sum + GL_EPS : sum;
    rc = rc_n;
    k_lo = k_lo_n;
    k_hi = k_hi_n;
  }
  return s >= GL_P ? s - GL_P : s;
}
#else
__device__ u64 poseidon_permute_coop(u64 s, int l);  // host compilation pass: the kernels only need the declaration
#endif

GL_HD void poseidon_two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
  u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
  poseidon_permute(s);
  out[0] = s[0];
  out[1] = s[1];
  out[2] = s[2];
  out[3] = s[3];
}
