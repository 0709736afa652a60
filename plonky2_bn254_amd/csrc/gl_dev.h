// Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension F2 = F[X]/(X^2-7) for gfx950.
// Host+device inline functions: the host side of the library (transcript, parameter tables) uses
// the same code as the kernels.  Values are kept canonical (< p) everywhere.
// Field definition: plonky2_field GoldilocksField / QuadraticExtension<GoldilocksField>, the field the
// reference instantiates at src/starks/curves/g1/scalar_mul_stark.rs:547-549 (F, D = 2).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

typedef uint64_t u64;
typedef uint32_t u32;

#define GL_HD __host__ __device__ __forceinline__

// First statement of the latency-bound kernels (few waves, long dependent chains: doubling chains, scans, batch inversions,
// small Merkle levels, FRI folds, proof-of-work search): their waves win the SIMD's issue arbitration against the waves of
// another proof's GPU-filling kernel that runs beside them, so they finish in about their stand-alone time.
#if defined(__HIP_DEVICE_COMPILE__)
#define LATENCY_KERNEL_PRIO() __builtin_amdgcn_s_setprio(3)
#else
#define LATENCY_KERNEL_PRIO() ((void)0)
#endif

static constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
static constexpr u64 GL_EPS = 0xFFFFFFFFULL;
static constexpr u64 GL_GEN = 0xc65c18b67785d900ULL;       // multiplicative generator = coset shift
static constexpr u64 GL_POW2_GEN = 0x64fdd1a46201e246ULL;  // element of order 2^32

// a + b: the wrapped case (+2^64 == +EPS) and the ">= p" case (-p) are the same 64-bit operation s + EPS
// (EPS + p = 2^64), so one add and one select canonicalise both.
GL_HD u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  u64 t = s + GL_EPS;
  return (s < a || s >= GL_P) ? t : s;
}
GL_HD u64 gl_sub(u64 a, u64 b) {
  u64 d = a - b;
  return (a < b) ? d - GL_EPS : d;  // borrowed 2^64: subtract (2^32-1) more, i.e. add p
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
GL_HD u64 gl_dbl(u64 a) { return gl_add(a, a); }

GL_HD u64 gl_mulhi(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// (hi:lo) mod p, canonical.  2^64 = EPS, 2^96 = -1 (mod p).
GL_HD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  t0 = (lo < hi_hi) ? t0 - GL_EPS : t0;
  u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
  u64 r = t0 + t1;
  u64 r2 = r + GL_EPS;             // wrapped sum (+2^64) and r >= p (-p) are the same correction
  return (r < t1 || r >= GL_P) ? r2 : r;
}
GL_HD u64 gl_mul(u64 a, u64 b) {
  unsigned __int128 x = (unsigned __int128)a * b;  // one 128-bit product: the compiler shares the partial products of lo and hi
  return gl_reduce128((u64)x, (u64)(x >> 64));
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
// a*b + c
GL_HD u64 gl_mad(u64 a, u64 b, u64 c) { return gl_add(gl_mul(a, b), c); }

GL_HD u64 gl_pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, a);
    a = gl_mul(a, a);
    e >>= 1;
  }
  return r;
}
// a^(p-2) with the addition chain 1, 2, 3, 6, 12, 24, 30, 31, 32 on the lengths of runs of ones (p - 2 = (2^31 - 1) 2^33 +
// 2^32 - 1): 64 squarings + 9 multiplications instead of the 64 + 63 of square-and-multiply; 0 -> 0 as before.
GL_HD u64 gl_sqr_n(u64 a, int n) {
  for (int i = 0; i < n; i++) a = gl_mul(a, a);
  return a;
}
GL_HD u64 gl_inv(u64 a) {
  const u64 t2 = gl_mul(gl_sqr_n(a, 1), a);        // 2^2 - 1
  const u64 t3 = gl_mul(gl_sqr_n(t2, 1), a);       // 2^3 - 1
  const u64 t6 = gl_mul(gl_sqr_n(t3, 3), t3);
  const u64 t12 = gl_mul(gl_sqr_n(t6, 6), t6);
  const u64 t24 = gl_mul(gl_sqr_n(t12, 12), t12);
  const u64 t30 = gl_mul(gl_sqr_n(t24, 6), t6);
  const u64 t31 = gl_mul(gl_sqr_n(t30, 1), a);
  const u64 t32 = gl_mul(gl_sqr_n(t31, 1), a);
  return gl_mul(gl_sqr_n(t31, 33), t32);
}
GL_HD u64 gl_root_of_unity(unsigned k) {  // primitive 2^k-th root, k <= 32
  u64 r = GL_POW2_GEN;
  for (unsigned i = k; i < 32; i++) r = gl_mul(r, r);
  return r;
}

// x * 2^S mod p for a compile-time S in [0,192).  2 has order 192 (2^96 = -1), and every root of unity
// of order <= 64 is a power of two (w_64 = 2^3, w_16 = 2^12), so small DFTs need shifts only.
template <int S>
GL_HD u64 gl_mul_2exp(u64 x) {
  static_assert(S >= 0 && S < 192, "shift out of range");
  if constexpr (S >= 96) {
    return gl_neg(gl_mul_2exp<S - 96>(x));
  } else if constexpr (S == 0) {
    return x;
  } else if constexpr (S < 64) {
    return gl_reduce128(x << S, x >> (64 - S));
  } else {
    // x*2^S = y*2^64 with y = x*2^(S-64) = (yh:yl), yh < 2^32.  2^64 = 2^32-1, 2^128 = -2^32.
    constexpr int T = S - 64;
    u64 yl = x << T;
    u64 yh = T ? (x >> (64 - T)) : 0;
    u64 r = gl_reduce128(0, yl);
    return gl_sub(r, yh << 32);
  }
}

// ---- quadratic extension ------------------------------------------------------------------------------
struct gl2 {
  u64 c0, c1;
};
GL_HD gl2 gl2_make(u64 a, u64 b) {
  gl2 r;
  r.c0 = a;
  r.c1 = b;
  return r;
}
GL_HD gl2 gl2_add(gl2 a, gl2 b) { return gl2_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
GL_HD gl2 gl2_sub(gl2 a, gl2 b) { return gl2_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
GL_HD gl2 gl2_mul(gl2 a, gl2 b) {
  u64 t = gl_mul(a.c1, b.c1);
  u64 t7 = gl_add(gl_add(gl_dbl(gl_dbl(t)), gl_dbl(t)), t);  // 7t
  return gl2_make(gl_add(gl_mul(a.c0, b.c0), t7), gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0)));
}
GL_HD gl2 gl2_mul_base(gl2 a, u64 s) { return gl2_make(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
GL_HD gl2 gl2_inv(gl2 a) {
  u64 t = gl_mul(a.c1, a.c1);
  u64 t7 = gl_add(gl_add(gl_dbl(gl_dbl(t)), gl_dbl(t)), t);
  u64 n = gl_sub(gl_mul(a.c0, a.c0), t7);
  u64 ni = gl_inv(n);
  return gl2_make(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
GL_HD gl2 gl2_pow(gl2 a, u64 e) {
  gl2 r = gl2_make(1, 0);
  while (e) {
    if (e & 1) r = gl2_mul(r, a);
    a = gl2_mul(a, a);
    e >>= 1;
  }
  return r;
}
GL_HD bool gl2_eq(gl2 a, gl2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }

GL_HD u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x) >> (32 - bits);
#else
  u32 r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}
