// BN254 base field Fq (4 x 64-bit Montgomery) for gfx950 device code, plus Jacobian G1 arithmetic.
// Replaces ark-bn254 / ark-ff at the reference call sites src/starks/curves/g1/add.rs:56,66,80
// (affine add, field division) and scalar_mul_stark.rs:105-106.  Only canonical affine coordinates ever
// leave the device, so the result does not depend on the coordinate system used internally.
#pragma once
#include "gl_dev.h"

typedef unsigned __int128 u128;

struct fq {
  u64 l[4];
};

__device__ static constexpr u64 FQ_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
__device__ static constexpr u64 FQ_ONE[4] = {0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL};
__device__ static constexpr u64 FQ_R2[4] = {0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL};
__device__ static constexpr u64 FQ_PM2[4] = {0x3c208c16d87cfd45ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static constexpr u64 FQ_NINV = 0x87d20782e4866389ULL;

__device__ __forceinline__ fq fq_zero() {
  fq r;
  r.l[0] = r.l[1] = r.l[2] = r.l[3] = 0;
  return r;
}
__device__ __forceinline__ fq fq_one() {
  fq r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.l[i] = FQ_ONE[i];
  return r;
}
__device__ __forceinline__ bool fq_is_zero(const fq& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
__device__ __forceinline__ bool fq_eq(const fq& a, const fq& b) {
  return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3];
}
__device__ __forceinline__ bool fq_geq_p(const u64 t[4]) {
#pragma unroll
  for (int i = 3; i >= 0; i--) {
    if (t[i] > FQ_P[i]) return true;
    if (t[i] < FQ_P[i]) return false;
  }
  return true;
}
__device__ __forceinline__ void fq_sub_p(u64 t[4]) {
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)t[i] - FQ_P[i] - borrow;
    t[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
}
__device__ __forceinline__ fq fq_add(const fq& a, const fq& b) {
  fq r;
  u128 c = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    r.l[i] = (u64)c;
    c >>= 64;
  }
  if ((u64)c || fq_geq_p(r.l)) fq_sub_p(r.l);  // p < 2^254: no carry out in practice
  return r;
}
__device__ __forceinline__ fq fq_sub(const fq& a, const fq& b) {
  fq r;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a.l[i] - b.l[i] - borrow;
    r.l[i] = (u64)d;
    borrow = (u64)(d >> 64) & 1;
  }
  if (borrow) {
    u128 c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      c += (u128)r.l[i] + FQ_P[i];
      r.l[i] = (u64)c;
      c >>= 64;
    }
  }
  return r;
}
__device__ __forceinline__ fq fq_dbl(const fq& a) { return fq_add(a, a); }
__device__ __forceinline__ fq fq_neg(const fq& a) { return fq_is_zero(a) ? a : fq_sub(fq_zero(), a); }

// CIOS Montgomery product a*b*R^-1 mod p.
__device__ __forceinline__ fq fq_mul(const fq& a, const fq& b) {
  u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u128 c;
    u64 bi = b.l[i];
    c = (u128)a.l[0] * bi + t0; t0 = (u64)c; c >>= 64;
    c += (u128)a.l[1] * bi + t1; t1 = (u64)c; c >>= 64;
    c += (u128)a.l[2] * bi + t2; t2 = (u64)c; c >>= 64;
    c += (u128)a.l[3] * bi + t3; t3 = (u64)c; c >>= 64;
    c += t4;
    t4 = (u64)c;
    u64 t5 = (u64)(c >> 64);
    u64 m = t0 * FQ_NINV;
    c = (u128)m * FQ_P[0] + t0; c >>= 64;
    c += (u128)m * FQ_P[1] + t1; t0 = (u64)c; c >>= 64;
    c += (u128)m * FQ_P[2] + t2; t1 = (u64)c; c >>= 64;
    c += (u128)m * FQ_P[3] + t3; t2 = (u64)c; c >>= 64;
    c += t4;
    t3 = (u64)c;
    t4 = t5 + (u64)(c >> 64);
  }
  fq r;
  r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3;
  if (t4 || fq_geq_p(r.l)) fq_sub_p(r.l);
  return r;
}
__device__ __forceinline__ fq fq_sqr(const fq& a) { return fq_mul(a, a); }

__device__ __forceinline__ fq fq_from_canonical(const u64* w) {  // w < p
  fq t, r2;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    t.l[i] = w[i];
    r2.l[i] = FQ_R2[i];
  }
  return fq_mul(t, r2);
}
__device__ __forceinline__ fq fq_to_canonical(const fq& a) {
  fq one;
  one.l[0] = 1;
  one.l[1] = one.l[2] = one.l[3] = 0;
  return fq_mul(a, one);
}
__device__ __forceinline__ fq fq_from_u32(u32 v) {
  u64 w[4] = {v, 0, 0, 0};
  return fq_from_canonical(w);
}

// a^(p-2); a != 0.  Not inlined: used once per batch-inversion thread.
__device__ __noinline__ fq fq_inv(const fq& a) {
  fq r = fq_one();
  for (int i = 253; i >= 0; i--) {  // p-2 < 2^254
    r = fq_sqr(r);
    if ((FQ_PM2[i >> 6] >> (i & 63)) & 1) r = fq_mul(r, a);
  }
  return r;
}

// ---- G1 (y^2 = x^3 + 3), Jacobian coordinates -----------------------------------------------------------
struct g1j {
  fq x, y, z;
};
__device__ __forceinline__ g1j g1_double(const g1j& p) {
  fq a = fq_sqr(p.x), b = fq_sqr(p.y), c = fq_sqr(b);
  fq xb = fq_add(p.x, b);
  fq d = fq_dbl(fq_sub(fq_sub(fq_sqr(xb), a), c));
  fq e = fq_add(fq_dbl(a), a);
  fq f = fq_sqr(e);
  g1j r;
  r.x = fq_sub(f, fq_dbl(d));
  fq c8 = fq_dbl(fq_dbl(fq_dbl(c)));
  r.y = fq_sub(fq_mul(e, fq_sub(d, r.x)), c8);
  r.z = fq_dbl(fq_mul(p.y, p.z));
  return r;
}
// Returns 0 ok, 1 if the points were equal (result = doubling), 2 if p == -q (point at infinity: the
// reference cannot generate a proof either, add.rs:49-51).
__device__ __forceinline__ int g1_add(const g1j& p, const g1j& q, g1j& r) {
  fq z1z1 = fq_sqr(p.z), z2z2 = fq_sqr(q.z);
  fq u1 = fq_mul(p.x, z2z2), u2 = fq_mul(q.x, z1z1);
  fq s1 = fq_mul(fq_mul(p.y, q.z), z2z2), s2 = fq_mul(fq_mul(q.y, p.z), z1z1);
  fq h = fq_sub(u2, u1), rr = fq_sub(s2, s1);
  if (fq_is_zero(h)) {
    if (fq_is_zero(rr)) {
      r = g1_double(p);
      return 1;
    }
    r = p;
    return 2;
  }
  fq hh = fq_sqr(h), hhh = fq_mul(h, hh), v = fq_mul(u1, hh);
  r.x = fq_sub(fq_sub(fq_sqr(rr), hhh), fq_dbl(v));
  r.y = fq_sub(fq_mul(rr, fq_sub(v, r.x)), fq_mul(s1, hhh));
  r.z = fq_mul(fq_mul(p.z, q.z), h);
  return 0;
}

// ---- Fq2 = Fq[u]/(u^2+1) and G2 (y^2 = x^3 + b2), Jacobian ------------------------------------------------
// Replaces ark-bn254 Fq2 / G2Affine at reference src/starks/curves/g2/add.rs:59-130.
struct fq2 {
  fq c0, c1;
};
__device__ __forceinline__ fq2 fq2_zero() {
  fq2 r;
  r.c0 = fq_zero();
  r.c1 = fq_zero();
  return r;
}
__device__ __forceinline__ fq2 fq2_one() {
  fq2 r;
  r.c0 = fq_one();
  r.c1 = fq_zero();
  return r;
}
__device__ __forceinline__ bool fq2_is_zero(const fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
__device__ __forceinline__ bool fq2_eq(const fq2& a, const fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
__device__ __forceinline__ fq2 fq2_add(const fq2& a, const fq2& b) {
  fq2 r;
  r.c0 = fq_add(a.c0, b.c0);
  r.c1 = fq_add(a.c1, b.c1);
  return r;
}
__device__ __forceinline__ fq2 fq2_sub(const fq2& a, const fq2& b) {
  fq2 r;
  r.c0 = fq_sub(a.c0, b.c0);
  r.c1 = fq_sub(a.c1, b.c1);
  return r;
}
__device__ __forceinline__ fq2 fq2_dbl(const fq2& a) { return fq2_add(a, a); }
__device__ __noinline__ fq2 fq2_mul(const fq2& a, const fq2& b) {  // Karatsuba: 3 Fq products
  fq t0 = fq_mul(a.c0, b.c0), t1 = fq_mul(a.c1, b.c1);
  fq t2 = fq_mul(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
  fq2 r;
  r.c0 = fq_sub(t0, t1);
  r.c1 = fq_sub(fq_sub(t2, t0), t1);
  return r;
}
__device__ __forceinline__ fq2 fq2_sqr(const fq2& a) { return fq2_mul(a, a); }
__device__ __forceinline__ fq fq2_norm(const fq2& a) { return fq_add(fq_sqr(a.c0), fq_sqr(a.c1)); }
// a^-1 given n^-1 with n = norm(a)
__device__ __forceinline__ fq2 fq2_inv_from_norm_inv(const fq2& a, const fq& ninv) {
  fq2 r;
  r.c0 = fq_mul(a.c0, ninv);
  r.c1 = fq_neg(fq_mul(a.c1, ninv));
  return r;
}

struct g2j {
  fq2 x, y, z;
};
__device__ __forceinline__ g2j g2_double(const g2j& p) {
  fq2 a = fq2_sqr(p.x), b = fq2_sqr(p.y), c = fq2_sqr(b);
  fq2 xb = fq2_add(p.x, b);
  fq2 d = fq2_dbl(fq2_sub(fq2_sub(fq2_sqr(xb), a), c));
  fq2 e = fq2_add(fq2_dbl(a), a);
  fq2 f = fq2_sqr(e);
  g2j r;
  r.x = fq2_sub(f, fq2_dbl(d));
  fq2 c8 = fq2_dbl(fq2_dbl(fq2_dbl(c)));
  r.y = fq2_sub(fq2_mul(e, fq2_sub(d, r.x)), c8);
  r.z = fq2_dbl(fq2_mul(p.y, p.z));
  return r;
}
// 0 ok, 1 equal points (doubled), 2 opposite points (infinity)
__device__ __forceinline__ int g2_add(const g2j& p, const g2j& q, g2j& r) {
  fq2 z1z1 = fq2_sqr(p.z), z2z2 = fq2_sqr(q.z);
  fq2 u1 = fq2_mul(p.x, z2z2), u2 = fq2_mul(q.x, z1z1);
  fq2 s1 = fq2_mul(fq2_mul(p.y, q.z), z2z2), s2 = fq2_mul(fq2_mul(q.y, p.z), z1z1);
  fq2 h = fq2_sub(u2, u1), rr = fq2_sub(s2, s1);
  if (fq2_is_zero(h)) {
    if (fq2_is_zero(rr)) {
      r = g2_double(p);
      return 1;
    }
    r = p;
    return 2;
  }
  fq2 hh = fq2_sqr(h), hhh = fq2_mul(h, hh), v = fq2_mul(u1, hh);
  r.x = fq2_sub(fq2_sub(fq2_sqr(rr), hhh), fq2_dbl(v));
  r.y = fq2_sub(fq2_mul(rr, fq2_sub(v, r.x)), fq2_mul(s1, hhh));
  r.z = fq2_mul(fq2_mul(p.z, q.z), h);
  return 0;
}
