// BN254 base field Fq for gfx950 device code, plus Jacobian G1 / G2 arithmetic.
// Replaces ark-bn254 / ark-ff at the reference call sites src/starks/curves/g1/add.rs:56,66,80
// (affine add, field division) and scalar_mul_stark.rs:105-106.  Only canonical affine coordinates ever
// leave the device, so the result does not depend on the representation used internally.
//
// Representation: ten 26-bit limbs in 32-bit registers, Montgomery form with R = 2^260, always canonical (< p, limbs < 2^26).
// gfx950 needs two wait states between a VALU instruction that writes VCC and an add-with-carry that reads it, so the usual
// 64-bit-limb carry chains (v_addc_co) spend more issue slots on s_nop than on arithmetic.  With 26-bit limbs a product is
// 100 + 100 v_mad_u64_u32 into ten 64-bit column sums that cannot overflow (< 2^56), carries are shifts and masks, and
// comparisons are sign bits: no VCC anywhere.  A multiplication is ~330 instructions instead of ~1000.
#pragma once
#include "gl_dev.h"

typedef unsigned __int128 u128;

static constexpr int FQ_NL = 10, FQ_LB = 26;
static constexpr u32 FQ_MASK = (1u << FQ_LB) - 1;
struct fq {
  u32 l[FQ_NL];
};
struct fqw {  // 256-bit value as four 64-bit words (storage / wire format)
  u64 l[4];
};

// p, R mod p, R^2 mod p (R = 2^260) in 26-bit limbs; -p^-1 mod 2^26; p - 2 in 64-bit words (exponent of the inversion)
__device__ static constexpr u32 FQ_P[FQ_NL] = {0x7cfd47, 0x2305b6, 0xa8d3c2, 0x245a1c7, 0x197816a, 0x605617, 0x1045b68, 0x280a6e1, 0x272e131, 0xc1913};
__device__ static constexpr u32 FQ_ONE[FQ_NL] = {0x2fce4b4, 0x82203d, 0x9a8455, 0x126eaa6, 0x2498908, 0x63c052, 0x29201d8, 0x1c93e16, 0x24e1bb7, 0x7c590};
__device__ static constexpr u32 FQ_R2[FQ_NL] = {0x166eb04, 0x22a0746, 0x16b86, 0x1865406, 0x98e615, 0x2d3e263, 0x1531600, 0x265a6ff, 0x1a30d3a, 0x2a11a};
static constexpr u32 FQ_NINV = 0x866389;
__device__ static constexpr u64 FQ_PM2[4] = {0x3c208c16d87cfd45ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

__device__ __forceinline__ fq fq_zero() {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) r.l[i] = 0;
  return r;
}
__device__ __forceinline__ fq fq_one() {
  fq r;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) r.l[i] = FQ_ONE[i];
  return r;
}
__device__ __forceinline__ bool fq_is_zero(const fq& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) o |= a.l[i];
  return o == 0;
}
__device__ __forceinline__ bool fq_eq(const fq& a, const fq& b) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) o |= a.l[i] ^ b.l[i];
  return o == 0;
}
// t[] = limbs of a 260-bit two's-complement value in (-p, p), borrow = -1 when it is negative: add p back in that case
__device__ __forceinline__ fq fq_fix_negative(const u32 t[FQ_NL], int borrow) {
  const u32 mask = (u32)borrow;
  fq r;
  int c = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) {
    int v = (int)t[j] + (int)(FQ_P[j] & mask) + c;
    r.l[j] = (u32)v & FQ_MASK;
    c = v >> FQ_LB;
  }
  return r;
}
__device__ __forceinline__ fq fq_add(const fq& a, const fq& b) {
  u32 t[FQ_NL];
  int c = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) {  // a + b - p with a signed carry
    int v = (int)(a.l[j] + b.l[j]) - (int)FQ_P[j] + c;
    t[j] = (u32)v & FQ_MASK;
    c = v >> FQ_LB;
  }
  return fq_fix_negative(t, c);
}
__device__ __forceinline__ fq fq_sub(const fq& a, const fq& b) {
  u32 t[FQ_NL];
  int c = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) {
    int v = (int)a.l[j] - (int)b.l[j] + c;
    t[j] = (u32)v & FQ_MASK;
    c = v >> FQ_LB;
  }
  return fq_fix_negative(t, c);
}
__device__ __forceinline__ fq fq_dbl(const fq& a) { return fq_add(a, a); }
__device__ __forceinline__ fq fq_neg(const fq& a) { return fq_sub(fq_zero(), a); }  // -0 = 0: the sum 0 - 0 is not negative

__device__ __forceinline__ fq fq_cond_sub_p(const u32 r[FQ_NL]);
// Montgomery product a*b*R^-1 mod p, operand scanning with interleaved reduction in radix 2^26.
__device__ __forceinline__ fq fq_mul(const fq& a, const fq& b) {
  u64 acc[FQ_NL + 1];
#pragma unroll
  for (int j = 0; j <= FQ_NL; j++) acc[j] = 0;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) {
    const u32 bi = b.l[i];
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] += (u64)a.l[j] * bi;
    const u32 m = ((u32)acc[0] * FQ_NINV) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] += (u64)m * FQ_P[j];
    // acc[0] is now a multiple of 2^26: divide by 2^26 (shift the window down by one limb)
    const u64 carry = acc[0] >> FQ_LB;
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] = acc[j + 1];
    acc[0] += carry;
    acc[FQ_NL] = 0;
  }
  // carry propagation: the result is below 2p
  u32 r[FQ_NL];
#pragma unroll
  for (int j = 0; j < FQ_NL - 1; j++) {
    acc[j + 1] += acc[j] >> FQ_LB;
    r[j] = (u32)acc[j] & FQ_MASK;
  }
  r[FQ_NL - 1] = (u32)acc[FQ_NL - 1];
  return fq_cond_sub_p(r);
}
// r[] (below 2p, limbs normalised) -> canonical
__device__ __forceinline__ fq fq_cond_sub_p(const u32 r[FQ_NL]) {
  u32 t[FQ_NL];
  int c = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) {
    int v = (int)r[j] - (int)FQ_P[j] + c;
    t[j] = (u32)v & FQ_MASK;
    c = v >> FQ_LB;
  }
  const u32 mask = (u32)c;  // all ones: r < p, keep r
  fq o;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) o.l[j] = (r[j] & mask) | (t[j] & ~mask);
  return o;
}
// a^2 R^-1: the 55 distinct limb products first (cross terms doubled), then the reduction: 155 multiply-adds instead of 200
__device__ __forceinline__ fq fq_sqr(const fq& a) {
  u64 c[2 * FQ_NL];
#pragma unroll
  for (int k = 0; k < 2 * FQ_NL; k++) c[k] = 0;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) {
    c[2 * i] += (u64)a.l[i] * a.l[i];
    const u32 d = a.l[i] << 1;
#pragma unroll
    for (int j = i + 1; j < FQ_NL; j++) c[i + j] += (u64)d * a.l[j];
  }
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) {
    const u32 m = ((u32)c[i] * FQ_NINV) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) c[i + j] += (u64)m * FQ_P[j];
    c[i + 1] += c[i] >> FQ_LB;
  }
  u32 r[FQ_NL];
#pragma unroll
  for (int j = 0; j < FQ_NL - 1; j++) {
    c[FQ_NL + j + 1] += c[FQ_NL + j] >> FQ_LB;
    r[j] = (u32)c[FQ_NL + j] & FQ_MASK;
  }
  r[FQ_NL - 1] = (u32)c[2 * FQ_NL - 1];
  return fq_cond_sub_p(r);
}

// ---- loose operands ---------------------------------------------------------------------------------------------
// fq_mul / fq_sqr / fq_mul2 accept "loose" operands: limbs up to 2^28.5 (a column sum of twenty limb products still fits 64
// bits) and any values whose products add up to at most 64 p^2 (R = 2^260 > 84 p: the Montgomery result stays below 2p and
// the final conditional subtraction makes it canonical).  Sums and differences that only feed a product therefore need no
// carry chain and no reduction - ten instructions instead of the eighty of fq_add / fq_sub.  Every function still RETURNS
// canonical values, so results are the same field elements bit for bit.
__device__ __forceinline__ fq fq_add_lazy(const fq& a, const fq& b) {  // a + b, limb-wise
  fq r;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) r.l[j] = a.l[j] + b.l[j];
  return r;
}
__device__ __forceinline__ fq fq_dbl_lazy(const fq& a) { return fq_add_lazy(a, a); }
__device__ __forceinline__ fq fq_tpl_lazy(const fq& a) {  // 3a
  fq r;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) r.l[j] = a.l[j] * 3u;
  return r;
}
// a - b + M p (M = 2, 4, 6), limb-wise and non-negative in every limb for b = a lazy sum of at most M/2 canonical values
// (limbs <= (M/2)(2^26 - 1), value < (M/2) p): M p is written with every limb below the top raised by (M/2) 2^26, borrowed
// from the limb above.  Result: value in (a, a + M p), limbs below a's + (M/2) 2^26 + 2^26.
template <int M>
struct FqSubConst {
  u32 k[FQ_NL];
  constexpr FqSubConst() : k{} {
    u64 c = 0;
    for (int j = 0; j < FQ_NL; j++) {  // digits of M p
      u64 v = (u64)M * FQ_P[j] + c;
      k[j] = (u32)(v & FQ_MASK);
      c = v >> FQ_LB;
    }
    k[FQ_NL - 1] += (u32)(c << FQ_LB);  // (nothing: M p < 2^260)
    for (int j = 0; j < FQ_NL - 1; j++) {
      k[j] += (u32)(M / 2) << FQ_LB;
      k[j + 1] -= (u32)(M / 2);
    }
  }
};
template <int M>
__device__ __forceinline__ fq fq_sub_lazy(const fq& a, const fq& b) {
  constexpr FqSubConst<M> K{};
  static_assert(K.k[FQ_NL - 1] >= (u32)(M / 2) * 0xc1913u, "top limb of M p too small");
  fq r;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) r.l[j] = a.l[j] + (K.k[j] - b.l[j]);
  return r;
}
// (a b + c d) R^-1 mod p, canonical: one interleaved reduction for both products (300 multiply-adds instead of the 400 of two
// products and an addition).  a b + c d <= 64 p^2.
__device__ __forceinline__ fq fq_mul2(const fq& a, const fq& b, const fq& c, const fq& d) {
  u64 acc[FQ_NL + 1];
#pragma unroll
  for (int j = 0; j <= FQ_NL; j++) acc[j] = 0;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) {
    const u32 bi = b.l[i], di = d.l[i];
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] += (u64)a.l[j] * bi;
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] += (u64)c.l[j] * di;
    const u32 m = ((u32)acc[0] * FQ_NINV) & FQ_MASK;
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] += (u64)m * FQ_P[j];
    const u64 carry = acc[0] >> FQ_LB;
#pragma unroll
    for (int j = 0; j < FQ_NL; j++) acc[j] = acc[j + 1];
    acc[0] += carry;
    acc[FQ_NL] = 0;
  }
  u32 r[FQ_NL];
#pragma unroll
  for (int j = 0; j < FQ_NL - 1; j++) {
    acc[j + 1] += acc[j] >> FQ_LB;
    r[j] = (u32)acc[j] & FQ_MASK;
  }
  r[FQ_NL - 1] = (u32)acc[FQ_NL - 1];
  return fq_cond_sub_p(r);
}

// 26-bit limbs <-> four 64-bit words (value below 2^256)
__device__ __forceinline__ fq fq_unpack(const u64 w[4]) {
  fq r;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) {
    const int bit = FQ_LB * j, k = bit >> 6, sh = bit & 63;
    u64 v = w[k] >> sh;
    if (sh + FQ_LB > 64 && k + 1 < 4) v |= w[k + 1] << (64 - sh);
    r.l[j] = (u32)v & FQ_MASK;
  }
  return r;
}
__device__ __forceinline__ fqw fq_pack(const fq& a) {
  fqw r;
#pragma unroll
  for (int k = 0; k < 4; k++) r.l[k] = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL; j++) {
    const int bit = FQ_LB * j, k = bit >> 6, sh = bit & 63;
    r.l[k] |= (u64)a.l[j] << sh;
    if (sh + FQ_LB > 64 && k + 1 < 4) r.l[k + 1] |= (u64)a.l[j] >> (64 - sh);
  }
  return r;
}
__device__ __forceinline__ fq fq_from_canonical(const u64* w) {  // w < p, plain (non-Montgomery) words
  fq r2;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) r2.l[i] = FQ_R2[i];
  return fq_mul(fq_unpack(w), r2);
}
__device__ __forceinline__ fqw fq_to_canonical(const fq& a) {
  fq one = fq_zero();
  one.l[0] = 1;
  return fq_pack(fq_mul(a, one));
}

// a^(p-2); a != 0 (Fermat: 254 squarings + 127 products, ~125 k instructions).  Kept for A/B runs (BN254S_FQ_INV_FERMAT at
// compile time) and as the definition fq_inv is tested against.
__device__ __noinline__ fq fq_inv_fermat(const fq& a) {
  fq r = fq_one();
  for (int i = 253; i >= 0; i--) {  // p-2 < 2^254
    r = fq_sqr(r);
    if ((FQ_PM2[i >> 6] >> (i & 63)) & 1) r = fq_mul(r, a);
  }
  return r;
}

// ---- inversion by divsteps (Bernstein-Yang "safegcd") -----------------------------------------------------------------
// f = p, g = x; a divstep is (delta, f, g) -> (1 - delta, g, (g - f)/2) if delta > 0 and g odd, else (1 + delta, f, (g + (g odd) f)/2);
// 586 of them bring any g < 2^254 to 0 and f to +-1 (the published bound for this variant: floor((45907 b + 26313)/19929),
// b = 254), and d, e follow along with (d, e) = x^-1 (f, g) mod p, so d = +-x^-1 at the end.  Twenty rounds of thirty divsteps:
// a round works on the low 30 bits of f and g only (32-bit registers) and yields a 2 x 2 integer matrix (u v; q r) with
// 2^30 (f', g') = (u v; q r)(f, g); the matrix is then applied to the full values - nine signed 30-bit limbs, 64-bit
// column sums from v_mad_i64_i32, no carry chain through VCC - and to (d, e) modulo p (a multiple of p makes the low 30 bits
// vanish before the shift).  ~12 k instructions per inversion against ~125 k of the Fermat form; branch-free, so the
// lanes of a wave stay together.  zeta = -(delta + 1/2) as a 32-bit integer.
static constexpr int FQ30_N = 9;
static constexpr int FQ30_MASK = (1 << 30) - 1;
__device__ static constexpr int FQ30_P[FQ30_N] = {0x187cfd47, 0x3082305b, 0x71ca8d3, 0x205aa45a, 0x1585d97, 0x116da06, 0x1a029b85, 0x139cb84c, 0x3064};
static constexpr u32 FQ30_PINV = 0x1b799c77;  // p^-1 mod 2^30
// R^3 mod p (R = 2^260) in 26-bit limbs: (x R)^-1 R^3 R^-1 = x^-1 R
__device__ static constexpr u32 FQ_R3[FQ_NL] = {0x3e3a1a8, 0xc91e0f, 0x16c513e, 0x25dab49, 0x2c59c28, 0x180fe5c, 0x3872393, 0x21fd714, 0x1ed8d19, 0x43c5b};

struct fq30 {
  int v[FQ30_N];
};
__device__ __forceinline__ void fq30_divsteps(int& zeta, u32 f, u32 g, int& u, int& v, int& q, int& r) {
  u32 uu = 1, vv = 0, qq = 0, rr = 1;
  int z = zeta;
#pragma unroll
  for (int i = 0; i < 30; i++) {
    u32 c1 = (u32)(z >> 31);          // delta > 0
    const u32 c2 = 0u - (g & 1u);     // g odd
    const u32 x = (f ^ c1) - c1, y = (uu ^ c1) - c1, w = (vv ^ c1) - c1;  // (f, u, v) negated when delta > 0
    g += x & c2;
    qq += y & c2;
    rr += w & c2;
    c1 &= c2;                          // swap
    z = (int)(((u32)z ^ c1) - 1u);
    f += g & c1;
    uu += qq & c1;
    vv += rr & c1;
    g >>= 1;
    uu <<= 1;
    vv <<= 1;
  }
  zeta = z;
  u = (int)uu;
  v = (int)vv;
  q = (int)qq;
  r = (int)rr;
}
// (f, g) <- (u f + v g, q f + r g) / 2^30 (exact)
__device__ __forceinline__ void fq30_update_fg(fq30& f, fq30& g, int u, int v, int q, int r) {
  long long cf = (long long)u * f.v[0] + (long long)v * g.v[0];
  long long cg = (long long)q * f.v[0] + (long long)r * g.v[0];
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < FQ30_N; i++) {
    cf += (long long)u * f.v[i] + (long long)v * g.v[i];
    cg += (long long)q * f.v[i] + (long long)r * g.v[i];
    f.v[i - 1] = (int)cf & FQ30_MASK;
    g.v[i - 1] = (int)cg & FQ30_MASK;
    cf >>= 30;
    cg >>= 30;
  }
  f.v[FQ30_N - 1] = (int)cf;
  g.v[FQ30_N - 1] = (int)cg;
}
// (d, e) <- (u d + v e, q d + r e) / 2^30 mod p, values kept in (-2p, p)
__device__ __forceinline__ void fq30_update_de(fq30& d, fq30& e, int u, int v, int q, int r) {
  const int sd = d.v[FQ30_N - 1] >> 31, se = e.v[FQ30_N - 1] >> 31;
  int md = (u & sd) + (v & se), me = (q & sd) + (r & se);  // + p times these: brings negative d, e back up
  long long cd = (long long)u * d.v[0] + (long long)v * e.v[0];
  long long ce = (long long)q * d.v[0] + (long long)r * e.v[0];
  md -= (int)((FQ30_PINV * (u32)cd + (u32)md) & (u32)FQ30_MASK);  // makes the low 30 bits of cd + p md vanish
  me -= (int)((FQ30_PINV * (u32)ce + (u32)me) & (u32)FQ30_MASK);
  cd += (long long)FQ30_P[0] * md;
  ce += (long long)FQ30_P[0] * me;
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < FQ30_N; i++) {
    cd += (long long)u * d.v[i] + (long long)v * e.v[i] + (long long)FQ30_P[i] * md;
    ce += (long long)q * d.v[i] + (long long)r * e.v[i] + (long long)FQ30_P[i] * me;
    d.v[i - 1] = (int)cd & FQ30_MASK;
    e.v[i - 1] = (int)ce & FQ30_MASK;
    cd >>= 30;
    ce >>= 30;
  }
  d.v[FQ30_N - 1] = (int)cd;
  e.v[FQ30_N - 1] = (int)ce;
}
// d in (-2p, p), negated when sign < 0 -> [0, p)
__device__ __forceinline__ void fq30_normalize(fq30& d, int sign) {
  int add = d.v[FQ30_N - 1] >> 31;
  const int neg = sign >> 31;
  int c = 0;
#pragma unroll
  for (int i = 0; i < FQ30_N; i++) {
    int x = d.v[i] + (FQ30_P[i] & add);
    x = (x ^ neg) - neg;
    c += x;
    d.v[i] = i < FQ30_N - 1 ? c & FQ30_MASK : c;
    if (i < FQ30_N - 1) c >>= 30;
  }
  add = d.v[FQ30_N - 1] >> 31;
  c = 0;
#pragma unroll
  for (int i = 0; i < FQ30_N; i++) {
    c += d.v[i] + (FQ30_P[i] & add);
    d.v[i] = i < FQ30_N - 1 ? c & FQ30_MASK : c;
    if (i < FQ30_N - 1) c >>= 30;
  }
}
// a^-1 for a Montgomery residue a != 0 (0 -> 0).  Inlined: its body is a rolled loop of twenty rounds (~600 instructions), and a
// call would force the caller's live field elements through scratch memory (k_fq_batch_inv kept 688 B per lane there).
__device__ __forceinline__ fq fq_inv(const fq& a) {
#if defined(BN254S_FQ_INV_FERMAT)
  return fq_inv_fermat(a);
#else
  const fqw w = fq_pack(a);  // the residue as a plain integer below p
  fq30 f, g, d, e;
#pragma unroll
  for (int i = 0; i < FQ30_N; i++) {
    const int bit = 30 * i, k = bit >> 6, sh = bit & 63;
    u64 x = w.l[k] >> sh;
    if (sh + 30 > 64 && k + 1 < 4) x |= w.l[k + 1] << (64 - sh);
    g.v[i] = (int)((u32)x & (u32)FQ30_MASK);
    f.v[i] = FQ30_P[i];
    d.v[i] = 0;
    e.v[i] = i == 0 ? 1 : 0;
  }
  int zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; it++) {
    int u, v, q, r;
    fq30_divsteps(zeta, (u32)f.v[0], (u32)g.v[0], u, v, q, r);
    fq30_update_de(d, e, u, v, q, r);
    fq30_update_fg(f, g, u, v, q, r);
  }
  fq30_normalize(d, f.v[FQ30_N - 1]);  // f = +-1: d = +-(a R)^-1
  // nine 30-bit limbs -> ten 26-bit limbs, then times R^3 (Montgomery): (a R)^-1 R^3 R^-1 = a^-1 R
  u64 ww[5] = {0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < FQ30_N; i++) {
    const int bit = 30 * i, k = bit >> 6, sh = bit & 63;
    ww[k] |= (u64)(u32)d.v[i] << sh;
    if (sh + 30 > 64) ww[k + 1] |= (u64)(u32)d.v[i] >> (64 - sh);
  }
  fq r3;
#pragma unroll
  for (int i = 0; i < FQ_NL; i++) r3.l[i] = FQ_R3[i];
  return fq_mul(fq_unpack(ww), r3);
#endif
}

// ---- G1 (y^2 = x^3 + 3), Jacobian coordinates -----------------------------------------------------------
struct g1j {
  fq x, y, z;
};
__device__ __forceinline__ g1j g1_double(const g1j& p) {
  fq a = fq_sqr(p.x), b = fq_sqr(p.y), c = fq_sqr(b);
  fq xb = fq_add_lazy(p.x, b);  // only squared: < 2p
  fq d = fq_dbl(fq_sub(fq_sub(fq_sqr(xb), a), c));
  fq e = fq_tpl_lazy(a);        // only multiplied: < 3p (e^2 <= 9 p^2, e (d - x3) < 3 p^2)
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
// a b for b canonical and a canonical or loose with a.c0, a.c1 < 3p (limbs <= 3 (2^26 - 1)): each component is one two-product
// Montgomery reduction, c0 = a0 b0 + a1 (2p - b1) <= 9 p^2, c1 = a0 b1 + a1 b0 <= 6 p^2.
__device__ __forceinline__ fq2 fq2_mul(const fq2& a, const fq2& b) {
  const fq nb1 = fq_sub_lazy<2>(fq_zero(), b.c1);
  fq2 r;
  r.c0 = fq_mul2(a.c0, b.c0, a.c1, nb1);
  r.c1 = fq_mul2(a.c0, b.c1, a.c1, b.c0);
  return r;
}
// a^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u: two Fq products.  M/2 = how many canonical values a component of a may be the lazy
// sum of (1: canonical; 2: x + y; 3: 3x): (a0 + a1)(a0 - a1 + M p) <= (M/2)(2p) (M/2 + M) p <= 54 p^2.
template <int M = 2>
__device__ __forceinline__ fq2 fq2_sqr(const fq2& a) {
  fq2 r;
  r.c0 = fq_mul(fq_add_lazy(a.c0, a.c1), fq_sub_lazy<M>(a.c0, a.c1));
  r.c1 = fq_mul(fq_dbl_lazy(a.c0), a.c1);
  return r;
}
__device__ __forceinline__ fq fq2_norm(const fq2& a) { return fq_mul2(a.c0, a.c0, a.c1, a.c1); }
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
  fq2 xb, e;  // loose (only multiplied): x + b < 2p, 3a < 3p
  xb.c0 = fq_add_lazy(p.x.c0, b.c0);
  xb.c1 = fq_add_lazy(p.x.c1, b.c1);
  fq2 d = fq2_dbl(fq2_sub(fq2_sub(fq2_sqr<4>(xb), a), c));
  e.c0 = fq_tpl_lazy(a.c0);
  e.c1 = fq_tpl_lazy(a.c1);
  fq2 f = fq2_sqr<6>(e);
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
