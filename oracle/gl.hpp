// ORACLE (test infrastructure, NOT the product path): CPU restatement of the reference algorithm.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Goldilocks field F (p = 2^64 - 2^32 + 1) and its quadratic extension F2 = F[X]/(X^2 - 7).
// Restates plonky2_field 0.2.2 `GoldilocksField` / `QuadraticExtension` (un-vendored dependency of the
// reference, Cargo.lock:613-616; constants validated in SURVEY.md App. B).  parity unpinned: the
// pinned fork source is not available offline, see DESIGN.md.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 GL_P = 0xFFFFFFFF00000001ULL;
static const u64 GL_EPS = 0xFFFFFFFFULL;                   // 2^64 mod p
static const u64 GL_GENERATOR = 0xc65c18b67785d900ULL;     // MULTIPLICATIVE_GROUP_GENERATOR = coset_shift()
static const u64 GL_POW2_GENERATOR = 0x64fdd1a46201e246ULL; // order 2^32
static const u64 GL_W = 7;                                  // X^2 = 7

// (written with overflow builtins and selects rather than branches: the conditions are data-dependent coin flips, and a
// mispredicted branch costs more than the arithmetic; same values as the obvious if-forms)
static inline u64 gl_add(u64 a, u64 b) {
  u64 s;
  const bool carry = __builtin_add_overflow(a, b, &s);
  return s - ((carry | (s >= GL_P)) ? GL_P : 0);
}
static inline u64 gl_sub(u64 a, u64 b) {
  u64 d;
  const bool borrow = __builtin_sub_overflow(a, b, &d);
  return d + (borrow ? GL_P : 0);
}
static inline u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
static inline u64 gl_reduce128(u128 x) {
  // x = lo + 2^64 (hi_lo + 2^32 hi_hi) = lo + (2^32 - 1) hi_lo - hi_hi  (mod p), since 2^64 = 2^32 - 1 and 2^96 = -1
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0;
  const bool borrow = __builtin_sub_overflow(lo, hi_hi, &t0);
  t0 -= borrow ? GL_EPS : 0;                 // borrow: add p, i.e. subtract 2^32 - 1 mod 2^64 (no second borrow: t0 >= 2^64 - 2^32)
  const u64 t1 = (hi_lo << 32) - hi_lo;      // hi_lo (2^32 - 1) < 2^64 - 2^33
  u64 r;
  const bool carry = __builtin_add_overflow(t0, t1, &r);
  // a wrapped sum stands for r + 2^64 = r + 2^32 - 1 (mod p), and r >= p is brought back by - p = + 2^32 - 1 (mod 2^64):
  // the two cases exclude each other (a wrapped r is below 2^64 - 2^33 + ... - 2^64 < p - 2^32) and share one correction
  return (carry | (r >= GL_P)) ? r + GL_EPS : r;
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_reduce128((u128)a * b); }
static inline u64 gl_pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, a);
    a = gl_mul(a, a);
    e >>= 1;
  }
  return r;
}
static inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
static inline u64 gl_from_i64(int64_t x) { return x >= 0 ? (u64)x % GL_P : GL_P - ((u64)(-x) % GL_P); }
// primitive_root_of_unity(k) = POWER_OF_TWO_GENERATOR^(2^(32-k))
static inline u64 gl_root_of_unity(unsigned k) {
  u64 r = GL_POW2_GENERATOR;
  for (unsigned i = k; i < 32; i++) r = gl_mul(r, r);
  return r;
}

// Operator-overloaded wrappers so that constraint code can be written once, generic over F / F2
// (the reference's `P: PackedField` genericity, e.g. scalar_mul_stark.rs:226-232).
struct F {
  u64 v;
  F() : v(0) {}
  explicit F(u64 x) : v(x) {}
  static F from_u64(u64 x) { return F(x % GL_P); }
  F operator+(F o) const { return F(gl_add(v, o.v)); }
  F operator-(F o) const { return F(gl_sub(v, o.v)); }
  F operator*(F o) const { return F(gl_mul(v, o.v)); }
  F operator-() const { return F(gl_neg(v)); }
  F& operator+=(F o) { v = gl_add(v, o.v); return *this; }
  F& operator-=(F o) { v = gl_sub(v, o.v); return *this; }
  F& operator*=(F o) { v = gl_mul(v, o.v); return *this; }
  bool operator==(F o) const { return v == o.v; }
  bool operator!=(F o) const { return v != o.v; }
  F inv() const { return F(gl_inv(v)); }
};

struct F2 {
  u64 c0, c1;
  F2() : c0(0), c1(0) {}
  explicit F2(u64 a) : c0(a), c1(0) {}
  F2(u64 a, u64 b) : c0(a), c1(b) {}
  static F2 from_u64(u64 x) { return F2(x % GL_P, 0); }
  static F2 from_base(F x) { return F2(x.v, 0); }
  F2 operator+(F2 o) const { return F2(gl_add(c0, o.c0), gl_add(c1, o.c1)); }
  F2 operator-(F2 o) const { return F2(gl_sub(c0, o.c0), gl_sub(c1, o.c1)); }
  F2 operator-() const { return F2(gl_neg(c0), gl_neg(c1)); }
  F2 operator*(F2 o) const {
    u64 a = gl_add(gl_mul(c0, o.c0), gl_mul(GL_W, gl_mul(c1, o.c1)));
    u64 b = gl_add(gl_mul(c0, o.c1), gl_mul(c1, o.c0));
    return F2(a, b);
  }
  F2 scalar_mul(u64 s) const { return F2(gl_mul(c0, s), gl_mul(c1, s)); }
  F2& operator+=(F2 o) { *this = *this + o; return *this; }
  F2& operator-=(F2 o) { *this = *this - o; return *this; }
  F2& operator*=(F2 o) { *this = *this * o; return *this; }
  bool operator==(F2 o) const { return c0 == o.c0 && c1 == o.c1; }
  bool operator!=(F2 o) const { return !(*this == o); }
  F2 inv() const {
    // 1/(a+bX) = (a-bX)/(a^2 - 7 b^2)
    u64 n = gl_sub(gl_mul(c0, c0), gl_mul(GL_W, gl_mul(c1, c1)));
    u64 ni = gl_inv(n);
    return F2(gl_mul(c0, ni), gl_mul(gl_neg(c1), ni));
  }
  F2 pow(u64 e) const {
    F2 r(1), a = *this;
    while (e) {
      if (e & 1) r = r * a;
      a = a * a;
      e >>= 1;
    }
    return r;
  }
  F2 exp_power_of_2(unsigned k) const {
    F2 r = *this;
    for (unsigned i = 0; i < k; i++) r = r * r;
    return r;
  }
};

static inline unsigned log2_strict(size_t n) {
  unsigned k = 0;
  while ((size_t(1) << k) < n) k++;
  assert((size_t(1) << k) == n);
  return k;
}
static inline size_t reverse_bits(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

// Montgomery batch inversion (F::batch_multiplicative_inverse): all inputs non-zero.
static inline std::vector<u64> gl_batch_inv(const std::vector<u64>& x) {
  size_t n = x.size();
  std::vector<u64> out(n);
  if (!n) return out;
  std::vector<u64> pre(n);
  u64 acc = 1;
  for (size_t i = 0; i < n; i++) {
    pre[i] = acc;
    acc = gl_mul(acc, x[i]);
  }
  u64 inv = gl_inv(acc);
  for (size_t i = n; i-- > 0;) {
    out[i] = gl_mul(inv, pre[i]);
    inv = gl_mul(inv, x[i]);
  }
  return out;
}

}  // namespace orc
