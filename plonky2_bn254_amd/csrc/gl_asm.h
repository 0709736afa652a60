// Hand-written gfx950 instruction sequences for the Goldilocks butterflies of the NTT kernels (device only).
//
// The compiler's code for a canonical butterfly (a + b, (a - b) 2^S) is 6 + 6 + 18 vector instructions plus hazard
// s_nops (it re-derives the borrow with a 64-bit compare, materialises 64-bit selects and reduces a shifted value through
// the general 128-bit reduction).  The blocks below are inline asm whose operands the compiler allocates and which it
// schedules as units; they are the sequences counted in profiles/r02_ntt_isa_budget.md:
//   sum        5 VALU + 1 SALU   64-bit add, two 64-bit compares, one select, one multiply-add (s + (2^32-1) if needed)
//   difference 5 VALU            two borrow chains (a - b, then - (2^32-1) if it borrowed)
//   x 2^S, S <= 32         6 VALU + 1 SALU   (x << S) + (x >> (64-S)) (2^32-1) as ONE multiply-add with carry-out
//   x 2^-K, K <= 32        9 VALU + 1 SALU   Montgomery-style: (x + m) >> K + m 2^(32-K) (2^32-1), m = -x mod 2^K
//   x 2^(32+S)            11 VALU + 2 SALU   x 2^S, then y0 2^32 + y1 (2^32-1)
//   a b                   19 VALU + 1 SALU   four multiply-adds chained through 64-bit shifts, no register moves
// Every function takes canonical operands (< p) and returns canonical results; all of them are exact for every input
// (tests/test_gpu_kernels.py::test_field_asm_edge_cases drives them with boundary values against Python integers).
//
// gfx950 rule used throughout: a VALU instruction may read an SGPR (carry, mask) written by another VALU instruction only
// two issue slots later; SALU reads and SALU-written masks need no padding.  Operand halves: the compiler coalesces
// "(u32)x" / "x >> 32" of a live 64-bit register pair into the pair's own registers, so halves are passed as operands.
#pragma once
#include "gl_dev.h"

#if defined(__HIP_DEVICE_COMPILE__)

static constexpr u64 GL_PM1 = GL_P - 1;

// s = a + b, d = a - b (REV: d = b - a) mod p
template <bool REV>
__device__ __forceinline__ void gl_bfly_asm(u64 a, u64 b, u64& s, u64& d) {
  u64 S, c0, c1, cb;
  u32 d0, d1, e, f;
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 m0 = REV ? b0 : a0, m1 = REV ? b1 : a1, n0 = REV ? a0 : b0, n1 = REV ? a1 : b1;  // d = m - n
  asm("v_lshl_add_u64 %0, %8, 0, %9\n"        // S = a + b (mod 2^64)
      "v_sub_co_u32 %1, %7, %10, %12\n"       // d0 = m0 - n0, borrow -> cb
      "v_cmp_lt_u64 %5, %0, %8\n"             // c0 = S < a: the sum wrapped
      "v_cmp_lt_u64 %6, %14, %0\n"            // c1 = p - 1 < S
      "v_subb_co_u32 %2, %7, %11, %13, %7\n"  // d1 = m1 - n1 - borrow, borrow -> cb
      "s_or_b64 %5, %5, %6\n"
      "v_cndmask_b32 %3, 0, -1, %5\n"         // e = 2^32 - 1 if the sum needs - p
      "v_cndmask_b32 %4, 0, -1, %7\n"         // f = 2^32 - 1 if the difference needs + p
      "v_sub_co_u32 %1, %7, %1, %4\n"         // d - f, low word
      "v_mad_u64_u32 %0, %6, %3, 1, %0\n"     // S + e (mod 2^64)
      "s_nop 0\n"
      "v_subb_co_u32 %2, %7, %2, 0, %7\n"     // d - f, high word
      : "=&v"(S), "=&v"(d0), "=&v"(d1), "=&v"(e), "=&v"(f), "=&s"(c0), "=&s"(c1), "=&s"(cb)
      : "v"(a), "v"(b), "v"(m0), "v"(m1), "v"(n0), "v"(n1), "s"(GL_PM1)
      : "scc");
  s = S;
  d = (u64)d0 | ((u64)d1 << 32);
}

// tail shared by the shift multiplications: T + carry 2^64 -> canonical, given T + carry 2^64 < 2 p
#define GL_ASM_TAIL(T, C0, C1, E, PM1)               \
  "v_cmp_lt_u64 " C1 ", " PM1 ", " T "\n"            \
  "s_or_b64 " C0 ", " C0 ", " C1 "\n"                \
  "v_cndmask_b32 " E ", 0, -1, " C0 "\n"             \
  "v_mad_u64_u32 " T ", " C1 ", " E ", 1, " T "\n"

// x 2^S, 0 < S <= 32: (x << S) mod 2^64 + (x >> (64 - S)) 2^64, and 2^64 = 2^32 - 1
template <int S>
__device__ __forceinline__ u64 gl_shl_small_asm(u64 x) {
  static_assert(S > 0 && S <= 32, "");
  u64 L, c0, c1;
  u32 M, e;
  const u32 x1 = (u32)(x >> 32);
  if constexpr (S == 32) {
    asm("v_lshlrev_b64 %0, 32, %5\n"
        "v_mad_u64_u32 %0, %2, %6, -1, %0\n" GL_ASM_TAIL("%0", "%2", "%3", "%4", "%7")
        : "=&v"(L), "=&v"(M), "=&s"(c0), "=&s"(c1), "=&v"(e)
        : "v"(x), "v"(x1), "s"(GL_PM1)
      : "scc");
  } else {
    asm("v_lshlrev_b64 %0, %7, %5\n"
        "v_lshrrev_b32 %1, %8, %6\n"
        "v_mad_u64_u32 %0, %2, %1, -1, %0\n" GL_ASM_TAIL("%0", "%2", "%3", "%4", "%9")
        : "=&v"(L), "=&v"(M), "=&s"(c0), "=&s"(c1), "=&v"(e)
        : "v"(x), "v"(x1), "n"(S), "n"(32 - S), "s"(GL_PM1)
      : "scc");
  }
  return L;
}

// x 2^-K, 0 < K < 32: with m = -x mod 2^K, x + m p is divisible by 2^K and (x + m p) / 2^K = ((x + m) >> K) + m 2^(32-K) (2^32 - 1)
template <int K>
__device__ __forceinline__ u64 gl_shr_small_asm(u64 x) {
  static_assert(K > 0 && K < 32, "");
  u64 Q, c0, c1;
  u32 mh, ml, e;
  const u32 x0 = (u32)x;
  asm("v_sub_u32 %1, 0, %7\n"                 // -x0
      "v_lshlrev_b32 %1, %9, %1\n"            // mh = m 2^(32-K)
      "v_lshrrev_b32 %2, %9, %1\n"            // ml = m
      "v_mad_u64_u32 %0, %3, %2, 1, %6\n"     // x + m < 2^64
      "v_lshrrev_b64 %0, %8, %0\n"
      "v_mad_u64_u32 %0, %3, %1, -1, %0\n" GL_ASM_TAIL("%0", "%3", "%4", "%5", "%10")
      : "=&v"(Q), "=&v"(mh), "=&v"(ml), "=&s"(c0), "=&s"(c1), "=&v"(e)
      : "v"(x), "v"(x0), "n"(K), "n"(32 - K), "s"(GL_PM1)
      : "scc");
  return Q;
}

// x 2^S for a compile-time S in (0, 96); canonical in, canonical out
template <int S>
__device__ __forceinline__ u64 gl_shl_asm(u64 x) {
  static_assert(S > 0 && S < 96, "");
  if constexpr (S <= 32) return gl_shl_small_asm<S>(x);
  else if constexpr (S <= 64) return gl_shl_small_asm<32>(gl_shl_small_asm<S - 32>(x));
  else return gl_shl_small_asm<32>(gl_shl_small_asm<32>(gl_shl_small_asm<S - 64>(x)));
}


// a * b mod p, canonical operands and result.  The 128-bit product is four multiply-adds (the second and fourth take the
// high word of the previous sum as addend through a 64-bit shift), words (w1:w0) = (R0:P0), (w3:w2) = U + carry 2^32;
// w0 + w1 2^32 + w2 (2^32-1) - w3 is one multiply-add with carry-out and the shared tail, then a canonical subtraction of w3.
__device__ __forceinline__ u64 gl_mul_asm(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 P, R, U, H, cc, cj;
  asm("v_mad_u64_u32 %0, %5, %6, %8, 0\n"     // P = a0 b0
      "v_lshrrev_b64 %3, 32, %0\n"
      "v_mad_u64_u32 %1, %5, %7, %8, %3\n"    // a1 b0 + P1 (no carry)
      "v_mad_u64_u32 %1, %4, %6, %9, %1\n"    // R = a0 b1 + that, carry -> cc
      "v_lshrrev_b64 %3, 32, %1\n"
      "v_mad_u64_u32 %2, %5, %7, %9, %3\n"    // U = a1 b1 + R1
      : "=&v"(P), "=&v"(R), "=&v"(U), "=&v"(H), "=&s"(cc), "=&s"(cj)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
  const u32 p0 = (u32)P, u0 = (u32)U, u1 = (u32)(U >> 32);
  u64 T, c0, c1;
  u32 w3, e;
  asm("v_lshlrev_b64 %0, 32, %5\n"            // (R0 : 0)
      "v_addc_co_u32 %1, %3, %8, 0, %9\n"     // w3 = U1 + carry
      "v_mad_u64_u32 %0, %3, %6, 1, %0\n"     // (R0 : P0)
      "v_mad_u64_u32 %0, %2, %7, -1, %0\n"    // + w2 (2^32 - 1), carry -> c0
      GL_ASM_TAIL("%0", "%2", "%3", "%4", "%10")
      : "=&v"(T), "=&v"(w3), "=&s"(c0), "=&s"(c1), "=&v"(e)
      : "v"(R), "v"(p0), "v"(u0), "v"(u1), "s"(cc), "s"(GL_PM1)
      : "scc");
  const u32 t0 = (u32)T, t1 = (u32)(T >> 32);
  u32 d0, d1, f;
  u64 cb;
  asm("v_sub_co_u32 %0, %3, %4, %6\n"
      "s_nop 1\n"
      "v_subb_co_u32 %1, %3, %5, 0, %3\n"
      "s_nop 1\n"
      "v_cndmask_b32 %2, 0, -1, %3\n"
      "v_sub_co_u32 %0, %3, %0, %2\n"
      "s_nop 1\n"
      "v_subb_co_u32 %1, %3, %1, 0, %3\n"
      : "=&v"(d0), "=&v"(d1), "=&v"(f), "=&s"(cb)
      : "v"(t0), "v"(t1), "v"(w3));
  return (u64)d0 | ((u64)d1 << 32);
}

#else  // host pass of the same translation units: declarations only (kernels are never compiled for the host)
template <bool REV>
__device__ void gl_bfly_asm(u64 a, u64 b, u64& s, u64& d);
template <int S>
__device__ u64 gl_shl_small_asm(u64 x);
template <int K>
__device__ u64 gl_shr_small_asm(u64 x);
template <int S>
__device__ u64 gl_shl_asm(u64 x);
__device__ u64 gl_mul_asm(u64 a, u64 b);
#endif
