// Shared pieces of the quotient (constraint-evaluation) kernels.
//
// starky's ConstraintConsumer folds the constraint stream c_0, c_1, ... Horner-style:
//   acc_j <- acc_j * alpha_j + c_e          (consumer.constraint)
// so after K constraints acc_j = sum_e c_e * alpha_j^(K-1-e).  The emission ORDER is therefore part of the
// transcript; the kernels keep it by numbering constraints exactly in the order of the reference's
// eval_packed_generic and multiplying each by the precomputed weight W_j[e] = alpha_j^(K-1-e).  Weights are
// uniform over the grid (scalar loads); products are accumulated un-reduced in 128+ bits.
#pragma once
#include "gl_dev.h"
#include "aux.h"

struct Acc {  // lazy accumulator: value = lo + hi*2^64 + ov*2^128
  u64 lo, hi;
  u32 ov;
};
__device__ __forceinline__ void acc_init(Acc& a) {
  a.lo = 0;
  a.hi = 0;
  a.ov = 0;
}
__device__ __forceinline__ void acc_mad(Acc& a, u64 x, u64 w) {
  const unsigned __int128 pr = (unsigned __int128)x * w;  // one 128-bit product (shared partial products)
  u64 pl = (u64)pr, ph = (u64)(pr >> 64);
  a.lo += pl;
  u64 c = a.lo < pl ? 1 : 0;
  u64 h = a.hi + ph;
  u32 o = h < ph ? 1 : 0;
  h += c;
  o += h < c ? 1 : 0;
  a.hi = h;
  a.ov += o;
}
__device__ __forceinline__ u64 acc_red(const Acc& a) {
  // 2^128 = -2^32 (mod p); ov stays far below 2^32
  return gl_sub(gl_reduce128(a.lo, a.hi), ((u64)a.ov) << 32);
}
struct Acc2 {
  Acc a0, a1;
};
__device__ __forceinline__ void acc2_init(Acc2& a) {
  acc_init(a.a0);
  acc_init(a.a1);
}
__device__ __forceinline__ void acc2_mad(Acc2& a, u64 x, u64 w0, u64 w1) {
  acc_mad(a.a0, x, w0);
  acc_mad(a.a1, x, w1);
}

// Carry-free accumulator for sums of products value * weight with weights that are uniform over the wave (powers of a
// challenge): both factors are cut into 22-bit limbs, the nine limb products go straight into five 64-bit column sums with
// v_mad_u64_u32 (no carries, no register shuffling; the weight limbs come through scalar loads), and the columns are folded
// once at the end.  Up to 2^17 terms per accumulator.  About half the instructions of acc_mad per term.
static constexpr u32 M22 = (1u << 22) - 1;
struct W3 {  // a weight cut in limbs (host- or table-side)
  u32 w0, w1, w2;
};
GL_HD W3 w3_split(u64 w) {
  W3 r;
  r.w0 = (u32)w & M22;
  r.w1 = (u32)(w >> 22) & M22;
  r.w2 = (u32)(w >> 44);
  return r;
}
struct Acc3 {
  u64 c[5];
};
__device__ __forceinline__ void acc3_init(Acc3& a) {
#pragma unroll
  for (int k = 0; k < 5; k++) a.c[k] = 0;
}
__device__ __forceinline__ void acc3_mad(Acc3& a, u32 v0, u32 v1, u32 v2, u32 w0, u32 w1, u32 w2) {
  a.c[0] += (u64)v0 * w0;
  a.c[1] += (u64)v0 * w1;
  a.c[1] += (u64)v1 * w0;
  a.c[2] += (u64)v0 * w2;
  a.c[2] += (u64)v1 * w1;
  a.c[2] += (u64)v2 * w0;
  a.c[3] += (u64)v1 * w2;
  a.c[3] += (u64)v2 * w1;
  a.c[4] += (u64)v2 * w2;
}
__device__ __forceinline__ u64 acc3_red(const Acc3& a) {
  // c0 + c1 2^22 + c2 2^44 fits 128 bits (columns stay below 2^62); c3 2^66 and c4 2^88 through their residues
  unsigned __int128 t = (unsigned __int128)a.c[0] + ((unsigned __int128)a.c[1] << 22) + ((unsigned __int128)a.c[2] << 44);
  u64 r = gl_reduce128((u64)t, (u64)(t >> 64));
  r = gl_add(r, gl_mul_2exp<66>(gl_reduce128(a.c[3], 0)));
  r = gl_add(r, gl_mul_2exp<88>(gl_reduce128(a.c[4], 0)));
  return r;
}

// Two Acc3 (one per challenge alpha) fed from a table of pre-cut weights: 8 u32 per term {alpha_0: w0 w1 w2 -, alpha_1: w0 w1 w2 -}
// (quotient_host_tables appends these to W and to the modulus-zero tables).  29 instead of 45 vector instructions per term
// against acc2_mad: the nine limb products need no carry handling and the weight limbs arrive through scalar loads.
struct AccW {
  Acc3 a0, a1;
};
__device__ __forceinline__ void accw_init(AccW& a) {
  acc3_init(a.a0);
  acc3_init(a.a1);
}
__device__ __forceinline__ void accw_mad(AccW& a, u64 x, const u32* __restrict__ w) {  // x canonical
  const u32 v0 = (u32)x & M22, v1 = (u32)(x >> 22) & M22, v2 = (u32)(x >> 44);
  acc3_mad(a.a0, v0, v1, v2, w[0], w[1], w[2]);
  acc3_mad(a.a1, v0, v1, v2, w[4], w[5], w[6]);
}
GL_HD void w3_store(u64* dst /* 4 words = 8 u32 */, u64 wa, u64 wb) {
  const W3 a = w3_split(wa), b = w3_split(wb);
  dst[0] = (u64)a.w0 | ((u64)a.w1 << 32);
  dst[1] = (u64)a.w2;
  dst[2] = (u64)b.w0 | ((u64)b.w1 << 32);
  dst[3] = (u64)b.w2;
}

// Per-LDE-point constants in bit-reversed (Merkle leaf) order, built once per context.
struct QPointTables {
  const u64* x;       // x_j = shift_h * w_N^k,  j = h*N + bitrev(k)
  const u64* lfirst;  // L_first(x_j)
  const u64* llast;   // L_last(x_j)
};

// position of the "next" row (natural index + 2, i.e. k + 1 on the same coset) in bit-reversed order
__device__ __forceinline__ size_t next_position(size_t j, unsigned log_n) {
  const size_t N = (size_t)1 << log_n;
  size_t h = j >> log_n;
  u32 k = bitrev32((u32)(j & (N - 1)), log_n);
  k = (k + 1) & (u32)(N - 1);
  return (h << log_n) | bitrev32(k, log_n);
}

// LogUp constraints of both challenges (starky eval_packed_lookups_generic) followed by the
// CTL constraints (eval_cross_table_lookup_checks, A.7).  e0 = index of the first of these constraints.
__device__ __forceinline__ void lookup_and_ctl_constraints(const StarkShape& sh, const u64* __restrict__ tl,
                                                           const u64* __restrict__ al, size_t M2, size_t j, size_t jn,
                                                           const u32* __restrict__ W3 /* cut weights, 8 u32 per constraint */, int e0,
                                                           const u64 betas[2], const u64 gammas[2], u64 lfirst, u64 llast,
                                                           u64 z_last, u64& tot0, u64& tot1) {
  const int m = sh.n_helpers(), n_rc = sh.n_rc();
  int e = e0;
  AccW acc;
  accw_init(acc);
  const u64 table = tl[(size_t)sh.table_col * M2 + j], freq = tl[(size_t)sh.freq_col * M2 + j];
  u64 hs[2] = {0, 0};
#pragma unroll 4
  for (int k = 0; k < m; k++) {  // unrolled: several columns' loads in flight
    u64 f0 = tl[(size_t)(sh.rc_begin + 2 * k) * M2 + j];
    bool two = (2 * k + 1) < n_rc;
    u64 f1 = two ? tl[(size_t)(sh.rc_begin + 2 * k + 1) * M2 + j] : 0;
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
      u64 h = al[(size_t)(ch * (m + 1) + k) * M2 + j];
      u64 g0 = gl_add(f0, betas[ch]);
      u64 c;
      if (two) {
        u64 g1 = gl_add(f1, betas[ch]);
        c = gl_sub(gl_sub(gl_mul(gl_mul(g1, g0), h), g1), g0);
      } else {
        c = gl_sub(gl_mul(g0, h), 1);
      }
      int ee = e0 + ch * (m + 2) + k;
      accw_mad(acc, c, W3 + 8 * (ee));
      hs[ch] = gl_add(hs[ch], h);
    }
  }
#pragma unroll
  for (int ch = 0; ch < 2; ch++) {
    u64 z = al[(size_t)(ch * (m + 1) + m) * M2 + j], nz = al[(size_t)(ch * (m + 1) + m) * M2 + jn];
    u64 twc = gl_add(table, betas[ch]);
    u64 y = gl_sub(gl_mul(hs[ch], twc), freq);
    int ee = e0 + ch * (m + 2) + m;
    accw_mad(acc, gl_mul(z, lfirst), W3 + 8 * (ee));
    accw_mad(acc, gl_sub(gl_mul(gl_sub(nz, z), twc), y), W3 + 8 * (ee + 1));
  }
  e = e0 + 2 * (m + 2);
  // CTL: one Z per (ctl, challenge), no helper columns
  for (int ctl = 0; ctl < sh.n_ctl; ctl++) {
    u64 a0 = 0, a1 = 0;
    for (int mm = sh.ctl.ncols[ctl] - 1; mm >= 0; mm--) {
      int start = sh.ctl.col_start[ctl][mm], nb = sh.ctl.col_bits[ctl][mm];
      u64 v = 0;
      for (int b = nb - 1; b >= 0; b--) v = gl_add(gl_dbl(v), tl[(size_t)(start + b) * M2 + j]);
      a0 = gl_add(gl_mul(a0, betas[0]), v);
      a1 = gl_add(gl_mul(a1, betas[1]), v);
    }
    u64 comb[2] = {gl_add(a0, gammas[0]), gl_add(a1, gammas[1])};
    u64 f0 = tl[(size_t)sh.ctl.filter_col[ctl] * M2 + j];
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
      int zc = 2 * (m + 1) + ctl * 2 + ch;
      u64 lz = al[(size_t)zc * M2 + j], nz = al[(size_t)zc * M2 + jn];
      u64 c_last = gl_mul(gl_sub(gl_mul(comb[ch], lz), f0), llast);
      u64 c_tr = gl_mul(gl_sub(gl_mul(comb[ch], gl_sub(lz, nz)), f0), z_last);
      accw_mad(acc, c_last, W3 + 8 * (e));
      accw_mad(acc, c_tr, W3 + 8 * (e + 1));
      e += 2;
    }
  }
  tot0 = gl_add(tot0, acc3_red(acc.a0));
  tot1 = gl_add(tot1, acc3_red(acc.a1));
}
