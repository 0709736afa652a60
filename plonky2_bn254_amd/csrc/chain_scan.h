// Phase A of the scalar-multiplication trace generators as a parallel prefix sum.
//
// generate_one_set (reference src/starks/curves/g1/scalar_mul_stark.rs:92-213, g2 twin) walks 256 double-and-add steps:
// C_k = S_k + D_k, S_{k+1} = bit_k ? C_k : S_k, D_{k+1} = 2 D_k.  Only the doublings are inherently sequential; the running
// sums are S_k = offset + sum_{j<k, bit_j} D_j, an inclusive scan over F_0 = offset, F_{k+1} = bit_k ? D_k : infinity.
// One 256-lane workgroup per instance runs that scan in LDS (8 Hillis-Steele steps with a complete addition law) and then
// adds D_k once more for C_k.  The trace only ever sees canonical affine coordinates, so the result is the same as the
// sequential walk; S_k = -D_k (the case the reference cannot prove, add.rs:49-51) is reported exactly where the
// sequential walk would meet it, in the final S_k + D_k.
#pragma once
#include "fq_dev.h"

__device__ __forceinline__ bool pt_inf(const g1j& p) { return fq_is_zero(p.z); }
__device__ __forceinline__ bool pt_inf(const g2j& p) { return fq2_is_zero(p.z); }
__device__ __forceinline__ g1j pt_infinity(const g1j*) {
  g1j r;
  r.x = fq_one();
  r.y = fq_one();
  r.z = fq_zero();
  return r;
}
__device__ __forceinline__ g2j pt_infinity(const g2j*) {
  g2j r;
  r.x = fq2_one();
  r.y = fq2_one();
  r.z = fq2_zero();
  return r;
}
__device__ __forceinline__ int pt_add(const g1j& a, const g1j& b, g1j& r) { return g1_add(a, b, r); }
__device__ __forceinline__ int pt_add(const g2j& a, const g2j& b, g2j& r) { return g2_add(a, b, r); }

// complete addition: either operand may be the point at infinity, equal points are doubled, opposite points give infinity
template <class P>
__device__ __forceinline__ P pt_add_complete(const P& p, const P& q) {
  if (pt_inf(p)) return q;
  if (pt_inf(q)) return p;
  P r;
  if (pt_add(p, q, r) == 2) r = pt_infinity((const P*)nullptr);
  return r;
}

// LDS image of a point: word w of lane k at sh[w * 256 + k] (conflict-free for consecutive lanes)
__device__ __forceinline__ void lds_put_fq(u64* sh, int w0, int k, const fq& v) {
  const fqw w = fq_pack(v);
#pragma unroll
  for (int l = 0; l < 4; l++) sh[(w0 + l) * 256 + k] = w.l[l];
}
__device__ __forceinline__ fq lds_get_fq(const u64* sh, int w0, int k) {
  u64 w[4];
#pragma unroll
  for (int l = 0; l < 4; l++) w[l] = sh[(w0 + l) * 256 + k];
  return fq_unpack(w);
}
__device__ __forceinline__ void lds_put(u64* sh, int k, const g1j& p) {
  lds_put_fq(sh, 0, k, p.x);
  lds_put_fq(sh, 4, k, p.y);
  lds_put_fq(sh, 8, k, p.z);
}
__device__ __forceinline__ void lds_get(const u64* sh, int k, g1j& p) {
  p.x = lds_get_fq(sh, 0, k);
  p.y = lds_get_fq(sh, 4, k);
  p.z = lds_get_fq(sh, 8, k);
}
__device__ __forceinline__ void lds_put(u64* sh, int k, const g2j& p) {
  lds_put_fq(sh, 0, k, p.x.c0);
  lds_put_fq(sh, 4, k, p.x.c1);
  lds_put_fq(sh, 8, k, p.y.c0);
  lds_put_fq(sh, 12, k, p.y.c1);
  lds_put_fq(sh, 16, k, p.z.c0);
  lds_put_fq(sh, 20, k, p.z.c1);
}
__device__ __forceinline__ void lds_get(const u64* sh, int k, g2j& p) {
  p.x.c0 = lds_get_fq(sh, 0, k);
  p.x.c1 = lds_get_fq(sh, 4, k);
  p.y.c0 = lds_get_fq(sh, 8, k);
  p.y.c1 = lds_get_fq(sh, 12, k);
  p.z.c0 = lds_get_fq(sh, 16, k);
  p.z.c1 = lds_get_fq(sh, 20, k);
}

// Inclusive scan of the 256 points held one per lane (lane k ends with F_0 + ... + F_k).  sh: sizeof(P)/8 * 256 words.
template <class P>
__device__ __forceinline__ void pt_scan256(P& f, u64* sh, int k) {
#pragma unroll 1
  for (int d = 1; d < 256; d <<= 1) {
    lds_put(sh, k, f);
    __syncthreads();
    if (k >= d) {
      P q;
      lds_get(sh, k - d, q);
      f = pt_add_complete(q, f);
    }
    __syncthreads();
  }
}
