// Pieces of the cooperative doubling chains (trace_g1.hip: k_g1_dbl_chain_coop, trace_g2fq.hip: k_g2_dbl_chain_coop): the
// lanes that share one instance exchange field elements through LDS and evaluate the sums between the products of a
// doubling as small integer combinations with ONE reduction each.
#pragma once
#include "fq_dev.h"

namespace chain_coop {
constexpr int SLOT_W = 12;  // dwords per Fq in LDS (ten limbs, padded to three 16-byte words)

__device__ __forceinline__ fq lds_ld(const u32* g, int idx) {
  const uint4* q = reinterpret_cast<const uint4*>(g + idx * SLOT_W);
  const uint4 a = q[0], b = q[1], c = q[2];
  fq r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x; r.l[9] = c.y;
  return r;
}
__device__ __forceinline__ void lds_st(u32* g, int idx, const fq& v) {
  uint4* q = reinterpret_cast<uint4*>(g + idx * SLOT_W);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  q[2] = make_uint4(v.l[8], v.l[9], 0, 0);
}
// sum_i k_i S_i + off p, canonical.  S_i canonical, |k_i| small; off p makes the value non-negative; it must stay below 32 p
// and every limb of the integer combination below 2^31 in magnitude.
__device__ __forceinline__ fq combine(const u32* g, int s0, int k0, int s1, int k1, int s2, int k2, int s3, int k3, int off) {
  const fq v0 = lds_ld(g, s0), v1 = lds_ld(g, s1), v2 = lds_ld(g, s2), v3 = lds_ld(g, s3);
  int t[FQ_NL];
#pragma unroll
  for (int j = 0; j < FQ_NL; j++)
    t[j] = (int)v0.l[j] * k0 + (int)v1.l[j] * k1 + (int)v2.l[j] * k2 + (int)v3.l[j] * k3 + (int)FQ_P[j] * off;
  int cy = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL - 1; j++) {
    const int v = t[j] + cy;
    t[j] = v & (int)FQ_MASK;
    cy = v >> FQ_LB;
  }
  t[FQ_NL - 1] += cy;  // value in [0, 32p): top limb below 2^25
  // quotient estimate from the top limb: never above floor(value / p) and at most one below it
  const u32 q = (u32)(((u64)(u32)t[FQ_NL - 1] * (u64)(((u64)1 << 40) / (FQ_P[FQ_NL - 1] + 1))) >> 40);
  u32 r[FQ_NL];
  cy = 0;
#pragma unroll
  for (int j = 0; j < FQ_NL - 1; j++) {
    const int v = t[j] - (int)(q * FQ_P[j]) + cy;
    r[j] = (u32)v & FQ_MASK;
    cy = v >> FQ_LB;
  }
  r[FQ_NL - 1] = (u32)(t[FQ_NL - 1] - (int)(q * FQ_P[FQ_NL - 1]) + cy);
  return fq_cond_sub_p(r);  // below 2p here
}
}  // namespace chain_coop
