// Shared device helpers of the three trace generators (G1 / G2 / Fq exp): SoA Fq vectors, batched inversion,
// limb conversion, generate_modulus_zero (reference src/starks/modular/modulus_zero.rs:77-123), round-flag
// table, and the range-check columns (generate_range_checks).
#pragma once
#include "fq_dev.h"
#include "../../include/bn254_stark.h"

static constexpr int NPTS = 514;  // per instance: 0 = offset | one, 1+k = C_k (k<256), 257+k = D_k (k<=256)

// ---- SoA vectors of Fq elements: element e, limb l at base[l*count + e] ---------------------------------
// (the element is stored as the four 64-bit words of its Montgomery residue; registers hold ten 26-bit limbs, fq_dev.h)
__device__ __forceinline__ fq ld_fq(const u64* base, size_t count, size_t e) {
  u64 w[4];
#pragma unroll
  for (int l = 0; l < 4; l++) w[l] = base[l * count + e];
  return fq_unpack(w);
}
__device__ __forceinline__ void st_fq(u64* base, size_t count, size_t e, const fq& v) {
  const fqw w = fq_pack(v);
#pragma unroll
  for (int l = 0; l < 4; l++) base[l * count + e] = w.l[l];
}

// index of the highest set bit of s below position k, or -1
__device__ __forceinline__ int last_set_below(const u64 s[4], int k) {
  for (int w = 3; w >= 0; w--) {
    int lo = w * 64;
    if (k <= lo) continue;
    u64 m = s[w];
    if (k < lo + 64) m &= (1ULL << (k - lo)) - 1;
    if (m) return lo + 63 - __clzll((long long)m);
  }
  return -1;
}
// point index of the running sum after step k-1 (S_{k-1}); k = 0 -> offset
__device__ __forceinline__ int sum_point(const u64 s[4], int k) {
  int j = last_set_below(s, k);
  return j < 0 ? 0 : 1 + j;
}

__device__ __forceinline__ void fq_to_limbs(const fq& mont, int limbs[16]) {
  const fqw c = fq_to_canonical(mont);
#pragma unroll
  for (int i = 0; i < 16; i++) limbs[i] = (int)((c.l[i >> 2] >> (16 * (i & 3))) & 0xFFFF);
}

__device__ static constexpr int MOD_LIMBS[16] = {64839, 55420, 35862, 15392, 51853, 26737, 27281, 38785,
                                                  22621, 33153, 17846, 47184, 41001, 57649, 20082, 12388};
// p^-1 mod 2^288, 32-bit words
__device__ static constexpr u32 PINV288[9] = {0x1b799c77u, 0x782df87du, 0xe1359536u, 0x6121829au, 0xe7cc257fu,
                                               0x2750342fu, 0x6e777394u, 0x0a85dd48u, 0x5b52d390u};

// generate_modulus_zero (modulus_zero.rs:77-123): in = 31 signed limb coefficients of a multiple of p.
// Writes the 80 witness values to columns col0.. of `row` (trace column-major, N rows).
// The function is called four to ten times per row and is not inlined; its 31 inputs travel through LDS (coefficient i of the
// calling lane at in[i * MZ_LANES]: one wave per workgroup, a lane reads only what it wrote) - as a by-pointer argument they
// sat in scratch memory (256 B per lane in round 2).
static constexpr int MZ_LANES = 64;
typedef __attribute__((address_space(3))) long long mz_lds_t;
static __device__ __noinline__ void gen_modulus_zero_lds(const mz_lds_t* in_lds, u64* __restrict__ trace, size_t N, size_t row, int col0,
                                                         int* err) {
  // Everything is streamed from the LDS slots (in(i)) instead of being held in arrays: the function is called from kernels that keep
  // a dozen coordinates alive across the call, and its own registers add to theirs.
  auto in = [&](int i) -> long long { return in_lds[i * MZ_LANES]; };
  // low 288 bits of V = sum in[i] 2^(16 i), two's complement
  u32 v[9];
  long long carry = 0;
#pragma unroll
  for (int w = 0; w < 9; w++) {
    long long t = carry + in(2 * w) + (in(2 * w + 1) << 16);
    v[w] = (u32)t;
    carry = t >> 32;
  }
  // q = V * p^-1 mod 2^288 (exact quotient, two's complement)
  u32 q[9];
  u128 acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)v[i] * PINV288[k - i];
    q[k] = (u32)acc;
    acc >>= 32;
  }
  bool neg = (q[8] >> 31) != 0;
  bool nonzero = false;
  if (neg) {  // |q| = -q
    u64 c = 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      c += (u64)(~q[k]);
      q[k] = (u32)c;
      c >>= 32;
    }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) nonzero |= q[k] != 0;
  if ((q[8] >> 16) != 0) atomicCAS(err, 0, BN254S_E_INTERNAL);  // quotient wider than 17 limbs
  u64* out = trace + (size_t)col0 * N + row;
  out[0] = (!neg && nonzero) ? 1 : 0;
  int qs[17];  // signed limbs of the quotient
  const int sgn = neg ? -1 : 1;
#pragma unroll
  for (int i = 0; i < 17; i++) {
    const int qa = (int)((q[i >> 1] >> (16 * (i & 1))) & 0xFFFF);
    out[(size_t)(1 + i) * N] = (u64)qa;
    qs[i] = sgn * qa;
  }
  // constr = in - quot(x) * m(x), coefficient d formed when the division below needs it;
  // aux = constr / (x - 2^16) (pol_remove_root_2exp), shifted by 2^29, split in 16-bit halves
  long long a = 0;
  bool bad = false;
#pragma unroll
  for (int d = 0; d < 32; d++) {  // (unrolled: rolled it is 5 % slower for the row kernels and frees no registers they need)
    long long cd = d < 31 ? in(d) : 0;
#pragma unroll
    for (int i = 0; i < 17; i++) {
      const int jdx = d - i;
      if (jdx >= 0 && jdx < 16) cd -= (long long)qs[i] * MOD_LIMBS[jdx];
    }
    if (d == 31) {  // exactness: the last carry must vanish, otherwise `in` was not a multiple of p
      bad |= a != cd;
      break;
    }
    a = d == 0 ? -(cd >> 16) : (a - cd) >> 16;
    const long long sh = a + (1LL << 29);
    bad |= (sh < 0) | (sh > (1LL << 30));
    out[(size_t)(18 + d) * N] = (u64)(sh & 0xFFFF);
    out[(size_t)(49 + d) * N] = (u64)((sh >> 16) & 0xFFFF);
  }
  if (bad) atomicCAS(err, 0, BN254S_E_INTERNAL);
}

// stores the coefficients to the calling lane's LDS slots and calls the function above; `slots` = &buffer[0][lane] of a
// __shared__ long long buffer[31][MZ_LANES]
__device__ __forceinline__ void gen_modulus_zero(const long long* in, long long* slots, u64* __restrict__ trace, size_t N, size_t row,
                                                 int col0, int* err) {
  mz_lds_t* l = (mz_lds_t*)slots;
#pragma unroll
  for (int i = 0; i < 31; i++) l[i * MZ_LANES] = in[i];
  gen_modulus_zero_lds(l, trace, N, row, col0, err);
}

// ---- row-kernel helpers: packed limbs, limb products accumulated in LDS -------------------------------------------------------
// The sixteen 16-bit limbs of a coordinate are kept two to a register (P16).  The 31-coefficient polynomial of a witness block lives
// in the lane's LDS slots (the argument buffer of gen_modulus_zero_lds, trace_common.h); a limb product is a called function that
// unpacks its operands, forms the 31 coefficients in its own registers and adds them to the slots.  The row kernel itself then
// holds little more than its packed coordinates.  (Round 2, k_g2_rows: fourteen unpacked arrays and two or three polynomials:
// 256 + 256 registers and 784 B of scratch per lane.)
struct P16 {
  u32 w[8];
};
__device__ __forceinline__ P16 p16_pack(const fq& m) {
  const fqw cw = fq_to_canonical(m);
  P16 p;
#pragma unroll
  for (int i = 0; i < 8; i++) p.w[i] = (u32)(cw.l[i >> 1] >> (32 * (i & 1)));
  return p;
}
__device__ __forceinline__ void p16_unpack(const P16& p, int* l) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    l[2 * i] = (int)(p.w[i] & 0xFFFF);
    l[2 * i + 1] = (int)(p.w[i] >> 16);
  }
}
// slots += coef * (x - [mode 1] sub) * (y - [mode 2] sub) as limb polynomials; cm = coef * 4 + mode.  The packed operands travel as
// vector-typed arguments (registers; a struct argument of this size goes through the stack, i.e. scratch memory).
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
static __device__ __noinline__ void mac_lds_v(mz_lds_t* slots, int cm, u32x4 xa, u32x4 xb, u32x4 ya, u32x4 yb, u32x4 sa, u32x4 sb) {
  const int mode = cm & 3, coef = cm >> 2;
  int a[16], b[16], t[16];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[2 * i] = (int)(xa[i] & 0xFFFF);
    a[2 * i + 1] = (int)(xa[i] >> 16);
    a[8 + 2 * i] = (int)(xb[i] & 0xFFFF);
    a[8 + 2 * i + 1] = (int)(xb[i] >> 16);
    b[2 * i] = (int)(ya[i] & 0xFFFF);
    b[2 * i + 1] = (int)(ya[i] >> 16);
    b[8 + 2 * i] = (int)(yb[i] & 0xFFFF);
    b[8 + 2 * i + 1] = (int)(yb[i] >> 16);
    t[2 * i] = (int)(sa[i] & 0xFFFF);
    t[2 * i + 1] = (int)(sa[i] >> 16);
    t[8 + 2 * i] = (int)(sb[i] & 0xFFFF);
    t[8 + 2 * i + 1] = (int)(sb[i] >> 16);
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    a[i] -= mode == 1 ? t[i] : 0;
    b[i] -= mode == 2 ? t[i] : 0;
    a[i] *= coef;
  }
  long long out[31];
#pragma unroll
  for (int i = 0; i < 31; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) out[i + j] += (long long)a[i] * b[j];
#pragma unroll
  for (int i = 0; i < 31; i++) slots[i * MZ_LANES] += out[i];
}
__device__ __forceinline__ void mac_lds(mz_lds_t* slots, int cm, const P16& x, const P16& y, const P16& sub) {
  mac_lds_v(slots, cm, u32x4{x.w[0], x.w[1], x.w[2], x.w[3]}, u32x4{x.w[4], x.w[5], x.w[6], x.w[7]}, u32x4{y.w[0], y.w[1], y.w[2], y.w[3]},
            u32x4{y.w[4], y.w[5], y.w[6], y.w[7]}, u32x4{sub.w[0], sub.w[1], sub.w[2], sub.w[3]},
            u32x4{sub.w[4], sub.w[5], sub.w[6], sub.w[7]});
}
__device__ __forceinline__ void lin_lds(mz_lds_t* slots, int coef, const P16& x) {
  int a[16];
  p16_unpack(x, a);
#pragma unroll
  for (int i = 0; i < 16; i++) slots[i * MZ_LANES] += (long long)(coef * a[i]);
}
__device__ __forceinline__ void zero_lds(mz_lds_t* slots) {
#pragma unroll
  for (int i = 0; i < 31; i++) slots[i * MZ_LANES] = 0;
}

__device__ __forceinline__ void pol_mul16(const int* a, const int* b, long long* out /*31*/) {
#pragma unroll
  for (int i = 0; i < 31; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) out[i + j] += (long long)a[i] * b[j];
}


// shared kernels live in trace_g1.hip
void launch_fq_batch_inv(const u64* in, u64* out, size_t count, hipStream_t st);
void launch_round_flag_table(u64* tbl, hipStream_t st);
// histogram of columns [rc_begin, rc_end) -> frequency column, and the range counter column
void launch_range_columns(u64* trace, size_t N, int rc_begin, int rc_end, int freq_col, int range_col, u32* hist, int* err,
                          hipStream_t st);
