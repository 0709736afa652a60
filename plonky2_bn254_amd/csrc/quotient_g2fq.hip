// Quotient-polynomial evaluation for the G2 scalar-mul STARK and the Fq-exp STARK.
//
// AIRs restated (emission order preserved, see quotient_common.h):
//   G2: eval_g2_add  src/starks/curves/g2/add.rs:132-196 with eval_is_ext_modulus_zero / eval_ext_modulus_zero
//       (g2/ext/is_modulus_zero.rs:48-75, ext/modulus_zero.rs:48-58) and the Fq2 limb products of g2/ext/mul.rs:14-32,
//       followed by the schedule of src/starks/curves/g2/scalar_mul_stark.rs:257-339           (396 + 1297 constraints)
//   Fq: eval_fq_mul  src/starks/fields/mul.rs:43-57, then src/starks/fields/exp_stark.rs:241-327   (33 + 737 constraints)
#include "quotient_sched.h"
#include "quotient.h"

// eval_modulus_zero blocks of eval_g2_add: c0/c1 of {delta_x is-zero, lambda (x != ), lambda (x ==), x, y}
static constexpr int G2_MZ_E0[10] = {1, 50, 100, 133, 166, 199, 264, 297, 330, 363};
static constexpr int FQ_MZ_E0[1] = {0};

// coefficient i of the Fq2 limb product (x0 + x1 u)(y0 + y1 u), u^2 = -1
__device__ __forceinline__ u64 econv_c0(const u64* x0, const u64* x1, const u64* y0, const u64* y1, int i) {
  return gl_sub(conv16(x0, y0, i), conv16(x1, y1, i));
}
__device__ __forceinline__ u64 econv_c1(const u64* x0, const u64* x1, const u64* y0, const u64* y1, int i) {
  return gl_add(conv16(x0, y1, i), conv16(x1, y0, i));
}

#define MZB(blk, auxcol, FILTER, FN, ...) \
  mz_block(tl, M2, j, (auxcol), A, G2_MZ_E0[blk], (blk), (FILTER), FN, tot0, tot1, ##__VA_ARGS__)

__global__ __launch_bounds__(256, 2) void k_quotient_g2_sched(QArgs A) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.count) return;
  schedule_part<G2L, false>(A, j, q_next(A, j), 396, 6);
}

// Parts of eval_g2_add: {0, 5: is-zero witnesses of delta_x.c0 / .c1, 1: lambda (x !=), 2: lambda (x ==) + a.y == b.y, 3: x, 4: y}.
template <int part>
__global__ __launch_bounds__(256, 2) void k_quotient_g2_add(QArgs A) {
  typedef G2L L;
  const size_t M2 = A.stride;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.count) return;
  const u64* __restrict__ tl = A.tl;
  const u64* __restrict__ W0 = A.W;
  const u64* __restrict__ W1 = A.W + A.K;
  const u32* __restrict__ W3 = (const u32*)(A.W + 2 * (size_t)A.K);  // cut weights (accw_mad)
  u64 tot0 = 0, tot1 = 0;
  int e = 0;
  const u64 filter = TL(L::FILTER);
  const int AUX = L::AUX;
  u64 is_x_eq_filter = 0;
  if constexpr (part == 1 || part == 2) is_x_eq_filter = TL(AUX + G2_AUX_IS_X_EQ_FILTER);
  // An eval_modulus_zero block is linear in its input polynomial, so the terms that are plain trace columns (b.y - a.y, the
  // x coordinates, ...) are summed into the block's accumulator first, one value at a time; only the operands of the Fq2 limb
  // products are ever held as arrays (four of them: 128 registers).  Round 1 held up to eight arrays and spilled.
  auto seed_w = [&](Acc2& sd, int blk, int i, u64 v) { acc2_mad(sd, v, W0[G2_MZ_E0[blk] + 1 + i], W1[G2_MZ_E0[blk] + 1 + i]); };
  if constexpr (part == 0 || part == 5) {
    // the two is-zero witnesses of delta_x are independent blocks over one coordinate each: a kernel of their own each
    // (parts 0 and 5), two operand arrays live instead of three
    constexpr int h = part == 0 ? 0 : 1;
    u64 dx[16], t0[16];
    const u64 z = TL(AUX + (h ? G2_AUX_IS_C1_ZERO : G2_AUX_IS_C0_ZERO));
    if constexpr (h == 0) {
      EMIT(gl_mul(filter, gl_sub(gl_mul(z, TL(AUX + G2_AUX_IS_C1_ZERO)), TL(AUX + G2_AUX_IS_X_EQ))));  // e = 0
    }
#pragma unroll
    for (int i = 0; i < 16; i++) dx[i] = gl_sub(TL(L::B + 16 * h + i), TL(L::A + 16 * h + i));
    ld16(tl, M2, j, AUX + (h ? G2_AUX_C1_AUX : G2_AUX_C0_AUX), t0);  // inv of delta_x.c0 / .c1
    const u64 c0 = gl_sub(z, 1);
    MZB(h, AUX + (h ? G2_AUX_C1_AUX : G2_AUX_C0_AUX) + 16, filter, [&](int i) __attribute__((always_inline)) { u64 v = conv16(dx, t0, i); return i == 0 ? gl_add(v, c0) : v; });
    e = h ? 83 : 34;
    AccW g;
    accw_init(g);
#pragma unroll
    for (int i = 0; i < 16; i++) accw_mad(g, dx[i], W3 + 8 * (e + i));
    const u64 f = gl_mul(filter, z);
    tot0 = gl_add(tot0, gl_mul(f, acc3_red(g.a0)));
    tot1 = gl_add(tot1, gl_mul(f, acc3_red(g.a1)));
    if constexpr (h == 1) {
      e = 99;
      EMIT(gl_sub(gl_mul(filter, TL(AUX + G2_AUX_IS_X_EQ)), TL(AUX + G2_AUX_IS_X_EQ_FILTER)));
    }
  } else {
    Acc2 sd0, sd1;
    acc2_init(sd0);
    acc2_init(sd1);
    u64 l0[16], l1[16], dx0[16], dx1[16];
    if constexpr (part == 1) {
      // lambda * delta_x - (b.y - a.y) under filter - is_x_eq_filter
#pragma unroll 4
      for (int i = 0; i < 16; i++) seed_w(sd0, 2, i, gl_sub(TL(L::A + 32 + i), TL(L::B + 32 + i)));
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA, l0);
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA + 16, l1);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        dx0[i] = gl_sub(TL(L::B + i), TL(L::A + i));
        dx1[i] = gl_sub(TL(L::B + 16 + i), TL(L::A + 16 + i));
      }
      const u64 f_ne = gl_sub(filter, is_x_eq_filter);
      MZB(2, AUX + G2_AUX_LAMBDA_AUX, f_ne, [&](int i) __attribute__((always_inline)) { return econv_c0(l0, l1, dx0, dx1, i); }, &sd0);
#pragma unroll 4
      for (int i = 0; i < 16; i++) seed_w(sd1, 3, i, gl_sub(TL(L::A + 48 + i), TL(L::B + 48 + i)));
      MZB(3, AUX + G2_AUX_LAMBDA_AUX + 80, f_ne, [&](int i) __attribute__((always_inline)) { return econv_c1(l0, l1, dx0, dx1, i); }, &sd1);
    } else if constexpr (part == 2) {
      // 2 * lambda * a.y - 3 * a.x^2 under is_x_eq_filter, then a.y == b.y
      ld16(tl, M2, j, L::A, dx0);  // a.x: its squares go into the seeds, then the arrays are reused for a.y
      ld16(tl, M2, j, L::A + 16, dx1);
#pragma unroll
      for (int i = 0; i < 31; i++) {
        const u64 xx0 = econv_c0(dx0, dx1, dx0, dx1, i), xx1 = econv_c1(dx0, dx1, dx0, dx1, i);
        seed_w(sd0, 4, i, gl_neg(gl_add(gl_dbl(xx0), xx0)));
        seed_w(sd1, 5, i, gl_neg(gl_add(gl_dbl(xx1), xx1)));
      }
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA, l0);
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA + 16, l1);
      ld16(tl, M2, j, L::A + 32, dx0);  // a.y
      ld16(tl, M2, j, L::A + 48, dx1);
      MZB(4, AUX + G2_AUX_LAMBDA_AUX, is_x_eq_filter, [&](int i) __attribute__((always_inline)) { return gl_dbl(econv_c0(l0, l1, dx0, dx1, i)); }, &sd0);
      MZB(5, AUX + G2_AUX_LAMBDA_AUX + 80, is_x_eq_filter, [&](int i) __attribute__((always_inline)) { return gl_dbl(econv_c1(l0, l1, dx0, dx1, i)); }, &sd1);
      e = 232;
      AccW g;
      accw_init(g);
#pragma unroll
      for (int i = 0; i < 16; i++) accw_mad(g, gl_sub(dx0[i], TL(L::B + 32 + i)), W3 + 8 * (e + i));
#pragma unroll
      for (int i = 0; i < 16; i++) accw_mad(g, gl_sub(dx1[i], TL(L::B + 48 + i)), W3 + 8 * (e + 16 + i));
      tot0 = gl_add(tot0, gl_mul(is_x_eq_filter, acc3_red(g.a0)));
      tot1 = gl_add(tot1, gl_mul(is_x_eq_filter, acc3_red(g.a1)));
    } else if constexpr (part == 3) {
      // lambda^2 - (a.x + b.x + c.x)
#pragma unroll 4
      for (int i = 0; i < 16; i++) {
        seed_w(sd0, 6, i, gl_neg(gl_add(gl_add(TL(L::A + i), TL(L::B + i)), TL(L::C + i))));
        seed_w(sd1, 7, i, gl_neg(gl_add(gl_add(TL(L::A + 16 + i), TL(L::B + 16 + i)), TL(L::C + 16 + i))));
      }
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA, l0);
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA + 16, l1);
      MZB(6, AUX + G2_AUX_X_AUX, filter, [&](int i) __attribute__((always_inline)) { return econv_c0(l0, l1, l0, l1, i); }, &sd0);
      MZB(7, AUX + G2_AUX_X_AUX + 80, filter, [&](int i) __attribute__((always_inline)) { return econv_c1(l0, l1, l0, l1, i); }, &sd1);
    } else {
      // lambda * (c.x - a.x) + c.y + a.y
#pragma unroll 4
      for (int i = 0; i < 16; i++) seed_w(sd0, 8, i, gl_add(TL(L::C + 32 + i), TL(L::A + 32 + i)));
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA, l0);
      ld16(tl, M2, j, AUX + G2_AUX_LAMBDA + 16, l1);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        dx0[i] = gl_sub(TL(L::C + i), TL(L::A + i));
        dx1[i] = gl_sub(TL(L::C + 16 + i), TL(L::A + 16 + i));
      }
      MZB(8, AUX + G2_AUX_Y_AUX, filter, [&](int i) __attribute__((always_inline)) { return econv_c0(l0, l1, dx0, dx1, i); }, &sd0);
#pragma unroll 4
      for (int i = 0; i < 16; i++) seed_w(sd1, 9, i, gl_add(TL(L::C + 48 + i), TL(L::A + 48 + i)));
      MZB(9, AUX + G2_AUX_Y_AUX + 80, filter, [&](int i) __attribute__((always_inline)) { return econv_c1(l0, l1, dx0, dx1, i); }, &sd1);
    }
  }
  store_part(A, part, j, tot0, tot1);
}

__global__ __launch_bounds__(256, 2) void k_quotient_fq_sched(QArgs A) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.count) return;
  schedule_part<FQL, true>(A, j, q_next(A, j), 33, 1);
}
__global__ __launch_bounds__(256) void k_quotient_fq_mul(QArgs A) {
  typedef FQL L;
  const size_t M2 = A.stride;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.count) return;
  const u64* __restrict__ tl = A.tl;
  const u64* __restrict__ W0 = A.W;
  const u64* __restrict__ W1 = A.W + A.K;
  u64 tot0 = 0, tot1 = 0;
  const u64 filter = TL(L::FILTER);
  u64 a[16], b[16], c[16];
  ld16(tl, M2, j, L::A, a);
  ld16(tl, M2, j, L::B, b);
  ld16(tl, M2, j, L::C, c);
  // eval_fq_mul: a*b - c
  mz_block(tl, M2, j, L::AUX, A, FQ_MZ_E0[0], 0, filter,
           [&](int i) __attribute__((always_inline)) {
             u64 v = conv16(a, b, i);
             return i < 16 ? gl_sub(v, c[i < 16 ? i : 0]) : v;
           },
           tot0, tot1);
  store_part(A, 0, j, tot0, tot1);
}

int g2_quotient_mz_blocks(const int** e0) {
  *e0 = G2_MZ_E0;
  return 10;
}
int fq_quotient_mz_blocks(const int** e0) {
  *e0 = FQ_MZ_E0;
  return 1;
}
void g2_quotient_launch(const QArgs& A0, const StarkShape& sh, hipStream_t st) {
  QArgs A = A0;
  A.n_parts = 7;
  const unsigned g = (unsigned)((A.count + 255) / 256);
  k_quotient_g2_add<0><<<g, 256, 0, st>>>(A);
  k_quotient_g2_add<5><<<g, 256, 0, st>>>(A);
  k_quotient_g2_add<1><<<g, 256, 0, st>>>(A);
  k_quotient_g2_add<2><<<g, 256, 0, st>>>(A);
  k_quotient_g2_add<3><<<g, 256, 0, st>>>(A);
  k_quotient_g2_add<4><<<g, 256, 0, st>>>(A);
  k_quotient_g2_sched<<<g, 256, 0, st>>>(A);
  quotient_finish_launch(A, sh, st);
}
void fq_quotient_launch(const QArgs& A0, const StarkShape& sh, hipStream_t st) {
  QArgs A = A0;
  A.n_parts = 2;
  const unsigned g = (unsigned)((A.count + 255) / 256);
  k_quotient_fq_mul<<<g, 256, 0, st>>>(A);
  k_quotient_fq_sched<<<g, 256, 0, st>>>(A);
  quotient_finish_launch(A, sh, st);
}

// loads this translation unit's code object (the HIP runtime defers that to the first launch otherwise)
void quotient_g2fq_module_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_quotient_fq_sched));
}
