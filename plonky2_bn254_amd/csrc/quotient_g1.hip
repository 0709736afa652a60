// Quotient-polynomial evaluation for the G1 scalar-mul STARK (K-quotient of DESIGN.md).
//
// Replaces starky `compute_quotient_polys` / `eval_vanishing_poly` calling the reference's AIR
//   G1ScalarMulStark::eval_packed_generic   src/starks/curves/g1/scalar_mul_stark.rs:226-339
//   eval_g1_add                             src/starks/curves/g1/add.rs:125-185
//   eval_is_modulus_zero / eval_modulus_zero  src/starks/modular/is_modulus_zero.rs:69-84, modulus_zero.rs:163-198
//   eval_round_flags / EvalEq               src/starks/common/round_flags.rs:46-81, common/eq.rs:25-51
// on every point of the 2N-point coset, one lane per point, columns read coalesced from the
// bit-reversed LDE.  Constraint numbering e follows the reference's emission order exactly (see
// quotient_common.h for why).  Within a `eval_modulus_zero` block the 32 limb constraints are linear in the
// witness, so their alpha-weighted sum is re-associated:
//   sum_i w_i*constr_i = qsign * sum_j quot_j * M_j  +  sum_d (lo_d*u_d + hi_d*B*u_d) - off*sum u_d  -  sum_i w_i*input_i
// with M_j = sum_t w_{j+t} m_t and u_d = w_{d+1} - B*w_d depending on alpha only (host-precomputed).
#include "quotient_common.h"
#include "trace_g1.h"
#include "quotient.h"

static constexpr int G1_K_AIR = 1111;
// first constraint index of the five eval_modulus_zero blocks inside eval_g1_add
static constexpr int G1_MZ_E0[5] = {0, 50, 83, 132, 165};

struct G1QArgs {
  const u64* tl;   // trace LDE [781][2N], bit-reversed order
  const u64* al;   // aux LDE [456][2N]
  const u64* W;    // weights [2][K]
  const u64* mzt;  // modulus-zero tables [5][2][80]
  QPointTables pt;
  u64 betas[2], gammas[2];
  u64 zh_inv[2];   // 1/Z_H on coset h
  u64 w_inv;       // w_N^-1 (last element of the subgroup)
  u64* out;        // [2 alphas][2 cosets][N] natural order
  unsigned log_n;
  int K;
};

#define TL(c) tl[(size_t)(c)*M2 + j]
#define TN(c) tl[(size_t)(c)*M2 + jn]

__device__ __forceinline__ void ld16(const u64* __restrict__ tl, size_t M2, size_t j, int col, u64* v) {
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = tl[(size_t)(col + i) * M2 + j];
}
// coefficient i of the limb product A*B (pol_mul_wide), reduced
__device__ __forceinline__ u64 conv16(const u64* A, const u64* B, int i) {
  Acc a;
  acc_init(a);
#pragma unroll
  for (int s = 0; s < 16; s++) {
    int t = i - s;
    if (t >= 0 && t < 16) acc_mad(a, A[s], B[t]);
  }
  return acc_red(a);
}

// One eval_modulus_zero block.  `in(i)` returns coefficient i (0..30) of the input polynomial.
template <class InFn>
__device__ __forceinline__ void mz_block(const u64* __restrict__ tl, size_t M2, size_t j, int auxcol, const u64* __restrict__ w0,
                                         const u64* __restrict__ w1, const u64* __restrict__ T0, const u64* __restrict__ T1,
                                         u64 filter, InFn in, u64& tot0, u64& tot1) {
  Acc2 pos, neg, q;
  acc2_init(pos);
  acc2_init(neg);
  acc2_init(q);
  const u64 iqp = TL(auxcol + MZ_IQP);
  acc2_mad(pos, gl_sub(gl_mul(iqp, iqp), iqp), w0[0], w1[0]);
  const u64 qsign = gl_sub(gl_dbl(iqp), 1);
#pragma unroll
  for (int jj = 0; jj < 17; jj++) acc2_mad(q, TL(auxcol + MZ_QUOT + jj), T0[jj], T1[jj]);
#pragma unroll
  for (int d = 0; d < 31; d++) {
    acc2_mad(pos, TL(auxcol + MZ_LO + d), T0[17 + d], T1[17 + d]);
    acc2_mad(pos, TL(auxcol + MZ_HI + d), T0[48 + d], T1[48 + d]);
  }
#pragma unroll
  for (int i = 0; i < 31; i++) acc2_mad(neg, in(i), w0[1 + i], w1[1 + i]);
  u64 s0 = gl_add(acc_red(pos.a0), gl_mul(qsign, acc_red(q.a0)));
  u64 s1 = gl_add(acc_red(pos.a1), gl_mul(qsign, acc_red(q.a1)));
  s0 = gl_sub(gl_sub(s0, T0[79]), acc_red(neg.a0));
  s1 = gl_sub(gl_sub(s1, T1[79]), acc_red(neg.a1));
  tot0 = gl_add(tot0, gl_mul(filter, s0));
  tot1 = gl_add(tot1, gl_mul(filter, s1));
}

// sum_i (a[i] - b[i]) * w[e+i] for both alphas, times `filter`
#define EQ_GROUP(filter, n, AEXPR, BEXPR)                         \
  {                                                               \
    Acc2 g_;                                                      \
    acc2_init(g_);                                                \
    for (int i = 0; i < (n); i++) {                               \
      acc2_mad(g_, gl_sub((AEXPR), (BEXPR)), W0[e + i], W1[e + i]); \
    }                                                             \
    e += (n);                                                     \
    u64 f_ = (filter);                                            \
    tot0 = gl_add(tot0, gl_mul(f_, acc_red(g_.a0)));              \
    tot1 = gl_add(tot1, gl_mul(f_, acc_red(g_.a1)));              \
  }
#define EMIT(c)                                  \
  {                                              \
    u64 c_ = (c);                                \
    tot0 = gl_add(tot0, gl_mul(c_, W0[e]));      \
    tot1 = gl_add(tot1, gl_mul(c_, W1[e]));      \
    e += 1;                                      \
  }

__global__ __launch_bounds__(256) void k_quotient_g1(G1QArgs A, StarkShape sh) {
  const unsigned log_n = A.log_n;
  const size_t N = (size_t)1 << log_n, M2 = 2 * N;
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M2) return;
  const size_t jn = next_position(j, log_n);
  const u64* __restrict__ tl = A.tl;
  const u64* __restrict__ W0 = A.W;
  const u64* __restrict__ W1 = A.W + A.K;
  u64 tot0 = 0, tot1 = 0;
  int e = 0;

  const u64 filter = TL(G1_COL_FILTER);
  // ---- eval_g1_add (add.rs:125-185) -----------------------------------------------------------------------
  {
    const int AUX = G1_COL_AUX;
    u64 ax[16], bx[16], dx[16], lam[16], t16[16];
    ld16(tl, M2, j, G1_COL_A, ax);
    ld16(tl, M2, j, G1_COL_B, bx);
#pragma unroll
    for (int i = 0; i < 16; i++) dx[i] = gl_sub(bx[i], ax[i]);
    ld16(tl, M2, j, AUX + G1_AUX_LAMBDA, lam);
    const u64 is_x_eq = TL(AUX + G1_AUX_IS_X_EQ);
    const u64 is_x_eq_filter = TL(AUX + G1_AUX_IS_X_EQ_FILTER);
    // block 0: delta_x * inv - 1 + is_x_eq   (eval_is_modulus_zero)
    {
      ld16(tl, M2, j, AUX + G1_AUX_IS_X_EQ_AUX, t16);  // inv limbs
      const u64 c0 = gl_sub(is_x_eq, 1);
      mz_block(tl, M2, j, AUX + G1_AUX_IS_X_EQ_AUX + 16, W0 + G1_MZ_E0[0], W1 + G1_MZ_E0[0], A.mzt + 0 * 160, A.mzt + 0 * 160 + 80,
               filter, [&](int i) { u64 v = conv16(dx, t16, i); return i == 0 ? gl_add(v, c0) : v; }, tot0, tot1);
      e = 33;
      // filter * (delta_x[i] * is_x_eq)
      Acc2 g;
      acc2_init(g);
#pragma unroll
      for (int i = 0; i < 16; i++) acc2_mad(g, dx[i], W0[e + i], W1[e + i]);
      e += 16;
      u64 f = gl_mul(filter, is_x_eq);
      tot0 = gl_add(tot0, gl_mul(f, acc_red(g.a0)));
      tot1 = gl_add(tot1, gl_mul(f, acc_red(g.a1)));
    }
    EMIT(gl_sub(gl_mul(filter, is_x_eq), is_x_eq_filter));  // e = 49
    const u64 is_not_eq_filter = gl_sub(filter, is_x_eq_filter);
    u64 ay[16];
    ld16(tl, M2, j, G1_COL_A + 16, ay);
    // block 1: lambda*delta_x - (b.y - a.y)
    {
      ld16(tl, M2, j, G1_COL_B + 16, t16);  // b.y
      mz_block(tl, M2, j, AUX + G1_AUX_LAMBDA_AUX, W0 + G1_MZ_E0[1], W1 + G1_MZ_E0[1], A.mzt + 1 * 160, A.mzt + 1 * 160 + 80,
               is_not_eq_filter,
               [&](int i) {
                 u64 v = conv16(lam, dx, i);
                 return i < 16 ? gl_sub(v, gl_sub(t16[i < 16 ? i : 0], ay[i < 16 ? i : 0])) : v;
               },
               tot0, tot1);
    }
    // block 2: 2*lambda*a.y - 3*a.x^2
    mz_block(tl, M2, j, AUX + G1_AUX_LAMBDA_AUX, W0 + G1_MZ_E0[2], W1 + G1_MZ_E0[2], A.mzt + 2 * 160, A.mzt + 2 * 160 + 80,
             is_x_eq_filter,
             [&](int i) {
               u64 ly = conv16(lam, ay, i), xx = conv16(ax, ax, i);
               return gl_sub(gl_dbl(ly), gl_add(gl_dbl(xx), xx));
             },
             tot0, tot1);
    e = 116;
    // a.y == b.y under is_x_eq_filter
    {
      ld16(tl, M2, j, G1_COL_B + 16, t16);
      Acc2 g;
      acc2_init(g);
#pragma unroll
      for (int i = 0; i < 16; i++) acc2_mad(g, gl_sub(ay[i], t16[i]), W0[e + i], W1[e + i]);
      e += 16;
      tot0 = gl_add(tot0, gl_mul(is_x_eq_filter, acc_red(g.a0)));
      tot1 = gl_add(tot1, gl_mul(is_x_eq_filter, acc_red(g.a1)));
    }
    // block 3: lambda^2 - (a.x + b.x + c.x)
    u64 cx[16];
    ld16(tl, M2, j, G1_COL_C, cx);
    mz_block(tl, M2, j, AUX + G1_AUX_X_AUX, W0 + G1_MZ_E0[3], W1 + G1_MZ_E0[3], A.mzt + 3 * 160, A.mzt + 3 * 160 + 80, filter,
             [&](int i) {
               u64 v = conv16(lam, lam, i);
               return i < 16 ? gl_sub(v, gl_add(gl_add(ax[i < 16 ? i : 0], bx[i < 16 ? i : 0]), cx[i < 16 ? i : 0])) : v;
             },
             tot0, tot1);
    // block 4: lambda*(c.x - a.x) + c.y + a.y
    {
#pragma unroll
      for (int i = 0; i < 16; i++) dx[i] = gl_sub(cx[i], ax[i]);  // reuse dx as c.x - a.x
      ld16(tl, M2, j, G1_COL_C + 16, t16);                          // c.y
      mz_block(tl, M2, j, AUX + G1_AUX_Y_AUX, W0 + G1_MZ_E0[4], W1 + G1_MZ_E0[4], A.mzt + 4 * 160, A.mzt + 4 * 160 + 80, filter,
               [&](int i) {
                 u64 v = conv16(lam, dx, i);
                 return i < 16 ? gl_add(v, gl_add(t16[i < 16 ? i : 0], ay[i < 16 ? i : 0])) : v;
               },
               tot0, tot1);
    }
    e = 198;
  }

  // ---- eval_packed_generic body (scalar_mul_stark.rs:257-339) ------------------------------------------
  const u64 is_first = TL(G1_COL_FLAGS + 0), is_last = TL(G1_COL_FLAGS + 1);
  const u64 n_filter = TN(G1_COL_FILTER), n_is_last = TN(G1_COL_FLAGS + 1);
  const u64 is_not_last_round = gl_sub(filter, is_last);
  const u64 is_next_not_last_round = gl_sub(n_filter, n_is_last);
  const u64 is_adding = TL(G1_COL_IS_ADDING), idnl = TL(G1_COL_IDNL);
  const u64 n_is_adding = TN(G1_COL_IS_ADDING), n_idnl = TN(G1_COL_IDNL);
  const u64 bit0 = TL(G1_COL_BITS), n_bit0 = TN(G1_COL_BITS);

  EMIT(gl_mul(is_first, gl_sub(is_adding, 1)));                                          // 198
  EQ_GROUP(is_first, 32, TL(G1_COL_DOUBLE + i), TL(G1_COL_B + i));                       // 199
  EQ_GROUP(gl_mul(bit0, is_first), 32, TL(G1_COL_SUM + i), TL(G1_COL_C + i));            // 231
  EQ_GROUP(gl_mul(gl_sub(1, bit0), is_first), 32, TL(G1_COL_SUM + i), TL(G1_COL_A + i)); // 263
  // doubling step -> addition step
  EQ_GROUP(idnl, 32, TN(G1_COL_A + i), TL(G1_COL_SUM + i));                              // 295
  EQ_GROUP(idnl, 32, TN(G1_COL_B + i), TL(G1_COL_DOUBLE + i));                           // 327
  EQ_GROUP(gl_mul(n_bit0, idnl), 32, TN(G1_COL_SUM + i), TN(G1_COL_C + i));              // 359
  EQ_GROUP(gl_mul(gl_sub(1, n_bit0), idnl), 32, TN(G1_COL_SUM + i), TN(G1_COL_A + i));   // 391
  EQ_GROUP(idnl, 32, TN(G1_COL_DOUBLE + i), TL(G1_COL_DOUBLE + i));                      // 423
  EMIT(gl_mul(idnl, gl_sub(n_is_adding, 1)));                                            // 455
  EMIT(gl_mul(idnl, n_idnl));                                                            // 456
  EQ_GROUP(idnl, 256, TN(G1_COL_BITS + i), TL(G1_COL_BITS + ((i + 1) & 255)));           // 457
  // addition step -> doubling step
  EQ_GROUP(is_adding, 32, TN(G1_COL_A + i), TL(G1_COL_DOUBLE + i));                      // 713
  EQ_GROUP(is_adding, 32, TN(G1_COL_B + i), TL(G1_COL_DOUBLE + i));                      // 745
  EQ_GROUP(is_adding, 32, TN(G1_COL_SUM + i), TL(G1_COL_SUM + i));                       // 777
  EQ_GROUP(is_adding, 32, TN(G1_COL_DOUBLE + i), TN(G1_COL_C + i));                      // 809
  EMIT(gl_mul(is_adding, n_is_adding));                                                  // 841
  EMIT(gl_mul(is_adding, gl_sub(n_idnl, is_next_not_last_round)));                       // 842
  EQ_GROUP(is_adding, 256, TN(G1_COL_BITS + i), TL(G1_COL_BITS + i));                    // 843
  // eval_round_flags (round_flags.rs:46-81), period 512
  {
    const u64 counter = TL(G1_COL_FLAGS + 2), inv_c = TL(G1_COL_FLAGS + 3), inv_cp = TL(G1_COL_FLAGS + 4);
    const u64 n_counter = TN(G1_COL_FLAGS + 2);
    const u64 not_filter = gl_sub(1, filter);
    EMIT(gl_mul(not_filter, is_first));
    EMIT(gl_mul(not_filter, is_last));
    EMIT(gl_mul(filter, gl_sub(gl_mul(counter, inv_c), gl_sub(1, is_first))));
    EMIT(gl_mul(gl_mul(filter, counter), is_first));
    const u64 cp = gl_sub(counter, 511);
    EMIT(gl_mul(filter, gl_sub(gl_mul(cp, inv_cp), gl_sub(1, is_last))));
    EMIT(gl_mul(gl_mul(filter, cp), is_last));
    EMIT(gl_mul(gl_mul(filter, gl_sub(1, is_last)), gl_sub(gl_sub(n_counter, counter), 1)));
    EMIT(gl_mul(gl_mul(filter, is_last), n_counter));
  }
  EMIT(gl_mul(is_not_last_round, gl_sub(TN(G1_COL_TIMESTAMP), TL(G1_COL_TIMESTAMP))));   // 1107
  EMIT(gl_mul(is_not_last_round, gl_sub(n_filter, filter)));                             // 1108
  const u64 x = A.pt.x[j], lfirst = A.pt.lfirst[j], llast = A.pt.llast[j];
  const u64 z_last = gl_sub(x, A.w_inv);
  {
    const u64 rc = TL(G1_COL_RANGE), diff = gl_sub(TN(G1_COL_RANGE), rc);
    EMIT(gl_mul(gl_sub(gl_mul(diff, diff), diff), z_last));                              // 1109 transition
    EMIT(gl_mul(gl_sub(rc, 65535), llast));                                              // 1110 last row
  }
  // ---- lookups + CTLs --------------------------------------------------------------------------------------
  lookup_and_ctl_constraints(sh, tl, A.al, M2, j, jn, W0, W1, e, A.betas, A.gammas, lfirst, llast, z_last, tot0, tot1);

  // divide by Z_H and store in natural order of the coset
  const size_t h = j >> log_n;
  const u32 k = bitrev32((u32)(j & (N - 1)), log_n);
  A.out[(0 * 2 + h) * N + k] = gl_mul(tot0, A.zh_inv[h]);
  A.out[(1 * 2 + h) * N + k] = gl_mul(tot1, A.zh_inv[h]);
}

// ---- per-context point tables ---------------------------------------------------------------------------------
__global__ void k_point_tables(u64* x, u64* lfirst, u64* llast, unsigned log_n, u64 w_n, u64 w_2n) {
  const size_t N = (size_t)1 << log_n;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * N) return;
  size_t h = j >> log_n;
  u32 k = bitrev32((u32)(j & (N - 1)), log_n);
  u64 shift = h ? gl_mul(GL_GEN, w_2n) : GL_GEN;
  u64 xv = gl_mul(shift, gl_pow(w_n, k));
  u64 gn = gl_pow(GL_GEN, N);            // x^N = g^N * (-1)^h
  u64 zh = gl_sub(h ? gl_neg(gn) : gn, 1);
  u64 ninv = gl_inv((u64)N % GL_P);
  u64 winv = gl_inv(w_n);
  x[j] = xv;
  lfirst[j] = gl_mul(gl_mul(zh, ninv), gl_inv(gl_sub(xv, 1)));
  llast[j] = gl_mul(gl_mul(gl_mul(zh, ninv), winv), gl_inv(gl_sub(xv, winv)));
}

void quotient_point_tables(u64* d_x, u64* d_lfirst, u64* d_llast, unsigned log_n, hipStream_t st) {
  size_t M2 = (size_t)2 << log_n;
  k_point_tables<<<(unsigned)((M2 + 255) / 256), 256, 0, st>>>(d_x, d_lfirst, d_llast, log_n, gl_root_of_unity(log_n),
                                                             gl_root_of_unity(log_n + 1));
}

// ---- host: alpha-dependent tables ---------------------------------------------------------------------------
// W[a][e] = alpha_a^(K-1-e); mzt[blk][a][80] = {M_j (17), u_d (31), B*u_d (31), off*sum(u_d)}
void g1_quotient_host_tables(const StarkShape& sh, const u64 alphas[2], std::vector<u64>& W, std::vector<u64>& mzt) {
  const int K = sh.n_total_constraints();
  W.assign(2 * (size_t)K, 0);
  for (int a = 0; a < 2; a++) {
    u64 v = 1;
    for (int e = K - 1; e >= 0; e--) {
      W[(size_t)a * K + e] = v;
      v = gl_mul(v, alphas[a]);
    }
  }
  static const u64 MOD[16] = {64839, 55420, 35862, 15392, 51853, 26737, 27281, 38785,
                              22621, 33153, 17846, 47184, 41001, 57649, 20082, 12388};
  const u64 B = 1ULL << 16, OFF = 1ULL << 29;
  mzt.assign(5 * 2 * 80, 0);
  for (int blk = 0; blk < 5; blk++)
    for (int a = 0; a < 2; a++) {
      const u64* w = &W[(size_t)a * K + G1_MZ_E0[blk] + 1];  // w[i], i = 0..31: weight of constr_i
      u64* T = &mzt[(size_t)(blk * 2 + a) * 80];
      for (int jj = 0; jj < 17; jj++) {
        u64 s = 0;
        for (int t = 0; t < 16; t++)
          if (jj + t < 32) s = gl_add(s, gl_mul(w[jj + t], MOD[t]));
        T[jj] = s;
      }
      u64 usum = 0;
      for (int d = 0; d < 31; d++) {
        u64 u = gl_sub(w[d + 1], gl_mul(B, w[d]));
        T[17 + d] = u;
        T[48 + d] = gl_mul(B, u);
        usum = gl_add(usum, u);
      }
      T[79] = gl_mul(OFF, usum);
    }
}

void g1_quotient_launch(const StarkShape& sh, const u64* d_tl, const u64* d_al, const u64* d_W, const u64* d_mzt,
                        const QPointTables& pt, const u64 betas[2], const u64 gammas[2], unsigned log_n, u64* d_out,
                        hipStream_t st) {
  G1QArgs A;
  A.tl = d_tl;
  A.al = d_al;
  A.W = d_W;
  A.mzt = d_mzt;
  A.pt = pt;
  for (int i = 0; i < 2; i++) {
    A.betas[i] = betas[i];
    A.gammas[i] = gammas[i];
  }
  const size_t N = (size_t)1 << log_n;
  u64 gn = gl_pow(GL_GEN, N);
  A.zh_inv[0] = gl_inv(gl_sub(gn, 1));
  A.zh_inv[1] = gl_inv(gl_sub(gl_neg(gn), 1));
  A.w_inv = gl_inv(gl_root_of_unity(log_n));
  A.out = d_out;
  A.log_n = log_n;
  A.K = sh.n_total_constraints();
  size_t M2 = 2 * N;
  k_quotient_g1<<<(unsigned)((M2 + 255) / 256), 256, 0, st>>>(A, sh);
}
