// Constraint-evaluation pieces shared by the three quotient kernels: kernel arguments, the
// eval_modulus_zero block with re-associated weights, and the 512-row schedule constraints
// (everything in eval_packed_generic after the add / mul check: reference
// src/starks/curves/g1/scalar_mul_stark.rs:257-339 == curves/g2/scalar_mul_stark.rs, src/starks/fields/exp_stark.rs:241-327).
#pragma once
#include <vector>
#include "quotient_common.h"
#include "layout.h"

// ModulusZeroAux block: [is_quot_positive, quot_abs[17], aux_lo[31], aux_hi[31]] (modulus_zero.rs:66-73)
#ifndef MZ_CONSTS
#define MZ_CONSTS
static constexpr int QMZ_IQP = 0, QMZ_QUOT = 1, QMZ_LO = 18, QMZ_HI = 49;
#endif

struct QArgs {
  const u64* tl;   // trace LDE [W][2N], bit-reversed order
  const u64* al;   // aux LDE [A][2N]
  const u64* W;    // weights [2][K]
  const u64* mzt;  // modulus-zero tables [n_blocks][2][80]
  QPointTables pt;
  u64 betas[2], gammas[2];
  u64 zh_inv[2];   // 1/Z_H on coset h
  u64 w_inv;       // w_N^-1 (last element of the subgroup)
  u64* out;        // [2 alphas][2 cosets][N] natural order
  u64* part;       // partial sums [n_parts][2 alphas][2N], bit-reversed order
  int n_parts;
  unsigned log_n;
  int K;
  // Which points a launch covers.  Whole domain (default): count = stride = 2N, point j is leaf j of the bit-reversed LDE, its
  // next row sits at next_position(j).  Window (natural = 1; proofs whose LDEs do not fit the device, prover.hip "stream"): tl / al
  // / part / pt hold `count` + 1 consecutive rows k0 .. k0 + count of coset `hsel` in NATURAL order with column stride `stride`
  // (the extra row serves the transition constraints of the last one), point j is row k0 + j and its next row is j + 1.
  size_t stride, count, k0;
  u32 natural, hsel;
};
// device tables: W = [2][K] weights + [K][8 u32] cut weights; mzt = [10 blocks][2][80] + [10][80][8 u32]
static constexpr size_t QUOTIENT_MZT3_OFF = 10 * 2 * 80, QUOTIENT_MZT_WORDS = 10 * 2 * 80 + 10 * 80 * 4;
static inline size_t QUOTIENT_W_WORDS(int K) { return 6 * (size_t)K; }
__device__ __forceinline__ size_t q_next(const QArgs& A, size_t j) { return A.natural ? j + 1 : next_position(j, A.log_n); }

#define TL(c) tl[(size_t)(c)*M2 + j]
#define TN(c) tl[(size_t)(c)*M2 + jn]

__device__ __forceinline__ void ld16(const u64* __restrict__ tl, size_t M2, size_t j, int col, u64* v) {
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = tl[(size_t)(col + i) * M2 + j];
}
// coefficient i of the limb product A*B (pol_mul_wide), reduced
// FULL: forced complete unrolling (G1: three or four operand arrays in registers, no scratch at two waves per SIMD); otherwise
// the unrolling is left to the compiler's heuristics (G2 / Fq: up to four Fq2 operand arrays; mz_block keeps the coefficients
// apart with scheduling barriers, which is what holds these kernels inside 256 registers)
template <bool FULL = false>
__device__ __forceinline__ u64 conv16(const u64* A, const u64* B, int i) {
  Acc a;
  acc_init(a);
  if constexpr (FULL) {
#pragma clang loop unroll(full)
    for (int s = 0; s < 16; s++) {
      int t = i - s;
      if (t >= 0 && t < 16) acc_mad(a, A[s], B[t]);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; s++) {
      int t = i - s;
      if (t >= 0 && t < 16) acc_mad(a, A[s], B[t]);
    }
  }
  return acc_red(a);
}

// One eval_modulus_zero block: first constraint index e0, tables of block `blk`.  `in(i)` returns coefficient i (0..30) of the
// input polynomial.
template <bool FULL = false, class InFn>
__device__ __forceinline__ void mz_block(const u64* __restrict__ tl, size_t M2, size_t j, int auxcol, const QArgs& A, int e0, int blk,
                                         u64 filter, InFn in, u64& tot0, u64& tot1, const Acc2* seed = nullptr) {
  const u64* __restrict__ w0 = A.W + e0;
  const u64* __restrict__ w1 = A.W + A.K + e0;
  const u64* __restrict__ T0 = A.mzt + (size_t)blk * 160;
  const u64* __restrict__ T1 = T0 + 80;
  const u32* __restrict__ w3 = (const u32*)(A.W + 2 * (size_t)A.K) + 8 * (size_t)e0;                   // cut weights (accw_mad)
  const u32* __restrict__ T3 = (const u32*)(A.mzt + QUOTIENT_MZT3_OFF) + 8 * (size_t)(blk * 80);
  // The input polynomial first (its limb products need the most registers), reduced to two field elements before the witness
  // columns of the block are summed.  `seed`: the weighted sum of a part of the input polynomial that the caller accumulated
  // beforehand (the block is linear in its input: terms that need other trace columns than the limb products are summed
  // first, with no operand arrays live).
  u64 n0, n1;
  {
    // (the 128-bit accumulator here: ten registers less than the carry-free one, where the operand arrays leave none)
    Acc2 neg;
    if (seed) neg = *seed;
    else acc2_init(neg);
    if constexpr (FULL) {
#pragma clang loop unroll(full)
      for (int i = 0; i < 31; i++) acc2_mad(neg, in(i), w0[1 + i], w1[1 + i]);
    } else {
#pragma unroll
      for (int i = 0; i < 31; i++) {
        acc2_mad(neg, in(i), w0[1 + i], w1[1 + i]);
        __builtin_amdgcn_sched_barrier(0);  // one coefficient at a time: interleaved, their partial sums do not fit the registers
      }
    }
    n0 = acc_red(neg.a0);
    n1 = acc_red(neg.a1);
  }
  __builtin_amdgcn_sched_barrier(0);
  // the witness columns of the block, weights cut in limbs (accw_mad); one accumulator pair live at a time
  const u64 iqp = TL(auxcol + QMZ_IQP);
  const u64 qsign = gl_sub(gl_dbl(iqp), 1);
  u64 s0, s1;
  {
    AccW q;
    accw_init(q);
#pragma unroll 1
    for (int jj = 0; jj < 17; jj++) accw_mad(q, TL(auxcol + QMZ_QUOT + jj), T3 + 8 * jj);
    s0 = gl_mul(qsign, acc3_red(q.a0));
    s1 = gl_mul(qsign, acc3_red(q.a1));
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    AccW pos;
    accw_init(pos);
    accw_mad(pos, gl_sub(gl_mul(iqp, iqp), iqp), w3);
#pragma unroll 1
    for (int d = 0; d < 31; d++) {
      accw_mad(pos, TL(auxcol + QMZ_LO + d), T3 + 8 * (17 + d));
      accw_mad(pos, TL(auxcol + QMZ_HI + d), T3 + 8 * (48 + d));
    }
    s0 = gl_add(s0, acc3_red(pos.a0));
    s1 = gl_add(s1, acc3_red(pos.a1));
  }
  s0 = gl_sub(gl_sub(s0, T0[79]), n0);
  s1 = gl_sub(gl_sub(s1, T1[79]), n1);
  tot0 = gl_add(tot0, gl_mul(filter, s0));
  tot1 = gl_add(tot1, gl_mul(filter, s1));
  asm volatile("" : "+v"(tot0), "+v"(tot1));  // (as in EQ_GROUP: the block's sums are folded here, not where the totals are stored)
}

// sum_i (a[i] - b[i]) * w[e+i] for both alphas, times `filter` (n: a multiple of 16; sixteen differences are formed first so that
// their thirty-two loads are in flight together, then accumulated with the cut weights)
#define EQ_GROUP(filter, n, AEXPR, BEXPR)                         \
  {                                                               \
    static_assert((n) % 16 == 0, "EQ_GROUP: n % 16");             \
    AccW g_;                                                      \
    accw_init(g_);                                                \
    constexpr int CH_ = 16;                                       \
    _Pragma("unroll 1") for (int i0_ = 0; i0_ < (n); i0_ += CH_) { \
      u64 d_[CH_];                                                \
      _Pragma("unroll") for (int k_ = 0; k_ < CH_; k_++) {        \
        const int i = i0_ + k_;                                   \
        d_[k_] = gl_sub((AEXPR), (BEXPR));                        \
      }                                                           \
      _Pragma("unroll") for (int k_ = 0; k_ < CH_; k_++) accw_mad(g_, d_[k_], W3 + 8 * (e + i0_ + k_)); \
    }                                                             \
    e += (n);                                                     \
    u64 f_ = (filter);                                            \
    tot0 = gl_add(tot0, gl_mul(f_, acc3_red(g_.a0)));             \
    tot1 = gl_add(tot1, gl_mul(f_, acc3_red(g_.a1)));             \
    asm volatile("" : "+v"(tot0), "+v"(tot1)); /* reduce here: sunk to the end, sixteen groups' column sums stay live */ \
  }
#define EMIT(c)                                  \
  {                                              \
    u64 c_ = (c);                                \
    tot0 = gl_add(tot0, gl_mul(c_, W0[e]));      \
    tot1 = gl_add(tot1, gl_mul(c_, W1[e]));      \
    e += 1;                                      \
  }


// The constraint stream is split over several kernels (add / mul blocks, schedule, lookups+CTLs): every part
// writes its alpha-weighted partial sums, k_quotient_finish adds them.  Since acc_j = sum_e c_e alpha_j^(K-1-e) is
// linear in the constraints, the split changes nothing but register pressure and the number of waves in flight.
__device__ __forceinline__ void store_part(const QArgs& A, int part, size_t j, u64 tot0, u64 tot1) {
  const size_t M2 = A.stride;
  A.part[((size_t)part * 2 + 0) * M2 + j] = tot0;
  A.part[((size_t)part * 2 + 1) * M2 + j] = tot1;
}

// Schedule constraints starting at constraint index e; writes partial `part`.  L = LayoutT<..>.
template <class L, bool FIRST_A_IS_ONE>
__device__ __forceinline__ void schedule_part(const QArgs& A, size_t j, size_t jn, int e, int part) {
  u64 tot0 = 0, tot1 = 0;
  const size_t M2 = A.stride;
  const u64* __restrict__ tl = A.tl;
  const u64* __restrict__ W0 = A.W;
  const u64* __restrict__ W1 = A.W + A.K;
  const u32* __restrict__ W3 = (const u32*)(A.W + 2 * (size_t)A.K);  // cut weights (accw_mad), 8 u32 per constraint
  const u64 filter = TL(L::FILTER);
  // ---- eval_packed_generic body (scalar_mul_stark.rs:257-339) ------------------------------------------
  const u64 is_first = TL(L::FLAGS + 0), is_last = TL(L::FLAGS + 1);
  const u64 n_filter = TN(L::FILTER), n_is_last = TN(L::FLAGS + 1);
  const u64 is_not_last_round = gl_sub(filter, is_last);
  const u64 is_next_not_last_round = gl_sub(n_filter, n_is_last);
  const u64 is_adding = TL(L::IS_ADDING), idnl = TL(L::IDNL);
  const u64 n_is_adding = TN(L::IS_ADDING), n_idnl = TN(L::IDNL);
  const u64 bit0 = TL(L::BITS), n_bit0 = TN(L::BITS);

  EMIT(gl_mul(is_first, gl_sub(is_adding, 1)));                                          // 198
  EQ_GROUP(is_first, L::PL, TL(L::DOUBLE + i), TL(L::B + i));                       // 199
  EQ_GROUP(gl_mul(bit0, is_first), L::PL, TL(L::SUM + i), TL(L::C + i));            // 231
  EQ_GROUP(gl_mul(gl_sub(1, bit0), is_first), L::PL, TL(L::SUM + i), TL(L::A + i)); // 263
  if (FIRST_A_IS_ONE) {  // Fq exp: first round, a = 1 (exp_stark.rs:250-256)
    EQ_GROUP(is_first, L::PL, TL(L::A + i), (u64)(i == 0 ? 1 : 0));
  }
  // doubling step -> addition step
  EQ_GROUP(idnl, L::PL, TN(L::A + i), TL(L::SUM + i));                              // 295
  EQ_GROUP(idnl, L::PL, TN(L::B + i), TL(L::DOUBLE + i));                           // 327
  EQ_GROUP(gl_mul(n_bit0, idnl), L::PL, TN(L::SUM + i), TN(L::C + i));              // 359
  EQ_GROUP(gl_mul(gl_sub(1, n_bit0), idnl), L::PL, TN(L::SUM + i), TN(L::A + i));   // 391
  EQ_GROUP(idnl, L::PL, TN(L::DOUBLE + i), TL(L::DOUBLE + i));                      // 423
  EMIT(gl_mul(idnl, gl_sub(n_is_adding, 1)));                                            // 455
  EMIT(gl_mul(idnl, n_idnl));                                                            // 456
  EQ_GROUP(idnl, 256, TN(L::BITS + i), TL(L::BITS + ((i + 1) & 255)));           // 457
  // addition step -> doubling step
  EQ_GROUP(is_adding, L::PL, TN(L::A + i), TL(L::DOUBLE + i));                      // 713
  EQ_GROUP(is_adding, L::PL, TN(L::B + i), TL(L::DOUBLE + i));                      // 745
  EQ_GROUP(is_adding, L::PL, TN(L::SUM + i), TL(L::SUM + i));                       // 777
  EQ_GROUP(is_adding, L::PL, TN(L::DOUBLE + i), TN(L::C + i));                      // 809
  EMIT(gl_mul(is_adding, n_is_adding));                                                  // 841
  EMIT(gl_mul(is_adding, gl_sub(n_idnl, is_next_not_last_round)));                       // 842
  EQ_GROUP(is_adding, 256, TN(L::BITS + i), TL(L::BITS + i));                    // 843
  // eval_round_flags (round_flags.rs:46-81), period 512
  {
    const u64 counter = TL(L::FLAGS + 2), inv_c = TL(L::FLAGS + 3), inv_cp = TL(L::FLAGS + 4);
    const u64 n_counter = TN(L::FLAGS + 2);
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
  EMIT(gl_mul(is_not_last_round, gl_sub(TN(L::TIMESTAMP), TL(L::TIMESTAMP))));   // 1107
  EMIT(gl_mul(is_not_last_round, gl_sub(n_filter, filter)));                             // 1108
  const u64 x = A.pt.x[j], llast = A.pt.llast[j];
  const u64 z_last = gl_sub(x, A.w_inv);
  {
    const u64 rc = TL(L::RANGE), diff = gl_sub(TN(L::RANGE), rc);
    EMIT(gl_mul(gl_sub(gl_mul(diff, diff), diff), z_last));                              // 1109 transition
    EMIT(gl_mul(gl_sub(rc, 65535), llast));                                              // 1110 last row
  }
  store_part(A, part, j, tot0, tot1);
}

// Lookups + CTL constraints (they start at constraint index sh.n_constraints), sum of all partials, division by Z_H
// and the store in natural order of the coset.
__device__ __forceinline__ void finish_point(const QArgs& A, const StarkShape& sh, size_t j) {
  const unsigned log_n = A.log_n;
  const size_t N = (size_t)1 << log_n, M2 = A.stride;
  const size_t jn = q_next(A, j);
  u64 tot0 = 0, tot1 = 0;
  for (int p = 0; p < A.n_parts; p++) {
    tot0 = gl_add(tot0, A.part[((size_t)p * 2 + 0) * M2 + j]);
    tot1 = gl_add(tot1, A.part[((size_t)p * 2 + 1) * M2 + j]);
  }
  const u64 x = A.pt.x[j], lfirst = A.pt.lfirst[j], llast = A.pt.llast[j];
  const u64 z_last = gl_sub(x, A.w_inv);
  lookup_and_ctl_constraints(sh, A.tl, A.al, M2, j, jn, (const u32*)(A.W + 2 * (size_t)A.K), sh.n_constraints, A.betas, A.gammas,
                             lfirst, llast, z_last, tot0, tot1);
  const size_t h = A.natural ? A.hsel : j >> log_n;
  const size_t k = A.natural ? A.k0 + j : (size_t)bitrev32((u32)(j & (N - 1)), log_n);
  A.out[(0 * 2 + h) * N + k] = gl_mul(tot0, A.zh_inv[h]);
  A.out[(1 * 2 + h) * N + k] = gl_mul(tot1, A.zh_inv[h]);
}

// W[a][e] = alpha_a^(K-1-e); mzt[blk][a][80] = {M_j (17), u_d (31), B*u_d (31), off*sum(u_d)} for the
// eval_modulus_zero blocks whose first constraint indices are mz_e0[0..n_blocks).
void quotient_host_tables(int K, const u64 alphas[2], const int* mz_e0, int n_blocks, std::vector<u64>& W, std::vector<u64>& mzt);
void quotient_fill_args(QArgs& A, const StarkShape& sh, const u64* d_tl, const u64* d_al, const u64* d_W, const u64* d_mzt,
                        const QPointTables& pt, const u64 betas[2], const u64 gammas[2], unsigned log_n, u64* d_out,
                        u64* d_part);
static constexpr int QUOTIENT_MAX_PARTS = 7;
void quotient_finish_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
