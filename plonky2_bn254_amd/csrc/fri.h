// Host interface of the openings / FRI kernels (fri.hip).
#pragma once
#include "quotient_common.h"

__host__ __device__ inline size_t merkle_level_offset_dev(int log_leaves, int level) {
  // sum_{l<level} 2^(log_leaves-l) = 2^(log_leaves+1) - 2^(log_leaves-level+1)
  return ((size_t)2 << log_leaves) - ((size_t)2 << (log_leaves - level));
}

// Partial evaluations per (polynomial, block k1 < R): out[(p*R + k1)*5 ..] = {S(zeta^R).c0,.c1, S((g zeta)^R).c0,.c1, sum};
// P(z) = sum_k1 z^k1 S_k1(z^R), P(1) = sum_k1 sums (coefficients in the transposed layout of ntt.h)
// power tables of the two opening points (zeta, g zeta) for k_openings: FRI_OPENING_TABLE_WORDS words, built on the device
static constexpr size_t FRI_OPENING_ZQ_WORDS = 2 * 256 * 4, FRI_OPENING_TABLE_WORDS = 2 * FRI_OPENING_ZQ_WORDS;
void fri_opening_tables(unsigned log_r, gl2 zeta, gl2 zeta_next, u64* d_tables, hipStream_t st);
void fri_openings(const u64* d_coeffs, size_t N, unsigned log_r, int npolys, const u64* d_tables, u64* d_out, hipStream_t st);

// d_out[2N][2]: sum_b alpha^(k_b) (F_b(x) - F_b(z_b)) / (x - z_b) on the LDE domain, bit-reversed order
void fri_combine(const StarkShape& sh, const u64* d_tl, const u64* d_al, const u64* d_ql, const u64* d_apow, const u64* d_xs,
                 gl2 zeta, gl2 zeta_next, gl2 r0, gl2 r1, gl2 r2, gl2 alpha, size_t M2, u64* d_out, hipStream_t st);

// The same from coefficient vectors (no resident LDE): comb[6][N] = the three alpha-weighted sums (f1, quotient part of f0, f2; c0 / c1
// each) formed on the coefficients; after the ordinary LDE of those six columns fri_combine_final applies the point-wise part.
void fri_combine_coeffs(const StarkShape& sh, const u64* d_tcoef, const u64* d_acoef, const u64* d_qcoef, const u64* d_apow, size_t N,
                        u64* d_comb, hipStream_t st);
void fri_combine_final(const StarkShape& sh, const u64* d_comb_lde, const u64* d_xs, gl2 zeta, gl2 zeta_next, gl2 r0, gl2 r1, gl2 r2,
                       gl2 alpha, size_t M2, u64* d_out, hipStream_t st);
// rows k0 .. k0 + rows - 1 (mod N) of coset h of ncols LDE columns (leaf order, column stride M2), natural order: out[ncols][rows]
void fri_extract_window(const u64* d_lde, size_t M2, unsigned log_n, int h, size_t k0, size_t rows, int ncols, u64* d_out, hipStream_t st);
// out[ncols][nq] = lde[c][indices[q]]
void fri_gather_rows(const u64* d_lde, size_t M2, int ncols, const u32* d_indices, int nq, u64* d_out, hipStream_t st);

void fri_fold(const u64* d_in, u64* d_out, unsigned log_m, u64 shift, gl2 beta, u64 inv16, hipStream_t st);

void fri_pow_launch(const u64 state[12], int pos, u64 base, unsigned pow_bits, size_t count, unsigned long long* d_result,
                    hipStream_t st);

static constexpr int FRI_MAX_LAYERS = 8;
struct QueryGatherArgs {
  const u64* lde[3];
  int lde_by_query[3] = {0, 0, 0};  // 1: lde[o] holds the leaf rows of the queries only, [width][n_queries] (fri_gather_rows)
  const u64* tree[3];
  int width[3];
  const u64* layer_vals[FRI_MAX_LAYERS];
  const u64* layer_tree[FRI_MAX_LAYERS];
  int n_layers;
  int log_m2, cap_height;
  size_t M2;
  const u32* indices;
  u64* out;
  size_t words_per_query;
};
void fri_gather_queries(const QueryGatherArgs& A, int n_queries, hipStream_t st);
