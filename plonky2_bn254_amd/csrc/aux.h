// STARK shape description shared by the auxiliary-column, quotient and FRI stages.
#pragma once
#include "gl_dev.h"
#include "../../include/bn254_stark.h"

static constexpr int CTL_MAX = 2, CTL_MAX_COLS = 160;

// A looked table: column m = sum_{b<bits} 2^b * trace[start+b]  (Column::single -> bits = 1,
// Column::le_bits -> bits = 16); reference src/starks/curves/g1/scalar_mul_ctl.rs:20-55.
struct CtlSpecDev {
  int ncols[CTL_MAX];
  int filter_col[CTL_MAX];
  short col_start[CTL_MAX][CTL_MAX_COLS];
  unsigned char col_bits[CTL_MAX][CTL_MAX_COLS];
};

struct StarkShape {
  int W = 0;                       // trace width
  int rc_begin = 0, rc_end = 0;    // LogUp-checked columns (Stark::lookups)
  int table_col = 0, freq_col = 0;
  int n_ctl = 2;
  int n_constraints = 0;           // AIR constraints emitted by eval_packed_generic
  CtlSpecDev ctl;
  GL_HD int n_rc() const { return rc_end - rc_begin; }
  GL_HD int n_helpers() const { return (n_rc() + 1) / 2; }              // per challenge, without Z
  GL_HD int n_aux() const { return 2 * (n_helpers() + 1) + 2 * n_ctl; }  // A
  GL_HD int n_lookup_cols() const { return 2 * (n_helpers() + 1); }
  GL_HD int n_total_constraints() const { return n_constraints + 2 * (n_helpers() + 2) + 2 * 2 * n_ctl; }
};

StarkShape g1_shape();

size_t aux_scratch_words(const StarkShape& sh, size_t N);
void aux_build(const StarkShape& sh, const u64* d_trace, size_t N, const u64 betas[2], const u64 gammas[2], u64* d_aux,
               u64* d_scratch, int* d_err, hipStream_t st);
