// G1 scalar-mul STARK column layout (reference src/starks/curves/g1/scalar_mul_view.rs:10-49,
// add.rs:36-46, modular/{is_modulus_zero.rs:27-32, modulus_zero.rs:66-73}, common/round_flags.rs:11-19)
// and the trace-generation driver.
#pragma once
#include "gl_dev.h"
#include "../../include/bn254_stark.h"

static constexpr int G1_W = 781;
static constexpr int G1_COL_DOUBLE = 0, G1_COL_SUM = 32, G1_COL_A = 64, G1_COL_B = 96, G1_COL_C = 128, G1_COL_AUX = 160;
// offsets inside G1AddAux (354 columns)
static constexpr int G1_AUX_IS_X_EQ = 0, G1_AUX_IS_X_EQ_AUX = 1 /* inv[16] then ModulusZeroAux[80] */,
                     G1_AUX_IS_X_EQ_FILTER = 97, G1_AUX_LAMBDA = 98, G1_AUX_LAMBDA_AUX = 114, G1_AUX_X_AUX = 194,
                     G1_AUX_Y_AUX = 274;
static constexpr int G1_COL_BITS = 514, G1_COL_FLAGS = 770 /* is_first, is_last, counter, inv_counter, inv_counter' */,
                     G1_COL_TIMESTAMP = 775, G1_COL_IS_ADDING = 776, G1_COL_IDNL = 777, G1_COL_FILTER = 778,
                     G1_COL_FREQ = 779, G1_COL_RANGE = 780;
static constexpr int G1_RC_BEGIN = 64, G1_RC_END = 514;  // range-checked (LogUp) columns
// ModulusZeroAux block: [is_quot_positive, quot_abs[17], aux_lo[31], aux_hi[31]]
static constexpr int MZ_IQP = 0, MZ_QUOT = 1, MZ_LO = 18, MZ_HI = 49, MZ_LEN = 80;

size_t g1_trace_scratch_words(size_t n);
// All pointers are device pointers; trace is column-major [G1_W][N]; outputs n x 8 canonical words.
int g1_generate_trace_device(const u64* d_scalars, const u64* d_x, const u64* d_off, size_t n, u64* d_trace, size_t N,
                             u64* d_scratch, u64* d_outputs, int* d_err, hipStream_t st, bool with_range = true);

// generate_range_checks on a finished trace: histogram of columns [rc_begin, rc_end) -> frequency column, and the range
// counter column (defined in trace_g1.hip; hist = 65536 u32 of scratch)
void launch_range_columns(u64* trace, size_t N, int rc_begin, int rc_end, int freq_col, int range_col, u32* hist, int* err,
                          hipStream_t st);
void launch_fq_inv_selftest(const u64* in, u64* out, size_t n, hipStream_t st);
