// Host interface of the Goldilocks NTT / LDE kernels (ntt.hip).
#pragma once
#include <vector>
#include "gl_dev.h"

static constexpr size_t NTT_N = 65536;  // kernel transform size; trace height of one 128-instance proof

struct NttTables {
  u64 *tw256_fwd = nullptr, *tw256_inv = nullptr;  // w_256^e
  u64 *twmat_fwd = nullptr, *twmat_inv = nullptr;  // w_N^(i2*k1) at [k1*256+i2]
  u64* twmat_inv_ninv = nullptr;                   // (1/N) w_N^-(i2*k1)
  u64* coset_pow[2] = {nullptr, nullptr};          // shift_h^i
  u64* coset_inv_pow[2] = {nullptr, nullptr};      // (1/N) shift_h^-k
  u64 n_inv = 0;
};

int ntt_tables_init(NttTables* T);
void ntt_tables_free(NttTables* T);
void ntt_inverse(const NttTables* T, const u64* values, u64* coeffs, u64* tmp, int ncols, hipStream_t s);
void ntt_coset_inverse(const NttTables* T, int h, const u64* values, u64* coeffs, u64* tmp, int ncols, hipStream_t s);
void ntt_lde(const NttTables* T, const u64* coeffs, u64* lde, u64* tmp, int ncols, hipStream_t s);
// from_values in one go (iNTT + both coset NTTs, middle passes fused): tmp[C][N], tmp2[2][C][N]; values == coeffs allowed
void ntt_inverse_lde(const NttTables* T, const u64* values, u64* coeffs, u64* lde, u64* tmp, u64* tmp2, int ncols, hipStream_t s);
void ntt_coset_forward_natural(const NttTables* T, int h, const u64* coeffs, u64* values, u64* tmp, int ncols, hipStream_t s);

// ---- N = R * 2^16 (tall traces) ---------------------------------------------------------------------------------
struct NttTallTables {
  unsigned log_n = 0;
  u64* wn_inv_pow = nullptr;              // w_N^-i2, i2 < 2^16
  u64* wn_pow_br = nullptr;               // w_N^bitrev16(p)
  u64* block_coset_pow[2] = {nullptr, nullptr};  // (shift_h^R)^i2
  u64 shift[2] = {0, 0};                  // g, g*w_2N
  u64 r_inv = 0;
};
int ntt_tall_tables_init(NttTallTables* T, unsigned log_n, u64 base_shift = GL_GEN);
void ntt_tall_tables_free(NttTallTables* T);
// Coefficients live in the "transposed" layout: coefficient k1 + R*k2 at position k1*2^16 + k2.
void ntt_inverse_tall(const NttTables* T, const NttTallTables* TT, const u64* values, u64* coeffs, u64* tmp, int ncols,
                      hipStream_t s);
void ntt_coset_inverse_tall(const NttTables* T, const NttTallTables* TT, int h, const u64* values, u64* coeffs, u64* tmp, int ncols,
                            hipStream_t s);
void ntt_inverse_lde_tall(const NttTables* T, const NttTallTables* TT, const u64* values, u64* coeffs, u64* lde, u64* tmp, int ncols,
                          hipStream_t s);
// coset_mask: bit h set = compute coset h (the lower / upper half of the bit-reversed output); 3 = both
void ntt_lde_tall(const NttTables* T, const NttTallTables* TT, const u64* coeffs, u64* lde, u64* tmp, int ncols, hipStream_t s,
                  int coset_mask = 3);

// ---- N = 2^23: one radix-2 level above the tall transforms (see ntt.hip) ---------------------------------------------
struct NttSplitTables {
  unsigned log_n = 0;
  u64 *fwd_lo = nullptr, *fwd_hi = nullptr, *inv_lo = nullptr, *inv_hi = nullptr;  // w_N^(+-b), w_N^(+-2048 a)
  u64 shift[2] = {0, 0};  // g, g*w_2N
  u64 half = 0;
};
int ntt_split_tables_init(NttSplitTables* S, unsigned log_n);
void ntt_split_tables_free(NttSplitTables* S);
void ntt_split_inverse(const NttSplitTables* S, const u64* values, u64* halves, int ncols, hipStream_t s);
void ntt_split_forward(const NttSplitTables* S, const u64* eo, u64* lde, size_t lde_stride, int ncols, hipStream_t s, int coset_mask = 3);
void ntt_coset_inverse_split(const NttTables* T, const NttTallTables* TT, const NttSplitTables* S, int h, const u64* values, u64* coeffs,
                             u64* tmp, int ncols, hipStream_t s);

// debug: runs the hand-written field sequences of gl_asm.h on n operand pairs (bn254s_selftest_field)
static constexpr int FIELD_SELFTEST_OUTS = 17;
void field_selftest(const u64* a, const u64* b, u64* out, size_t n, hipStream_t s);
