// Host interface of the Goldilocks NTT / LDE kernels (ntt.hip).
#pragma once
#include <vector>
#include "gl_dev.h"

static constexpr size_t NTT_N = 65536;  // kernel transform size; trace height of one 128-instance proof

struct NttTables {
  u64 *tw256_fwd = nullptr, *tw256_inv = nullptr;  // w_256^e
  u64 *twmat_fwd = nullptr, *twmat_inv = nullptr;  // w_N^(i2*k1) at [k1*256+i2]
  u64* coset_pow[2] = {nullptr, nullptr};          // shift_h^i
  u64* coset_inv_pow[2] = {nullptr, nullptr};      // (1/N) shift_h^-k
  u64 n_inv = 0;
};

int ntt_tables_init(NttTables* T);
void ntt_tables_free(NttTables* T);
void ntt_inverse(const NttTables* T, const u64* values, u64* coeffs, u64* tmp, int ncols, hipStream_t s);
void ntt_coset_inverse(const NttTables* T, int h, const u64* values, u64* coeffs, u64* tmp, int ncols, hipStream_t s);
void ntt_lde(const NttTables* T, const u64* coeffs, u64* lde, u64* tmp, int ncols, hipStream_t s);
void ntt_coset_forward_natural(const NttTables* T, int h, const u64* coeffs, u64* values, u64* tmp, int ncols, hipStream_t s);
