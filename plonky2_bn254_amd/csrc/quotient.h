// Host interface of the quotient kernels.
#pragma once
#include <vector>
#include "quotient_sched.h"

void quotient_point_tables(u64* d_x, u64* d_lfirst, u64* d_llast, unsigned log_n, hipStream_t st);
// window mode (QArgs in quotient_sched.h): the constants of rows k0 .. k0 + count - 1 of coset h in natural order, and the
// arguments of a launch over `count` of the count + 1 rows held in d_tw / d_aw (column stride `stride`)
void quotient_point_tables_window(u64* d_x, u64* d_lfirst, u64* d_llast, unsigned log_n, int h, size_t k0, size_t count, hipStream_t st);
void quotient_window_args(QArgs& A, const u64* d_tw, const u64* d_aw, const QPointTables& pt, u64* d_part, size_t stride, size_t count,
                          size_t k0, int h);
// first constraint indices of the eval_modulus_zero blocks of each AIR (for quotient_host_tables)
int g1_quotient_mz_blocks(const int** e0);
int g2_quotient_mz_blocks(const int** e0);
int fq_quotient_mz_blocks(const int** e0);
// A.out: [2 alphas][2 cosets][N] quotient values, natural order on each coset
void g1_quotient_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
void g2_quotient_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
void fq_quotient_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
