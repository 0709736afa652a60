// Host interface of the quotient kernels.
#pragma once
#include <vector>
#include "quotient_sched.h"

void quotient_point_tables(u64* d_x, u64* d_lfirst, u64* d_llast, unsigned log_n, hipStream_t st);
// first constraint indices of the eval_modulus_zero blocks of each AIR (for quotient_host_tables)
int g1_quotient_mz_blocks(const int** e0);
int g2_quotient_mz_blocks(const int** e0);
int fq_quotient_mz_blocks(const int** e0);
// A.out: [2 alphas][2 cosets][N] quotient values, natural order on each coset
void g1_quotient_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
void g2_quotient_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
void fq_quotient_launch(const QArgs& A, const StarkShape& sh, hipStream_t st);
