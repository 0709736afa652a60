// Host interface of the quotient kernels.
#pragma once
#include <vector>
#include "quotient_common.h"

void quotient_point_tables(u64* d_x, u64* d_lfirst, u64* d_llast, unsigned log_n, hipStream_t st);
void g1_quotient_host_tables(const StarkShape& sh, const u64 alphas[2], std::vector<u64>& W, std::vector<u64>& mzt);
// d_out: [2 alphas][2 cosets][N] quotient values, natural order on each coset
void g1_quotient_launch(const StarkShape& sh, const u64* d_tl, const u64* d_al, const u64* d_W, const u64* d_mzt,
                        const QPointTables& pt, const u64 betas[2], const u64 gammas[2], unsigned log_n, u64* d_out,
                        hipStream_t st);
