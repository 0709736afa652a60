// Host-side pieces shared by the prover (prover.hip) and the native verifier (verify.hip).
#pragma once
#include <vector>
#include "aux.h"
#include "layout.h"

// Stark::lookups, looked CTL tables and constraint counts of the three AIRs (kind = KIND_G1 / KIND_G2 / KIND_FQ)
StarkShape shape_for(int kind);
int point_words(int kind);  // words per input point: 8 (G1), 16 (G2), 4 (Fq element)
// FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits) for a polynomial of 2^degree_bits coefficients
std::vector<int> fri_arities(const bn254s_params& P, int degree_bits);
