// Compile-time column layout shared by the three STARKs (reference `#[repr(C)]` views:
// src/starks/curves/g1/scalar_mul_view.rs:34-49, curves/g2/scalar_mul_view.rs:34-49, fields/exp_view.rs:31-48):
//   [double|square (PL)] [sum|product (PL)] [a (PL)] [b (PL)] [c (PL)] [op aux (AUXL)] [bits 256]
//   [is_first, is_last, counter, inv_counter, inv_counter'] [timestamp, is_adding|is_mul,
//   is_doubling_not_last|is_sq_not_last, filter, frequency, range_counter]
#pragma once

template <int PL_, int AUXL_>
struct LayoutT {
  static constexpr int PL = PL_, AUXL = AUXL_;
  static constexpr int DOUBLE = 0, SUM = PL_, A = 2 * PL_, B = 3 * PL_, C = 4 * PL_, AUX = 5 * PL_;
  static constexpr int BITS = AUX + AUXL_, FLAGS = BITS + 256, TIMESTAMP = FLAGS + 5, IS_ADDING = TIMESTAMP + 1,
                       IDNL = TIMESTAMP + 2, FILTER = TIMESTAMP + 3, FREQ = TIMESTAMP + 4, RANGE = TIMESTAMP + 5, W = RANGE + 1;
  static constexpr int RC_BEGIN = 2 * PL_, RC_END = 5 * PL_ + AUXL_;
};
typedef LayoutT<32, 354> G1L;   // W = 781
typedef LayoutT<64, 708> G2L;   // W = 1295
typedef LayoutT<16, 80> FQL;    // W = 427

// G2AddAux offsets (src/starks/curves/g2/add.rs:46-56, ext/is_modulus_zero.rs:21-28, ext/modulus_zero.rs:22-26)
static constexpr int G2_AUX_IS_X_EQ = 0, G2_AUX_IS_C0_ZERO = 1, G2_AUX_IS_C1_ZERO = 2, G2_AUX_C0_AUX = 3 /* inv16 + mz80 */,
                     G2_AUX_C1_AUX = 99, G2_AUX_IS_X_EQ_FILTER = 195, G2_AUX_LAMBDA = 196 /* c0 16, c1 16 */,
                     G2_AUX_LAMBDA_AUX = 228 /* c0 mz80, c1 mz80 */, G2_AUX_X_AUX = 388, G2_AUX_Y_AUX = 548;
enum { KIND_G1 = 0, KIND_G2 = 1, KIND_FQ = 2 };
