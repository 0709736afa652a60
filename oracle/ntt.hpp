// ORACLE (test infrastructure, NOT the product path).
// Radix-2 FFT conventions of plonky2_field 0.2.2 `fft.rs` / `polynomial/mod.rs` (un-vendored;
// SURVEY.md App. A.2), used through PolynomialBatch::from_values at reference
// src/starks/common/prover.rs:31-38:
//   fft(c)[j]  = sum_i c_i w^(ij)            (natural order in, natural order out)
//   ifft(v)[i] = 1/N sum_j v_j w^(-ij)
//   coset_fft(c, s) = fft(c_i * s^i);  coset_ifft(v, s)[i] = ifft(v)[i] * s^(-i)
// Pinned against an O(N^2) DFT in tests (the conventions themselves are "parity unpinned").
#pragma once
#include "gl.hpp"

namespace orc {

// In-place iterative DIT on base-field data; `root` must be a primitive N-th root of unity.
// Twiddles: stage `len` uses w^(k n/len), k < len/2; they are laid out stage by stage (tw[half + k] for the stage of half-size
// `half`: n - 1 words) so that the inner loop reads them contiguously, and kept per (n, root) for the life of the process -
// a proof runs thousands of transforms of the same three sizes.
struct TwiddleTable {
  std::vector<u64> tw;  // tw[half + k] = root^(k n / (2 half)), half = 1, 2, 4, ..., n/2
};
static inline const TwiddleTable& twiddles_for(size_t n, u64 root) {
  struct Entry {
    size_t n;
    u64 root;
    TwiddleTable* t;
  };
  static std::vector<Entry> cache;
  const TwiddleTable* hit = nullptr;
#pragma omp critical(orc_twiddle_cache)
  {
    for (auto& e : cache)
      if (e.n == n && e.root == root) hit = e.t;
    if (!hit) {
      TwiddleTable* t = new TwiddleTable();
      t->tw.assign(n > 1 ? n : 2, 1);
      std::vector<u64> pw(n / 2 ? n / 2 : 1);
      pw[0] = 1;
      for (size_t k = 1; k < n / 2; k++) pw[k] = gl_mul(pw[k - 1], root);
      for (size_t half = 1; half < n; half <<= 1) {
        const size_t step = n / (2 * half);
        for (size_t k = 0; k < half; k++) t->tw[half + k] = pw[k * step];
      }
      cache.push_back({n, root, t});
      hit = t;
    }
  }
  return *hit;
}
template <class T>
static void fft_inplace_root(std::vector<T>& a, u64 root) {
  size_t n = a.size();
  unsigned lg = log2_strict(n);
  for (size_t i = 0; i < n; i++) {
    size_t j = reverse_bits(i, lg);
    if (i < j) std::swap(a[i], a[j]);
  }
  const u64* tw = twiddles_for(n, root).tw.data();
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t half = len / 2;
    const u64* w = tw + half;
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < half; k++) {
        T u = a[i + k];
        T v = mul_base(a[i + k + half], w[k]);
        a[i + k] = u + v;
        a[i + k + half] = u - v;
      }
  }
}
static inline F mul_base(F x, u64 s) { return F(gl_mul(x.v, s)); }
static inline F2 mul_base(F2 x, u64 s) { return x.scalar_mul(s); }

template <class T>
static void fft(std::vector<T>& a) {
  fft_inplace_root(a, gl_root_of_unity(log2_strict(a.size())));
}
template <class T>
static void ifft(std::vector<T>& a) {
  unsigned lg = log2_strict(a.size());
  fft_inplace_root(a, gl_inv(gl_root_of_unity(lg)));
  u64 ninv = gl_inv((u64)a.size() % GL_P);
  for (auto& x : a) x = mul_base(x, ninv);
}
template <class T>
static void coset_fft(std::vector<T>& a, u64 shift) {
  u64 s = 1;
  for (auto& x : a) {
    x = mul_base(x, s);
    s = gl_mul(s, shift);
  }
  fft(a);
}
template <class T>
static void coset_ifft(std::vector<T>& a, u64 shift) {
  ifft(a);
  u64 si = gl_inv(shift), s = 1;
  for (auto& x : a) {
    x = mul_base(x, s);
    s = gl_mul(s, si);
  }
}

}  // namespace orc
