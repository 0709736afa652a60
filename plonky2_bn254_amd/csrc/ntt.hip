// Goldilocks radix-2^16 NTT / low-degree extension kernels for gfx950 (the north-star kernel).
//
// What it replaces: plonky2 `PolynomialBatch::from_values` / `from_coeffs` as called by the reference at
// src/starks/common/prover.rs:31-38 (per column: iFFT(N) -> zero-pad to 2N -> coset FFT with shift g ->
// rows in bit-reversed order), and `coset_ifft` in starky's compute_quotient_polys.
//
// Design (DESIGN.md "K-ntt"): a size-2^16 transform is a 256 x 256 four-step.  Each 256-point DFT is two
// radix-16 passes held in registers (16 values per thread) with one LDS transpose in between.  All
// 16th roots of unity in Goldilocks are powers of two (w_16 = 2^12), so the radix-16 butterflies use
// shifts only; full 64-bit modular multiplications are needed just for the inter-pass twiddles.
//   pass 1 (k_ntt_pass1): tile = 16 adjacent matrix columns i2 x all 256 i1; DFT over i1; multiply by
//                         w_N^(i2*k1); write Y[k1][i2].
//   pass 2 (k_ntt_pass2): tile = 16 matrix rows k1 x all 256 i2; DFT over i2; write X[k1 + 256*k2]
//                         either in natural order or in bit-reversed (Merkle-leaf) order.
// The 2N-point coset LDE is computed as two N-point coset NTTs (cosets g and g*w_2N); in bit-reversed
// order they are the lower and upper half of the output, so no separate zero-padded 2N transform and
// no bit-reversal pass exist.  Global accesses are 128-byte segments (16 lanes x 8 B), or 1 KB per store
// instruction where the output is staged through LDS; twiddle / coset tables have the same access pattern
// as the data and stay L2-resident.  The arithmetic is the hand-written code of gl_asm.h.
#include "gl_dev.h"
#include "gl_asm.h"
#include "ntt.h"

// Issue priority of the NTT/LDE kernels' waves (0 = default).  The stage runs alone among the wide kernels, but the latency-bound
// kernels of the other proofs in flight (priority 3, gl_dev.h) share its SIMDs; BN254S_NTT_PRIO is a compile-time A/B knob.
#if defined(__HIP_DEVICE_COMPILE__) && defined(BN254S_NTT_PRIO) && BN254S_NTT_PRIO > 0
#define NTT_KERNEL_PRIO() __builtin_amdgcn_s_setprio(BN254S_NTT_PRIO)
#else
#define NTT_KERNEL_PRIO() ((void)0)
#endif

// ---- radix-16 DFT in registers --------------------------------------------------------------------------
template <bool INV, int SPAN, int G, int J>
__device__ __forceinline__ void bfly16(u64* x) {
  constexpr int E = J * (8 / SPAN);
  constexpr int S0 = E == 0 ? 0 : (!INV ? 12 * E : 192 - 12 * E);  // twiddle = 2^S0
  // 2^96 = -1: a negative twiddle is folded into the subtraction (b - a); 2^S with S > 64 is -2^-(96 - S), a Montgomery-style
  // right shift (gl_asm.h) that is cheaper than three left shifts
  constexpr bool NEG0 = S0 >= 96;
  constexpr int S1 = NEG0 ? S0 - 96 : S0;
  constexpr bool RIGHT = S1 > 64;
  constexpr bool NEG = NEG0 != RIGHT;
  u64 s, d;
  gl_bfly_asm<NEG>(x[G + J], x[G + J + SPAN], s, d);
  x[G + J] = s;
  if constexpr (S1 == 0) x[G + J + SPAN] = d;
  else if constexpr (RIGHT) x[G + J + SPAN] = gl_shr_small_asm<96 - S1>(d);
  else x[G + J + SPAN] = gl_shl_asm<S1>(d);
}
// DIF network: natural-order input, output X[k] ends up in x[bitrev4(k)].
template <bool INV>
__device__ __forceinline__ void dft16(u64* x) {
#define BF(SPAN, G, J) bfly16<INV, SPAN, G, J>(x)
  BF(8, 0, 0); BF(8, 0, 1); BF(8, 0, 2); BF(8, 0, 3); BF(8, 0, 4); BF(8, 0, 5); BF(8, 0, 6); BF(8, 0, 7);
  BF(4, 0, 0); BF(4, 0, 1); BF(4, 0, 2); BF(4, 0, 3); BF(4, 8, 0); BF(4, 8, 1); BF(4, 8, 2); BF(4, 8, 3);
  BF(2, 0, 0); BF(2, 0, 1); BF(2, 4, 0); BF(2, 4, 1); BF(2, 8, 0); BF(2, 8, 1); BF(2, 12, 0); BF(2, 12, 1);
  BF(1, 0, 0); BF(1, 2, 0); BF(1, 4, 0); BF(1, 6, 0); BF(1, 8, 0); BF(1, 10, 0); BF(1, 12, 0); BF(1, 14, 0);
#undef BF
}
__device__ __forceinline__ constexpr int br4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

// 16 independent 256-point DFTs per 256-thread workgroup.  On entry thread (d, g) holds
// x[m] = in_d[g + 16 m]; on exit thread (d, ka) holds x[br4(kb)] = X_d[ka + 16 kb].
// The LDS transpose uses a padded image so that the write and the transposed read are both
// bank-conflict free; MODE selects the lane layouts:
//   0: lanes d-fast before and after (d = t&15)            slot = ka*272 + g*16 + d
//   1: lanes g-fast before and after (g/ka = t&15)          slot = ka*257 + d*16 + g
//   2: lanes g-fast before, d-fast after (re-maps d, ka)    slot = ka*272 + d*17 + g
// The exchange moves the low and the high 32-bit words in two rounds through the same 17 KB image (slots are 4-byte
// words, 64 banks): half the LDS of a one-round 64-bit exchange, so that eight workgroups fit on a CU and the loads of one
// workgroup overlap the arithmetic of the others (the passes are neither HBM- nor VALU-bound alone).
static constexpr int LDS_TILE_WORDS = 16 * 272;  // u32 words
template <int MODE>
__device__ __forceinline__ int lds_slot(int ka, int g, int d) {
  return MODE == 0 ? ka * 272 + g * 16 + d : MODE == 1 ? ka * 257 + d * 16 + g : ka * 272 + d * 17 + g;
}
template <bool INV, int MODE>
__device__ __forceinline__ void dft256_tile(u64* x, u32* lds, const u64* __restrict__ tw256, int& d, int& g) {
  dft16<INV>(x);
  // inner twiddle w_256^(g*ka), then transpose (g <-> ka) through LDS
#if defined(BN254S_NTT_AB) && BN254S_NTT_AB == 1
  // TIMING-ONLY build (tools/gpu_ntt_ab.sh, wrong results): the inner twiddles of the pass-1 tiles dropped = the upper bound of
  // what a 64 x 64 x 16 decomposition (one general twiddle layer fewer per transform) could save
  if (MODE != 0)
#endif
#if defined(BN254S_NTT_AB) && BN254S_NTT_AB == 2
  // TIMING-ONLY build: ... replaced by power-of-two twiddles (what a radix-64 stage would put there instead)
  if (MODE == 0) {
#pragma unroll
    for (int ka = 1; ka < 16; ka++) x[br4(ka)] = (ka & 1) ? gl_shl_asm<36>(x[br4(ka)]) : gl_shl_asm<12>(x[br4(ka)]);
  } else
#endif
#pragma unroll
  for (int ka = 1; ka < 16; ka++) x[br4(ka)] = gl_mul_asm(x[br4(ka)], tw256[g * ka]);
  int d2 = d, g2 = g;
  if (MODE == 2) {
    d2 = threadIdx.x & 15;
    g2 = threadIdx.x >> 4;
  }
  // MODE 1: the sixteen lanes of a 256-point transform sit in one wave (g = lane & 15) and a wave's four transforms own their
  // slots, so the exchange needs no workgroup barrier: LDS operations of one wave complete in order.
  u32 lo[16], hi[16];
#pragma unroll
  for (int ka = 0; ka < 16; ka++) lds[lds_slot<MODE>(ka, g, d)] = (u32)x[br4(ka)];
  if (MODE == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
#pragma unroll
  for (int gg = 0; gg < 16; gg++) lo[gg] = lds[lds_slot<MODE>(g2, gg, d2)];  // the thread now owns (d2, ka = g2)
  if (MODE == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
#pragma unroll
  for (int ka = 0; ka < 16; ka++) lds[lds_slot<MODE>(ka, g, d)] = (u32)(x[br4(ka)] >> 32);
  if (MODE == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();
#pragma unroll
  for (int gg = 0; gg < 16; gg++) hi[gg] = lds[lds_slot<MODE>(g2, gg, d2)];
#pragma unroll
  for (int gg = 0; gg < 16; gg++) x[gg] = (u64)lo[gg] | ((u64)hi[gg] << 32);
  d = d2;
  g = g2;
  dft16<INV>(x);
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------
// grid = (16 tiles, ncols); block = 256.  in/out column strides in elements.
// For traces taller than 2^16 rows a column of N = R*2^16 words is R consecutive blocks; blockIdx.y then counts
// (column, block) pairs: column = y >> log_r, block = y & (R-1)  (log_r = 0: one block per column).
__device__ __forceinline__ size_t col_offset(unsigned y, unsigned log_r, size_t stride) {
  return (size_t)(y >> log_r) * stride + (size_t)(y & ((1u << log_r) - 1)) * NTT_N;
}

template <bool INV>
__global__ __launch_bounds__(256) void k_ntt_pass1(const u64* __restrict__ in, size_t in_stride, u64* __restrict__ out,
                                                   size_t out_stride, const u64* __restrict__ pre,
                                                   const u64* __restrict__ twmat, const u64* __restrict__ tw256,
                                                   unsigned log_r) {
  NTT_KERNEL_PRIO();
  __shared__ u32 lds[LDS_TILE_WORDS];
  const int t = threadIdx.x;
  int d = t & 15, g = t >> 4;
  const int i2 = blockIdx.x * 16 + d;
  const u64* col = in + col_offset(blockIdx.y, log_r, in_stride);
  u64 x[16];
#pragma unroll
  for (int m = 0; m < 16; m++) x[m] = col[(g + 16 * m) * 256 + i2];
  if (pre) {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = gl_mul_asm(x[m], pre[(g + 16 * m) * 256 + i2]);
  }
  dft256_tile<INV, 0>(x, lds, tw256, d, g);
  u64* ocol = out + col_offset(blockIdx.y, log_r, out_stride);
  const int ka = g;
#pragma unroll
  for (int kb = 0; kb < 16; kb++) {
    int k1 = ka + 16 * kb;
    u64 v = x[br4(kb)];
    v = gl_mul_asm(v, twmat[k1 * 256 + i2]);
    ocol[k1 * 256 + i2] = v;
  }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------
// grid = (16 tiles, ncols); block = 256.  Tile b handles rows k1 = b + 16*d (d = 0..15), so that in
// bit-reversed output order the 16 rows br8(k1) = br4(b)*16 + br4(d) form one contiguous 32 KB region.
// post: optional per-output-index (natural k) scale table; post_scalar multiplies everything (1/N).
static constexpr int STAGE_ROW = 16 * 9;  // 16-byte units per staged output row: sixteen chunks of 8 pieces + 1 of padding
template <bool INV, bool OUT_BITREV>
__global__ __launch_bounds__(256, 4) void k_ntt_pass2(const u64* __restrict__ in, size_t in_stride, u64* __restrict__ out,
                                                   size_t out_stride, const u64* __restrict__ post, u64 post_scalar,
                                                   const u64* __restrict__ tw256, unsigned log_r) {
  NTT_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) u32 lds[OUT_BITREV ? 16 * STAGE_ROW * 4 : LDS_TILE_WORDS];  // exchange image, then the store staging
  const int t = threadIdx.x;
  int g = t & 15, d = t >> 4;
  // bit-reversed output: rows k1 = b + 16 d (their images br8(k1) are 16 consecutive output rows);
  // natural output: rows k1 = 16 b + d (consecutive, so that stores along d are contiguous)
  const int k1_load = OUT_BITREV ? blockIdx.x + 16 * d : blockIdx.x * 16 + d;
  const u64* col = in + col_offset(blockIdx.y, log_r, in_stride);
  u64 x[16];
#pragma unroll
  for (int m = 0; m < 16; m++) x[m] = col[k1_load * 256 + g + 16 * m];
  dft256_tile<INV, OUT_BITREV ? 1 : 2>(x, lds, tw256, d, g);
  const int k1 = OUT_BITREV ? blockIdx.x + 16 * d : blockIdx.x * 16 + d;
  const int ka = g;
  u64* ocol = out + col_offset(blockIdx.y, log_r, out_stride);
  if (OUT_BITREV) {
    // position = br8(k1)*256 + br8(k2), k2 = ka + 16*kb  ->  br4(ka)*16 + br4(kb): x[] is already in br4(kb) order, so the
    // thread owns 16 consecutive words (chunk br4(ka) of row br4(d) of the workgroup's 16 consecutive output rows).  Stored
    // from there, a wave writes 64 separate 16-byte pieces per instruction and the lines reach HBM in parts (WRITE_SIZE 795
    // MB for 648 MB, profiles/r02_pmc_ntt.md); staged through LDS, every store instruction of a wave is 1 KB contiguous.
    __syncthreads();  // every wave is done with the exchange image (the staging image overlays it)
    ulonglong2* stage = reinterpret_cast<ulonglong2*>(lds);
    const int slot = br4(d) * STAGE_ROW + br4(ka) * 9;  // 16-byte units; chunk stride 144 B keeps the writes conflict free
#pragma unroll
    for (int p = 0; p < 16; p += 2) {
      u64 v0 = x[p], v1 = x[p + 1];
      if (post_scalar != 1) {
        v0 = gl_mul_asm(v0, post_scalar);
        v1 = gl_mul_asm(v1, post_scalar);
      }
      ulonglong2 w;
      w.x = v0;
      w.y = v1;
      stage[slot + (p >> 1)] = w;
    }
    __syncthreads();
    ulonglong2* o16 = reinterpret_cast<ulonglong2*>(ocol + (size_t)br4(blockIdx.x) * 16 * 256);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int piece = i * 256 + t;  // row = piece >> 7, chunk = (piece >> 3) & 15, part = piece & 7
      o16[piece] = stage[(piece >> 7) * STAGE_ROW + ((piece >> 3) & 15) * 9 + (piece & 7)];
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < 16; kb++) {
      int k = k1 + 256 * (ka + 16 * kb);
      u64 v = x[br4(kb)];
      if (post) v = gl_mul_asm(v, post[k]);
      else if (post_scalar != 1) v = gl_mul_asm(v, post_scalar);
      ocol[k] = v;
    }
  }
}

// ---- fused: pass 2 of the iNTT + pass 1 of both coset NTTs ------------------------------------------------------
// The tile that pass 2 of the inverse transform finishes (16 rows k1 x all k2, coefficient index k1 + 256 k2) is exactly the
// tile pass 1 of a forward transform starts from (16 adjacent columns i2 = k1 x all i1 = k2), and after the MODE-2 tile
// the lane that owns (d, ka) holds the coefficients k2 = ka + 16 kb it needs there (a register renaming, x[br4(kb)] -> m = kb).
// So the coefficients are stored once (they are kept for the openings) and the two coset transforms continue from registers:
// the commitment moves 80 N bytes per column instead of 96 N and needs four launches instead of six.
// grid = (16 tiles, ncols); y0 / y1 = pass-1 output of the cosets g and g*w_2N ([k1][i2] images for k_ntt_pass2).
template <int NPARK>
__global__ __launch_bounds__(256, NPARK ? 4 : 3) void k_ntt_intt2_lde1(const u64* __restrict__ in, size_t in_stride, u64* __restrict__ coef,
                                                        size_t coef_stride, u64* __restrict__ y0, u64* __restrict__ y1,
                                                        size_t y_stride, const u64* __restrict__ tw256_inv,
                                                        const u64* __restrict__ pre0, const u64* __restrict__ pre1,
                                                        const u64* __restrict__ twmat, const u64* __restrict__ tw256_fwd,
                                                        unsigned log_r) {
  NTT_KERNEL_PRIO();
  __shared__ u32 lds[LDS_TILE_WORDS];
  // NPARK of the thread's sixteen coefficients wait in LDS between the two coset transforms: with 8 of them there the kernel
  // needs 122 registers and 33 KB of LDS, four workgroups per CU (1.76 ms per 1237-column stage against 1.84 ms with all
  // sixteen in registers and three workgroups)
  __shared__ u64 park[NPARK ? NPARK * 256 : 1];
  u64 c[16];
  const int t = threadIdx.x;
  int g = t & 15, d = t >> 4;
  const u64* col = in + col_offset(blockIdx.y, log_r, in_stride);
  u64 x[16];
  {
    const int k1_load = blockIdx.x * 16 + d;
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = col[k1_load * 256 + g + 16 * m];
  }
  dft256_tile<true, 2>(x, lds, tw256_inv, d, g);  // now d = t & 15 (row k1 of the tile), g = t >> 4 = ka
  const int i2 = blockIdx.x * 16 + d;               // k1 of the inverse transform = matrix column i2 of the forward ones
  u64* ocol = coef + col_offset(blockIdx.y, log_r, coef_stride);
#pragma unroll
  for (int kb = 0; kb < 16; kb++) {
    c[kb] = x[br4(kb)];  // 1/N is already in the twiddle table of the pass before (twmat_inv_ninv)
    ocol[i2 + 256 * (g + 16 * kb)] = c[kb];
    if (kb >= 16 - NPARK) park[(kb - (16 - NPARK)) * 256 + t] = c[kb];
  }
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    const u64* __restrict__ pre = h ? pre1 : pre0;
    u64* ycol = (h ? y1 : y0) + col_offset(blockIdx.y, log_r, y_stride);
    asm volatile("" : "+v"(g), "+v"(d));  // the twiddle loads stay inside the loop (hoisted, they hold 62 registers)
#pragma unroll
    for (int m = 0; m < 16; m++) {
      x[m] = gl_mul_asm(m >= 16 - NPARK ? park[(m - (16 - NPARK)) * 256 + t] : c[m], pre[(g + 16 * m) * 256 + i2]);
      if ((m & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four products at a time: operands of later ones are not loaded early
    }
    __syncthreads();  // the previous tile's LDS reads are done
    dft256_tile<false, 0>(x, lds, tw256_fwd, d, g);
#pragma unroll
    for (int kb = 0; kb < 16; kb++) {
      int k1 = g + 16 * kb;
      ycol[k1 * 256 + i2] = gl_mul_asm(x[br4(kb)], twmat[k1 * 256 + i2]);
      if ((kb & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
static void fill_pow_table(std::vector<u64>& t, u64 base, size_t n, u64 first = 1) {
  t.resize(n);
  u64 v = first;
  for (size_t i = 0; i < n; i++) {
    t[i] = v;
    v = gl_mul(v, base);
  }
}

int ntt_tables_init(NttTables* T) {
  const size_t N = NTT_N;
  std::vector<u64> h;
  auto upload = [&](u64** dst, const std::vector<u64>& src) -> int {
    if (hipMalloc((void**)dst, src.size() * 8) != hipSuccess) return -1;
    if (hipMemcpy(*dst, src.data(), src.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
  };
  u64 wN = gl_root_of_unity(16), wNi = gl_inv(wN);
  u64 w256 = gl_root_of_unity(8), w256i = gl_inv(w256);
  fill_pow_table(h, w256, 256);
  if (upload(&T->tw256_fwd, h)) return -1;
  fill_pow_table(h, w256i, 256);
  if (upload(&T->tw256_inv, h)) return -1;
  // twmat[k1*256 + i2] = w_N^(i2*k1)
  for (int inv = 0; inv < 2; inv++) {
    u64 w = inv ? wNi : wN;
    h.resize(N);
    u64 wk = 1;  // w^k1
    for (int k1 = 0; k1 < 256; k1++) {
      u64 v = 1;
      for (int i2 = 0; i2 < 256; i2++) {
        h[k1 * 256 + i2] = v;
        v = gl_mul(v, wk);
      }
      wk = gl_mul(wk, w);
    }
    if (upload(inv ? &T->twmat_inv : &T->twmat_fwd, h)) return -1;
    if (inv) {  // (1/N) w_N^-(i2 k1): the fused from_values path folds the 1/N of the inverse transform into its first pass
      const u64 ninv0 = gl_inv((u64)N);
      for (auto& v : h) v = gl_mul(v, ninv0);
      if (upload(&T->twmat_inv_ninv, h)) return -1;
    }
  }
  // coset power tables: shift_h^i, shift_0 = g, shift_1 = g * w_2N
  u64 w2N = gl_root_of_unity(17);
  u64 shifts[2] = {GL_GEN, gl_mul(GL_GEN, w2N)};
  u64 ninv = gl_inv((u64)N);
  for (int hh = 0; hh < 2; hh++) {
    fill_pow_table(h, shifts[hh], N);
    if (upload(&T->coset_pow[hh], h)) return -1;
    fill_pow_table(h, gl_inv(shifts[hh]), N, ninv);  // (1/N) * shift^-k
    if (upload(&T->coset_inv_pow[hh], h)) return -1;
  }
  T->n_inv = ninv;
  return 0;
}

void ntt_tables_free(NttTables* T) {
  hipFree(T->tw256_fwd);
  hipFree(T->tw256_inv);
  hipFree(T->twmat_fwd);
  hipFree(T->twmat_inv);
  hipFree(T->twmat_inv_ninv);
  for (int h = 0; h < 2; h++) {
    hipFree(T->coset_pow[h]);
    hipFree(T->coset_inv_pow[h]);
  }
}

// values[C][N] (natural) -> coefficients[C][N] (natural); in place allowed (uses tmp[C][N]).
void ntt_inverse(const NttTables* T, const u64* values, u64* coeffs, u64* tmp, int ncols, hipStream_t s) {
  dim3 grid(16, ncols), block(256);
  k_ntt_pass1<true><<<grid, block, 0, s>>>(values, NTT_N, tmp, NTT_N, nullptr, T->twmat_inv, T->tw256_inv, 0);
  k_ntt_pass2<true, false><<<grid, block, 0, s>>>(tmp, NTT_N, coeffs, NTT_N, nullptr, T->n_inv, T->tw256_inv, 0);
}
// coset iNTT of values given on coset `h` (natural order) -> coefficients (natural)
void ntt_coset_inverse(const NttTables* T, int h, const u64* values, u64* coeffs, u64* tmp, int ncols, hipStream_t s) {
  dim3 grid(16, ncols), block(256);
  k_ntt_pass1<true><<<grid, block, 0, s>>>(values, NTT_N, tmp, NTT_N, nullptr, T->twmat_inv, T->tw256_inv, 0);
  k_ntt_pass2<true, false><<<grid, block, 0, s>>>(tmp, NTT_N, coeffs, NTT_N, T->coset_inv_pow[h], 1, T->tw256_inv, 0);
}
// coefficients[C][N] -> lde[C][2N] in bit-reversed order (both cosets); tmp[C][N].
void ntt_lde(const NttTables* T, const u64* coeffs, u64* lde, u64* tmp, int ncols, hipStream_t s) {
  dim3 grid(16, ncols), block(256);
  for (int h = 0; h < 2; h++) {
    k_ntt_pass1<false><<<grid, block, 0, s>>>(coeffs, NTT_N, tmp, NTT_N, T->coset_pow[h], T->twmat_fwd, T->tw256_fwd, 0);
    k_ntt_pass2<false, true><<<grid, block, 0, s>>>(tmp, NTT_N, lde + (size_t)h * NTT_N, 2 * NTT_N, nullptr, 1, T->tw256_fwd, 0);
  }
}
// values[C][N] -> coefficients[C][N] and lde[C][2N] (bit-reversed) with the fused middle kernel; tmp[C][N], tmp2[2][C][N].
void ntt_inverse_lde(const NttTables* T, const u64* values, u64* coeffs, u64* lde, u64* tmp, u64* tmp2, int ncols, hipStream_t s) {
  dim3 grid(16, ncols), block(256);
  u64* y0 = tmp2;
  u64* y1 = tmp2 + (size_t)ncols * NTT_N;
  k_ntt_pass1<true><<<grid, block, 0, s>>>(values, NTT_N, tmp, NTT_N, nullptr, T->twmat_inv_ninv, T->tw256_inv, 0);
  k_ntt_intt2_lde1<8><<<grid, block, 0, s>>>(tmp, NTT_N, coeffs, NTT_N, y0, y1, NTT_N, T->tw256_inv, T->coset_pow[0],
                                          T->coset_pow[1], T->twmat_fwd, T->tw256_fwd, 0);
  k_ntt_pass2<false, true><<<grid, block, 0, s>>>(y0, NTT_N, lde, 2 * NTT_N, nullptr, 1, T->tw256_fwd, 0);
  k_ntt_pass2<false, true><<<grid, block, 0, s>>>(y1, NTT_N, lde + NTT_N, 2 * NTT_N, nullptr, 1, T->tw256_fwd, 0);
}
// forward coset NTT on coset h, natural output (used for small FRI-side transforms and tests)
void ntt_coset_forward_natural(const NttTables* T, int h, const u64* coeffs, u64* values, u64* tmp, int ncols, hipStream_t s) {
  dim3 grid(16, ncols), block(256);
  k_ntt_pass1<false><<<grid, block, 0, s>>>(coeffs, NTT_N, tmp, NTT_N, T->coset_pow[h], T->twmat_fwd, T->tw256_fwd, 0);
  k_ntt_pass2<false, false><<<grid, block, 0, s>>>(tmp, NTT_N, values, NTT_N, nullptr, 1, T->tw256_fwd, 0);
}


// ---- traces taller than 2^16 rows: N = R * 2^16, R = 2^log_r <= 64 -----------------------------------------------
// An N-point transform is an R-point DFT across the R blocks of a column (shift-only twiddles, one lane per
// in-block position, all accesses contiguous across lanes) combined with the 2^16-point kernels above on each block.
//   inverse (values natural -> coefficients):  outer DIF pass first, then the block iNTTs.  Coefficient k1 + R*k2 ends
//   up at [block k1][k2] ("transposed" coefficient layout; every consumer below and in fri.hip knows it).
//   forward coset LDE (transposed coefficients -> bit-reversed values): block NTTs first (coset (shift^R)), then the
//   outer DIT pass with twiddle (shift * w_N^k2)^i1; output position bitrev(k2)*R + bitrev_r(k1) = p*R + register index.
__device__ __forceinline__ u64 mul_2exp3(u64 x, int e) {  // x * 2^(3 e) = x * w_64^e, e in [0,64); e is a compile-time constant after unrolling
  switch (e) {
    case 0: return x;
    case 1: return gl_mul_2exp<3>(x);
    case 2: return gl_mul_2exp<6>(x);
    case 3: return gl_mul_2exp<9>(x);
    case 4: return gl_mul_2exp<12>(x);
    case 5: return gl_mul_2exp<15>(x);
    case 6: return gl_mul_2exp<18>(x);
    case 7: return gl_mul_2exp<21>(x);
    case 8: return gl_mul_2exp<24>(x);
    case 9: return gl_mul_2exp<27>(x);
    case 10: return gl_mul_2exp<30>(x);
    case 11: return gl_mul_2exp<33>(x);
    case 12: return gl_mul_2exp<36>(x);
    case 13: return gl_mul_2exp<39>(x);
    case 14: return gl_mul_2exp<42>(x);
    case 15: return gl_mul_2exp<45>(x);
    case 16: return gl_mul_2exp<48>(x);
    case 17: return gl_mul_2exp<51>(x);
    case 18: return gl_mul_2exp<54>(x);
    case 19: return gl_mul_2exp<57>(x);
    case 20: return gl_mul_2exp<60>(x);
    case 21: return gl_mul_2exp<63>(x);
    case 22: return gl_mul_2exp<66>(x);
    case 23: return gl_mul_2exp<69>(x);
    case 24: return gl_mul_2exp<72>(x);
    case 25: return gl_mul_2exp<75>(x);
    case 26: return gl_mul_2exp<78>(x);
    case 27: return gl_mul_2exp<81>(x);
    case 28: return gl_mul_2exp<84>(x);
    case 29: return gl_mul_2exp<87>(x);
    case 30: return gl_mul_2exp<90>(x);
    case 31: return gl_mul_2exp<93>(x);
    case 32: return gl_mul_2exp<96>(x);
    case 33: return gl_mul_2exp<99>(x);
    case 34: return gl_mul_2exp<102>(x);
    case 35: return gl_mul_2exp<105>(x);
    case 36: return gl_mul_2exp<108>(x);
    case 37: return gl_mul_2exp<111>(x);
    case 38: return gl_mul_2exp<114>(x);
    case 39: return gl_mul_2exp<117>(x);
    case 40: return gl_mul_2exp<120>(x);
    case 41: return gl_mul_2exp<123>(x);
    case 42: return gl_mul_2exp<126>(x);
    case 43: return gl_mul_2exp<129>(x);
    case 44: return gl_mul_2exp<132>(x);
    case 45: return gl_mul_2exp<135>(x);
    case 46: return gl_mul_2exp<138>(x);
    case 47: return gl_mul_2exp<141>(x);
    case 48: return gl_mul_2exp<144>(x);
    case 49: return gl_mul_2exp<147>(x);
    case 50: return gl_mul_2exp<150>(x);
    case 51: return gl_mul_2exp<153>(x);
    case 52: return gl_mul_2exp<156>(x);
    case 53: return gl_mul_2exp<159>(x);
    case 54: return gl_mul_2exp<162>(x);
    case 55: return gl_mul_2exp<165>(x);
    case 56: return gl_mul_2exp<168>(x);
    case 57: return gl_mul_2exp<171>(x);
    case 58: return gl_mul_2exp<174>(x);
    case 59: return gl_mul_2exp<177>(x);
    case 60: return gl_mul_2exp<180>(x);
    case 61: return gl_mul_2exp<183>(x);
    case 62: return gl_mul_2exp<186>(x);
    default: return gl_mul_2exp<189>(x);
  }
}
// R-point DIF network, R = 2^LOGR <= 64; output X[k] lands in x[bitrev_LOGR(k)].  w_R = 2^(192/R) = w_64^(64/R).
template <int LOGR, bool INV>
__device__ __forceinline__ void dft_small(u64* x) {
  constexpr int R = 1 << LOGR;
#pragma unroll
  for (int s = 0; s < LOGR; s++) {
    const int span = R >> (s + 1);
#pragma unroll
    for (int g0 = 0; g0 < R; g0 += 2 * span) {
#pragma unroll
      for (int jj = 0; jj < span; jj++) {
        u64 a = x[g0 + jj], b = x[g0 + jj + span];
        x[g0 + jj] = gl_add(a, b);
        int e = (jj << s) * (64 >> LOGR);  // twiddle w_R^(jj * 2^s) = 2^(3 e)
        if (INV) e = (64 - e) & 63;
        x[g0 + jj + span] = mul_2exp3(gl_sub(a, b), e);
      }
    }
  }
}
__device__ __forceinline__ constexpr int brn(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

// values[col][i1][i2] -> y[col][k1][i2] = (1/R) * w_N^-(i2 k1) * sum_i1 x[i1][i2] w_R^-(i1 k1)
template <int LOGR>
__global__ __launch_bounds__(256) void k_ntt_outer_inv(const u64* __restrict__ in, u64* __restrict__ out, size_t N,
                                                       const u64* __restrict__ wn_inv_pow /* w_N^-i2 */, u64 r_inv) {
  constexpr int R = 1 << LOGR;
  const size_t i2 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u64* c = in + (size_t)blockIdx.y * N;
  u64* o = out + (size_t)blockIdx.y * N;
  u64 x[R];
#pragma unroll
  for (int i1 = 0; i1 < R; i1++) x[i1] = c[(size_t)i1 * NTT_N + i2];
  dft_small<LOGR, true>(x);
  const u64 w = wn_inv_pow[i2];
  u64 f = r_inv;
#pragma unroll
  for (int k1 = 0; k1 < R; k1++) {
    o[(size_t)k1 * NTT_N + i2] = gl_mul(x[brn(k1, LOGR)], f);
    f = gl_mul(f, w);
  }
}

// z[col][i1][p] (block NTT outputs, p = bitrev16(k2)) -> out[col*out_stride + p*R + bitrev_r(k1)]
template <int LOGR>
__global__ __launch_bounds__(256) void k_ntt_outer_fwd(const u64* __restrict__ z, size_t N, u64* __restrict__ out,
                                                       size_t out_stride, const u64* __restrict__ wn_pow_br /* w_N^bitrev16(p) */,
                                                       u64 shift) {
  constexpr int R = 1 << LOGR;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u64* c = z + (size_t)blockIdx.y * N;
  u64 x[R];
  const u64 b = gl_mul(shift, wn_pow_br[p]);
  u64 f = 1;
#pragma unroll
  for (int i1 = 0; i1 < R; i1++) {
    u64 v = c[(size_t)i1 * NTT_N + p];
    x[i1] = i1 ? gl_mul(v, f) : v;
    f = gl_mul(f, b);
  }
  dft_small<LOGR, false>(x);
  // A lane owns R consecutive output words; stored from there a wave writes 64 pieces of 16 bytes R*8 bytes apart per
  // instruction (the pattern that made pass 2 store-issue bound).  Staged through LDS in chunks of at most 16 words per lane,
  // eight neighbouring lanes write one 128-byte line per instruction.
  constexpr int CH = R < 16 ? R : 16;        // words per lane and round
  constexpr int U = CH / 2;                  // 16-byte units per lane and round
  constexpr int PITCH = U + 1;               // padded: conflict-free writes
  __shared__ ulonglong2 stage[256 * PITCH];
  const int t = threadIdx.x;
  ulonglong2* o16 = reinterpret_cast<ulonglong2*>(out + (size_t)blockIdx.y * out_stride + (size_t)blockIdx.x * 256 * R);
#pragma unroll
  for (int c0 = 0; c0 < R; c0 += CH) {
    if (c0) __syncthreads();
#pragma unroll
    for (int u = 0; u < U; u++) {
      ulonglong2 w2;
      w2.x = x[c0 + 2 * u];
      w2.y = x[c0 + 2 * u + 1];
      stage[t * PITCH + u] = w2;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < U; it++) {
      const int piece = it * 256 + t, owner = piece / U, part = piece % U;
      o16[(size_t)owner * (R / 2) + c0 / 2 + part] = stage[owner * PITCH + part];
    }
  }
}

// c[col][k1][k2] *= shift^-(k1 + R k2)   (coset_ifft post-scaling in the transposed layout)
__global__ __launch_bounds__(256) void k_coset_unscale(u64* __restrict__ c, size_t N, unsigned log_r, u64 shift_inv) {
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= N) return;
  const size_t k1 = pos >> 16, k2 = pos & (NTT_N - 1);
  const u64 k = k1 + (k2 << log_r);
  u64* col = c + (size_t)blockIdx.y * N;
  col[pos] = gl_mul(col[pos], gl_pow(shift_inv, k));
}

int ntt_tall_tables_init(NttTallTables* T, unsigned log_n, u64 base_shift) {
  const unsigned log_r = log_n - 16;
  const size_t N = (size_t)1 << log_n, R = (size_t)1 << log_r;
  std::vector<u64> h;
  auto upload = [&](u64** dst, const std::vector<u64>& src) -> int {
    if (hipMalloc((void**)dst, src.size() * 8) != hipSuccess) return -1;
    if (hipMemcpy(*dst, src.data(), src.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
  };
  T->log_n = log_n;
  const u64 wN = gl_root_of_unity(log_n), wNi = gl_inv(wN);
  fill_pow_table(h, wNi, NTT_N);
  if (upload(&T->wn_inv_pow, h)) return -1;
  {
    std::vector<u64> nat;
    fill_pow_table(nat, wN, NTT_N);
    h.resize(NTT_N);
    for (size_t p = 0; p < NTT_N; p++) h[p] = nat[bitrev32((u32)p, 16)];
    if (upload(&T->wn_pow_br, h)) return -1;
  }
  const u64 w2N = gl_root_of_unity(log_n + 1);
  T->shift[0] = base_shift;
  T->shift[1] = gl_mul(base_shift, w2N);
  for (int hh = 0; hh < 2; hh++) {
    fill_pow_table(h, gl_pow(T->shift[hh], R), NTT_N);  // (shift^R)^i2: coset of the block transforms
    if (upload(&T->block_coset_pow[hh], h)) return -1;
  }
  T->r_inv = gl_inv((u64)R);
  (void)N;
  return 0;
}
void ntt_tall_tables_free(NttTallTables* T) {
  hipFree(T->wn_inv_pow);
  hipFree(T->wn_pow_br);
  hipFree(T->block_coset_pow[0]);
  hipFree(T->block_coset_pow[1]);
}

template <int LOGR>
static void outer_inv_launch(const u64* in, u64* out, size_t N, const NttTallTables* TT, int ncols, hipStream_t s) {
  k_ntt_outer_inv<LOGR><<<dim3(NTT_N / 256, ncols), 256, 0, s>>>(in, out, N, TT->wn_inv_pow, TT->r_inv);
}
template <int LOGR>
static void outer_fwd_launch(const u64* z, size_t N, u64* out, size_t out_stride, const NttTallTables* TT, int h, int ncols,
                             hipStream_t s) {
  k_ntt_outer_fwd<LOGR><<<dim3(NTT_N / 256, ncols), 256, 0, s>>>(z, N, out, out_stride, TT->wn_pow_br, TT->shift[h]);
}

// values[C][N] natural -> coefficients[C][N] in the transposed layout; may run in place; tmp[C][N].
void ntt_inverse_tall(const NttTables* T, const NttTallTables* TT, const u64* values, u64* coeffs, u64* tmp, int ncols,
                      hipStream_t s) {
  const unsigned log_r = TT->log_n - 16;
  const size_t N = (size_t)1 << TT->log_n;
  switch (log_r) {
    case 1: outer_inv_launch<1>(values, coeffs, N, TT, ncols, s); break;
    case 2: outer_inv_launch<2>(values, coeffs, N, TT, ncols, s); break;
    case 3: outer_inv_launch<3>(values, coeffs, N, TT, ncols, s); break;
    case 4: outer_inv_launch<4>(values, coeffs, N, TT, ncols, s); break;
    case 5: outer_inv_launch<5>(values, coeffs, N, TT, ncols, s); break;
    default: outer_inv_launch<6>(values, coeffs, N, TT, ncols, s); break;
  }
  dim3 grid(16, (unsigned)(ncols << log_r)), block(256);
  k_ntt_pass1<true><<<grid, block, 0, s>>>(coeffs, N, tmp, N, nullptr, T->twmat_inv, T->tw256_inv, log_r);
  k_ntt_pass2<true, false><<<grid, block, 0, s>>>(tmp, N, coeffs, N, nullptr, T->n_inv, T->tw256_inv, log_r);
}
// values on coset h (natural) -> transposed coefficients of the interpolant
void ntt_coset_inverse_tall(const NttTables* T, const NttTallTables* TT, int h, const u64* values, u64* coeffs, u64* tmp, int ncols,
                            hipStream_t s) {
  ntt_inverse_tall(T, TT, values, coeffs, tmp, ncols, s);
  const size_t N = (size_t)1 << TT->log_n;
  k_coset_unscale<<<dim3((unsigned)(N / 256), ncols), 256, 0, s>>>(coeffs, N, TT->log_n - 16, gl_inv(TT->shift[h]));
}
// from_values of a tall commitment: outer inverse pass, block iNTT pass 1, the fused middle kernel (its two pass-1 images go
// straight into the two halves of the LDE buffer), then pass 2 + outer forward pass per coset.  values == coeffs allowed.
void ntt_inverse_lde_tall(const NttTables* T, const NttTallTables* TT, const u64* values, u64* coeffs, u64* lde, u64* tmp, int ncols,
                          hipStream_t s) {
  const unsigned log_r = TT->log_n - 16;
  const size_t N = (size_t)1 << TT->log_n;
  switch (log_r) {
    case 1: outer_inv_launch<1>(values, coeffs, N, TT, ncols, s); break;
    case 2: outer_inv_launch<2>(values, coeffs, N, TT, ncols, s); break;
    case 3: outer_inv_launch<3>(values, coeffs, N, TT, ncols, s); break;
    case 4: outer_inv_launch<4>(values, coeffs, N, TT, ncols, s); break;
    case 5: outer_inv_launch<5>(values, coeffs, N, TT, ncols, s); break;
    default: outer_inv_launch<6>(values, coeffs, N, TT, ncols, s); break;
  }
  dim3 grid(16, (unsigned)(ncols << log_r)), block(256);
  k_ntt_pass1<true><<<grid, block, 0, s>>>(coeffs, N, tmp, N, nullptr, T->twmat_inv_ninv, T->tw256_inv, log_r);
  k_ntt_intt2_lde1<8><<<grid, block, 0, s>>>(tmp, N, coeffs, N, lde, lde + N, 2 * N, T->tw256_inv, TT->block_coset_pow[0],
                                          TT->block_coset_pow[1], T->twmat_fwd, T->tw256_fwd, log_r);
  for (int h = 0; h < 2; h++) {
    u64* half = lde + (size_t)h * N;
    k_ntt_pass2<false, true><<<grid, block, 0, s>>>(half, 2 * N, tmp, N, nullptr, 1, T->tw256_fwd, log_r);
    switch (log_r) {
      case 1: outer_fwd_launch<1>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 2: outer_fwd_launch<2>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 3: outer_fwd_launch<3>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 4: outer_fwd_launch<4>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 5: outer_fwd_launch<5>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      default: outer_fwd_launch<6>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
    }
  }
}
// transposed coefficients[C][N] -> lde[C][2N] bit-reversed; tmp[C][N].
void ntt_lde_tall(const NttTables* T, const NttTallTables* TT, const u64* coeffs, u64* lde, u64* tmp, int ncols, hipStream_t s,
                  int coset_mask) {
  const unsigned log_r = TT->log_n - 16;
  const size_t N = (size_t)1 << TT->log_n;
  dim3 grid(16, (unsigned)(ncols << log_r)), block(256);
  for (int h = 0; h < 2; h++) {
    if (!((coset_mask >> h) & 1)) continue;  // (the streaming quotient needs one coset at a time)
    u64* half = lde + (size_t)h * N;
    k_ntt_pass1<false><<<grid, block, 0, s>>>(coeffs, N, half, 2 * N, TT->block_coset_pow[h], T->twmat_fwd, T->tw256_fwd, log_r);
    k_ntt_pass2<false, true><<<grid, block, 0, s>>>(half, 2 * N, tmp, N, nullptr, 1, T->tw256_fwd, log_r);
    switch (log_r) {
      case 1: outer_fwd_launch<1>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 2: outer_fwd_launch<2>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 3: outer_fwd_launch<3>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 4: outer_fwd_launch<4>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      case 5: outer_fwd_launch<5>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
      default: outer_fwd_launch<6>(tmp, N, half, 2 * N, TT, h, ncols, s); break;
    }
  }
}

// ---- one radix-2 level above the tall transforms: N = 2 H (2^23-row traces: H = 2^22) --------------------------------
// P(x) = Pe(x^2) + x Po(x^2).  A column of N words holds the two halves [Pe | Po], each in the tall layout of an H-point
// transform, so every tall routine above runs on "2 C columns of H words" unchanged:
//   values -> halves:  Y0[p] = (x[p] + x[p+H]) / 2 = Pe(w_H^p),  Y1[p] = (x[p] - x[p+H]) w_N^-p / 2 = Po(w_H^p)      (in place)
//   LDE: the point g w_2N^(2k+h) squares to (g^2 w_N^h) w_H^k, so the half transforms run on the cosets g^2, g^2 w_N (tall
//   tables built with base shift g^2) and out[h N + 2q + {0,1}] = E[h H + q] +- t O[h H + q], t = g w_2N^h w_N^bitrev(q):
//   w_2N^(2(k+H)+h) = -w_2N^(2k+h) and bitrev(k + H) = bitrev(k) + 1, the two results are neighbours.
// w_N^e for e < 2^22 comes from two 2048-entry tables (e = 2048 a + b).
__device__ __forceinline__ u64 split_pow(const u64* __restrict__ hi, const u64* __restrict__ lo, u32 e) {
  return gl_mul(hi[e >> 11], lo[e & 2047]);
}
__global__ __launch_bounds__(256) void k_split_inv(const u64* __restrict__ in, u64* __restrict__ out, size_t N,
                                                   const u64* __restrict__ hi, const u64* __restrict__ lo, u64 half, u64 odd_scale) {
  const size_t H = N >> 1, p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u64* x = in + (size_t)blockIdx.y * N;
  u64* y = out + (size_t)blockIdx.y * N;
  const u64 a = x[p], b = x[p + H];
  y[p] = gl_mul(gl_add(a, b), half);
  y[p + H] = gl_mul(gl_mul(gl_sub(a, b), split_pow(hi, lo, (u32)p)), odd_scale);  // odd_scale = 1/2 (or s^-1 / 2 on a coset)
}
__global__ __launch_bounds__(256) void k_split_fwd(const u64* __restrict__ eo, u64* __restrict__ out, size_t N, unsigned log_h,
                                                   size_t out_stride, const u64* __restrict__ hi, const u64* __restrict__ lo, u64 shift0,
                                                   u64 shift1, size_t i0) {
  const size_t H = N >> 1, i = i0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // i = h H + q < N
  const u32 h = (u32)(i >> log_h), q = (u32)(i & (H - 1));
  const u64* e = eo + (size_t)blockIdx.y * 2 * N;  // [E (N words) | O (N words)]: the LDEs of the two halves
  const u32 k = bitrev32(q, log_h);
  const u64 t = gl_mul(h ? shift1 : shift0, split_pow(hi, lo, k));
  const u64 a = e[i], b = gl_mul(e[N + i], t);
  ulonglong2 w;
  w.x = gl_add(a, b);
  w.y = gl_sub(a, b);
  *reinterpret_cast<ulonglong2*>(out + (size_t)blockIdx.y * out_stride + (size_t)h * N + 2 * (size_t)q) = w;
}
// c[half][e] *= base^-e (e = k1 + R k2 in the tall layout): the unscaling of a coset interpolant whose halves hold q_(2e+par) s^(2e+par)
__global__ __launch_bounds__(256) void k_coset_unscale_half(u64* __restrict__ c, size_t H, unsigned log_r, u64 base_inv) {
  const size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t k1 = pos >> 16, k2 = pos & (NTT_N - 1);
  u64* col = c + (size_t)blockIdx.y * H;
  col[pos] = gl_mul(col[pos], gl_pow(base_inv, k1 + (k2 << log_r)));
}

void ntt_split_tables_free(NttSplitTables* S);
int ntt_split_tables_init(NttSplitTables* S, unsigned log_n) {
  if (log_n < 18 || log_n > 23) return -1;  // e < 2^22 in split_pow
  S->log_n = log_n;
  const u64 w = gl_root_of_unity(log_n), wi = gl_inv(w);
  std::vector<u64> h;
  auto upload = [&](u64** dst) -> int {
    if (hipMalloc((void**)dst, h.size() * 8) != hipSuccess) return -1;
    return hipMemcpy(*dst, h.data(), h.size() * 8, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
  };
  S->fwd_lo = S->fwd_hi = S->inv_lo = S->inv_hi = nullptr;
  const u64 bases[4] = {w, gl_pow(w, 2048), wi, gl_pow(wi, 2048)};
  u64** dsts[4] = {&S->fwd_lo, &S->fwd_hi, &S->inv_lo, &S->inv_hi};
  for (int t = 0; t < 4; t++) {
    fill_pow_table(h, bases[t], 2048);
    if (upload(dsts[t])) {
      ntt_split_tables_free(S);  // the tables uploaded so far (hipFree(nullptr) is a no-op)
      S->fwd_lo = S->fwd_hi = S->inv_lo = S->inv_hi = nullptr;
      return -1;
    }
  }
  const u64 w2N = gl_root_of_unity(log_n + 1);
  S->shift[0] = GL_GEN;
  S->shift[1] = gl_mul(GL_GEN, w2N);
  S->half = gl_inv(2);
  return 0;
}
void ntt_split_tables_free(NttSplitTables* S) {
  hipFree(S->fwd_lo);
  hipFree(S->fwd_hi);
  hipFree(S->inv_lo);
  hipFree(S->inv_hi);
}
// values[C][N] -> halves (in place allowed); the caller then runs the tall routines on 2 C columns of H words
void ntt_split_inverse(const NttSplitTables* S, const u64* values, u64* halves, int ncols, hipStream_t s) {
  const size_t N = (size_t)1 << S->log_n;
  k_split_inv<<<dim3((unsigned)(N / 512), ncols), 256, 0, s>>>(values, halves, N, S->inv_hi, S->inv_lo, S->half, S->half);
}
// eo[C][2][N] (bit-reversed LDEs of the two halves of each column) -> lde[C][2N] bit-reversed
void ntt_split_forward(const NttSplitTables* S, const u64* eo, u64* lde, size_t lde_stride, int ncols, hipStream_t s, int coset_mask) {
  const size_t N = (size_t)1 << S->log_n, H = N >> 1;
  if (coset_mask == 3) {
    k_split_fwd<<<dim3((unsigned)(N / 256), ncols), 256, 0, s>>>(eo, lde, N, S->log_n - 1, lde_stride, S->fwd_hi, S->fwd_lo, S->shift[0], S->shift[1], 0);
    return;
  }
  const int h = coset_mask == 2 ? 1 : 0;  // one coset: the i = h H + q half of the index range
  k_split_fwd<<<dim3((unsigned)(H / 256), ncols), 256, 0, s>>>(eo, lde, N, S->log_n - 1, lde_stride, S->fwd_hi, S->fwd_lo, S->shift[0], S->shift[1],
                                                             (size_t)h * H);
}
// values on coset h of the N-point domain (natural order) -> the halves of the interpolant's coefficients; TT = tall tables of H
void ntt_coset_inverse_split(const NttTables* T, const NttTallTables* TT, const NttSplitTables* S, int h, const u64* values, u64* coeffs,
                             u64* tmp, int ncols, hipStream_t s) {
  const size_t N = (size_t)1 << S->log_n, H = N >> 1;
  const u64 si = gl_inv(S->shift[h]);
  // Q(s x) has coefficients q_k s^k: halves hold q_(2e) s^(2e) and q_(2e+1) s^(2e+1); the odd half's s^-1 rides on its 1/2
  k_split_inv<<<dim3((unsigned)(N / 512), ncols), 256, 0, s>>>(values, coeffs, N, S->inv_hi, S->inv_lo, S->half, gl_mul(S->half, si));
  ntt_inverse_tall(T, TT, coeffs, coeffs, tmp, 2 * ncols, s);
  k_coset_unscale_half<<<dim3((unsigned)(H / 256), 2 * ncols), 256, 0, s>>>(coeffs, H, TT->log_n - 16, gl_mul(si, si));
}

// ---- self test of the hand-written field sequences (gl_asm.h) -------------------------------------------------------
// out[i][0..FIELD_SELFTEST_OUTS): a+b, a-b, b-a, a*b, then a 2^S for S in GL_SELFTEST_SHIFTS, then a 2^-K for K in GL_SELFTEST_RSHIFTS
__global__ void k_field_selftest(const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 x = a[i], y = b[i];
  u64* o = out + i * FIELD_SELFTEST_OUTS;
  u64 s, d, s2, d2;
  gl_bfly_asm<false>(x, y, s, d);
  gl_bfly_asm<true>(x, y, s2, d2);
  o[0] = s;
  o[1] = d;
  o[2] = d2;
  o[3] = s2 == s ? gl_mul_asm(x, y) : ~0ull;
  o[4] = gl_shl_asm<12>(x);
  o[5] = gl_shl_asm<24>(x);
  o[6] = gl_shl_asm<32>(x);
  o[7] = gl_shl_asm<36>(x);
  o[8] = gl_shl_asm<48>(x);
  o[9] = gl_shl_asm<60>(x);
  o[10] = gl_shl_asm<64>(x);
  o[11] = gl_shl_asm<1>(x);
  o[12] = gl_shl_asm<31>(x);
  o[13] = gl_shr_small_asm<12>(x);
  o[14] = gl_shr_small_asm<24>(x);
  o[15] = gl_shr_small_asm<1>(x);
  o[16] = gl_shr_small_asm<31>(x);
}
void field_selftest(const u64* a, const u64* b, u64* out, size_t n, hipStream_t s) {
  k_field_selftest<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a, b, out, n);
}

// loads this translation unit's code object (the HIP runtime defers that to the first launch otherwise)
void ntt_module_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_coset_unscale));
}
