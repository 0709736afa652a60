// Openings and FRI for gfx950 (K-eval, K-fricombine, K-frifold, K-pow, query gather of DESIGN.md).
//
// Replaces (un-vendored starky 0.4.0 / plonky2 0.2.2, reached from reference src/starks/common/prover.rs:55-65):
//   StarkOpeningSet::new            - every committed polynomial at zeta and g*zeta (F2), CTL Z's at 1
//   PolynomialBatch::prove_openings - alpha-batched quotients (F_b(X) - F_b(z_b)) / (X - z_b)
//   fri_committed_trees             - arity-16 folding with a Merkle tree per layer
//   fri_proof_of_work               - grinding (smallest witness; upstream takes any)
//   fri_prover_query_rounds         - leaf rows and Merkle paths at the 84 query indices
// The reference prover works on coefficients (divide_by_linear, reduce_with_powers, coset FFT per layer).
// Here everything after the openings stays in the evaluation domain: the batched quotient is formed
// point-wise on the 2N-point coset from the LDE values that are already resident in HBM, and a layer is
// folded by interpolating each 16-point coset {x w_16^i} (one leaf = 16 consecutive bit-reversed values)
// and evaluating at beta.  Both give the same field elements as the coefficient route (the folded
// polynomial sum_j beta^j P_j(Y) restricted to Y = x^16), with no extension-field FFT at all.
#include "fri.h"
#include "poseidon_dev.h"

// ---- openings -----------------------------------------------------------------------------------------------
// Coefficient layout: coefficient k1 + R*k2 at position k1*M + k2 (R = N / 2^16 blocks of M = 2^16; R = 1 is the
// plain natural order).  P(z) = sum_k1 z^k1 * sum_k2 c[k1][k2] (z^R)^k2: one 256-thread block per (polynomial, k1)
// evaluates the inner sum at zeta^R and (g zeta)^R; the host combines the R partial values.
// Inside a block lane t owns the coefficients t + 256 q: S(z) = sum_t z^t * sum_q c[t + 256 q] (z^256)^q.  The powers
// (z^256)^q are the same for every lane and every polynomial, so they come from a 256-entry table through scalar loads and
// a coefficient costs four base-field multiply-accumulates into un-reduced 128+32-bit sums (no extension-field Horner).
// out[(p*R + k1)*5 ..] = {S0.c0, S0.c1, S1.c0, S1.c1, sum of the block's coefficients}
static constexpr int OPEN_Q = (int)(65536 / 256);

// zq3[(pt*OPEN_Q + q)*8 ..] = 22-bit limbs of the two components of ((z_pt^R)^256)^q (3 + 3 u32, padded to 8),
// zt[(pt*256 + t)*2 ..] = (z_pt^R)^t
__global__ __launch_bounds__(256) void k_opening_tables(gl2 z0r, gl2 z1r, u64* __restrict__ zq, u64* __restrict__ zt) {
  LATENCY_KERNEL_PRIO();
  const int t = threadIdx.x;
  u32* zq3 = reinterpret_cast<u32*>(zq);
  for (int pt = 0; pt < 2; pt++) {
    const gl2 z = pt ? z1r : z0r;
    gl2 a = gl2_pow(gl2_pow(z, 256), (u64)t), b = gl2_pow(z, (u64)t);
    const W3 a0 = w3_split(a.c0), a1 = w3_split(a.c1);
    u32* e = zq3 + (size_t)(pt * OPEN_Q + t) * 8;
    e[0] = a0.w0; e[1] = a0.w1; e[2] = a0.w2; e[3] = 0;
    e[4] = a1.w0; e[5] = a1.w1; e[6] = a1.w2; e[7] = 0;
    zt[(pt * 256 + t) * 2] = b.c0;
    zt[(pt * 256 + t) * 2 + 1] = b.c1;
  }
}

__global__ __launch_bounds__(256) void k_openings(const u64* __restrict__ coeffs, size_t N, unsigned log_r, const u64* __restrict__ zq,
                                                  const u64* __restrict__ zt, u64* __restrict__ out) {
  __shared__ u64 red[256 * 5];
  const int t = threadIdx.x;
  const size_t M = N >> log_r;  // = 65536
  const u64* c = coeffs + (size_t)blockIdx.x * N + (size_t)blockIdx.y * M;
  Acc3 a00, a01, a10, a11, as;  // (point, component); `as` sums the coefficients themselves (weight 1)
  acc3_init(a00);
  acc3_init(a01);
  acc3_init(a10);
  acc3_init(a11);
  u64 s0 = 0, s1 = 0, s2 = 0;
  const u32* zq3 = reinterpret_cast<const u32*>(zq);
#pragma unroll 8
  for (int q = 0; q < OPEN_Q; q++) {
    const u64 v = c[(size_t)t + 256 * (size_t)q];
    const u32 v0 = (u32)v & M22, v1 = (u32)(v >> 22) & M22, v2 = (u32)(v >> 44);
    const u32* e0 = zq3 + (size_t)q * 8;
    const u32* e1 = zq3 + (size_t)(OPEN_Q + q) * 8;
    acc3_mad(a00, v0, v1, v2, e0[0], e0[1], e0[2]);
    acc3_mad(a01, v0, v1, v2, e0[4], e0[5], e0[6]);
    acc3_mad(a10, v0, v1, v2, e1[0], e1[1], e1[2]);
    acc3_mad(a11, v0, v1, v2, e1[4], e1[5], e1[6]);
    s0 += v0;
    s1 += v1;
    s2 += v2;
  }
  acc3_init(as);
  as.c[0] = s0;
  as.c[1] = s1;
  as.c[2] = s2;
  gl2 r0 = gl2_mul(gl2_make(acc3_red(a00), acc3_red(a01)), gl2_make(zt[2 * t], zt[2 * t + 1]));
  gl2 r1 = gl2_mul(gl2_make(acc3_red(a10), acc3_red(a11)), gl2_make(zt[2 * (256 + t)], zt[2 * (256 + t) + 1]));
  red[t] = r0.c0;
  red[256 + t] = r0.c1;
  red[512 + t] = r1.c0;
  red[768 + t] = r1.c1;
  red[1024 + t] = acc3_red(as);
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off)
      for (int k = 0; k < 5; k++) red[k * 256 + t] = gl_add(red[k * 256 + t], red[k * 256 + t + off]);
    __syncthreads();
  }
  if (t < 5) out[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 5 + t] = red[t * 256];
}

void fri_opening_tables(unsigned log_r, gl2 zeta, gl2 zeta_next, u64* d_tables, hipStream_t st) {
  const u64 R = (u64)1 << log_r;
  k_opening_tables<<<1, 256, 0, st>>>(gl2_pow(zeta, R), gl2_pow(zeta_next, R), d_tables, d_tables + FRI_OPENING_ZQ_WORDS);
}
void fri_openings(const u64* d_coeffs, size_t N, unsigned log_r, int npolys, const u64* d_tables, u64* d_out, hipStream_t st) {
  const u64 R = (u64)1 << log_r;
  k_openings<<<dim3(npolys, (unsigned)R), 256, 0, st>>>(d_coeffs, N, log_r, d_tables, d_tables + FRI_OPENING_ZQ_WORDS, d_out);
}

// ---- batched quotient on the LDE domain ---------------------------------------------------------------
struct CombineArgs {
  const u64* tl;   // trace LDE [W][2N]
  const u64* al;   // aux LDE [A][2N]
  const u64* ql;   // quotient LDE [4][2N]
  const u32* apow3; // alpha^j cut in 22-bit limbs: 8 u32 per j = (c0: w0 w1 w2 -, c1: w0 w1 w2 -), j < W + A + 4
  const u64* xs;   // x_j (bit-reversed order)
  int W, A, num_lookup;
  gl2 zeta, zeta_next;
  gl2 r0, r1, r2;      // F_b(z_b) from the openings
  gl2 a_n1n2, a_n2;    // alpha^(n1+n2), alpha^(n2)
  u64* out;            // [2N][2] extension values, bit-reversed order
  size_t M2;
};

__global__ __launch_bounds__(256) void k_fri_combine(CombineArgs A) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.M2) return;
  const size_t M2 = A.M2;
  // sum alpha^j v_j, components c0 / c1 (Acc3: carry-free columns, weights through scalar loads)
  Acc3 g0, g1, z0, z1, q0, q1;
  acc3_init(g0);
  acc3_init(g1);
  acc3_init(z0);
  acc3_init(z1);
  acc3_init(q0);
  acc3_init(q1);
  auto mad2 = [&](Acc3& a0, Acc3& a1, u64 v, const u32* e) {
    const u32 v0 = (u32)v & M22, v1 = (u32)(v >> 22) & M22, v2 = (u32)(v >> 44);
    acc3_mad(a0, v0, v1, v2, e[0], e[1], e[2]);
    acc3_mad(a1, v0, v1, v2, e[4], e[5], e[6]);
  };
#pragma unroll 8
  for (int c = 0; c < A.W; c++) mad2(g0, g1, A.tl[(size_t)c * M2 + j], A.apow3 + 8 * (size_t)c);  // unrolled: loads in flight
  const u32* ap = A.apow3 + 8 * (size_t)A.W;
#pragma unroll 8
  for (int c = 0; c < A.num_lookup; c++) mad2(g0, g1, A.al[(size_t)c * M2 + j], ap + 8 * (size_t)c);
  for (int c = A.num_lookup; c < A.A; c++) {  // the CTL Z columns also enter the batch opened at 1
    u64 v = A.al[(size_t)c * M2 + j];
    mad2(g0, g1, v, ap + 8 * (size_t)c);
    mad2(z0, z1, v, A.apow3 + 8 * (size_t)(c - A.num_lookup));
  }
  gl2 f1 = gl2_make(acc3_red(g0), acc3_red(g1));
  ap = A.apow3 + 8 * (size_t)(A.W + A.A);
  for (int c = 0; c < 4; c++) mad2(q0, q1, A.ql[(size_t)c * M2 + j], ap + 8 * (size_t)c);
  gl2 f0 = gl2_add(f1, gl2_make(acc3_red(q0), acc3_red(q1)));
  gl2 f2 = gl2_make(acc3_red(z0), acc3_red(z1));
  const u64 x = A.xs[j];
  gl2 xe = gl2_make(x, 0);
  gl2 t0 = gl2_mul(gl2_sub(f0, A.r0), gl2_inv(gl2_sub(xe, A.zeta)));
  gl2 t1 = gl2_mul(gl2_sub(f1, A.r1), gl2_inv(gl2_sub(xe, A.zeta_next)));
  gl2 t2 = gl2_mul_base(gl2_sub(f2, A.r2), gl_inv(gl_sub(x, 1)));
  gl2 r = gl2_add(gl2_add(gl2_mul(t0, A.a_n1n2), gl2_mul(t1, A.a_n2)), t2);
  reinterpret_cast<ulonglong2*>(A.out)[j] = make_ulonglong2(r.c0, r.c1);
}

// ---- the same batch polynomial from COEFFICIENTS (prover.hip "stream": no LDE of the commitments is resident) -----------------
// The three alpha-weighted sums of k_fri_combine are linear in the polynomials, so they can be formed on the coefficient vectors
// (all commitments share one coefficient layout): comb[0,1] = f1 (trace + aux), comb[2,3] = the quotient part of f0, comb[4,5]
// = f2 (CTL Z columns), components c0 / c1.  The six columns then go through the ordinary LDE and k_fri_combine_final applies the
// point-wise part.  Same field elements as k_fri_combine (sums of the same products).
__global__ __launch_bounds__(256) void k_fri_combine_coeffs(CombineArgs A, u64* __restrict__ comb) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.M2) return;
  const size_t M2 = A.M2;  // (= N here: the arrays are coefficient vectors)
  Acc3 g0, g1, z0, z1, q0, q1;
  acc3_init(g0);
  acc3_init(g1);
  acc3_init(z0);
  acc3_init(z1);
  acc3_init(q0);
  acc3_init(q1);
  auto mad2 = [&](Acc3& a0, Acc3& a1, u64 v, const u32* e) {
    const u32 v0 = (u32)v & M22, v1 = (u32)(v >> 22) & M22, v2 = (u32)(v >> 44);
    acc3_mad(a0, v0, v1, v2, e[0], e[1], e[2]);
    acc3_mad(a1, v0, v1, v2, e[4], e[5], e[6]);
  };
#pragma unroll 8
  for (int c = 0; c < A.W; c++) mad2(g0, g1, A.tl[(size_t)c * M2 + j], A.apow3 + 8 * (size_t)c);
  const u32* ap = A.apow3 + 8 * (size_t)A.W;
#pragma unroll 8
  for (int c = 0; c < A.num_lookup; c++) mad2(g0, g1, A.al[(size_t)c * M2 + j], ap + 8 * (size_t)c);
  for (int c = A.num_lookup; c < A.A; c++) {
    u64 v = A.al[(size_t)c * M2 + j];
    mad2(g0, g1, v, ap + 8 * (size_t)c);
    mad2(z0, z1, v, A.apow3 + 8 * (size_t)(c - A.num_lookup));
  }
  ap = A.apow3 + 8 * (size_t)(A.W + A.A);
  for (int c = 0; c < 4; c++) mad2(q0, q1, A.ql[(size_t)c * M2 + j], ap + 8 * (size_t)c);
  comb[0 * M2 + j] = acc3_red(g0);
  comb[1 * M2 + j] = acc3_red(g1);
  comb[2 * M2 + j] = acc3_red(q0);
  comb[3 * M2 + j] = acc3_red(q1);
  comb[4 * M2 + j] = acc3_red(z0);
  comb[5 * M2 + j] = acc3_red(z1);
}
__global__ __launch_bounds__(256) void k_fri_combine_final(CombineArgs A, const u64* __restrict__ cl /* [6][M2] LDE of comb */) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.M2) return;
  const size_t M2 = A.M2;
  const gl2 f1 = gl2_make(cl[0 * M2 + j], cl[1 * M2 + j]);
  const gl2 f0 = gl2_add(f1, gl2_make(cl[2 * M2 + j], cl[3 * M2 + j]));
  const gl2 f2 = gl2_make(cl[4 * M2 + j], cl[5 * M2 + j]);
  const u64 x = A.xs[j];
  gl2 xe = gl2_make(x, 0);
  gl2 t0 = gl2_mul(gl2_sub(f0, A.r0), gl2_inv(gl2_sub(xe, A.zeta)));
  gl2 t1 = gl2_mul(gl2_sub(f1, A.r1), gl2_inv(gl2_sub(xe, A.zeta_next)));
  gl2 t2 = gl2_mul_base(gl2_sub(f2, A.r2), gl_inv(gl_sub(x, 1)));
  gl2 r = gl2_add(gl2_add(gl2_mul(t0, A.a_n1n2), gl2_mul(t1, A.a_n2)), t2);
  reinterpret_cast<ulonglong2*>(A.out)[j] = make_ulonglong2(r.c0, r.c1);
}
static CombineArgs combine_args(const StarkShape& sh, const u64* d_tl, const u64* d_al, const u64* d_ql, const u64* d_apow, const u64* d_xs,
                                gl2 zeta, gl2 zeta_next, gl2 r0, gl2 r1, gl2 r2, gl2 alpha, size_t M2, u64* d_out) {
  CombineArgs A;
  A.tl = d_tl;
  A.al = d_al;
  A.ql = d_ql;
  A.apow3 = reinterpret_cast<const u32*>(d_apow);
  A.xs = d_xs;
  A.W = sh.W;
  A.A = sh.n_aux();
  A.num_lookup = sh.n_lookup_cols();
  A.zeta = zeta;
  A.zeta_next = zeta_next;
  A.r0 = r0;
  A.r1 = r1;
  A.r2 = r2;
  const int n1 = sh.W + sh.n_aux(), n2 = 2 * sh.n_ctl;
  A.a_n2 = gl2_pow(alpha, n2);
  A.a_n1n2 = gl2_pow(alpha, n1 + n2);
  A.out = d_out;
  A.M2 = M2;
  return A;
}
void fri_combine_coeffs(const StarkShape& sh, const u64* d_tcoef, const u64* d_acoef, const u64* d_qcoef, const u64* d_apow, size_t N,
                        u64* d_comb, hipStream_t st) {
  const gl2 z = gl2_make(0, 0);
  CombineArgs A = combine_args(sh, d_tcoef, d_acoef, d_qcoef, d_apow, nullptr, z, z, z, z, z, gl2_make(1, 0), N, nullptr);
  k_fri_combine_coeffs<<<(unsigned)((N + 255) / 256), 256, 0, st>>>(A, d_comb);
}
void fri_combine_final(const StarkShape& sh, const u64* d_comb_lde, const u64* d_xs, gl2 zeta, gl2 zeta_next, gl2 r0, gl2 r1, gl2 r2,
                       gl2 alpha, size_t M2, u64* d_out, hipStream_t st) {
  CombineArgs A = combine_args(sh, nullptr, nullptr, nullptr, nullptr, d_xs, zeta, zeta_next, r0, r1, r2, alpha, M2, d_out);
  k_fri_combine_final<<<(unsigned)((M2 + 255) / 256), 256, 0, st>>>(A, d_comb_lde);
}

// rows k0 .. k0 + rows - 1 (mod N) of coset h of `ncols` LDE columns (leaf order, stride M2) in natural order: out[c][t]
__global__ __launch_bounds__(256) void k_extract_window(const u64* __restrict__ lde, size_t M2, unsigned log_n, u32 h, size_t k0,
                                                        size_t rows, u64* __restrict__ out) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows) return;
  const size_t N = (size_t)1 << log_n;
  const u32 k = (u32)((k0 + t) & (N - 1));
  out[(size_t)blockIdx.y * rows + t] = lde[(size_t)blockIdx.y * M2 + ((size_t)h << log_n) + bitrev32(k, log_n)];
}
void fri_extract_window(const u64* d_lde, size_t M2, unsigned log_n, int h, size_t k0, size_t rows, int ncols, u64* d_out, hipStream_t st) {
  k_extract_window<<<dim3((unsigned)((rows + 255) / 256), ncols), 256, 0, st>>>(d_lde, M2, log_n, (u32)h, k0, rows, d_out);
}
// out[c][q] = lde[c][indices[q]]: the leaf rows of the queries of `ncols` columns
__global__ void k_gather_rows(const u64* __restrict__ lde, size_t M2, const u32* __restrict__ indices, int nq, u64* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  out[(size_t)blockIdx.y * nq + q] = lde[(size_t)blockIdx.y * M2 + indices[q]];
}
void fri_gather_rows(const u64* d_lde, size_t M2, int ncols, const u32* d_indices, int nq, u64* d_out, hipStream_t st) {
  k_gather_rows<<<dim3((unsigned)((nq + 63) / 64), ncols), 64, 0, st>>>(d_lde, M2, d_indices, nq, d_out);
}

void fri_combine(const StarkShape& sh, const u64* d_tl, const u64* d_al, const u64* d_ql, const u64* d_apow, const u64* d_xs,
                 gl2 zeta, gl2 zeta_next, gl2 r0, gl2 r1, gl2 r2, gl2 alpha, size_t M2, u64* d_out, hipStream_t st) {
  CombineArgs A = combine_args(sh, d_tl, d_al, d_ql, d_apow, d_xs, zeta, zeta_next, r0, r1, r2, alpha, M2, d_out);
  k_fri_combine<<<(unsigned)((M2 + 255) / 256), 256, 0, st>>>(A);
}

// ---- arity-16 fold in the evaluation domain ---------------------------------------------------------------
// in: M extension values in bit-reversed order on the coset shift*<w_M>; out: M/16 values on shift^16*<w_{M/16}>.
// inverse 16-point DFT with w_16^-1 = 2^180 (shifts only), component-wise on (c0, c1).
template <int E>
__device__ __forceinline__ u64 mul_w16inv(u64 x) {
  if constexpr (E == 0) return x;
  else return gl_mul_2exp<192 - 12 * E>(x);
}
template <int SPAN, int G, int J>
__device__ __forceinline__ void ibfly(u64* x) {
  constexpr int E = J * (8 / SPAN);
  u64 a = x[G + J], b = x[G + J + SPAN];
  x[G + J] = gl_add(a, b);
  if constexpr (E == 0) x[G + J + SPAN] = gl_sub(a, b);
  else x[G + J + SPAN] = gl_mul_2exp<96 - 12 * E>(gl_sub(b, a));  // w_16^-E = 2^(192-12E) = -2^(96-12E)
}
__device__ __forceinline__ void idft16(u64* x) {  // natural in, X[k] in x[br4(k)], unscaled
#define BF(SPAN, G, J) ibfly<SPAN, G, J>(x)
  BF(8, 0, 0); BF(8, 0, 1); BF(8, 0, 2); BF(8, 0, 3); BF(8, 0, 4); BF(8, 0, 5); BF(8, 0, 6); BF(8, 0, 7);
  BF(4, 0, 0); BF(4, 0, 1); BF(4, 0, 2); BF(4, 0, 3); BF(4, 8, 0); BF(4, 8, 1); BF(4, 8, 2); BF(4, 8, 3);
  BF(2, 0, 0); BF(2, 0, 1); BF(2, 4, 0); BF(2, 4, 1); BF(2, 8, 0); BF(2, 8, 1); BF(2, 12, 0); BF(2, 12, 1);
  BF(1, 0, 0); BF(1, 2, 0); BF(1, 4, 0); BF(1, 6, 0); BF(1, 8, 0); BF(1, 10, 0); BF(1, 12, 0); BF(1, 14, 0);
#undef BF
}
__device__ __forceinline__ constexpr int fbr4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

__global__ __launch_bounds__(256) void k_fri_fold(const u64* __restrict__ in, u64* __restrict__ out, unsigned log_m, u64 shift,
                                                  gl2 beta, u64 inv16) {
  LATENCY_KERNEL_PRIO();
  const size_t Mo = (size_t)1 << (log_m - 4);
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Mo) return;
  u64 r0[16], r1[16];
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(in) + 16 * c;
#pragma unroll
  for (int t = 0; t < 16; t++) {  // position 16c+t holds natural coset index i = br4(t)
    ulonglong2 v = p[t];
    r0[fbr4(t)] = v.x;
    r1[fbr4(t)] = v.y;
  }
  idft16(r0);
  idft16(r1);
  // coefficients (times 16) of R(x0 * Y): r[br4(m)], m = 0..15.  R(beta) = sum_m coef_m (beta/x0)^m
  const u32 n0 = bitrev32((u32)c, log_m - 4);
  const u64 x0 = gl_mul(shift, gl_pow(gl_root_of_unity(log_m), n0));
  const gl2 y = gl2_mul_base(beta, gl_inv(x0));
  gl2 acc = gl2_make(0, 0);
#pragma unroll
  for (int m = 15; m >= 0; m--) {
    acc = gl2_mul(acc, y);
    acc.c0 = gl_add(acc.c0, r0[fbr4(m)]);
    acc.c1 = gl_add(acc.c1, r1[fbr4(m)]);
  }
  acc = gl2_mul_base(acc, inv16);
  reinterpret_cast<ulonglong2*>(out)[c] = make_ulonglong2(acc.c0, acc.c1);
}

void fri_fold(const u64* d_in, u64* d_out, unsigned log_m, u64 shift, gl2 beta, u64 inv16, hipStream_t st) {
  size_t Mo = (size_t)1 << (log_m - 4);
  k_fri_fold<<<(unsigned)((Mo + 255) / 256), 256, 0, st>>>(d_in, d_out, log_m, shift, beta, inv16);
}

// ---- proof of work --------------------------------------------------------------------------------------------
struct PowArgs {
  u64 state[12];
  int pos;
  u64 base;
  unsigned pow_bits;
  unsigned long long* result;  // min over valid candidates, initialised to ~0
};
__global__ __launch_bounds__(256) void k_pow(PowArgs A) {
  LATENCY_KERNEL_PRIO();
  u64 cand = A.base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = A.state[i];
  // set_elt(candidate, pos): pos is uniform
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (i == A.pos) s[i] = cand;
  poseidon_permute(s);
  if ((s[7] >> (64 - A.pow_bits)) == 0) atomicMin(A.result, (unsigned long long)cand);
}

void fri_pow_launch(const u64 state[12], int pos, u64 base, unsigned pow_bits, size_t count, unsigned long long* d_result,
                    hipStream_t st) {
  PowArgs A;
  for (int i = 0; i < 12; i++) A.state[i] = state[i];
  A.pos = pos;
  A.base = base;
  A.pow_bits = pow_bits;
  A.result = d_result;
  k_pow<<<(unsigned)(count / 256), 256, 0, st>>>(A);
}

// ---- query gather ---------------------------------------------------------------------------------------------
// One block per query.  Writes, in proof order: for each initial oracle: leaf row + path; for each layer:
// 16 extension values + path.
__global__ __launch_bounds__(256) void k_gather_queries(QueryGatherArgs A) {
  LATENCY_KERNEL_PRIO();
  const int q = blockIdx.x, t = threadIdx.x;
  u64* out = A.out + (size_t)q * A.words_per_query;
  size_t x_index = A.indices[q];
  for (int o = 0; o < 3; o++) {
    const u64* lde = A.lde[o];
    const int width = A.width[o];
    if (A.lde_by_query[o]) {  // leaf rows gathered beforehand: [width][n_queries]
      for (int c = t; c < width; c += blockDim.x) out[c] = lde[(size_t)c * gridDim.x + q];
    } else {
      for (int c = t; c < width; c += blockDim.x) out[c] = lde[(size_t)c * A.M2 + x_index];
    }
    out += width;
    // Merkle path: sibling at each level
    const int plen = A.log_m2 - A.cap_height;
    for (int w = t; w < plen * 4; w += blockDim.x) {
      int l = w >> 2;
      size_t node = (x_index >> l) ^ 1;
      out[w] = A.tree[o][4 * (merkle_level_offset_dev(A.log_m2, l) + node) + (w & 3)];
    }
    out += plen * 4;
  }
  size_t xi = x_index;
  int log_m = A.log_m2;
  for (int l = 0; l < A.n_layers; l++) {
    size_t ci = xi >> 4;
    if (t < 32) out[t] = A.layer_vals[l][32 * ci + t];
    out += 32;
    const int lg = log_m - 4;  // leaves of this layer's tree
    const int plen = lg - A.cap_height;
    for (int w = t; w < plen * 4; w += blockDim.x) {
      int lv = w >> 2;
      size_t node = (ci >> lv) ^ 1;
      out[w] = A.layer_tree[l][4 * (merkle_level_offset_dev(lg, lv) + node) + (w & 3)];
    }
    out += plen * 4;
    xi = ci;
    log_m = lg;
  }
}

void fri_gather_queries(const QueryGatherArgs& A, int n_queries, hipStream_t st) {
  k_gather_queries<<<n_queries, 256, 0, st>>>(A);
}

// loads this translation unit's code object (the HIP runtime defers that to the first launch otherwise)
void fri_module_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_opening_tables));
}
