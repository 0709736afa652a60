// Auxiliary polynomial columns: LogUp range-check helpers + running sum, and cross-table-lookup Z columns.
//
// Replaces (un-vendored starky 0.4.0, driven from reference src/starks/common/prover.rs:46-65):
//   lookup_helper_columns  for Stark::lookups() (scalar_mul_stark.rs:493-500):
//       h_k = 1/(beta+f_2k) + 1/(beta+f_2k+1),  Z_0 = 0,  Z_{i+1} = Z_i + sum_k h_k(i) - freq(i)/(beta+table(i))
//   get_ctl_data / partial_sums for looked tables with no looking tables (scalar_mul_ctl.rs:20-55):
//       Z_i = sum_{j>=i} filter_j / (sum_m col_m(j) beta^m + gamma)
// Output column order (starky prove_with_commitment): per challenge [h_0..h_{m-1}, Z], then the CTL Z's
// in (ctl, challenge) order.
#include "aux.h"

static constexpr int LOGUP_CHUNK = 30;  // helper columns (pairs of range-checked columns) per thread for the in-register batch inversion

// h_k = 1/(a + beta) + 1/(b + beta) = (a + b + 2 beta) / ((a + beta)(b + beta)): ONE inverse per helper.  A thread inverts the
// pair products of LOGUP_CHUNK helpers with one field inversion (Montgomery's trick: prefix products kept in registers): five
// products per helper plus 1/30 of an inversion (~90 products), against six per helper plus 1/15 of one when every column was
// inverted by itself.  The last helper of an odd column count is 1/(a + beta).
__global__ __launch_bounds__(256) void k_logup_helpers(const u64* __restrict__ trace, size_t N, int rc_begin, int n_rc,
                                                       u64 beta0, u64 beta1, u64* __restrict__ aux, int helpers_per_ch,
                                                       u64* __restrict__ psum, int nchunks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int chunk = blockIdx.y, ch = blockIdx.z;
  const u64 beta = ch ? beta1 : beta0;
  const int k0 = chunk * LOGUP_CHUNK;  // first helper of this thread
  const int cnt = min(LOGUP_CHUNK, helpers_per_ch - k0);
  const u64* col = trace + (size_t)(rc_begin + 2 * k0) * N + i;
  u64 prod[LOGUP_CHUNK], pre[LOGUP_CHUNK];
  u64 acc = 1;
#pragma unroll
  for (int j = 0; j < LOGUP_CHUNK; j++) {
    if (j < cnt) {
      const u64 a = gl_add(col[(size_t)(2 * j) * N], beta);
      const bool two = 2 * (k0 + j) + 1 < n_rc;
      prod[j] = two ? gl_mul(a, gl_add(col[(size_t)(2 * j + 1) * N], beta)) : a;
      pre[j] = acc;
      acc = gl_mul(acc, prod[j]);
    }
  }
  u64 inv = gl_inv(acc);
  u64 sum = 0;
  const u64 beta2 = gl_dbl(beta);
  u64* out = aux + (size_t)ch * (helpers_per_ch + 1) * N + i;
#pragma unroll
  for (int j = LOGUP_CHUNK - 1; j >= 0; j--) {
    if (j < cnt) {
      u64 h = gl_mul(inv, pre[j]);  // 1 / ((a + beta)(b + beta))
      inv = gl_mul(inv, prod[j]);
      if (2 * (k0 + j) + 1 < n_rc) h = gl_mul(h, gl_add(gl_add(col[(size_t)(2 * j) * N], col[(size_t)(2 * j + 1) * N]), beta2));
      out[(size_t)(k0 + j) * N] = h;
      sum = gl_add(sum, h);
    }
  }
  psum[(size_t)(ch * nchunks + chunk) * N + i] = sum;
}

// term[i] = sum_chunks psum[i] - freq[i] / (beta + table[i])
__global__ __launch_bounds__(256) void k_logup_terms(const u64* __restrict__ trace, size_t N, int table_col, int freq_col,
                                                     u64 beta0, u64 beta1, const u64* __restrict__ psum, int nchunks,
                                                     u64* __restrict__ terms) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ch = blockIdx.y;
  const u64 beta = ch ? beta1 : beta0;
  u64 s = 0;
  for (int c = 0; c < nchunks; c++) s = gl_add(s, psum[(size_t)(ch * nchunks + c) * N + i]);
  u64 t = gl_inv(gl_add(beta, trace[(size_t)table_col * N + i]));
  s = gl_sub(s, gl_mul(trace[(size_t)freq_col * N + i], t));
  terms[(size_t)ch * N + i] = s;
}

// One 1024-thread block per column.  mode 0: out[0] = 0, out[i+1] = out[i] + in[i] (exclusive prefix);
// mode 1: out[i] = sum_{j >= i} in[j] (inclusive suffix).
__global__ __launch_bounds__(1024) void k_scan(const u64* __restrict__ in, size_t in_stride, u64* __restrict__ out,
                                               size_t out_stride, size_t N, int mode) {
  LATENCY_KERNEL_PRIO();
  __shared__ u64 part[1024];
  const int t = threadIdx.x;
  const u64* src = in + (size_t)blockIdx.x * in_stride;
  u64* dst = out + (size_t)blockIdx.x * out_stride;
  const size_t per = N / 1024;
  auto idx = [&](size_t q) { return mode ? N - 1 - q : q; };  // scan order position -> memory index
  size_t q0 = (size_t)t * per;
  u64 s = 0;
  for (size_t q = 0; q < per; q++) s = gl_add(s, src[idx(q0 + q)]);
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    u64 v = t >= off ? part[t - off] : 0;
    __syncthreads();
    part[t] = gl_add(part[t], v);
    __syncthreads();
  }
  u64 run = t ? part[t - 1] : 0;  // sum of everything before this thread's chunk
  for (size_t q = 0; q < per; q++) {
    u64 x = src[idx(q0 + q)];
    if (mode == 0) {
      dst[idx(q0 + q)] = run;
      run = gl_add(run, x);
    } else {
      run = gl_add(run, x);
      dst[idx(q0 + q)] = run;
    }
  }
}

// CTL terms: filter/combine for every (ctl, challenge); terms[(ctl*2+ch)*N + i].
__global__ __launch_bounds__(256) void k_ctl_terms(const u64* __restrict__ trace, size_t N, CtlSpecDev spec, u64 beta0,
                                                   u64 gamma0, u64 beta1, u64 gamma1, u64* __restrict__ terms,
                                                   int* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int ctl = blockIdx.y;
  u64 f = trace[(size_t)spec.filter_col[ctl] * N + i];
  u64 t0 = 0, t1 = 0;
  if (f == 1) {
    u64 a0 = 0, a1 = 0;
    for (int m = spec.ncols[ctl] - 1; m >= 0; m--) {
      int start = spec.col_start[ctl][m], nb = spec.col_bits[ctl][m];
      u64 v = 0;
      for (int b = nb - 1; b >= 0; b--) v = gl_add(gl_dbl(v), trace[(size_t)(start + b) * N + i]);
      a0 = gl_add(gl_mul(a0, beta0), v);
      a1 = gl_add(gl_mul(a1, beta1), v);
    }
    t0 = gl_inv(gl_add(a0, gamma0));
    t1 = gl_inv(gl_add(a1, gamma1));
  } else if (f != 0) {
    atomicCAS(err, 0, BN254S_E_INTERNAL);  // starky: "Non-binary filter?"
  }
  terms[(size_t)(ctl * 2 + 0) * N + i] = t0;
  terms[(size_t)(ctl * 2 + 1) * N + i] = t1;
}

size_t aux_scratch_words(const StarkShape& sh, size_t N) {
  int nchunks = (sh.n_helpers() + LOGUP_CHUNK - 1) / LOGUP_CHUNK;
  return (size_t)(2 * nchunks + 2 + 2 * sh.n_ctl) * N;
}

void aux_build(const StarkShape& sh, const u64* d_trace, size_t N, const u64 betas[2], const u64 gammas[2], u64* d_aux,
               u64* d_scratch, int* d_err, hipStream_t st) {
  const int n_rc = sh.n_rc(), m = sh.n_helpers();
  const int nchunks = (m + LOGUP_CHUNK - 1) / LOGUP_CHUNK;
  u64* psum = d_scratch;
  u64* terms = psum + (size_t)2 * nchunks * N;
  u64* cterms = terms + 2 * N;
  dim3 g1((unsigned)((N + 255) / 256), nchunks, 2);
  k_logup_helpers<<<g1, 256, 0, st>>>(d_trace, N, sh.rc_begin, n_rc, betas[0], betas[1], d_aux, m, psum, nchunks);
  dim3 g2((unsigned)((N + 255) / 256), 2);
  k_logup_terms<<<g2, 256, 0, st>>>(d_trace, N, sh.table_col, sh.freq_col, betas[0], betas[1], psum, nchunks, terms);
  // Z columns sit after the helpers of each challenge: column ch*(m+1) + m
  k_scan<<<2, 1024, 0, st>>>(terms, N, d_aux + (size_t)m * N, (size_t)(m + 1) * N, N, 0);
  dim3 g3((unsigned)((N + 255) / 256), sh.n_ctl);
  k_ctl_terms<<<g3, 256, 0, st>>>(d_trace, N, sh.ctl, betas[0], gammas[0], betas[1], gammas[1], cterms, d_err);
  k_scan<<<2 * sh.n_ctl, 1024, 0, st>>>(cterms, N, d_aux + (size_t)2 * (m + 1) * N, N, N, 1);
}

// loads this translation unit's code object (the HIP runtime defers that to the first launch otherwise)
void aux_module_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_logup_terms));
}
