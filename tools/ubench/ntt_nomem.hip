// Micro-benchmark (tuning only): the arithmetic of NTT pass 2 without its global memory traffic (same instruction
// stream: two radix-16 stages, inner twiddles, LDS exchange), and the memory traffic without the arithmetic.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../plonky2_bn254_amd/csrc -I../../include ntt_nomem.hip -o ntt_nomem
#include "../../plonky2_bn254_amd/csrc/ntt.hip"
#include <cstdio>

// MEM: 0 = no global loads/stores of data (inputs synthesised, outputs stored only under a condition that never holds),
//      1 = the real kernel's traffic
template <int MEM>
__global__ __launch_bounds__(256, 8) void k_pass2_variant(const u64* __restrict__ in, u64* __restrict__ out,
                                                          const u64* __restrict__ tw256, u64 never) {
  __shared__ u32 lds[LDS_TILE_WORDS];
  const int t = threadIdx.x;
  int g = t & 15, d = t >> 4;
  const int k1_load = blockIdx.x + 16 * d;
  const u64* col = in + (size_t)blockIdx.y * NTT_N;
  u64 x[16];
  if (MEM) {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = col[k1_load * 256 + g + 16 * m];
  } else {
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = (u64)(t * 16 + m) * 0x9E3779B97F4A7C15ULL % GL_P + blockIdx.y;
  }
  dft256_tile<false, 1>(x, lds, tw256, d, g);
  const int k1 = blockIdx.x + 16 * d;
  u64* ocol = out + (size_t)blockIdx.y * 2 * NTT_N;
  size_t base = (size_t)bitrev32(k1, 8) * 256 + br4(g) * 16;
  if (MEM) {
#pragma unroll
    for (int p = 0; p < 16; p += 2) {
      ulonglong2 w;
      w.x = x[p];
      w.y = x[p + 1];
      *reinterpret_cast<ulonglong2*>(ocol + base + p) = w;
    }
  } else {
    u64 acc = 0;
#pragma unroll
    for (int p = 0; p < 16; p++) acc ^= x[p];
    if (acc == never) ocol[base] = acc;
  }
}

int main() {
  const unsigned ny = 1237;
  NttTables T;
  if (ntt_tables_init(&T)) return 1;
  u64 *a, *b;
  (void)hipMalloc(&a, (size_t)ny * NTT_N * 8);
  (void)hipMalloc(&b, (size_t)ny * 2 * NTT_N * 8);
  (void)hipMemset(a, 1, (size_t)ny * NTT_N * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int variant = 0; variant < 3; variant++) {
    for (int rep = 0; rep < 2; rep++) {
      (void)hipEventRecord(e0);
      for (int w = 0; w < 10; w++) {
        if (variant == 0) k_pass2_variant<0><<<dim3(16, ny), 256>>>(a, b, T.tw256_fwd, 0x123456789ULL);
        else if (variant == 1) k_pass2_variant<1><<<dim3(16, ny), 256>>>(a, b, T.tw256_fwd, 0);
        else k_ntt_pass2<false, true><<<dim3(16, ny), 256>>>(a, NTT_N, b, 2 * NTT_N, nullptr, 1, T.tw256_fwd, 0);
      }
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("pass 2 on %u columns, %s: %7.1f us\n", ny, variant == 2 ? "the library kernel" : variant ? "with its loads and stores" : "arithmetic and LDS only", ms * 100);
    }
  }
  // the whole from_values stage of the library on the same columns
  u64 *tmp, *tmp2;
  (void)hipMalloc(&tmp, (size_t)ny * NTT_N * 8);
  (void)hipMalloc(&tmp2, (size_t)2 * ny * NTT_N * 8);
  for (int rep = 0; rep < 2; rep++) {
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; w++) ntt_inverse_lde(&T, a, a, b, tmp, tmp2, ny, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("ntt_inverse_lde on %u columns: %7.1f us\n", ny, ms * 200);
  }
  return 0;
}
