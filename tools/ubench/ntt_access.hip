// Micro-benchmark (tuning only): the memory side of the NTT passes without their arithmetic - same tiles, same addresses,
// same access widths (1237 columns of 2^16 words: 648 MB read + 648 MB written per launch).
// Build: hipcc -O3 --offload-arch=gfx950 ntt_access.hip -o ntt_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
static constexpr size_t N = 65536;
__device__ __forceinline__ int br4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

// pass-2 pattern: 16 rows k1 = b + 16 d of 2 KB each in, the 32 KB bit-reversed region out (16 B per lane, 128 B apart)
__global__ __launch_bounds__(256) void k_pass2(const u64* __restrict__ in, u64* __restrict__ out, unsigned ny) {
  const int t = threadIdx.x, g = t & 15, d = t >> 4;
  const int k1 = blockIdx.x + 16 * d;
  for (unsigned y = blockIdx.y; y < ny; y += gridDim.y) {
    const u64* col = in + (size_t)y * N;
    u64 x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = col[k1 * 256 + g + 16 * m];
    u64* o = out + (size_t)y * N + (size_t)(__brev(k1) >> 24) * 256 + br4(g) * 16;
#pragma unroll
    for (int p = 0; p < 16; p += 2) {
      ulonglong2 w;
      w.x = x[p] + 1;
      w.y = x[p + 1] + 1;
      *reinterpret_cast<ulonglong2*>(o + p) = w;
    }
  }
}
// pass-1 pattern: 16 adjacent columns i2 x 256 rows in (128 B per row), the same shape out
__global__ __launch_bounds__(256) void k_pass1(const u64* __restrict__ in, u64* __restrict__ out, unsigned ny) {
  const int t = threadIdx.x, d = t & 15, g = t >> 4;
  const int i2 = blockIdx.x * 16 + d;
  for (unsigned y = blockIdx.y; y < ny; y += gridDim.y) {
    const u64* col = in + (size_t)y * N;
    u64 x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = col[(g + 16 * m) * 256 + i2];
    u64* o = out + (size_t)y * N;
#pragma unroll
    for (int kb = 0; kb < 16; kb++) o[(g + 16 * kb) * 256 + i2] = x[kb] + 1;
  }
}
// the same bytes as full 2 KB rows per 256 lanes, 8 B per lane
__global__ __launch_bounds__(256) void k_linear(const u64* __restrict__ in, u64* __restrict__ out, unsigned ny) {
  const int t = threadIdx.x;
  for (unsigned y = blockIdx.y; y < ny; y += gridDim.y) {
    const u64* col = in + (size_t)y * N + (size_t)blockIdx.x * 4096;
    u64* o = out + (size_t)y * N + (size_t)blockIdx.x * 4096;
    u64 x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = col[m * 256 + t];
#pragma unroll
    for (int m = 0; m < 16; m++) o[m * 256 + t] = x[m] + 1;
  }
}
// 16 B per lane, fully linear
__global__ __launch_bounds__(256) void k_linear16(const u64* __restrict__ in, u64* __restrict__ out, unsigned ny) {
  const int t = threadIdx.x;
  for (unsigned y = blockIdx.y; y < ny; y += gridDim.y) {
    const ulonglong2* col = reinterpret_cast<const ulonglong2*>(in + (size_t)y * N + (size_t)blockIdx.x * 4096);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + (size_t)y * N + (size_t)blockIdx.x * 4096);
    ulonglong2 x[8];
#pragma unroll
    for (int m = 0; m < 8; m++) x[m] = col[m * 256 + t];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      x[m].x += 1;
      o[m * 256 + t] = x[m];
    }
  }
}

template <typename K>
void run(const char* name, K kern, unsigned gy, unsigned ny, u64* a, u64* b) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) kern<<<dim3(16, gy), 256>>>(a, b, ny);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; w++) kern<<<dim3(16, gy), 256>>>(a, b, ny);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  printf("%-28s grid 16 x %4u: %7.1f us  %6.0f GB/s\n", name, gy, ms * 1e3, 2.0 * ny * N * 8 / (ms * 1e-3) / 1e9);
  fflush(stdout);
}

int main() {
  const unsigned ny = 1237;
  u64 *a, *b;
  (void)hipMalloc(&a, (size_t)ny * N * 8);
  (void)hipMalloc(&b, (size_t)ny * N * 8);
  (void)hipMemset(a, 1, (size_t)ny * N * 8);
  for (unsigned gy : {1237u, 128u, 64u}) {
    run("pass2 pattern", k_pass2, gy, ny, a, b);
    run("pass1 pattern", k_pass1, gy, ny, a, b);
    run("linear 8 B/lane", k_linear, gy, ny, a, b);
    run("linear 16 B/lane", k_linear16, gy, ny, a, b);
  }
  return 0;
}
