// Micro-benchmark: issue rate of the integer ops that Goldilocks arithmetic is built from (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 intops.hip -o intops ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define ITERS 4096
template <int OP> __global__ void k(u64* out, u32 a0, u32 b0) {
  u32 a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  u64 x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = a * 3, x5 = b * 5, x6 = a * 7, x7 = b * 9;
  u32 y0 = a, y1 = b, y2 = a ^ b, y3 = a + b, y4 = a * 3, y5 = b * 5, y6 = a * 7, y7 = b * 9;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) {  // v_mad_u64_u32
#define M(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
      M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#undef M
    } else if (OP == 1) {  // v_mul_lo_u32
#define M(y) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(y) : "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 2) {  // v_mul_hi_u32
#define M(y) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(y) : "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 3) {  // v_mad_u32_u24
#define M(y) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(y) : "v"(a), "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 4) {  // v_add_co_u32 + v_addc_co_u32 (64-bit add)
#define M(x) { u32 lo_ = (u32)x, hi_ = (u32)(x >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo_), "+v"(hi_) : "v"(a), "v"(b) : "vcc"); x = ((u64)hi_ << 32) | lo_; }
      M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#undef M
    } else if (OP == 5) {  // v_lshlrev_b64
#define M(x) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x));
      M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#undef M
    } else if (OP == 6) {  // v_add_u32 (baseline full rate)
#define M(y) asm volatile("v_add_u32 %0, %0, %1" : "+v"(y) : "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 7) {  // v_dot4_u32_u8
#define M(y) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(y) : "v"(a), "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 8) {  // v_mad_u32_u16? use v_mad_u16? skip: v_mul_u32_u24
#define M(y) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(y) : "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 10) {  // v_perm_b32
#define M(y) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(y) : "v"(a), "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 11) {  // v_lshl_add_u64
#define M(x) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(x7));
      M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x0)
#undef M
    } else if (OP == 12) {  // v_lshl_add_u32
#define M(y) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(y) : "v"(b));
      M(y0) M(y1) M(y2) M(y3) M(y4) M(y5) M(y6) M(y7)
#undef M
    } else if (OP == 9) {  // v_mad_i64_i32
#define M(x) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");
      M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
#undef M
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}
template <int OP> void run(const char* name, int insts_per_macro) {
  u64* d; hipMalloc(&d, 256 * 1024 * 256 * 8);
  int blocks = 256 * 8, threads = 256;  // 8 waves/SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(d, 1, 2); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<blocks, threads>>>(d, 1, 2); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_insts = (double)blocks * (threads / 64) * ITERS * 8 * insts_per_macro;
  double per_simd = wave_insts / 1024;  // 256 CUs x 4 SIMDs
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-22s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cycles / per_simd);
  hipFree(d);
}
int main() {
  run<6>("v_add_u32", 1); run<3>("v_mad_u32_u24", 1); run<8>("v_mul_u32_u24", 1); run<1>("v_mul_lo_u32", 1); run<2>("v_mul_hi_u32", 1);
  run<0>("v_mad_u64_u32", 1); run<9>("v_mad_i64_i32", 1); run<4>("add64 (2 insts)", 2); run<5>("v_lshlrev_b64", 1); run<7>("v_dot4_u32_u8", 1); run<10>("v_perm_b32", 1); run<11>("v_lshl_add_u64", 1); run<12>("v_lshl_add_u32", 1);
  return 0;
}
