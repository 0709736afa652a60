// Micro-benchmark of the Poseidon permutation variants (tuning only; not part of the library).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include poseidon_ub.hip -o poseidon_ub ; run on the GPU box.
// Prints, per variant and occupancy, the time per permutation and the cycles per wave-permutation and SIMD, plus the
// shader clock measured with s_memtime against the wall clock (a power-capped clock shows here).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../plonky2_bn254_amd/csrc/poseidon_dev.h"
#include "gen/poseidon_asm_nosbox.inc"
#include "gen/poseidon_asm_nomds.inc"
#include "gen/poseidon_asm_nofold.inc"

#define RUN_ASM(BODY)                                                                                                          \
  asm volatile(BODY                                                                                                            \
               : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]),      \
                 [x6] "+v"(x[6]), [x7] "+v"(x[7]), [x8] "+v"(x[8]), [x9] "+v"(x[9]), [x10] "+v"(x[10]), [x11] "+v"(x[11]),  \
                 [flag] "+v"(flag)                                                                                             \
               : [tab] "s"(POSEIDON_INIT_DEV), [blk] "s"(POSEIDON_BLK_DEV)                                                     \
               : POSEIDON_ASM_CLOBBERS)

template <int V>
__global__ __launch_bounds__(256) void k(u64* out, unsigned long long* clk, int iters) {
  u64 x[12];
  for (int i = 0; i < 12; i++) x[i] = 0x9E3779B97F4A7C15ull * (threadIdx.x + 256ull * blockIdx.x + 1) + i;
  u32 flag = 1;
  u32 nflag = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (V == 0) poseidon_permute_plain(x);
    else if (V == 1) RUN_ASM(POSEIDON_ASM_PERMUTE);
    else if (V == 2) RUN_ASM(POSEIDON_ASM_EXP_NOSBOX);
    else if (V == 3) RUN_ASM(POSEIDON_ASM_EXP_NOMDS);
    else if (V == 4) RUN_ASM(POSEIDON_ASM_EXP_NOFOLD);
#endif
    nflag += (flag == 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long w1 = wall_clock64();
  u64 acc = flag;
  for (int i = 0; i < 12; i++) acc ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (nflag) atomicAdd(&clk[2], (unsigned long long)nflag);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = w1 - w0;
  }
}

template <int V>
void run(const char* name, int blocks, int iters) {
  u64* d;
  unsigned long long* c;
  hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipMalloc(&c, 24); hipMemset(c, 0, 24);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<V><<<blocks, 256>>>(d, c, 2);
  hipDeviceSynchronize();
  hipMemset(c, 0, 24);
  hipEventRecord(e0);
  k<V><<<blocks, 256>>>(d, c, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[3];
  hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
  double perms = (double)blocks * 256 * iters;
  double waves_per_simd = blocks * 4.0 / 1024.0;
  double mhz = (double)h[0] / ((double)h[1] / 100.0);  // wall_clock64 ticks at 100 MHz
  printf("%-8s blocks %5d (%.0f waves/SIMD): %8.3f ms  %.3f G perm/s  %7.0f cycles(@2.4GHz)/wave-perm/SIMD  shader clock %.0f MHz  flagged %llu\n", name,
         blocks, waves_per_simd, ms, perms / ms / 1e6, ms * 1e-3 * 2.4e9 / (iters * waves_per_simd), mhz, h[2]);
  hipFree(d);
  hipFree(c);
}

int main() {
  for (int blocks : {512, 1024}) {
    run<0>("plain", blocks, 60);
    run<1>("asm", blocks, 60);
    run<2>("nosbox", blocks, 60);
    run<3>("nomds", blocks, 60);
    run<4>("nofold", blocks, 60);
  }
  return 0;
}
