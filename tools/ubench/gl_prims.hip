// Micro-benchmark (tuning only): issue cost of the 64-bit instructions and of the hand-written Goldilocks sequences of
// csrc/gl_asm.h.  Build: hipcc -O3 --offload-arch=gfx950 -I../../plonky2_bn254_amd/csrc gl_prims.hip -o gl_prims
#include <hip/hip_runtime.h>
#include <cstdio>
#include "gl_asm.h"
#define REP8(X) X X X X X X X X
#define ITERS 1024

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, u64 seed) {
  u64 a = seed * (threadIdx.x + 1) % GL_P, b = (seed + 77) * (threadIdx.x + 3) % GL_P, c = a ^ 5, d = b ^ 9;
  c %= GL_P;
  d %= GL_P;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) {
      asm volatile(REP8("v_cmp_lt_u64 s[40:41], v[40:41], v[42:43]\n v_cmp_lt_u64 s[42:43], v[44:45], v[46:47]\n v_cmp_lt_u64 s[44:45], v[48:49], v[50:51]\n v_cmp_lt_u64 s[46:47], v[52:53], v[54:55]\n")
                   ::: "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if (OP == 1) {
      asm volatile(REP8("v_lshlrev_b64 v[40:41], 12, v[42:43]\n v_lshlrev_b64 v[44:45], 12, v[46:47]\n v_lshlrev_b64 v[48:49], 12, v[50:51]\n v_lshlrev_b64 v[52:53], 12, v[54:55]\n")
                   ::: "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53");
    } else if (OP == 2) {
      asm volatile(REP8("v_lshrrev_b64 v[40:41], 12, v[42:43]\n v_lshrrev_b64 v[44:45], 12, v[46:47]\n v_lshrrev_b64 v[48:49], 12, v[50:51]\n v_lshrrev_b64 v[52:53], 12, v[54:55]\n")
                   ::: "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53");
    } else if (OP == 3) {
      asm volatile(REP8("v_lshl_add_u64 v[40:41], v[42:43], 0, v[40:41]\n v_lshl_add_u64 v[44:45], v[46:47], 0, v[44:45]\n v_lshl_add_u64 v[48:49], v[50:51], 0, v[48:49]\n v_lshl_add_u64 v[52:53], v[54:55], 0, v[52:53]\n")
                   ::: "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53");
    } else if (OP == 4) {
      asm volatile(REP8("v_sub_co_u32 v40, s[40:41], v42, v43\n v_sub_co_u32 v44, s[42:43], v46, v47\n v_subb_co_u32 v41, s[40:41], v42, v43, s[40:41]\n v_subb_co_u32 v45, s[42:43], v46, v47, s[42:43]\n")
                   ::: "v40", "v41", "v44", "v45", "s40", "s41", "s42", "s43");
    } else if (OP == 5) {
      asm volatile(REP8("v_cndmask_b32 v40, 0, -1, s[40:41]\n v_cndmask_b32 v41, 0, -1, s[42:43]\n v_cndmask_b32 v44, 0, -1, s[40:41]\n v_cndmask_b32 v45, 0, -1, s[42:43]\n")
                   ::: "v40", "v41", "v44", "v45");
    } else if (OP == 6) {
      asm volatile(REP8("v_mad_u64_u32 v[40:41], s[40:41], v42, 1, v[40:41]\n v_mad_u64_u32 v[44:45], s[42:43], v46, -1, v[44:45]\n v_mad_u64_u32 v[48:49], s[44:45], v50, 1, v[48:49]\n v_mad_u64_u32 v[52:53], s[46:47], v54, -1, v[52:53]\n")
                   ::: "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if (OP == 7) {
      asm volatile(REP8("v_lshrrev_b32 v40, 12, v42\n v_lshlrev_b32 v44, 12, v46\n v_sub_u32 v48, 0, v50\n v_lshrrev_b32 v52, 5, v54\n")
                   ::: "v40", "v44", "v48", "v52");
    } else if (OP == 8) {
      asm volatile(REP8("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
    } else if (OP == 9) {
      asm volatile(REP8("s_or_b64 s[40:41], s[42:43], s[44:45]\n s_or_b64 s[46:47], s[42:43], s[44:45]\n s_or_b64 s[48:49], s[42:43], s[44:45]\n s_or_b64 s[50:51], s[42:43], s[44:45]\n")
                   ::: "s40", "s41", "s46", "s47", "s48", "s49", "s50", "s51", "scc");
    } else if (OP == 10) {  // 32 butterflies
#pragma unroll
      for (int r = 0; r < 16; r++) {
        u64 s, t;
        gl_bfly_asm<false>(a, b, s, t);
        a = s;
        b = t;
        gl_bfly_asm<false>(c, d, s, t);
        c = s;
        d = t;
      }
    } else if (OP == 11) {  // 32 shifts <= 32
#pragma unroll
      for (int r = 0; r < 8; r++) {
        a = gl_shl_small_asm<12>(a);
        b = gl_shl_small_asm<24>(b);
        c = gl_shl_small_asm<12>(c);
        d = gl_shl_small_asm<24>(d);
      }
    } else if (OP == 12) {  // 32 right shifts
#pragma unroll
      for (int r = 0; r < 8; r++) {
        a = gl_shr_small_asm<12>(a);
        b = gl_shr_small_asm<24>(b);
        c = gl_shr_small_asm<12>(c);
        d = gl_shr_small_asm<24>(d);
      }
    } else if (OP == 13) {  // 32 products
#pragma unroll
      for (int r = 0; r < 8; r++) {
        a = gl_mul_asm(a, b);
        c = gl_mul_asm(c, d);
        b = gl_mul_asm(b, c);
        d = gl_mul_asm(d, a);
      }
    } else if (OP == 14) {  // 32 compiler butterflies
#pragma unroll
      for (int r = 0; r < 16; r++) {
        u64 s = gl_add(a, b), t = gl_sub(a, b);
        a = s;
        b = t;
        s = gl_add(c, d), t = gl_sub(c, d);
        c = s;
        d = t;
      }
    } else if (OP == 15) {  // 32 compiler products
#pragma unroll
      for (int r = 0; r < 8; r++) {
        a = gl_mul(a, b);
        c = gl_mul(c, d);
        b = gl_mul(b, c);
        d = gl_mul(d, a);
      }
    } else if (OP == 16) {
      asm volatile(REP8("v_cmp_lt_u64 vcc, v[40:41], v[42:43]\n v_cmp_lt_u64 vcc, v[44:45], v[46:47]\n v_cmp_lt_u64 vcc, v[48:49], v[50:51]\n v_cmp_lt_u64 vcc, v[52:53], v[54:55]\n")
                   ::: "vcc");
    } else if (OP == 17) {
      asm volatile(REP8("v_cmp_lt_u32 s[40:41], v40, v42\n v_cmp_lt_u32 s[42:43], v44, v46\n v_cmp_lt_u32 s[44:45], v48, v50\n v_cmp_lt_u32 s[46:47], v52, v54\n")
                   ::: "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

template <int OP>
void run(const char* name, int blocks, double units_per_iter) {
  u64* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 10; w++) k<OP><<<blocks, 256>>>(d, 12345);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; w++) k<OP><<<blocks, 256>>>(d, 12345);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  double per_wave = (double)ITERS * units_per_iter;
  double waves_per_simd = blocks * 4.0 / 1024.0;
  printf("%-40s %2.0f waves/SIMD %8.3f ms -> %7.2f cycles(@2.4GHz) per unit per SIMD\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / (per_wave * waves_per_simd));
  fflush(stdout);
  (void)hipFree(d);
}

int main() {
  for (int blocks : {1024, 2048}) {
    run<0>("v_cmp_lt_u64 -> s[pair]", blocks, 32);
    run<16>("v_cmp_lt_u64 -> vcc", blocks, 32);
    run<17>("v_cmp_lt_u32 -> s[pair]", blocks, 32);
    run<1>("v_lshlrev_b64", blocks, 32);
    run<2>("v_lshrrev_b64", blocks, 32);
    run<3>("v_lshl_add_u64", blocks, 32);
    run<4>("v_sub_co / v_subb_co -> s[pair]", blocks, 32);
    run<5>("v_cndmask 0,-1,s[pair]", blocks, 32);
    run<6>("v_mad_u64_u32 x1 / x-1 -> s[pair]", blocks, 32);
    run<7>("32-bit shifts / v_sub_u32", blocks, 32);
    run<8>("s_nop 0", blocks, 32);
    run<9>("s_or_b64", blocks, 32);
    run<10>("butterfly (asm, 10 VALU)", blocks, 32);
    run<14>("butterfly (compiler)", blocks, 32);
    run<11>("x 2^S, S<=32 (asm, 6 VALU)", blocks, 32);
    run<12>("x 2^-K (asm, 9 VALU)", blocks, 32);
    run<13>("product (asm, 19 VALU)", blocks, 32);
    run<15>("product (compiler)", blocks, 32);
  }
  return 0;
}
