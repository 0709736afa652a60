// Micro-benchmark (tuning only): issue cost of the scalar-coupled vector instructions the hand-scheduled Poseidon uses.
// Build: hipcc -O3 --offload-arch=gfx950 sgpr_ops.hip -o sgpr_ops ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
typedef uint32_t u32;
#define REP8(X) X X X X X X X X
#define ITERS 2048

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, u32 a0) {
  u32 a = a0 + threadIdx.x;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) {  // baseline: 8 x v_add_u32 on independent registers
      asm volatile(REP8("v_add_u32 v40, v40, v41\n v_add_u32 v42, v42, v43\n v_add_u32 v44, v44, v45\n v_add_u32 v46, v46, v47\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if (OP == 1) {  // v_add_co_u32 to four distinct SGPR pairs
      asm volatile(REP8("v_add_co_u32 v40, s[40:41], v40, v41\n v_add_co_u32 v42, s[42:43], v42, v43\n v_add_co_u32 v44, s[44:45], v44, v45\n v_add_co_u32 v46, s[46:47], v46, v47\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if (OP == 2) {  // v_add_co_u32 all to vcc
      asm volatile(REP8("v_add_co_u32 v40, vcc, v40, v41\n v_add_co_u32 v42, vcc, v42, v43\n v_add_co_u32 v44, vcc, v44, v45\n v_add_co_u32 v46, vcc, v46, v47\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    } else if (OP == 3) {  // v_cndmask with a (constant) SGPR mask
      asm volatile(REP8("v_cndmask_b32 v40, v40, v41, s[40:41]\n v_cndmask_b32 v42, v42, v43, s[42:43]\n v_cndmask_b32 v44, v44, v45, s[44:45]\n v_cndmask_b32 v46, v46, v47, s[46:47]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if (OP == 4) {  // add_co -> (3 other add_co) -> cndmask reading its carry: the fold pattern
      asm volatile(REP8("v_add_co_u32 v40, s[40:41], v40, v41\n v_add_co_u32 v42, s[42:43], v42, v43\n v_add_co_u32 v44, s[44:45], v44, v45\n v_add_co_u32 v46, s[46:47], v46, v47\n"
                        "v_cndmask_b32 v48, 0, -1, s[40:41]\n v_cndmask_b32 v49, 0, -1, s[42:43]\n v_cndmask_b32 v50, 0, -1, s[44:45]\n v_cndmask_b32 v51, 0, -1, s[46:47]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if (OP == 5) {  // v_mad_u64_u32 with carry-out to vcc
      asm volatile(REP8("v_mad_u64_u32 v[40:41], vcc, v56, v57, v[40:41]\n v_mad_u64_u32 v[42:43], vcc, v56, v57, v[42:43]\n v_mad_u64_u32 v[44:45], vcc, v56, v57, v[44:45]\n v_mad_u64_u32 v[46:47], vcc, v56, v57, v[46:47]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    } else if (OP == 6) {  // v_mad_u64_u32 with carry-out to distinct SGPR pairs
      asm volatile(REP8("v_mad_u64_u32 v[40:41], s[40:41], v56, v57, v[40:41]\n v_mad_u64_u32 v[42:43], s[42:43], v56, v57, v[42:43]\n v_mad_u64_u32 v[44:45], s[44:45], v56, v57, v[44:45]\n v_mad_u64_u32 v[46:47], s[46:47], v56, v57, v[46:47]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if (OP == 7) {  // v_lshl_add_u64
      asm volatile(REP8("v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if (OP == 8) {  // sub_co + subb_co pairs (distinct SGPR pairs), as in the exact reduction
      asm volatile(REP8("v_sub_co_u32 v40, s[40:41], v40, v48\n v_sub_co_u32 v42, s[42:43], v42, v48\n v_subb_co_u32 v41, s[40:41], v41, 0, s[40:41]\n v_subb_co_u32 v43, s[42:43], v43, 0, s[42:43]\n")
                   ::: "v40", "v41", "v42", "v43", "s40", "s41", "s42", "s43");
    } else if (OP == 9) {  // scalar mask logic between vector adds: 2 valu + 2 salu
      asm volatile(REP8("v_add_u32 v40, v40, v41\n s_xor_b64 s[44:45], s[40:41], s[42:43]\n v_add_u32 v42, v42, v43\n s_and_b64 s[46:47], s[44:45], s[40:41]\n")
                   ::: "v40", "v41", "v42", "v43", "s44", "s45", "s46", "s47", "scc");
    } else if (OP == 14) {  // 3 scalar ops per 5 vector ops (the exact reduction's mask logic)
      asm volatile(REP8("v_add_u32 v40, v40, v41\n s_xor_b64 s[44:45], s[40:41], s[42:43]\n s_and_b64 s[46:47], s[44:45], s[40:41]\n s_and_b64 s[48:49], s[44:45], s[42:43]\n v_add_u32 v42, v42, v43\n v_add_u32 v44, v44, v45\n v_add_u32 v46, v46, v47\n v_add_u32 v48, v48, v49\n")
                   ::: "v40", "v42", "v44", "v46", "v48", "s44", "s45", "s46", "s47", "s48", "s49", "scc");
    } else if (OP == 15) {  // v_cndmask e32 (vcc mask)
      asm volatile(REP8("v_cndmask_b32 v40, v40, v41, vcc\n v_cndmask_b32 v42, v42, v43, vcc\n v_cndmask_b32 v44, v44, v45, vcc\n v_cndmask_b32 v46, v46, v47, vcc\n")
                   ::: "v40", "v42", "v44", "v46");
    } else if (OP == 16) {  // v_add3_u32 alone
      asm volatile(REP8("v_add3_u32 v40, v40, v41, v50\n v_add3_u32 v42, v42, v43, v50\n v_add3_u32 v44, v44, v45, v50\n v_add3_u32 v46, v46, v47, v50\n")
                   ::: "v40", "v42", "v44", "v46");
    } else if (OP == 17) {  // v_sub_u32 e64 form (forced VOP3 by an SGPR second source)
      asm volatile(REP8("v_sub_u32_e64 v40, v40, v41\n v_sub_u32_e64 v42, v42, v43\n v_sub_u32_e64 v44, v44, v45\n v_sub_u32_e64 v46, v46, v47\n")
                   ::: "v40", "v42", "v44", "v46");
    } else if (OP == 10) {  // v_mov_b32
      asm volatile(REP8("v_mov_b32 v40, v41\n v_mov_b32 v42, v43\n v_mov_b32 v44, v45\n v_mov_b32 v46, v47\n")
                   ::: "v40", "v42", "v44", "v46");
    } else if (OP == 11) {  // v_mad_u64_u32, inline-constant multiplier, SGPR-pair-free (the MDS form)
      asm volatile(REP8("v_mad_u64_u32 v[40:41], vcc, v56, 17, v[40:41]\n v_mad_u64_u32 v[42:43], vcc, v57, 41, v[42:43]\n v_mad_u64_u32 v[44:45], vcc, v58, 13, v[44:45]\n v_mad_u64_u32 v[46:47], vcc, v59, 39, v[46:47]\n")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "vcc");
    } else if (OP == 12) {  // v_sub_u32 / v_add3_u32 / v_min3_u32 mix (the flagged fold)
      asm volatile(REP8("v_sub_u32 v40, v40, v41\n v_add3_u32 v42, v42, v43, v44\n v_min3_u32 v45, v45, v46, v47\n v_max3_u32 v48, v48, v49, v50\n")
                   ::: "v40", "v42", "v45", "v48");
    } else if (OP == 13) {  // cndmask chain through one carry: add_co, 2 fillers, cndmask (single stream, the tight case)
      asm volatile(REP8("v_add_co_u32 v40, s[40:41], v40, v41\n v_add_u32 v42, v42, v43\n v_add_u32 v44, v44, v45\n v_cndmask_b32 v46, 0, -1, s[40:41]\n")
                   ::: "v40", "v42", "v44", "v46", "s40", "s41");
    }
  }
  u64 r;
  asm volatile("v_mov_b32 %0, v40" : "=v"(a));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

template <int OP>
void run(const char* name, int blocks) {
  printf("%-44s ...", name);
  fflush(stdout);
  u64* d;
  (void)hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 20; w++) k<OP><<<blocks, 256>>>(d, 1);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int w = 0; w < 10; w++) k<OP><<<blocks, 256>>>(d, 1);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  double insts_per_wave = (double)ITERS * 32 * (OP == 14 ? 2 : 1);
  double waves_per_simd = blocks * 4.0 / 1024.0;
  printf("\r%-44s %2.0f waves/SIMD %8.3f ms -> %6.2f cycles(@2.4GHz) per instruction per SIMD\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd));
  fflush(stdout);
  (void)hipFree(d);
}

int main() {
  for (int blocks : {512, 2048}) {
    run<0>("v_add_u32", blocks);
    run<10>("v_mov_b32", blocks);
    run<12>("sub/add3/min3/max3", blocks);
    run<7>("v_lshl_add_u64", blocks);
    run<5>("v_mad_u64_u32 -> vcc", blocks);
    run<11>("v_mad_u64_u32 x inline const -> vcc", blocks);
    run<6>("v_mad_u64_u32 -> s[pair]", blocks);
    run<2>("v_add_co_u32 -> vcc", blocks);
    run<1>("v_add_co_u32 -> s[pair]", blocks);
    run<3>("v_cndmask_b32 s[pair]", blocks);
    run<4>("4 add_co + 4 cndmask (fold pattern)", blocks);
    run<13>("add_co, add, add, cndmask (one stream)", blocks);
    run<8>("sub_co + subb_co pairs", blocks);
    run<9>("2 v_add_u32 + 2 salu (per 4 instr)", blocks);
    run<14>("5 v_add_u32 + 3 salu (per 8 instr)", blocks);
    run<15>("v_cndmask_b32 vcc (e32)", blocks);
    run<16>("v_add3_u32", blocks);
    run<17>("v_sub_u32_e64", blocks);
  }
  return 0;
}
