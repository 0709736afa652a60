// Poseidon-Goldilocks Merkle commitment for gfx950.
// Replaces plonky2 `MerkleTree::new(leaves, cap_height)` inside PolynomialBatch::from_coeffs (reached from
// reference src/starks/common/prover.rs:31-38) and the FRI commit-phase trees: leaf digest =
// hash_or_noop(leaf) (leaves of <= 4 elements are not hashed), inner nodes = two_to_one, cap = the 2^cap_height
// nodes at that depth.  One lane per leaf, sponge state in VGPRs, round constants read through the
// scalar/constant path; leaves are read column-major so that consecutive lanes touch consecutive words.
#include <algorithm>
#include "merkle.h"
#include "poseidon_dev.h"

// Element c of leaf j sits at data[c * elem_stride + j * leaf_stride]: the column part of the address is wave-uniform (scalar
// registers), the lane part one 32-bit byte offset.  The whole hash_no_pad loop (loads of the next eight columns while the
// current eight are being absorbed, round 0's constants, permutation) is the hand-scheduled statement POSEIDON_ASM_SPONGE,
// which keeps the sponge state in its own registers from the first column to the last.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_leaf_hash(
    const u64* __restrict__ data, size_t leaf_stride, size_t elem_stride, int leaf_len, size_t n_leaves, u64* __restrict__ digests) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the signature)
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_leaves) return;
  const u32 lane_off = (u32)(j * leaf_stride);  // n_leaves * leaf_stride <= 2^29 per launch (merkle_leaves cuts larger inputs)
  // leaf_len > 4: hash_no_pad (leaves of <= 4 elements are not hashed: k_leaf_copy below)
#if defined(BN254S_POSEIDON_PLAIN)
  u64 t[12];
#pragma unroll
  for (int i = 0; i < 12; i++) t[i] = 0;
#pragma unroll 1
  for (int c = 0; c < leaf_len; c += 8) {
    const u64* col = data + (size_t)c * elem_stride;
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (c + i < leaf_len) t[i] = (col + (size_t)i * elem_stride)[lane_off];
    poseidon_permute_plain(t);
  }
  ulonglong2* o = reinterpret_cast<ulonglong2*>(digests + 4 * j);
  o[0] = make_ulonglong2(t[0], t[1]);
  o[1] = make_ulonglong2(t[2], t[3]);
#else
  // the input of every permutation is parked here for the (rare) exact repeat: [lane of the state][thread], 96 B per thread
  __shared__ u64 parked[12][256];
  const u32 lds_addr = (u32)(uintptr_t)(__attribute__((address_space(3))) u64*)&parked[0][threadIdx.x];
  const u32 byte_off = lane_off * 8u;
  const u64 stride_bytes = (u64)elem_stride * 8u;
  const u32 digest_off = (u32)j * 32u;  // n_leaves <= 2^27 per launch
  // The statement canonicalises the digest and stores it itself: it owns v10..v126, and with the digest coming back in four
  // 64-bit outputs beside its inputs the compiler ran out of its ten registers and spilled to scratch memory (round 2: 24 B per
  // lane, off the hot loop, but a private segment for every wave).
  asm volatile(POSEIDON_ASM_SPONGE_STORE
               :
               : [col] "s"(data), [off] "v"(byte_off), [stride] "s"(stride_bytes), [len] "s"(leaf_len),
                 [rc] "s"(POSEIDON_RC_DEV), [tab] "s"(POSEIDON_INIT_DEV), [blk] "s"(POSEIDON_BLK_DEV),
                 [lds] "v"(lds_addr), [dig] "s"(digests), [doff] "v"(digest_off)
               : POSEIDON_ASM_CLOBBERS, POSEIDON_ASM_SPONGE_CLOBBERS, "memory");
#endif
#endif
}

// hash_or_noop for leaves of at most four elements (the quotient commitment: 4 chunk polynomials): the digest is the leaf itself,
// zero-padded.  Same addressing as k_leaf_hash.
__global__ __launch_bounds__(256) void k_leaf_copy(const u64* __restrict__ data, size_t leaf_stride, size_t elem_stride, int leaf_len,
                                                   size_t n_leaves, u64* __restrict__ digests) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_leaves) return;
  u64 s[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < leaf_len) s[i] = data[(size_t)i * elem_stride + j * leaf_stride];
  ulonglong2* o = reinterpret_cast<ulonglong2*>(digests + 4 * j);
  o[0] = make_ulonglong2(s[0], s[1]);
  o[1] = make_ulonglong2(s[2], s[3]);
}

// ASM = false: the compiler's permutation (one wave per SIMD here, 2^16 / 2^15 nodes: it keeps twelve independent S-boxes in flight
// and finishes a lone wave in 60 us; the hand-scheduled one is built for issue-bound launches and takes 90 us alone).
// ASM = true: the hand-scheduled permutation, half the instructions: for proofs that run beside many others, where the issue slots
// count and the latency of one level does not (merkle_set_throughput_mode).
template <bool ASM>
__global__ __launch_bounds__(256) void k_merkle_level(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out) {
  LATENCY_KERNEL_PRIO();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const ulonglong2* p = reinterpret_cast<const ulonglong2*>(in + 8 * i);
  ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
  u64 s[12] = {a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y, 0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (ASM) poseidon_permute(s);
  else poseidon_permute_plain(s);
#endif
  ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 4 * i);
  o[0] = make_ulonglong2(s[0], s[1]);
  o[1] = make_ulonglong2(s[2], s[3]);
}

// ---- small trees: one permutation per 16 lanes (poseidon_permute_coop) ----------------------------------------------------
// Levels of at most this many nodes use the cooperative (16 lanes per permutation) kernels: a fifth of the latency at five times the
// instructions.  Two modes, chosen per call by the prover (same digests either way):
//   latency (a proof alone or with few others): cooperative up to 16384 nodes, the compiler's permutation for the two one-lane levels;
//   throughput (many proofs in flight: issue slots count, the latency of one level does not): cooperative up to 512 nodes, the
//   hand-scheduled permutation for all one-lane levels.  bench.py, 32 proofs in flight (tools/gpu_merkle_mode.sh): 87.7-88.8 ->
//   90.0-90.6 proofs/s with thresholds 1024 / 256 (0: 89.2; the level kernel alone: no change).
// BN254S_COOP_MAX_NODES / BN254S_MERKLE_LEVEL_ASM override both modes (tuning).
static long COOP_OVERRIDE = -1;
static int LEVEL_ASM_OVERRIDE = -1;
void merkle_set_throughput_mode(long coop_max_nodes, int level_asm) {
  COOP_OVERRIDE = coop_max_nodes;
  LEVEL_ASM_OVERRIDE = level_asm;
}
static size_t coop_max_nodes(int mode) { return COOP_OVERRIDE >= 0 ? (size_t)COOP_OVERRIDE : mode == MERKLE_THROUGHPUT ? 512 : 16384; }
static bool level_asm(int mode) { return LEVEL_ASM_OVERRIDE >= 0 ? LEVEL_ASM_OVERRIDE != 0 : mode == MERKLE_THROUGHPUT; }

__global__ __launch_bounds__(256) void k_merkle_level_coop(const u64* __restrict__ in, u64* __restrict__ out, size_t n_out) {
  LATENCY_KERNEL_PRIO();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t node = t >> 4;
  const int l = (int)(t & 15);
  const bool live = node < n_out;  // whole 16-lane groups are live or not; every lane still runs the permutation
  u64 s = (live && l < 8) ? in[8 * node + l] : 0;
  s = poseidon_permute_coop(s, l);
  if (live && l < 4) out[4 * node + l] = s;
}

__global__ __launch_bounds__(256) void k_leaf_hash_coop(const u64* __restrict__ data, size_t leaf_stride, size_t elem_stride,
                                                        int leaf_len, size_t n_leaves, u64* __restrict__ digests) {
  LATENCY_KERNEL_PRIO();
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j = t >> 4;
  const int l = (int)(t & 15);
  const bool live = j < n_leaves;
  const u64* p = data + (live ? j : 0) * leaf_stride;
  u64 s = 0;
  for (int c = 0; c < leaf_len; c += 8) {  // leaf_len > 4 (hash_no_pad: overwrite the rate lanes, permute)
    if (l < 8 && c + l < leaf_len) s = p[(size_t)(c + l) * elem_stride];
    s = poseidon_permute_coop(s, l);
  }
  if (live && l < 4) digests[4 * j + l] = s;
}

void merkle_leaves(const u64* data, size_t leaf_stride, size_t elem_stride, int leaf_len, int log_leaves, u64* tree,
                   hipStream_t s, int mode) {
  const size_t COOP_MAX_NODES = coop_max_nodes(mode);
  size_t n = (size_t)1 << log_leaves;
  if (leaf_len <= 4) {
    k_leaf_copy<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(data, leaf_stride, elem_stride, leaf_len, n, tree);
    return;
  }
  if (n * (size_t)((leaf_len + 7) / 8) <= 4 * COOP_MAX_NODES) {  // small trees (FRI layers): latency matters
    k_leaf_hash_coop<<<(unsigned)((16 * n + 255) / 256), 256, 0, s>>>(data, leaf_stride, elem_stride, leaf_len, n, tree);
    return;
  }
  // k_leaf_hash addresses a leaf with a 32-bit byte offset: a launch covers at most 2^29 words of leaf offsets (whole
  // workgroups of 256 leaves); anything larger is cut into several launches over consecutive leaf ranges
  size_t per_launch = std::min(n, (size_t)1 << 27);  // (32-bit byte offsets of the digests, too)
  if (leaf_stride && per_launch * leaf_stride > ((size_t)1 << 29)) per_launch = std::max((size_t)256, ((((size_t)1 << 29) / leaf_stride) / 256) * 256);
  for (size_t j0 = 0; j0 < n; j0 += per_launch) {
    const size_t cnt = std::min(per_launch, n - j0);
    k_leaf_hash<<<(unsigned)((cnt + 255) / 256), 256, 0, s>>>(data + j0 * leaf_stride, leaf_stride, elem_stride, leaf_len, cnt, tree + 4 * j0);
  }
}

// ---- streaming commitment (prover.hip "stream": the LDE of a commitment never exists as a whole) --------------------------------
// hash_no_pad over the columns of a leaf, a chunk of columns at a time: the sponge states of all leaves stay resident between
// chunks (state lane l of leaf j at state[l * n_leaves + j]); a chunk brings `ncols` columns (a multiple of the rate 8, except for
// the last chunk of a commitment), each absorbed block is overwritten into lanes 0.. and permuted, exactly the sequence
// hash_no_pad runs.  `first`: start from the zero state; digests != nullptr (last chunk): write the digest instead of the state.
__global__ __launch_bounds__(256) void k_leaf_absorb(const u64* __restrict__ data, size_t elem_stride, int ncols, size_t n_leaves,
                                                     u64* __restrict__ state, int first, u64* __restrict__ digests) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_leaves) return;
  u64 s[12];
#pragma unroll
  for (int l = 0; l < 12; l++) s[l] = first ? 0 : state[(size_t)l * n_leaves + j];
#pragma unroll 1
  for (int c = 0; c < ncols; c += 8) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (c + i < ncols) s[i] = data[(size_t)(c + i) * elem_stride + j];
    poseidon_permute(s);
  }
  if (digests) {
    ulonglong2* o = reinterpret_cast<ulonglong2*>(digests + 4 * j);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
  } else {
#pragma unroll
    for (int l = 0; l < 12; l++) state[(size_t)l * n_leaves + j] = s[l];
  }
}
// leaves [j0, j0 + cnt) only (cnt a multiple of 256; leaf_len > 4): a commitment's leaf hash cut into several launches
void merkle_leaves_range(const u64* data, size_t leaf_stride, size_t elem_stride, int leaf_len, size_t j0, size_t cnt, u64* tree,
                         hipStream_t s) {
  size_t per_launch = std::min(cnt, (size_t)1 << 27);
  if (leaf_stride && per_launch * leaf_stride > ((size_t)1 << 29)) per_launch = std::max((size_t)256, ((((size_t)1 << 29) / leaf_stride) / 256) * 256);
  for (size_t a = 0; a < cnt; a += per_launch) {
    const size_t n = std::min(per_launch, cnt - a);
    k_leaf_hash<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(data + (j0 + a) * leaf_stride, leaf_stride, elem_stride, leaf_len, n, tree + 4 * (j0 + a));
  }
}
void merkle_absorb(const u64* data, size_t elem_stride, int ncols, int log_leaves, u64* state, bool first, u64* digests, hipStream_t s) {
  const size_t n = (size_t)1 << log_leaves;
  k_leaf_absorb<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(data, elem_stride, ncols, n, state, first ? 1 : 0, digests);
}

void merkle_upper(int log_leaves, int cap_height, u64* tree, hipStream_t s, int mode) {
  const size_t COOP_MAX_NODES = coop_max_nodes(mode);
  const bool MERKLE_LEVEL_ASM = level_asm(mode);
  for (int l = 0; l < log_leaves - cap_height; l++) {
    size_t n_out = (size_t)1 << (log_leaves - l - 1);
    const u64* in = tree + 4 * merkle_level_offset(log_leaves, l);
    u64* out = tree + 4 * merkle_level_offset(log_leaves, l + 1);
    if (n_out <= COOP_MAX_NODES) k_merkle_level_coop<<<(unsigned)((16 * n_out + 255) / 256), 256, 0, s>>>(in, out, n_out);
    else if (MERKLE_LEVEL_ASM) k_merkle_level<true><<<(unsigned)((n_out + 255) / 256), 256, 0, s>>>(in, out, n_out);
    else k_merkle_level<false><<<(unsigned)((n_out + 255) / 256), 256, 0, s>>>(in, out, n_out);
  }
}

void merkle_build(const u64* data, size_t leaf_stride, size_t elem_stride, int leaf_len, int log_leaves, int cap_height,
                  u64* tree, hipStream_t s, int mode) {
  merkle_leaves(data, leaf_stride, elem_stride, leaf_len, log_leaves, tree, s, mode);
  merkle_upper(log_leaves, cap_height, tree, s, mode);
}

// loads this translation unit's code object (the HIP runtime defers that to the first launch otherwise)
void merkle_module_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_merkle_level<false>));
}
