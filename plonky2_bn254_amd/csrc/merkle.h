// Poseidon Merkle commitment kernels (merkle.hip): leaf hashing + tree levels with a cap.
#pragma once
#include "gl_dev.h"

// Digest storage: 4 consecutive u64 per digest.  A tree over 2^L leaves with cap height H keeps levels
// 0..L-H concatenated: level l starts at digest index merkle_level_offset(L, l) and has 2^(L-l) digests.
static inline size_t merkle_level_offset(int log_leaves, int level) {
  size_t off = 0;
  for (int l = 0; l < level; l++) off += (size_t)1 << (log_leaves - l);
  return off;
}
static inline size_t merkle_tree_digests(int log_leaves, int cap_height) { return merkle_level_offset(log_leaves, log_leaves - cap_height + 1); }

// Hash 2^log_leaves leaves of `leaf_len` elements (element e of leaf j at data[j*leaf_stride + e*elem_stride])
// and build all levels up to the cap.  tree must hold merkle_tree_digests() * 4 words.
// mode: MERKLE_LATENCY (a proof alone) or MERKLE_THROUGHPUT (many proofs in flight) picks the kernels of the small levels
// (merkle.hip); the digests are the same.
enum { MERKLE_LATENCY = 0, MERKLE_THROUGHPUT = 1 };
void merkle_build(const u64* data, size_t leaf_stride, size_t elem_stride, int leaf_len, int log_leaves, int cap_height,
                  u64* tree, hipStream_t s, int mode = MERKLE_LATENCY);
// The two halves of merkle_build: leaf digests only (GPU-saturating) and the upper levels (latency-bound).
void merkle_leaves(const u64* data, size_t leaf_stride, size_t elem_stride, int leaf_len, int log_leaves, u64* tree,
                   hipStream_t s, int mode = MERKLE_LATENCY);
void merkle_upper(int log_leaves, int cap_height, u64* tree, hipStream_t s, int mode = MERKLE_LATENCY);
// the leaf hash of leaves [j0, j0 + cnt) of a wide commitment (leaf_len > 4, cnt a multiple of 256)
void merkle_leaves_range(const u64* data, size_t leaf_stride, size_t elem_stride, int leaf_len, size_t j0, size_t cnt, u64* tree,
                         hipStream_t s);
// Streaming leaf hash: absorbs `ncols` more columns (data[c * elem_stride + j]) into the resident sponge states [12][2^log_leaves];
// `first` starts from the zero state, digests != nullptr (last chunk of the commitment) writes the leaf digests (tree level 0).
void merkle_absorb(const u64* data, size_t elem_stride, int ncols, int log_leaves, u64* state, bool first, u64* digests, hipStream_t s);
// tuning: overrides both modes: largest level handled by the cooperative kernels (-1: per mode) and whether the one-lane level kernel
// uses the hand-scheduled permutation (-1: per mode).  Process-wide.
void merkle_set_throughput_mode(long coop_max_nodes, int level_asm);
