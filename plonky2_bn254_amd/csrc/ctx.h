// Library context: device, stream, tables and pooled device workspace.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <mutex>
#include <condition_variable>
#include <hip/hip_runtime.h>
#include "../../include/bn254_stark.h"
#include "gl_dev.h"
#include "ntt.h"

// Named, grow-only device buffers (kept for the life of their owner).
struct BufPool {
  std::map<std::string, std::pair<void*, size_t>> pool;
  std::string err;
  void* buf(const std::string& name, size_t bytes) {
    auto it = pool.find(name);
    if (it != pool.end() && it->second.second >= bytes) return it->second.first;
    if (it != pool.end()) {
      hipFree(it->second.first);
      pool.erase(it);
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      err = "hipMalloc failed for " + name + " (" + std::to_string(bytes) + " bytes)";
      return nullptr;
    }
    pool[name] = {p, bytes};
    return p;
  }
  u64* words(const std::string& name, size_t n) { return (u64*)buf(name, n * 8); }
  bool has(const std::string& name) const { return pool.find(name) != pool.end(); }
  void drop(const std::string& name) {
    auto it = pool.find(name);
    if (it == pool.end()) return;
    hipFree(it->second.first);
    pool.erase(it);
  }
  void release() {
    for (auto& kv : pool) hipFree(kv.second.first);
    pool.clear();
  }
};

// A slot = stream + workspace of one proof in flight (bn254s_prove_g1_batch pipelines several).
struct Slot {
  hipStream_t st = nullptr;
  BufPool mem;
  void* pinned = nullptr;  // pinned host staging buffer
  size_t pinned_bytes = 0;
  std::vector<hipEvent_t> events;  // stage begin / end pairs, created on first use and kept for the slot's lifetime
};

enum { BIG_NTT = 0, BIG_EXCL = 1, BIG_HASH = 2 };

struct bn254s_ctx : BufPool {
  int device = 0;
  hipStream_t stream = nullptr;
  NttTables ntt;
  std::vector<Slot*> slots;
  std::map<unsigned, NttTallTables*> tall;  // per log_n (+100: halves of a split transform)
  std::map<unsigned, NttSplitTables*> split;  // per log_n
  // The GPU-filling sections of the proofs in flight take turns through the semaphore below; latency-bound kernels
  // (doubling chain, upper Merkle levels, scans, PoW, FRI folds) run outside it and overlap freely.
  std::mutex big_mu;
  std::condition_variable big_cv;
  // Weighted semaphore over the GPU-filling sections of all proofs in flight.  Classes (cost out of big_cap = 6):
  //   BIG_NTT  (6): the NTT/LDE stage runs alone - it is the stage the roofline figure is quoted on;
  //   BIG_EXCL (3): quotient, auxiliary columns, range-check histogram, openings, FRI combine;
  //   BIG_HASH (3): Poseidon leaf hashing - one 2^17-leaf launch puts two waves on a SIMD; two such sections run together
  //                 (round 2 sweep with the hand-scheduled hash, tools/gpu_locksweep.sh: 70.1 proofs/s against 68.2 for the
  //                 round-1 costs 3 / 2 / 1 out of 3 with six slots; letting the NTT stage share the GPU gives 71.2).
  // BN254S_BIG_CAP / BN254S_BIG_COST_NTT / BN254S_BIG_COST_EXCL / BN254S_BIG_COST_HASH override the costs (tuning only).
  int big_cap = 6, big_cost[3] = {6, 3, 3}, big_used = 0;
  void big_lock(int cls) {
    std::unique_lock<std::mutex> lk(big_mu);
    big_cv.wait(lk, [&] { return big_used + big_cost[cls] <= big_cap; });
    big_used += big_cost[cls];
  }
  void big_unlock(int cls) {
    {
      std::lock_guard<std::mutex> lk(big_mu);
      big_used -= big_cost[cls];
    }
    big_cv.notify_all();
  }
  Slot* slot(size_t i) {
    while (slots.size() <= i) {
      Slot* s = new Slot();
      if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess) {
        delete s;
        return nullptr;
      }
      slots.push_back(s);
    }
    return slots[i];
  }
};

#define HIP_TRY(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return BN254S_E_HIP;                                                                  \
    }                                                                                       \
  } while (0)
