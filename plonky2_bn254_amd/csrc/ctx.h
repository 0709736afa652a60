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
};

struct bn254s_ctx : BufPool {
  int device = 0;
  hipStream_t stream = nullptr;
  NttTables ntt;
  std::vector<Slot*> slots;
  std::map<unsigned, NttTallTables*> tall;  // per log_n
  // GPU-saturating kernels (NTT, leaf hashing, quotient, ...) of different slots are serialised with this lock:
  // they cannot run faster side by side, and alone they give clean per-kernel timings.  Latency-bound
  // kernels (EC chain, upper Merkle levels, scans, PoW) run outside it and overlap freely.
  std::mutex big_mu;
  std::condition_variable big_cv;
  int big_excl = 0, big_shared = 0;  // holders of the exclusive class / of the shared (hash) class
  // Exclusive sections (NTT, quotient, openings, FRI combine) run alone; the Poseidon leaf-hash kernels of up to
  // two proofs may run side by side (one 2^17-leaf launch puts only two waves on a SIMD).
  void big_lock(bool shared) {
    std::unique_lock<std::mutex> lk(big_mu);
    if (shared) {
      big_cv.wait(lk, [&] { return big_excl == 0 && big_shared < 2; });
      big_shared++;
    } else {
      big_cv.wait(lk, [&] { return big_excl == 0 && big_shared == 0; });
      big_excl = 1;
    }
  }
  void big_unlock(bool shared) {
    {
      std::lock_guard<std::mutex> lk(big_mu);
      if (shared) big_shared--;
      else big_excl = 0;
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
