// Library context: device, stream, tables and pooled device workspace.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <hip/hip_runtime.h>
#include "../../include/bn254_stark.h"
#include "gl_dev.h"
#include "ntt.h"

struct bn254s_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  NttTables ntt;
  std::string err;
  // pooled device buffers, keyed by name; grown on demand and kept for the life of the context
  std::map<std::string, std::pair<void*, size_t>> pool;

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
};

#define HIP_TRY(ctx, call)                                                                  \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return BN254S_E_HIP;                                                                  \
    }                                                                                       \
  } while (0)
