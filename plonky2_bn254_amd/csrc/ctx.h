// Library context: device, stream, tables and pooled device workspace.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <deque>
#include <functional>
#include <thread>
#include <atomic>
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
      (void)hipGetLastError();  // the failure is reported through the return value: do not leave it as the thread's sticky error
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

enum { BIG_NTT = 0, BIG_EXCL = 1, BIG_HASH = 2, BIG_HASH_PART = 3 };

// The host threads that drive the slots: a worker takes the next proof queued by the batch entry points (arrival order, across
// calls) and the lowest free slot - the first proofs of the next batch start while the last proofs of the current one finish
// (bn254s_prove_batch_begin / _end).  Started on first use, joined by bn254s_ctx_destroy.
struct WorkPool {
  std::mutex mu;
  std::condition_variable cv, done_cv;
  std::deque<std::function<void(size_t)>> q;
  std::vector<std::thread> th;
  std::vector<char> busy;  // per slot; a task runs on the LOWEST free slot, so k proofs in flight only ever touch slots 0..k-1
  bool stop = false;
  size_t limit = 0, n_busy = 0;  // proofs in flight <= limit (BN254S_SLOTS of the latest batch call)
  size_t n_waiting = 0, completions = 0;
  std::mutex retry_mu;  // one task at a time gives idle workspaces back and allocates again after BN254S_E_OOM
  std::atomic<int> in_flight{0};  // = n_busy, readable without the mutex (the provers pick throughput / latency kernels by it)
  void start(size_t n, int device) {
    std::lock_guard<std::mutex> lk(mu);
    limit = n;
    if (busy.size() < n) busy.resize(n, 0);
    while (th.size() < n) {
      th.emplace_back([this, device]() {
        (void)hipSetDevice(device);
        for (;;) {
          std::function<void(size_t)> f;
          size_t s = 0;
          {
            std::unique_lock<std::mutex> lk2(mu);
            auto free_slot = [&]() -> size_t {  // lowest slot that is neither running a task nor claimed by for_idle_slots
              for (size_t i = 0; i < limit && i < busy.size(); i++)
                if (!busy[i]) return i;
              return (size_t)-1;
            };
            cv.wait(lk2, [&] { return (stop && q.empty()) || (!q.empty() && n_busy < limit && free_slot() != (size_t)-1); });
            if (q.empty()) return;  // stop requested and nothing left
            f = std::move(q.front());
            q.pop_front();
            s = free_slot();
            busy[s] = 1;
            n_busy++;
            in_flight.store((int)n_busy, std::memory_order_relaxed);
          }
          f(s);
          {
            std::lock_guard<std::mutex> lk3(mu);
            busy[s] = 0;
            n_busy--;
            in_flight.store((int)n_busy, std::memory_order_relaxed);
            completions++;
          }
          cv.notify_one();  // a worker may be waiting for a free slot below the limit
          done_cv.notify_all();
        }
      });
    }
  }
  void push(std::function<void(size_t)> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      q.push_back(std::move(f));
    }
    cv.notify_one();
  }
  // calls g(i) for every slot that is idle right now.  The idle slots are marked busy while g runs, so no task can start on
  // one of them meanwhile, and g (hipFree: it synchronises the device) runs OUTSIDE the pool's mutex: running tasks finish and
  // queue as usual.
  template <class G>
  void for_idle_slots(G g) {
    std::vector<size_t> idle;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < busy.size(); i++)
        if (!busy[i]) {
          busy[i] = 2;  // claimed (not a running task: n_busy is unchanged)
          idle.push_back(i);
        }
    }
    for (size_t i : idle) g(i);
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i : idle) busy[i] = 0;
    }
    cv.notify_all();
  }
  // tasks that are running and not parked in wait_for_a_completion (the caller counts itself)
  size_t active() {
    std::lock_guard<std::mutex> lk(mu);
    return n_busy - n_waiting;
  }
  size_t completions_now() {
    std::lock_guard<std::mutex> lk(mu);
    return completions;
  }
  // parks the calling task until a task has ended since `c0` was sampled (completions_now() BEFORE the failed attempt: a
  // completion between the attempt and this call is not lost), or a minute has passed
  void wait_for_a_completion(size_t c0) {
    std::unique_lock<std::mutex> lk(mu);
    n_waiting++;
    done_cv.wait_for(lk, std::chrono::seconds(60), [&] { return completions != c0; });
    n_waiting--;
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
    th.clear();
  }
};

struct bn254s_ctx : BufPool {
  int device = 0;
  hipStream_t stream = nullptr;
  NttTables ntt;
  std::vector<Slot*> slots;
  WorkPool workers;
  std::map<unsigned, NttTallTables*> tall;  // per log_n (+100: halves of a split transform)
  std::map<unsigned, NttSplitTables*> split;  // per log_n
  // The GPU-filling sections of the proofs in flight take turns through the semaphore below; latency-bound kernels
  // (doubling chain, upper Merkle levels, scans, PoW, FRI folds) run outside it and overlap freely.
  std::mutex big_mu;
  std::condition_variable big_cv;
  // Weighted semaphore over the GPU-filling sections of all proofs in flight, admission in arrival order.  Classes (cost out
  // of big_cap = 12):
  //   BIG_NTT  (12): the NTT/LDE stage runs alone - it is the stage the roofline figure is quoted on, and it fills the machine;
  //   BIG_HASH  (4): Poseidon leaf hashing (a 2^17-leaf launch puts two 127-register waves on every SIMD: three at a time);
  //   BIG_EXCL  (2): quotient, auxiliary columns, range-check histogram, openings, FRI combine;
  // Arrival order matters: without it a waiting NTT stage (which needs the whole capacity) is overtaken by later, cheaper
  // sections - at the start of a call by the range-check histograms of all the other proofs - and the first wide arithmetic
  // of a step starts milliseconds late.  Round 3 (bench.py, 32 proofs queued on 32 slots, tools/gpu_cost_sweep.sh): capacity 9
  // with costs 9 / 3 / 3 (round 2) 86.0-86.5 proofs/s; 12 with 12 / 3 / 3: 87.4-88.5; 12 with 12 / 4 / 2: 88.5-88.9; 15 with
  // 15 / 3 / 3: 88.2; 18: 85.6; an NTT stage that shares the GPU (cost 6 of 9) gains 3 % and doubles its own time.
  // BN254S_BIG_CAP / BN254S_BIG_COST_NTT / BN254S_BIG_COST_EXCL / BN254S_BIG_COST_HASH / BN254S_SCHED_FIFO override (tuning only; FIFO
  // admission is the default, BN254S_SCHED_FIFO=0 switches it off).
  int big_cap = 12, big_cost[4] = {12, 2, 4, 2}, big_used = 0;  // indexed by BIG_NTT, BIG_EXCL, BIG_HASH, BIG_HASH_PART
  int hash_split = 1;  // BN254S_HASH_SPLIT: a 2^16-row proof's leaf hash as this many sections of cost big_cost[BIG_HASH] / split
  bool big_fifo = true;  // BN254S_SCHED_FIFO=0: the unordered semaphore (A/B runs)
  // Waiters in arrival order.  The head is admitted as soon as it fits.  One exception ("convoy"): an NTT stage needs the whole
  // capacity, i.e. the GPU drains before it starts; when it ends and other NTT stages are waiting further back, up to
  // big_convoy of them run back to back right away (the capacity is free at that moment) instead of letting cheaper sections
  // in and draining again for each of them.  Bounded, so nobody starves: BN254S_NTT_CONVOY (0 = strict arrival order).
  // bench.py, 32 proofs in flight (tools/gpu_convoy_ab.sh): 0: 89.8-90.2, 3: 91.2-91.4, 8: 91.0-91.5, 16: 91.8 proofs/s.
  struct BigWaiter {
    unsigned long id;
    int cls;
  };
  std::deque<BigWaiter> big_q;
  unsigned long big_next_id = 0;
  int big_convoy = 4, big_convoy_run = 0;  // NTT stages admitted out of order since the last in-order admission
  bool big_last_was_ntt = false;           // the capacity was last released by an NTT stage and nothing was admitted since
  void big_lock(int cls) {
    std::unique_lock<std::mutex> lk(big_mu);
    if (big_fifo) {
      const unsigned long my = big_next_id++;
      big_q.push_back({my, cls});
      big_cv.wait(lk, [&] {
        if (big_q.front().id == my) {
          if (big_used + big_cost[cls] > big_cap) return false;
          // the head yields once to a convoy: an NTT stage just ended, this waiter is not one, an NTT waiter stands behind it
          if (big_last_was_ntt && cls != BIG_NTT && big_convoy_run < big_convoy)
            for (const BigWaiter& w : big_q)
              if (w.cls == BIG_NTT) return false;
          return true;
        }
        // not the head: only an NTT waiter, directly after an NTT stage, with the whole capacity free, first of its class in line
        if (cls != BIG_NTT || !big_last_was_ntt || big_used != 0 || big_convoy_run >= big_convoy) return false;
        if (big_q.front().cls == BIG_NTT) return false;  // the head is an NTT stage itself: it goes first, in order
        for (const BigWaiter& w : big_q) {
          if (w.cls == BIG_NTT) return w.id == my;
        }
        return false;
      });
      const bool in_order = big_q.front().id == my;
      for (auto it = big_q.begin(); it != big_q.end(); ++it)
        if (it->id == my) {
          big_q.erase(it);
          break;
        }
      if (in_order) big_convoy_run = 0;
      else big_convoy_run++;
      big_last_was_ntt = false;
      big_used += big_cost[cls];
      lk.unlock();
      big_cv.notify_all();  // the next waiter may fit beside this one
      return;
    }
    big_cv.wait(lk, [&] { return big_used + big_cost[cls] <= big_cap; });
    big_used += big_cost[cls];
  }
  void big_unlock(int cls) {
    {
      std::lock_guard<std::mutex> lk(big_mu);
      big_used -= big_cost[cls];
      big_last_was_ntt = cls == BIG_NTT && big_used == 0;
    }
    big_cv.notify_all();
  }
  // Slots are created on demand by the entry points (caller's thread) while workers of an earlier batch read slots[s]: the
  // vector never reallocates (capacity reserved in the constructor, BN254S_SLOTS is clamped to MAX_SLOTS) and grows under slots_mu.
  static constexpr size_t MAX_SLOTS = 64;
  std::mutex slots_mu, err_mu;
  bn254s_ctx() { slots.reserve(MAX_SLOTS); }
  void set_err(const std::string& e) {
    std::lock_guard<std::mutex> lk(err_mu);
    err = e;
  }
  size_t n_slots_now() {
    std::lock_guard<std::mutex> lk(slots_mu);
    return slots.size();
  }
  Slot* slot(size_t i) {
    if (i >= MAX_SLOTS) return nullptr;
    std::lock_guard<std::mutex> lk(slots_mu);
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
      (ctx)->set_err(std::string(#call) + ": " + hipGetErrorString(e_));                    \
      return BN254S_E_HIP;                                                                  \
    }                                                                                       \
  } while (0)
