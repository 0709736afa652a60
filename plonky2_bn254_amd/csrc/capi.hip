// C ABI of libbn254stark.so (include/bn254_stark.h): context management and kernel-level entry points.
// The proving entry points live in prover.hip.
#include <algorithm>
#include <cstring>
#include "ctx.h"
#include "merkle.h"
#include "trace_g1.h"
#include "poseidon_dev.h"

// one per translation unit: loads its code object (defined at the end of each .hip file)
void ntt_module_warm();
void merkle_module_warm();
void aux_module_warm();
void fri_module_warm();
void quotient_g1_module_warm();
void quotient_g2fq_module_warm();
void trace_g1_module_warm();
void trace_g2fq_module_warm();
void prover_module_warm();

extern "C" {

int bn254s_abi_version(void) { return BN254S_ABI_VERSION; }

void bn254s_params_default(bn254s_params* p) {
  p->struct_size = sizeof(bn254s_params);
  p->security_bits = 100;
  p->num_challenges = 2;
  p->rate_bits = 1;
  p->cap_height = 4;
  p->pow_bits = 16;
  p->arity_bits = 4;
  p->final_poly_bits = 5;
  p->num_queries = 84;
  p->min_rows_log2 = 16;
}

// Six proofs in flight use six HIP streams; the runtime's default of four hardware queues would make pairs of streams share
// one queue and serialise behind each other's long kernels (measured: 36 -> 40 proofs/s).  The variable is read when the HIP
// runtime initialises, so it is set when the library is loaded and never overrides a value the user chose.
__attribute__((constructor)) static void bn254s_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }


int bn254s_ctx_create(int device_id, bn254s_ctx** out) {
  if (!out) return BN254S_E_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return BN254S_E_HIP;
  // BN254S_SYNC=blocking: host threads waiting for the GPU sleep on an interrupt instead of spinning (takes effect only if this is
  // the first use of the device in the process; with N ranks x 32 proof threads on one host the spinning threads outnumber the cores)
  if (const char* e = getenv("BN254S_SYNC"))
    if (!strcmp(e, "blocking")) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
  (void)hipGetLastError();
  if (hipSetDevice(device_id) != hipSuccess) return BN254S_E_HIP;
  bn254s_ctx* c = new bn254s_ctx();
  c->device = device_id;
  merkle_set_throughput_mode(getenv("BN254S_COOP_MAX_NODES") ? atol(getenv("BN254S_COOP_MAX_NODES")) : -1,
                             getenv("BN254S_MERKLE_LEVEL_ASM") ? atoi(getenv("BN254S_MERKLE_LEVEL_ASM")) : -1);
  if (const char* e = getenv("BN254S_BIG_CAP")) c->big_cap = std::max(1, atoi(e));
  if (const char* e = getenv("BN254S_SCHED_FIFO")) c->big_fifo = atoi(e) != 0;
  if (const char* e = getenv("BN254S_NTT_CONVOY")) c->big_convoy = std::max(0, atoi(e));
  if (const char* e = getenv("BN254S_HASH_SPLIT")) c->hash_split = std::min(8, std::max(1, atoi(e)));
  c->big_cost[BIG_NTT] = c->big_cap;
  const char* cost_env[3] = {"BN254S_BIG_COST_NTT", "BN254S_BIG_COST_EXCL", "BN254S_BIG_COST_HASH"};
  for (int k = 0; k < 3; k++) {
    if (const char* e = getenv(cost_env[k])) c->big_cost[k] = atoi(e);
    c->big_cost[k] = std::min(c->big_cap, std::max(0, c->big_cost[k]));
  }
  c->big_cost[BIG_HASH_PART] = std::max(1, c->big_cost[BIG_HASH] / c->hash_split);
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return BN254S_E_HIP;
  }
  if (ntt_tables_init(&c->ntt) != 0) {
    delete c;
    return BN254S_E_OOM;
  }
  {  // first use of the pinned device -> host copy path sets up runtime state (~8 ms): not inside the first proof either
    void* h = nullptr;
    if (hipHostMalloc(&h, 256 << 10) == hipSuccess) {
      (void)hipMemcpyAsync(h, c->ntt.twmat_fwd, 256 << 10, hipMemcpyDeviceToHost, c->stream);  // (a 512 KB table)
      (void)hipStreamSynchronize(c->stream);
      hipHostFree(h);
    }
  }
  // code objects are loaded lazily, one per translation unit: do it now rather than inside the first proof
  ntt_module_warm();
  merkle_module_warm();
  aux_module_warm();
  fri_module_warm();
  quotient_g1_module_warm();
  quotient_g2fq_module_warm();
  trace_g1_module_warm();
  trace_g2fq_module_warm();
  prover_module_warm();
  *out = c;
  return BN254S_OK;
}

void bn254s_ctx_destroy(bn254s_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  c->workers.shutdown();  // finishes what is queued
  hipStreamSynchronize(c->stream);
  for (Slot* s : c->slots) {
    hipStreamSynchronize(s->st);
    s->mem.release();
    if (s->pinned) hipHostFree(s->pinned);
    for (auto& e : s->events) hipEventDestroy(e);
    hipStreamDestroy(s->st);
    delete s;
  }
  for (auto& kv : c->tall) {
    ntt_tall_tables_free(kv.second);
    delete kv.second;
  }
  for (auto& kv : c->split) {
    ntt_split_tables_free(kv.second);
    delete kv.second;
  }
  c->release();
  ntt_tables_free(&c->ntt);
  hipStreamDestroy(c->stream);
  delete c;
}

int bn254s_ctx_trim(bn254s_ctx* c) {
  if (!c) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  auto give_back = [&](size_t i) {
    if (i < c->n_slots_now()) c->slot(i)->mem.release();
  };
  if (c->workers.busy.empty()) {  // no batch call yet: no worker can be running
    for (size_t i = 0; i < c->n_slots_now(); i++) give_back(i);
  } else {
    c->workers.for_idle_slots(give_back);
  }
  // the context's own pool (buffers of the kernel-level entry points, cached point tables: rebuilt on demand) - only while no
  // proof is queued or running, since running proofs hold pointers into it
  bool idle;
  {
    std::lock_guard<std::mutex> lk(c->workers.mu);
    idle = c->workers.n_busy == 0 && c->workers.q.empty();
  }
  if (idle) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->release();
  }
  return BN254S_OK;
}

const char* bn254s_last_error(const bn254s_ctx* c) { return c ? c->err.c_str() : "null context"; }

int bn254s_commit_values(bn254s_ctx* c, const uint64_t* values, size_t ncols, uint64_t* coeffs, uint64_t* lde,
                         uint64_t* cap) {
  if (!c || !values || ncols == 0) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t N = NTT_N;
  u64* d_vals = c->words("cv.vals", ncols * N);
  u64* d_tmp = c->words("cv.tmp", 3 * ncols * N);
  u64* d_lde = c->words("cv.lde", ncols * 2 * N);
  u64* d_tree = c->words("cv.tree", merkle_tree_digests(17, 4) * 4);
  if (!d_vals || !d_tmp || !d_lde || !d_tree) return BN254S_E_OOM;
  HIP_TRY(c, hipMemcpyAsync(d_vals, values, ncols * N * 8, hipMemcpyHostToDevice, c->stream));
  ntt_inverse_lde(&c->ntt, d_vals, d_vals, d_lde, d_tmp, d_tmp + ncols * N, (int)ncols, c->stream);
  merkle_build(d_lde, 1, 2 * N, (int)ncols, 17, 4, d_tree, c->stream);
  HIP_TRY(c, hipGetLastError());
  if (coeffs) HIP_TRY(c, hipMemcpyAsync(coeffs, d_vals, ncols * N * 8, hipMemcpyDeviceToHost, c->stream));
  if (lde) HIP_TRY(c, hipMemcpyAsync(lde, d_lde, ncols * 2 * N * 8, hipMemcpyDeviceToHost, c->stream));
  if (cap)
    HIP_TRY(c, hipMemcpyAsync(cap, d_tree + 4 * merkle_level_offset(17, 13), 16 * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BN254S_OK;
}

__global__ void k_fill_pseudo(u64* p, size_t n, u64 seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 z = seed + i * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  p[i] = z >= GL_P ? z - GL_P : z;
}

int bn254s_bench_ntt(bn254s_ctx* c, size_t ncols, int iters, float* ms) {
  if (!c || !ms || ncols == 0 || iters <= 0) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t N = NTT_N;
  u64* d_vals = c->words("cv.vals", ncols * N);
  u64* d_tmp = c->words("cv.tmp", 3 * ncols * N);
  u64* d_lde = c->words("cv.lde", ncols * 2 * N);
  if (!d_vals || !d_tmp || !d_lde) return BN254S_E_OOM;
  size_t n = ncols * N;
  k_fill_pseudo<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(d_vals, n, 0x1234);
  hipEvent_t e0, e1;
  HIP_TRY(c, hipEventCreate(&e0));
  HIP_TRY(c, hipEventCreate(&e1));
  // warm-up
  ntt_inverse_lde(&c->ntt, d_vals, d_vals, d_lde, d_tmp, d_tmp + n, (int)ncols, c->stream);
  HIP_TRY(c, hipEventRecord(e0, c->stream));
  for (int it = 0; it < iters; it++) ntt_inverse_lde(&c->ntt, d_vals, d_vals, d_lde, d_tmp, d_tmp + n, (int)ncols, c->stream);
  HIP_TRY(c, hipEventRecord(e1, c->stream));
  HIP_TRY(c, hipEventSynchronize(e1));
  float t = 0;
  HIP_TRY(c, hipEventElapsedTime(&t, e0, e1));
  *ms = t / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  HIP_TRY(c, hipGetLastError());
  return BN254S_OK;
}

// Shader clock while a stage runs: ONE wave samples the core-clock counter (s_memtime) against the constant 100 MHz counter
// (s_memrealtime) every ~10 us until `stop` is set by the measured stream (or max_samples are taken: the loop is bounded).
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* __restrict__ samples, const int* __restrict__ stop, int max_samples,
                                                    int* __restrict__ n_taken) {
  if (threadIdx.x != 0) return;
  int i = 0;
  for (; i < max_samples; i++) {
    samples[2 * i] = __builtin_readcyclecounter();
    samples[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
    if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      i++;
      break;
    }
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
  }
  *n_taken = i;
}
// Issue cost of the half-rate vector instructions the NTT and Poseidon kernels are made of (v_lshl_add_u64 here; v_mad_u64_u32,
// 64-bit shifts / compares, carry instructions cost the same: tools/ubench/sgpr_ops.hip): 8 waves per SIMD, 4 independent
// chains per wave, 2048 x 32 instructions per wave.
__global__ __launch_bounds__(256) void k_issue_probe(u64* out) {
  for (int i = 0; i < 2048; i++) {
    asm volatile(
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n"
        "v_lshl_add_u64 v[40:41], v[40:41], 0, v[48:49]\n v_lshl_add_u64 v[42:43], v[42:43], 0, v[48:49]\n"
        "v_lshl_add_u64 v[44:45], v[44:45], 0, v[48:49]\n v_lshl_add_u64 v[46:47], v[46:47], 0, v[48:49]\n" ::
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
  }
  u32 a;
  asm volatile("v_mov_b32 %0, v40" : "=v"(a));
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// ns per wave-instruction and SIMD of the half-rate class with 8 waves per SIMD, and the shader clock held meanwhile.
int bn254s_bench_issue(bn254s_ctx* c, float* ns_per_issue, float* mhz) {
  if (!c || !ns_per_issue || !mhz) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
  const int simds = prop.multiProcessorCount * 4, blocks = simds * 8 / 4;  // 256-thread blocks = 4 waves; 8 waves per SIMD
  const int MAXS = 20000;
  u64* d_out = c->words("iss.out", (size_t)blocks * 256);
  unsigned long long* d_s = (unsigned long long*)c->words("clk.samples", 2 * (size_t)MAXS + 2);
  if (!d_out || !d_s) return BN254S_E_OOM;
  int* d_flags = (int*)(d_s + 2 * (size_t)MAXS);
  hipStream_t ps = nullptr;
  HIP_TRY(c, hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  HIP_TRY(c, hipMemsetAsync(d_flags, 0, 8, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  hipEvent_t e0, e1;
  HIP_TRY(c, hipEventCreate(&e0));
  HIP_TRY(c, hipEventCreate(&e1));
  k_clock_probe<<<1, 64, 0, ps>>>(d_s, d_flags, MAXS, d_flags + 1);
  const int reps = 12;
  for (int i = 0; i < 4; i++) k_issue_probe<<<blocks, 256, 0, c->stream>>>(d_out);  // clock ramp
  HIP_TRY(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < reps; i++) k_issue_probe<<<blocks, 256, 0, c->stream>>>(d_out);
  HIP_TRY(c, hipEventRecord(e1, c->stream));
  (void)hipMemsetAsync(d_flags, 1, 1, c->stream);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipStreamSynchronize(ps);
  hipStreamDestroy(ps);
  float t = 0;
  HIP_TRY(c, hipEventElapsedTime(&t, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ns_per_issue = (float)((double)t * 1e6 / reps / (2048.0 * 32 * 8));
  std::vector<unsigned long long> h(2 * (size_t)MAXS + 2);
  HIP_TRY(c, hipMemcpy(h.data(), d_s, h.size() * 8, hipMemcpyDeviceToHost));
  const int n = ((const int*)&h[2 * (size_t)MAXS])[1];
  *mhz = 0;
  if (n >= 8) {
    const int lo = n / 4;
    const double dt = (double)(h[2 * (size_t)(n - 1) + 1] - h[2 * (size_t)lo + 1]), dc = (double)(h[2 * (size_t)(n - 1)] - h[2 * (size_t)lo]);
    if (dt > 0) *mhz = (float)(dc / dt * 100.0);
  }
  return BN254S_OK;
}

// bn254s_bench_ntt with the shader clock of the GPU measured while the stage runs (mhz: mean over the timed iterations,
// mhz_min: the slowest ~10 us interval).  The probe is one wave on a second stream.
int bn254s_bench_ntt_clock(bn254s_ctx* c, size_t ncols, int iters, float* ms, float* mhz, float* mhz_min) {
  if (!c || !ms || !mhz || ncols == 0 || iters <= 0) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  const int MAXS = 20000;
  unsigned long long* d_s = (unsigned long long*)c->words("clk.samples", 2 * (size_t)MAXS + 2);
  if (!d_s) return BN254S_E_OOM;
  int* d_flags = (int*)(d_s + 2 * (size_t)MAXS);  // [0] stop, [1] samples taken
  hipStream_t ps = nullptr;
  HIP_TRY(c, hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  HIP_TRY(c, hipMemsetAsync(d_flags, 0, 8, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  k_clock_probe<<<1, 64, 0, ps>>>(d_s, d_flags, MAXS, d_flags + 1);
  int rc = bn254s_bench_ntt(c, ncols, iters, ms);
  (void)hipMemsetAsync(d_flags, 1, 1, c->stream);  // stop (stream-ordered after the timed launches)
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(ps);
  hipStreamDestroy(ps);
  if (rc != BN254S_OK) return rc;
  std::vector<unsigned long long> h(2 * (size_t)MAXS + 2);
  HIP_TRY(c, hipMemcpy(h.data(), d_s, h.size() * 8, hipMemcpyDeviceToHost));
  const int n = ((const int*)&h[2 * (size_t)MAXS])[1];
  if (n < 8) {
    *mhz = 0;
    if (mhz_min) *mhz_min = 0;
    return BN254S_OK;
  }
  const int lo = n / 8;  // the first samples precede the first launch (and the warm-up iteration's clock ramp)
  const double dt = (double)(h[2 * (size_t)(n - 1) + 1] - h[2 * (size_t)lo + 1]), dc = (double)(h[2 * (size_t)(n - 1)] - h[2 * (size_t)lo]);
  *mhz = dt > 0 ? (float)(dc / dt * 100.0) : 0.f;
  float mn = 1e9f;
  for (int i = lo + 1; i < n; i++) {
    const double t = (double)(h[2 * (size_t)i + 1] - h[2 * (size_t)i - 1]), cy = (double)(h[2 * (size_t)i] - h[2 * (size_t)i - 2]);
    if (t > 0) mn = std::min(mn, (float)(cy / t * 100.0));
  }
  if (mhz_min) *mhz_min = mn;
  return BN254S_OK;
}

// Calibration kernel for the PMC traffic counters: a plain 8-byte-per-lane copy of `words` u64 (reads and writes
// words*8 bytes), the same access width as the NTT kernels.
__global__ void k_copy_u64(const u64* __restrict__ in, u64* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
int bn254s_bench_copy(bn254s_ctx* c, size_t words, int iters) {
  if (!c || words == 0 || iters <= 0) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  u64* a = c->words("cp.a", words);
  u64* b = c->words("cp.b", words);
  if (!a || !b) return BN254S_E_OOM;
  k_fill_pseudo<<<(unsigned)((words + 255) / 256), 256, 0, c->stream>>>(a, words, 5);
  for (int i = 0; i < iters; i++) k_copy_u64<<<(unsigned)((words + 255) / 256), 256, 0, c->stream>>>(a, b, words);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BN254S_OK;
}

// Times the leaf-hash kernel alone on `ncols` columns of 2^log_leaves synthetic rows (ms per run).
int bn254s_bench_leafhash(bn254s_ctx* c, size_t ncols, int log_leaves, int iters, float* ms) {
  if (!c || !ms || ncols == 0 || iters <= 0 || log_leaves < 4 || log_leaves > 24) return BN254S_E_INVALID_ARG;
  HIP_TRY(c, hipSetDevice(c->device));
  size_t n = (size_t)1 << log_leaves;
  u64* d = c->words("lh.data", ncols * n);
  u64* t = c->words("lh.tree", 4 * n);
  if (!d || !t) return BN254S_E_OOM;
  k_fill_pseudo<<<(unsigned)((ncols * n + 255) / 256), 256, 0, c->stream>>>(d, ncols * n, 77);
  hipEvent_t e0, e1;
  HIP_TRY(c, hipEventCreate(&e0));
  HIP_TRY(c, hipEventCreate(&e1));
  merkle_leaves(d, 1, n, (int)ncols, log_leaves, t, c->stream);
  HIP_TRY(c, hipEventRecord(e0, c->stream));
  for (int i = 0; i < iters; i++) merkle_leaves(d, 1, n, (int)ncols, log_leaves, t, c->stream);
  HIP_TRY(c, hipEventRecord(e1, c->stream));
  HIP_TRY(c, hipEventSynchronize(e1));
  float tt = 0;
  HIP_TRY(c, hipEventElapsedTime(&tt, e0, e1));
  *ms = tt / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return BN254S_OK;
}

__global__ __launch_bounds__(256) void k_poseidon_states(u64* st, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
  for (int k = 0; k < 12; k++) s[k] = st[12 * i + k];
  poseidon_permute(s);
  for (int k = 0; k < 12; k++) st[12 * i + k] = s[k];
}

int bn254s_poseidon_permute(bn254s_ctx* c, uint64_t* states, size_t n) {
  if (!c || !states) return BN254S_E_INVALID_ARG;
  if (n == 0) return BN254S_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  u64* d = c->words("pp.states", 12 * n);
  if (!d) return BN254S_E_OOM;
  HIP_TRY(c, hipMemcpyAsync(d, states, 96 * n, hipMemcpyHostToDevice, c->stream));
  k_poseidon_states<<<(unsigned)((n + 63) / 64), 64, 0, c->stream>>>(d, n);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(states, d, 96 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BN254S_OK;
}

// Debug: the hand-written gfx950 field sequences (csrc/gl_asm.h) on n operand pairs; out[n][17], see k_field_selftest.
int bn254s_selftest_field(bn254s_ctx* c, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  if (!c || !a || !b || !out) return BN254S_E_INVALID_ARG;
  if (n == 0) return BN254S_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  u64* d = c->words("fs.io", (2 + FIELD_SELFTEST_OUTS) * n);
  if (!d) return BN254S_E_OOM;
  HIP_TRY(c, hipMemcpyAsync(d, a, 8 * n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d + n, b, 8 * n, hipMemcpyHostToDevice, c->stream));
  field_selftest(d, d + n, d + 2 * n, n, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, d + 2 * n, 8 * FIELD_SELFTEST_OUTS * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BN254S_OK;
}

// Debug: BN254 Fq inversion on the device, x[n][4] canonical words -> out[n][8] = x^-1 by divsteps (the product's fq_inv) and by
// Fermat's little theorem (0 -> 0).
int bn254s_selftest_fq_inv(bn254s_ctx* c, const uint64_t* x, size_t n, uint64_t* out) {
  if (!c || !x || !out) return BN254S_E_INVALID_ARG;
  if (n == 0) return BN254S_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  u64* d = c->words("fi.io", 12 * n);
  if (!d) return BN254S_E_OOM;
  HIP_TRY(c, hipMemcpyAsync(d, x, 32 * n, hipMemcpyHostToDevice, c->stream));
  launch_fq_inv_selftest(d, d + 4 * n, n, c->stream);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, d + 4 * n, 64 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return BN254S_OK;
}

}  // extern "C"
