// Host orchestration of one STARK proof: the GPU counterpart of the reference's
//   starks::common::prover::prove                 src/starks/common/prover.rs:18-72
// followed by starky's prove_with_commitment, as called from G1StarkProofGenerator::run_once
// (src/generators/g1/stark_proof.rs:151-163).  All polynomial data stays in HBM; the host only runs the
// Fiat-Shamir transcript on caps / openings (a few KB per round trip) and assembles the proof words.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include "ctx.h"
#include "trace_g1.h"
#include "trace_g2fq.h"
#include "aux.h"
#include "merkle.h"
#include "quotient.h"
#include "fri.h"
#include "transcript.h"
#include "prover.h"

// ---- shapes -------------------------------------------------------------------------------------------------
// Stark::lookups (range-checked columns), the two looked CTL tables and the constraint counts of the three AIRs:
// G1 scalar_mul_{view,stark,ctl}.rs, G2 twins, fields/exp_{view,stark,ctl}.rs.
template <class L>
static StarkShape make_shape(bool input_has_a, int n_constraints) {
  StarkShape s;
  s.W = L::W;
  s.rc_begin = L::RC_BEGIN;
  s.rc_end = L::RC_END;
  s.table_col = L::RANGE;
  s.freq_col = L::FREQ;
  s.n_ctl = 2;
  s.n_constraints = n_constraints;
  memset(&s.ctl, 0, sizeof(s.ctl));
  int m = 0;
  for (int i = 0; i < L::PL; i++, m++) { s.ctl.col_start[0][m] = L::B + i; s.ctl.col_bits[0][m] = 1; }
  if (input_has_a)
    for (int i = 0; i < L::PL; i++, m++) { s.ctl.col_start[0][m] = L::A + i; s.ctl.col_bits[0][m] = 1; }
  for (int k = 0; k < 16; k++, m++) { s.ctl.col_start[0][m] = L::BITS + 16 * k; s.ctl.col_bits[0][m] = 16; }
  s.ctl.col_start[0][m] = L::TIMESTAMP; s.ctl.col_bits[0][m] = 1; m++;
  s.ctl.ncols[0] = m;
  s.ctl.filter_col[0] = L::FLAGS + 0;
  m = 0;
  for (int i = 0; i < L::PL; i++, m++) { s.ctl.col_start[1][m] = L::SUM + i; s.ctl.col_bits[1][m] = 1; }
  s.ctl.col_start[1][m] = L::TIMESTAMP; s.ctl.col_bits[1][m] = 1; m++;
  s.ctl.ncols[1] = m;
  s.ctl.filter_col[1] = L::FLAGS + 1;
  return s;
}
StarkShape g1_shape() { return make_shape<G1L>(true, 1111); }
StarkShape shape_for(int kind) {
  return kind == KIND_G1 ? make_shape<G1L>(true, 1111) : kind == KIND_G2 ? make_shape<G2L>(true, 1693) : make_shape<FQL>(false, 770);
}
int point_words(int kind) { return kind == KIND_G1 ? 8 : kind == KIND_G2 ? 16 : 4; }

// ---- proof object ---------------------------------------------------------------------------------------------
enum { ST_TRACE = 0, ST_TRACE_NTT, ST_TRACE_MERKLE, ST_AUX, ST_AUX_NTT, ST_AUX_MERKLE, ST_QUOTIENT, ST_QUOTIENT_COMMIT,
       ST_OPENINGS, ST_FRI, ST_TOTAL, ST_COUNT };
static const char* STAGE_NAMES[ST_COUNT] = {"trace_gen", "trace_ntt", "trace_merkle", "aux_columns", "aux_ntt", "aux_merkle",
                                            "quotient", "quotient_commit", "openings", "fri", "total"};

struct bn254s_proof {
  std::vector<u64> words, outputs;
  size_t sec_off[BN254S_SEC_COUNT + 1] = {0};  // bn254s_proof_section: start of every field of the word layout
  int degree_bits = 0;
  float stage_ms[ST_COUNT] = {0};
};

static size_t rows_for(size_t n, uint32_t min_rows_log2) {
  size_t r = std::max((size_t)1 << min_rows_log2, n * 512), p = 1;
  while (p < r) p <<= 1;
  return p;
}
std::vector<int> fri_arities(const bn254s_params& P, int degree_bits) {  // ConstantArityBits(arity, final)
  std::vector<int> a;
  int d = degree_bits;
  while (d > (int)P.final_poly_bits && d + (int)P.rate_bits - (int)P.arity_bits >= (int)P.cap_height) {
    a.push_back(P.arity_bits);
    d -= P.arity_bits;
  }
  return a;
}

__global__ __launch_bounds__(256) void k_quotient_chunks(const u64* __restrict__ ab, u64* __restrict__ out, size_t N, u64 inv2,
                                                         u64 inv2c) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int a = blockIdx.y;
  u64 A = ab[(size_t)(a * 2 + 0) * N + i], B = ab[(size_t)(a * 2 + 1) * N + i];
  out[(size_t)(a * 2 + 0) * N + i] = gl_mul(gl_add(A, B), inv2);
  out[(size_t)(a * 2 + 1) * N + i] = gl_mul(gl_sub(A, B), inv2c);
}

#define CHK(call)                                              \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) {                                    \
      err = std::string(#call) + ": " + hipGetErrorString(e_); \
      return BN254S_E_HIP;                                     \
    }                                                          \
  } while (0)

// Per-degree tables of the tall-trace NTT, built on first use.
// `half_of_split`: the tables of the H-point halves under the radix-2 level of ntt.hip (cosets g^2, g^2 w_2H instead of g, g w_2H)
static const NttTallTables* get_tall_tables(bn254s_ctx* c, unsigned log_n, bool half_of_split = false) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  const unsigned key = log_n + (half_of_split ? 100 : 0);
  auto it = c->tall.find(key);
  if (it != c->tall.end()) return it->second;
  NttTallTables* t = new NttTallTables();
  if (ntt_tall_tables_init(t, log_n, half_of_split ? gl_mul(GL_GEN, GL_GEN) : GL_GEN) != 0) {
    delete t;
    return nullptr;
  }
  c->tall[key] = t;
  return t;
}
static const NttSplitTables* get_split_tables(bn254s_ctx* c, unsigned log_n) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  auto it = c->split.find(log_n);
  if (it != c->split.end()) return it->second;
  NttSplitTables* t = new NttSplitTables();
  if (ntt_split_tables_init(t, log_n) != 0) {
    delete t;
    return nullptr;
  }
  c->split[log_n] = t;
  return t;
}

// Holds the context's big-kernel lock; on release waits until the slot's stream has drained.
struct BigSection {
  bn254s_ctx* c;
  hipStream_t st;
  int cls;
  BigSection(bn254s_ctx* c_, hipStream_t st_, int cls_) : c(c_), st(st_), cls(cls_) { c->big_lock(cls); }
  ~BigSection() {
    hipStreamSynchronize(st);
    c->big_unlock(cls);
  }
};

// Point tables are per degree; built once per context (bn254s_ctx_create thread or first prove call).
static int get_point_tables(bn254s_ctx* c, unsigned log_n, QPointTables& pt) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  std::string key = "pt." + std::to_string(log_n);
  bool fresh = !c->has(key);
  size_t M2 = (size_t)2 << log_n;
  u64* p = c->words(key, 3 * M2);
  if (!p) return BN254S_E_OOM;
  if (fresh) {
    quotient_point_tables(p, p + M2, p + 2 * M2, log_n, c->stream);
    if (hipStreamSynchronize(c->stream) != hipSuccess) return BN254S_E_HIP;
  }
  pt.x = p;
  pt.lfirst = p + M2;
  pt.llast = p + 2 * M2;
  return 0;
}

// `after_alloc` (optional) is called once the workspace of the proof is complete: the out-of-memory retry of the batch entry
// points holds WorkPool::retry_mu up to that point, so that allocation attempts never interleave.
static int prove_on_slot(bn254s_ctx* c, Slot& sl, int kind, const bn254s_params& P, const u64* scalars, const u64* x,
                         const u64* off, size_t n, bn254s_proof* pr, std::string& err,
                         const std::function<void()>* after_alloc = nullptr) {
  const StarkShape sh = shape_for(kind);
  const int PW = point_words(kind);
  const size_t N = rows_for(n, P.min_rows_log2);
  unsigned log_n = 0;
  while (((size_t)1 << log_n) < N) log_n++;
  if (log_n < 16 || log_n > 23) {
    err = "supported trace heights: 2^16 .. 2^23 rows (up to 16384 instances in one proof)";
    return BN254S_E_UNSUPPORTED;
  }
  if (P.num_challenges != 2 || P.rate_bits != 1 || P.cap_height != 4 || P.arity_bits != 4 || P.min_rows_log2 < 16 ||
      P.pow_bits == 0 || P.pow_bits > 32 || P.num_queries == 0 || P.num_queries > 256) {
    err = "unsupported StarkConfig (only standard_fast_config shapes)";
    return BN254S_E_UNSUPPORTED;
  }
  // 2^23 rows: one radix-2 level above the tall transforms (ntt.hip), and the workspace is laid out to fit one GPU: the NTT
  // stage runs in column chunks (its scratch is a chunk, not a commitment), the auxiliary coefficients overwrite their values,
  // and the trace values are released before the auxiliary LDE is allocated.  BN254S_FORCE_SPLIT=1 takes
  // the same path from 2^18 rows on (tests: word-for-word against the oracle at 2^18).
  const bool force_split = getenv("BN254S_FORCE_SPLIT") && atoi(getenv("BN254S_FORCE_SPLIT"));  // read per call (tests)
  const unsigned log_s = (log_n == 23 || (force_split && log_n >= 18)) ? 1 : 0;
  const unsigned log_m2 = log_n + 1, log_r = log_n - 16 - log_s;
  const size_t R = (size_t)1 << log_r;
  const size_t M2 = 2 * N, NH = N >> log_s;  // NH: words of one half (= N without the extra level)
  const int NSPL = 1 << log_s;              // coefficient arrays hold NSPL halves per polynomial
  // The compact workspace (NTT stage in column chunks, auxiliary commitment in place, auxiliary LDE in the memory of the dead
  // trace values) is used whenever the plain layout would not fit: 2^23 rows, and G2 from 2^22 rows on (339 GB plain, 211 GB
  // compact).  BN254S_FORCE_LOWMEM=1 selects it for any tall proof (tests).
  const bool force_lowmem = getenv("BN254S_FORCE_LOWMEM") && atoi(getenv("BN254S_FORCE_LOWMEM"));
  const double plain_bytes = 8.0 * (double)N * (4.0 * sh.W + 4.0 * sh.n_aux() + std::max(sh.W, sh.n_aux()));
  // "stream": even the compact workspace does not fit (G2 at 2^23 rows: its two LDEs alone are 296 GB).  Only the coefficients
  // stay resident; a commitment streams 16-column chunks through the NTT into a leaf hash that absorbs into resident sponge
  // states, and wherever LDE rows are needed they are recomputed from the coefficients: the quotient runs over windows of
  // consecutive rows of a coset (natural order, QArgs window mode), the FRI batch polynomial is formed on the coefficient
  // vectors (it is linear in them) and extended once, the query rows are gathered from one more pass of chunk LDEs.
  // BN254S_FORCE_STREAM=1 takes this path for any proof above 2^16 rows (tests: word for word against the oracle at 2^17 / 2^18);
  // BN254S_STREAM_WIN_LOG = log2 of the rows per window (default: at most 2^22, at least two windows per coset).
  const bool force_stream = getenv("BN254S_FORCE_STREAM") && atoi(getenv("BN254S_FORCE_STREAM"));
  const double compact_bytes = 8.0 * ((double)std::max((size_t)sh.W * N, (size_t)sh.n_aux() * 2 * N) + (double)sh.W * N * 3.0 + (double)sh.n_aux() * N);
  size_t dev_free_b = 0, dev_total_b = 0;
  (void)hipMemGetInfo(&dev_free_b, &dev_total_b);
  const bool stream = log_n > 16 && (force_stream || (dev_total_b > 0 && compact_bytes + 20e9 > (double)dev_total_b));
  const bool lowmem = stream || log_s || (log_n > 16 && (force_lowmem || plain_bytes > 200e9));
  unsigned log_bk = std::min(22u, log_n - 1);
  if (const char* e = getenv("BN254S_STREAM_WIN_LOG")) log_bk = std::min(log_n, std::max(8u, (unsigned)atoi(e)));
  const size_t BK = (size_t)1 << log_bk, WROWS = BK + 1;  // rows per quotient window (+ the next row of the last one)
  const int W = sh.W, A = sh.n_aux(), NQ = 4, CAPW = 64;
  const int K = sh.n_total_constraints();
  hipStream_t st = sl.st;
  BufPool& mem = sl.mem;
  u64* d_tmp_fwd = nullptr;  // = d_tmp once the workspace is set up (used by the NTT dispatch lambdas)
  QPointTables pt;
  if (int rc = get_point_tables(c, log_n, pt)) {
    err = "point tables";
    return rc;
  }
  const NttTallTables* TT = nullptr;
  const NttSplitTables* SP = nullptr;
  if (log_r) {
    TT = get_tall_tables(c, log_n - log_s, log_s != 0);
    if (!TT) {
      err = "tall NTT tables";
      return BN254S_E_OOM;
    }
  }
  if (log_s) {
    SP = get_split_tables(c, log_n);
    if (!SP) {
      err = "split NTT tables";
      return BN254S_E_OOM;
    }
  }
  const int CH = 16;  // columns per chunk of the split commitment
  // iNTT / LDE / coset iNTT dispatch: 2^16-row traces use the four-step kernels directly, taller ones add the outer pass
  // from_values of a whole commitment: the fused four-launch form for 2^16 rows, iNTT then LDE otherwise
  auto do_commit_ntt = [&](const u64* vals, u64* coef, u64* lde, int nc) {
    if (log_s) {
      // per chunk: values -> [Pe | Po] (radix-2 level), the tall commitment of the 2 CH half columns, halves' LDEs -> LDE
      u64* eo = d_tmp_fwd;                          // [CH][2][N]
      u64* ntmp = d_tmp_fwd + (size_t)CH * 2 * N;   // [2 CH][NH]
      for (int c0 = 0; c0 < nc; c0 += CH) {
        const int cn = std::min(CH, nc - c0);
        ntt_split_inverse(SP, vals + (size_t)c0 * N, coef + (size_t)c0 * N, cn, st);
        ntt_inverse_lde_tall(&c->ntt, TT, coef + (size_t)c0 * N, coef + (size_t)c0 * N, eo, ntmp, 2 * cn, st);
        ntt_split_forward(SP, eo, lde + (size_t)c0 * M2, M2, cn, st);
      }
    } else if (log_r && lowmem) {
      for (int c0 = 0; c0 < nc; c0 += CH) {
        const int cn = std::min(CH, nc - c0);
        ntt_inverse_lde_tall(&c->ntt, TT, vals + (size_t)c0 * N, coef + (size_t)c0 * N, lde + (size_t)c0 * M2, d_tmp_fwd, cn, st);
      }
    } else if (log_r) {
      ntt_inverse_lde_tall(&c->ntt, TT, vals, coef, lde, d_tmp_fwd, nc, st);
    } else {
      ntt_inverse_lde(&c->ntt, vals, coef, lde, d_tmp_fwd, d_tmp_fwd + (size_t)nc * N, nc, st);
    }
  };
  auto do_lde = [&](const u64* coef, u64* lde, int nc) {
    if (log_s) {
      u64* eo = d_tmp_fwd;
      u64* ntmp = d_tmp_fwd + (size_t)CH * 2 * N;
      for (int c0 = 0; c0 < nc; c0 += CH) {
        const int cn = std::min(CH, nc - c0);
        ntt_lde_tall(&c->ntt, TT, coef + (size_t)c0 * N, eo, ntmp, 2 * cn, st);
        ntt_split_forward(SP, eo, lde + (size_t)c0 * M2, M2, cn, st);
      }
    } else if (log_r) ntt_lde_tall(&c->ntt, TT, coef, lde, d_tmp_fwd, nc, st);
    else ntt_lde(&c->ntt, coef, lde, d_tmp_fwd, nc, st);
  };
  // stream mode: the LDE of ONE column chunk (coefficients at `coef`, cn <= CH columns) into d_ldechunk
  u64* d_ldechunk_fwd = nullptr;  // (= d_ldechunk once the workspace is set up)
  // (coset_mask: 1 / 2 = only coset 0 / 1 of the output, 3 = both; under the split level coset h of the N-point domain comes from
  // coset h of both half transforms)
  auto lde_chunk = [&](const u64* coef, int cn, int coset_mask = 3) {
    if (log_s) {
      u64* eo = d_tmp_fwd;
      u64* ntmp = d_tmp_fwd + (size_t)CH * 2 * N;
      ntt_lde_tall(&c->ntt, TT, coef, eo, ntmp, 2 * cn, st, coset_mask);
      ntt_split_forward(SP, eo, d_ldechunk_fwd, M2, cn, st, coset_mask);
    } else {
      ntt_lde_tall(&c->ntt, TT, coef, d_ldechunk_fwd, d_tmp_fwd, cn, st, coset_mask);
    }
  };
  // stream mode: from_values of a commitment, chunk by chunk: coefficients stay, the chunk's LDE is absorbed by the leaf hash
  u64* d_sponge_fwd = nullptr;
  const unsigned log_m2_fwd = log_n + 1;
  auto commit_stream = [&](const u64* vals, u64* coef, int nc, u64* tree) {
    for (int c0 = 0; c0 < nc; c0 += CH) {
      const int cn = std::min(CH, nc - c0);
      if (log_s) {
        u64* eo = d_tmp_fwd;
        u64* ntmp = d_tmp_fwd + (size_t)CH * 2 * N;
        ntt_split_inverse(SP, vals + (size_t)c0 * N, coef + (size_t)c0 * N, cn, st);
        ntt_inverse_lde_tall(&c->ntt, TT, coef + (size_t)c0 * N, coef + (size_t)c0 * N, eo, ntmp, 2 * cn, st);
        ntt_split_forward(SP, eo, d_ldechunk_fwd, M2, cn, st);
      } else {
        ntt_inverse_lde_tall(&c->ntt, TT, vals + (size_t)c0 * N, coef + (size_t)c0 * N, d_ldechunk_fwd, d_tmp_fwd, cn, st);
      }
      merkle_absorb(d_ldechunk_fwd, M2, cn, (int)log_m2_fwd, d_sponge_fwd, c0 == 0, c0 + cn == nc ? tree : nullptr, st);
    }
  };
  // leaf hash of a resident commitment: one GPU-filling section, or (BN254S_HASH_SPLIT, 2^16-row proofs) several shorter ones
  auto hash_sections = [&](const u64* lde, int width, u64* tree, int stage) {
    const int S = log_n == 16 ? c->hash_split : 1;
    if (S <= 1) {
      BigSection big(c, st, BIG_HASH);
      hipEventRecord(sl.events[2 * stage], st);  // (the stage's clock starts once the section is admitted)
      merkle_leaves(lde, 1, M2, width, log_m2_fwd, tree, st);
      return;
    }
    const size_t cnt = M2 / S;
    for (int k = 0; k < S; k++) {
      BigSection big(c, st, BIG_HASH_PART);
      if (k == 0) hipEventRecord(sl.events[2 * stage], st);
      merkle_leaves_range(lde, 1, M2, width, (size_t)k * cnt, cnt, tree, st);
    }
  };
  // proofs running beside this one right now: the small Merkle levels use their throughput kernels then (merkle.h; same digests)
  auto mmode = [&]() { return c->workers.in_flight.load(std::memory_order_relaxed) >= 4 ? MERKLE_THROUGHPUT : MERKLE_LATENCY; };
  const std::vector<int> arities = fri_arities(P, log_n);
  const int L = (int)arities.size();
  if (L > FRI_MAX_LAYERS) return BN254S_E_UNSUPPORTED;

  // ---- workspace ------------------------------------------------------------------------------------------
  const size_t in_words = n * (4 + 2 * (size_t)PW);
  if (lowmem) {  // buffers of the plain layout left in the slot by an earlier, smaller proof would not fit beside this one
    mem.drop("acoef");
    mem.drop("alde");
  }
  u64* d_in = mem.words("in", in_words + 16);
  // compact workspace: the trace values are dead once the auxiliary values exist, the auxiliary LDE takes their place
  u64* d_tvals = mem.words("tvals", lowmem && !stream ? std::max((size_t)W * N, (size_t)A * M2) : (size_t)W * N);
  u64* d_tcoef = mem.words("tcoef", (size_t)W * N);
  u64* d_hist = mem.words("hist", 65536 / 2);
  // [tmp | y0 | y1] for the fused commitment; one tmp for a tall one; [E,O LDEs | tmp] of one column chunk under the split level
  u64* d_tmp = mem.words("tmp", log_s ? (size_t)3 * CH * N : lowmem ? (size_t)CH * N : (size_t)(log_r ? 1 : 3) * std::max(W, A) * N);
  d_tmp_fwd = d_tmp;
  mem.drop("win");  // (allocated in the middle of a streaming proof, in the memory of its dead trace values)
  if (stream) {  // no resident LDE: whatever an earlier proof left in the slot under these names goes first
    mem.drop("tlde");
    mem.drop("alde");
    mem.drop("acoef");
  } else {
    mem.drop("ldechunk");
    mem.drop("sponge");
    mem.drop("comb");
  }
  u64* d_ldechunk = stream ? mem.words("ldechunk", (size_t)CH * M2) : nullptr;  // the LDE of one column chunk
  u64* d_sponge = stream ? mem.words("sponge", (size_t)12 * M2) : nullptr;      // sponge states of the streaming leaf hash
  u64* d_comb = stream ? mem.words("comb", (size_t)6 * N + (size_t)6 * M2) : nullptr;  // FRI batch polynomial: coefficients | LDE
  u64* d_tlde = stream ? d_ldechunk : mem.words("tlde", (size_t)W * M2);
  const size_t tree_words = merkle_tree_digests(log_m2, P.cap_height) * 4;
  u64* d_trees = mem.words("trees", 3 * tree_words);
  u64* d_avals = mem.words("avals", (size_t)A * N);
  u64* d_acoef = lowmem ? d_avals : mem.words("acoef", (size_t)A * N);         // compact workspace: the commitment runs in place
  u64* d_alde = stream ? d_ldechunk : lowmem ? d_tvals : mem.words("alde", (size_t)A * M2);
  const size_t trace_scr = kind == KIND_G1 ? g1_trace_scratch_words(n) : kind == KIND_G2 ? g2_trace_scratch_words(n) : fq_trace_scratch_words(n);
  u64* d_scr = mem.words("scratch", std::max(trace_scr, aux_scratch_words(sh, N)));
  u64* d_q = mem.words("quot", (size_t)(3 * NQ) * N + (size_t)NQ * M2 + (size_t)QUOTIENT_MAX_PARTS * 2 * (stream ? WROWS : M2));  // qv, ab, qcoef, qlde, partials
  u64* d_tabs = mem.words("tabs", QUOTIENT_W_WORDS(K) + QUOTIENT_MZT_WORDS + 4 * (size_t)(W + A + NQ));  // W, mzt, apow (8 u32 per power)
  u64* d_open = mem.words("open", std::max((size_t)(W + A + NQ) * NSPL * R * 5 + FRI_OPENING_TABLE_WORDS, n * (size_t)PW));
  // FRI layer values (extension, 2 words) and trees
  size_t fri_words = 0, fri_tree_words = 0;
  {
    size_t m = M2;
    for (int l = 0; l <= L; l++) {
      fri_words += 2 * m;
      if (l < L) fri_tree_words += merkle_tree_digests((int)log_m2 - 4 * (l + 1), P.cap_height) * 4;
      m >>= 4;
    }
  }
  u64* d_fri = mem.words("fri", fri_words);
  u64* d_fritrees = mem.words("fritrees", fri_tree_words);
  // per-query proof words
  size_t wpq = (size_t)(W + A + NQ) + 3 * 4 * (log_m2 - P.cap_height);
  {
    int lg = log_m2;
    for (int l = 0; l < L; l++) {
      lg -= arities[l];
      wpq += 32 + 4 * (lg - P.cap_height);
    }
  }
  u64* d_qout = mem.words("qout", wpq * P.num_queries + 64);
  {  // pinned staging for the two larger device -> host copies (opening partials, query rounds), part of the workspace
    const size_t need = std::max((size_t)(W + A + NQ) * NSPL * R * 5, wpq * P.num_queries) * 8;
    if (sl.pinned_bytes < need) {
      if (sl.pinned) hipHostFree(sl.pinned);
      sl.pinned = nullptr;
      sl.pinned_bytes = 0;
      CHK(hipHostMalloc(&sl.pinned, need));
      sl.pinned_bytes = need;
    }
  }
  if (!d_hist || !d_in || !d_tvals || !d_tcoef || !d_tmp || !d_tlde || !d_trees || !d_avals || !d_acoef || !d_alde || !d_scr || !d_q ||
      !d_tabs || !d_open || !d_fri || !d_fritrees || !d_qout || (stream && (!d_ldechunk || !d_sponge || !d_comb))) {
    err = mem.err;
    mem.release();  // a workspace that could not be completed is given back: the context stays usable for smaller proofs
    return BN254S_E_OOM;
  }
  {
    // The HIP runtime allocates device memory of its own while kernels run (scratch for the kernels that spill, per hardware
    // queue; signals, kernel arguments): when that fails the queue aborts the whole process (HSA_STATUS_ERROR_OUT_OF_RESOURCES).
    // A workspace that leaves less than the reserve free is therefore given back and reported as out of memory, which the batch
    // entry points turn into "wait for a running proof to finish".  BN254S_MEM_RESERVE_MB overrides (default 6144).
    static const size_t reserve = (size_t)(getenv("BN254S_MEM_RESERVE_MB") ? atol(getenv("BN254S_MEM_RESERVE_MB")) : 6144) << 20;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < reserve) {
      err = "device memory: " + std::to_string(free_b >> 20) + " MB would be left free, below the reserve of " +
            std::to_string(reserve >> 20) + " MB kept for the runtime's own allocations";
      mem.release();
      return BN254S_E_OOM;
    }
  }
  if (after_alloc) (*after_alloc)();
  d_ldechunk_fwd = d_ldechunk;
  d_sponge_fwd = d_sponge;
  int* d_err = (int*)(d_in + in_words);
  unsigned long long* d_pow = (unsigned long long*)(d_in + in_words + 2);
  u32* d_qidx = (u32*)(d_qout + wpq * P.num_queries);
  u64* d_ttree = d_trees;
  u64* d_atree = d_trees + tree_words;
  u64* d_qtree = d_trees + 2 * tree_words;
  u64* d_qv = d_q;                       // [2][2][N]
  u64* d_ab = d_q + (size_t)NQ * N;      // [2][2][N]
  u64* d_qcoef = d_q + (size_t)2 * NQ * N;  // [4][N]
  u64* d_qlde = d_q + (size_t)3 * NQ * N;   // [4][2N]
  u64* d_qpart = d_qlde + (size_t)NQ * M2;  // [parts][2][2N]
  u64* d_W = d_tabs;
  u64* d_mzt = d_tabs + QUOTIENT_W_WORDS(K);
  u64* d_apow = d_mzt + QUOTIENT_MZT_WORDS;
  const size_t cap_off = 4 * merkle_level_offset(log_m2, log_m2 - P.cap_height);

  if (sl.events.empty()) {
    sl.events.resize(2 * ST_COUNT);
    for (auto& e : sl.events) CHK(hipEventCreate(&e));
  }
  hipEvent_t* ev = sl.events.data();
  // one begin/end event pair per stage; a stage's time never includes waiting for the big-kernel lock
  auto sb = [&](int stage) { hipEventRecord(ev[2 * stage], st); };
  auto se = [&](int stage) { hipEventRecord(ev[2 * stage + 1], st); };
  auto cleanup_events = [&]() {};  // the events belong to the slot

  // ---- trace generation (scalar_mul_stark.rs:55-69) ---------------------------------------------------------
  sb(ST_TOTAL);
  sb(ST_TRACE);
  CHK(hipMemsetAsync(d_err, 0, 16, st));
  u64* d_sc = d_in;
  u64* d_x = d_in + 4 * n;
  u64* d_off = d_x + (size_t)PW * n;
  CHK(hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, st));
  CHK(hipMemcpyAsync(d_x, x, n * (size_t)PW * 8, hipMemcpyHostToDevice, st));
  if (kind != KIND_FQ) CHK(hipMemcpyAsync(d_off, off, n * (size_t)PW * 8, hipMemcpyHostToDevice, st));
  u64* d_outs = d_open;  // n*PW words fit (reused later)
  int trc = kind == KIND_G1   ? g1_generate_trace_device(d_sc, d_x, d_off, n, d_tvals, N, d_scr, d_outs, d_err, st, false)
            : kind == KIND_G2 ? g2_generate_trace_device(d_sc, d_x, d_off, n, d_tvals, N, d_scr, d_outs, d_err, st, false)
                              : fq_generate_trace_device(d_sc, d_x, n, d_tvals, N, d_scr, d_outs, d_err, st, false);
  if (trc) {
    err = "trace generation launch failed";
    return BN254S_E_HIP;
  }
  {
    // generate_range_checks: the histogram keeps 128 KB of LDS per workgroup on every CU, which would lock another proof's
    // NTT tiles (35 KB each) out of the CUs it shares: it takes its turn like the other wide kernels
    BigSection big(c, st, BIG_EXCL);
    launch_range_columns(d_tvals, N, sh.rc_begin, sh.rc_end, sh.freq_col, sh.table_col, (u32*)d_hist, d_err, st);
  }
  pr->outputs.resize(n * (size_t)PW);
  int h_err = 0;
  CHK(hipMemcpyAsync(pr->outputs.data(), d_outs, n * (size_t)PW * 8, hipMemcpyDeviceToHost, st));
  CHK(hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
  se(ST_TRACE);

  // ---- trace commitment (prover.rs:31-38) -----------------------------------------------------------------
  if (stream) {  // (the NTT and the leaf hash alternate chunk by chunk: one section, timed as the NTT stage)
    BigSection big(c, st, BIG_NTT);
    sb(ST_TRACE_NTT);
    commit_stream(d_tvals, d_tcoef, W, d_ttree);
    se(ST_TRACE_NTT);
    sb(ST_TRACE_MERKLE);
  } else {
    {
      BigSection big(c, st, BIG_NTT);
      sb(ST_TRACE_NTT);
      do_commit_ntt(d_tvals, d_tcoef, d_tlde, W);
      se(ST_TRACE_NTT);
    }
    hash_sections(d_tlde, W, d_ttree, ST_TRACE_MERKLE);
  }
  merkle_upper(log_m2, P.cap_height, d_ttree, st, mmode());
  u64 caps[3][64];
  CHK(hipMemcpyAsync(caps[0], d_ttree + cap_off, CAPW * 8, hipMemcpyDeviceToHost, st));
  se(ST_TRACE_MERKLE);
  CHK(hipStreamSynchronize(st));
  if (h_err) {
    err = "trace generation reported device error " + std::to_string(h_err);
    cleanup_events();
    return h_err;
  }

  // ---- transcript: trace cap, CTL challenges, compact (prover.rs:40-54) --------------------------------
  Challenger ch;
  ch.observe_n(caps[0], CAPW);
  u64 betas[2], gammas[2];
  for (int i = 0; i < 2; i++) {
    betas[i] = ch.challenge();
    gammas[i] = ch.challenge();
  }
  u64 init_state[12];
  ch.compact(init_state);

  // ---- auxiliary columns + commitment ---------------------------------------------------------------------
  {
    BigSection big(c, st, BIG_EXCL);
    sb(ST_AUX);
    aux_build(sh, d_tvals, N, betas, gammas, d_avals, d_scr, d_err, st);
    se(ST_AUX);
  }
  u64 *d_tw = nullptr, *d_aw = nullptr, *d_wpt = nullptr, *d_qrows = nullptr;
  if (stream) {
    // the trace values are dead: their memory becomes the quotient windows (all columns x WROWS rows), the window's point tables
    // and the leaf rows of the queries
    CHK(hipStreamSynchronize(st));
    mem.drop("tvals");
    d_tvals = nullptr;
    u64* win = mem.words("win", (size_t)(W + A) * WROWS + 3 * WROWS + (size_t)(W + A) * P.num_queries);
    if (!win) {
      err = mem.err;
      mem.release();
      return BN254S_E_OOM;
    }
    d_tw = win;
    d_aw = win + (size_t)W * WROWS;
    d_wpt = d_aw + (size_t)A * WROWS;
    d_qrows = d_wpt + 3 * WROWS;
    BigSection big(c, st, BIG_NTT);
    sb(ST_AUX_NTT);
    commit_stream(d_avals, d_acoef, A, d_atree);
    se(ST_AUX_NTT);
    sb(ST_AUX_MERKLE);
  } else {
    {
      BigSection big(c, st, BIG_NTT);
      sb(ST_AUX_NTT);
      do_commit_ntt(d_avals, d_acoef, d_alde, A);
      se(ST_AUX_NTT);
    }
    hash_sections(d_alde, A, d_atree, ST_AUX_MERKLE);
  }
  merkle_upper(log_m2, P.cap_height, d_atree, st, mmode());
  CHK(hipMemcpyAsync(caps[1], d_atree + cap_off, CAPW * 8, hipMemcpyDeviceToHost, st));
  se(ST_AUX_MERKLE);
  CHK(hipStreamSynchronize(st));
  ch.observe_n(caps[1], CAPW);
  u64 alphas[2] = {ch.challenge(), 0};
  alphas[1] = ch.challenge();

  // ---- quotient ---------------------------------------------------------------------------------------------
  {
    std::vector<u64> hW, hmzt;
    const int* mz_e0 = nullptr;
    int nblk = kind == KIND_G1 ? g1_quotient_mz_blocks(&mz_e0) : kind == KIND_G2 ? g2_quotient_mz_blocks(&mz_e0) : fq_quotient_mz_blocks(&mz_e0);
    quotient_host_tables(K, alphas, mz_e0, nblk, hW, hmzt);
    CHK(hipMemcpyAsync(d_W, hW.data(), hW.size() * 8, hipMemcpyHostToDevice, st));
    CHK(hipMemcpyAsync(d_mzt, hmzt.data(), hmzt.size() * 8, hipMemcpyHostToDevice, st));
    CHK(hipStreamSynchronize(st));  // host vectors go out of scope
  }
  if (stream) {
    // per coset and window of BK consecutive rows: the LDE of every column chunk is recomputed and the window's rows (plus the next
    // one) are taken out in natural order; the quotient kernels then run on the window (QArgs window mode)
    BigSection big(c, st, BIG_NTT);
    sb(ST_QUOTIENT);
    QPointTables wpt;
    wpt.x = d_wpt;
    wpt.lfirst = d_wpt + WROWS;
    wpt.llast = d_wpt + 2 * WROWS;
    for (int h = 0; h < 2; h++)
      for (size_t k0 = 0; k0 < N; k0 += BK) {
        for (int c0 = 0; c0 < W; c0 += CH) {
          const int cn = std::min(CH, W - c0);
          lde_chunk(d_tcoef + (size_t)c0 * N, cn, 1 << h);
          fri_extract_window(d_ldechunk, M2, log_n, h, k0, WROWS, cn, d_tw + (size_t)c0 * WROWS, st);
        }
        for (int c0 = 0; c0 < A; c0 += CH) {
          const int cn = std::min(CH, A - c0);
          lde_chunk(d_acoef + (size_t)c0 * N, cn, 1 << h);
          fri_extract_window(d_ldechunk, M2, log_n, h, k0, WROWS, cn, d_aw + (size_t)c0 * WROWS, st);
        }
        quotient_point_tables_window(d_wpt, d_wpt + WROWS, d_wpt + 2 * WROWS, log_n, h, k0, WROWS, st);
        QArgs QA;
        quotient_fill_args(QA, sh, nullptr, nullptr, d_W, d_mzt, pt, betas, gammas, log_n, d_qv, d_qpart);
        quotient_window_args(QA, d_tw, d_aw, wpt, d_qpart, WROWS, BK, k0, h);
        if (kind == KIND_G1) g1_quotient_launch(QA, sh, st);
        else if (kind == KIND_G2) g2_quotient_launch(QA, sh, st);
        else fq_quotient_launch(QA, sh, st);
      }
  } else {
    BigSection big(c, st, BIG_EXCL);
    sb(ST_QUOTIENT);
    QArgs QA;
    quotient_fill_args(QA, sh, d_tlde, d_alde, d_W, d_mzt, pt, betas, gammas, log_n, d_qv, d_qpart);
    if (kind == KIND_G1) g1_quotient_launch(QA, sh, st);
    else if (kind == KIND_G2) g2_quotient_launch(QA, sh, st);
    else fq_quotient_launch(QA, sh, st);
  }
  for (int a = 0; a < 2; a++)
    for (int h = 0; h < 2; h++)
      if (log_s) ntt_coset_inverse_split(&c->ntt, TT, SP, h, d_qv + (size_t)(a * 2 + h) * N, d_ab + (size_t)(a * 2 + h) * N, d_tmp, 1, st);
      else if (log_r) ntt_coset_inverse_tall(&c->ntt, TT, h, d_qv + (size_t)(a * 2 + h) * N, d_ab + (size_t)(a * 2 + h) * N, d_tmp, 1, st);
      else ntt_coset_inverse(&c->ntt, h, d_qv + (size_t)(a * 2 + h) * N, d_ab + (size_t)(a * 2 + h) * N, d_tmp, 1, st);
  {
    u64 gn = gl_pow(GL_GEN, N);
    u64 inv2 = gl_inv(2), inv2c = gl_inv(gl_mul(2, gn));
    dim3 g((unsigned)((N + 255) / 256), 2);
    k_quotient_chunks<<<g, 256, 0, st>>>(d_ab, d_qcoef, N, inv2, inv2c);
  }
  se(ST_QUOTIENT);
  sb(ST_QUOTIENT_COMMIT);
  do_lde(d_qcoef, d_qlde, NQ);
  merkle_build(d_qlde, 1, M2, NQ, log_m2, P.cap_height, d_qtree, st, mmode());
  CHK(hipMemcpyAsync(caps[2], d_qtree + cap_off, CAPW * 8, hipMemcpyDeviceToHost, st));
  se(ST_QUOTIENT_COMMIT);
  CHK(hipStreamSynchronize(st));
  ch.observe_n(caps[2], CAPW);
  gl2 zeta = ch.challenge_ext();
  {
    gl2 zp = zeta;
    for (unsigned i = 0; i < log_n; i++) zp = gl2_mul(zp, zp);
    if (gl2_eq(zp, gl2_make(1, 0))) {
      err = "Opening point is in the subgroup.";
      cleanup_events();
      return BN254S_E_TRANSCRIPT;
    }
  }
  const u64 g = gl_root_of_unity(log_n);
  gl2 zeta_next = gl2_mul_base(zeta, g);

  // ---- openings (StarkOpeningSet::new) ---------------------------------------------------------------------
  {
    BigSection big(c, st, BIG_EXCL);
    sb(ST_OPENINGS);
    // split level: P(z) = Pe(z^2) + z Po(z^2), the halves are polynomials of NH coefficients opened at z^2 and (g z)^2
    u64* d_otab = d_open + (size_t)(W + A + NQ) * NSPL * R * 5;
    fri_opening_tables(log_r, log_s ? gl2_mul(zeta, zeta) : zeta, log_s ? gl2_mul(zeta_next, zeta_next) : zeta_next, d_otab, st);
    fri_openings(d_tcoef, NH, log_r, W * NSPL, d_otab, d_open, st);
    fri_openings(d_acoef, NH, log_r, A * NSPL, d_otab, d_open + (size_t)W * NSPL * R * 5, st);
    fri_openings(d_qcoef, NH, log_r, NQ * NSPL, d_otab, d_open + (size_t)(W + A) * NSPL * R * 5, st);
  }
  // device -> host copies land in the slot's pinned staging buffer (pageable destinations go through a runtime-internal
  // staging allocation that costs ~8 ms the first time and an extra copy every time)
  const size_t n_part = (size_t)(W + A + NQ) * NSPL * R * 5, n_q = wpq * P.num_queries;
  u64* h_part = (u64*)sl.pinned;
  std::vector<u64> h_open((size_t)(W + A + NQ) * 5);
  CHK(hipMemcpyAsync(h_part, d_open, n_part * 8, hipMemcpyDeviceToHost, st));
  se(ST_OPENINGS);
  CHK(hipStreamSynchronize(st));
  {  // P(z) = sum_k1 z^k1 S_k1(z^R) (transposed coefficient layout); P(1) = sum of the block sums
    std::vector<gl2> zp0(R), zp1(R);
    gl2 a = gl2_make(1, 0), b = gl2_make(1, 0);
    const gl2 y0 = log_s ? gl2_mul(zeta, zeta) : zeta, y1 = log_s ? gl2_mul(zeta_next, zeta_next) : zeta_next;
    for (size_t k1 = 0; k1 < R; k1++) {
      zp0[k1] = a;
      zp1[k1] = b;
      a = gl2_mul(a, y0);
      b = gl2_mul(b, y1);
    }
    for (int p = 0; p < W + A + NQ; p++) {
      gl2 v0 = gl2_make(0, 0), v1 = gl2_make(0, 0);
      u64 s1 = 0;
      for (int half = 0; half < NSPL; half++) {
        gl2 e0 = gl2_make(0, 0), e1 = gl2_make(0, 0);
        for (size_t k1 = 0; k1 < R; k1++) {
          const u64* q = &h_part[(((size_t)p * NSPL + half) * R + k1) * 5];
          e0 = gl2_add(e0, gl2_mul(zp0[k1], gl2_make(q[0], q[1])));
          e1 = gl2_add(e1, gl2_mul(zp1[k1], gl2_make(q[2], q[3])));
          s1 = gl_add(s1, q[4]);
        }
        v0 = gl2_add(v0, half ? gl2_mul(zeta, e0) : e0);
        v1 = gl2_add(v1, half ? gl2_mul(zeta_next, e1) : e1);
      }
      u64* o5 = &h_open[(size_t)p * 5];
      o5[0] = v0.c0; o5[1] = v0.c1; o5[2] = v1.c0; o5[3] = v1.c1; o5[4] = s1;
    }
  }
  const int num_lookup = sh.n_lookup_cols(), n_ctlz = 2 * sh.n_ctl;
  auto op = [&](int p, int k) { return h_open[(size_t)p * 5 + k]; };
  // to_fri_openings order: (local | aux | quotient) at zeta, (next | aux_next) at g*zeta, ctl_zs_first
  for (int p = 0; p < W + A + NQ; p++) {
    ch.observe(op(p, 0));
    ch.observe(op(p, 1));
  }
  for (int p = 0; p < W + A; p++) {
    ch.observe(op(p, 2));
    ch.observe(op(p, 3));
  }
  for (int i = 0; i < n_ctlz; i++) {
    ch.observe(op(W + num_lookup + i, 4));
    ch.observe(0);
  }

  // ---- FRI (prove_openings) ------------------------------------------------------------------------------------
  gl2 fri_alpha = ch.challenge_ext();
  {
    std::vector<u32> apow(8 * (size_t)(W + A + NQ), 0);  // alpha^p cut in 22-bit limbs (quotient_common.h W3), 8 u32 per power
    gl2 ap = gl2_make(1, 0), r0 = gl2_make(0, 0), r1 = r0, r2 = r0;
    for (int p = 0; p < W + A + NQ; p++) {
      const W3 s0 = w3_split(ap.c0), s1 = w3_split(ap.c1);
      u32* e = &apow[8 * (size_t)p];
      e[0] = s0.w0; e[1] = s0.w1; e[2] = s0.w2;
      e[4] = s1.w0; e[5] = s1.w1; e[6] = s1.w2;
      r0 = gl2_add(r0, gl2_mul(ap, gl2_make(op(p, 0), op(p, 1))));
      if (p < W + A) r1 = gl2_add(r1, gl2_mul(ap, gl2_make(op(p, 2), op(p, 3))));
      if (p < n_ctlz) r2 = gl2_add(r2, gl2_mul_base(ap, op(W + num_lookup + p, 4)));
      ap = gl2_mul(ap, fri_alpha);
    }
    CHK(hipMemcpyAsync(d_apow, apow.data(), apow.size() * 4, hipMemcpyHostToDevice, st));
    CHK(hipStreamSynchronize(st));
    BigSection big(c, st, BIG_EXCL);
    sb(ST_FRI);
    if (stream) {  // the batch polynomial on the coefficient vectors, one LDE of its six columns, then the point-wise part
      fri_combine_coeffs(sh, d_tcoef, d_acoef, d_qcoef, d_apow, N, d_comb, st);
      do_lde(d_comb, d_comb + (size_t)6 * N, 6);
      fri_combine_final(sh, d_comb + (size_t)6 * N, pt.x, zeta, zeta_next, r0, r1, r2, fri_alpha, M2, d_fri, st);
    } else {
      fri_combine(sh, d_tlde, d_alde, d_qlde, d_apow, pt.x, zeta, zeta_next, r0, r1, r2, fri_alpha, M2, d_fri, st);
    }
  }
  std::vector<std::vector<u64>> layer_caps(L, std::vector<u64>(CAPW));
  std::vector<const u64*> layer_vals(L), layer_trees(L);
  u64* vals = d_fri;
  u64* ftree = d_fritrees;
  u64 shift = GL_GEN;
  unsigned lm = log_m2;
  const u64 inv16 = gl_inv(16);
  for (int l = 0; l < L; l++) {
    const int lg = (int)lm - 4;
    merkle_build(vals, 32, 1, 32, lg, P.cap_height, ftree, st, mmode());
    CHK(hipMemcpyAsync(layer_caps[l].data(), ftree + 4 * merkle_level_offset(lg, lg - P.cap_height), CAPW * 8,
                       hipMemcpyDeviceToHost, st));
    CHK(hipStreamSynchronize(st));
    ch.observe_n(layer_caps[l].data(), CAPW);
    gl2 beta = ch.challenge_ext();
    u64* next = vals + 2 * ((size_t)1 << lm);
    fri_fold(vals, next, lm, shift, beta, inv16, st);
    layer_vals[l] = vals;
    layer_trees[l] = ftree;
    ftree += merkle_tree_digests(lg, P.cap_height) * 4;
    vals = next;
    shift = gl_pow(shift, 16);
    lm -= 4;
  }
  // final polynomial: coset iDFT of the last (tiny) layer on the host, upper half must vanish
  const size_t Mf = (size_t)1 << lm;
  std::vector<u64> h_final(2 * Mf);
  CHK(hipMemcpyAsync(h_final.data(), vals, h_final.size() * 8, hipMemcpyDeviceToHost, st));
  CHK(hipStreamSynchronize(st));
  std::vector<gl2> final_poly(Mf >> P.rate_bits);
  {
    const u64 w = gl_root_of_unity(lm), winv = gl_inv(w), sinv = gl_inv(shift), minv = gl_inv((u64)Mf);
    for (size_t k = 0; k < Mf; k++) {
      gl2 acc = gl2_make(0, 0);
      u64 wk = gl_pow(winv, k), t = 1;
      for (size_t nn = 0; nn < Mf; nn++) {  // value at natural index nn sits at position bitrev(nn)
        size_t pos = bitrev32((u32)nn, lm);
        acc = gl2_add(acc, gl2_mul_base(gl2_make(h_final[2 * pos], h_final[2 * pos + 1]), t));
        t = gl_mul(t, wk);
      }
      acc = gl2_mul_base(acc, gl_mul(minv, gl_pow(sinv, k)));
      if (k < final_poly.size()) final_poly[k] = acc;
      else if (acc.c0 | acc.c1) {
        err = "FRI final polynomial has non-zero high coefficients";
        cleanup_events();
        return BN254S_E_INTERNAL;
      }
    }
  }
  for (auto& cf : final_poly) {
    ch.observe(cf.c0);
    ch.observe(cf.c1);
  }
  // proof of work (fri_proof_of_work): smallest witness with >= pow_bits leading zeros in the response
  u64 pow_witness = 0;
  {
    u64 stt[12];
    memcpy(stt, ch.state, sizeof(stt));
    for (int i = 0; i < ch.in_len; i++) stt[i] = ch.in_buf[i];
    const size_t WIN = (size_t)1 << (P.pow_bits + 1);  // 86 % of the searches end in the first window
    unsigned long long found = ~0ULL;
    for (u64 base = 0; found == ~0ULL; base += WIN) {
      CHK(hipMemsetAsync(d_pow, 0xFF, 8, st));
      fri_pow_launch(stt, ch.in_len, base, P.pow_bits, WIN, d_pow, st);
      CHK(hipMemcpyAsync(&found, d_pow, 8, hipMemcpyDeviceToHost, st));
      CHK(hipStreamSynchronize(st));
      if (base > ((u64)1 << 40)) {
        err = "proof of work not found";
        cleanup_events();
        return BN254S_E_INTERNAL;
      }
    }
    pow_witness = found;
    ch.observe(pow_witness);
    u64 resp = ch.challenge();
    if (resp >> (64 - P.pow_bits)) {
      err = "proof of work self-check failed";
      cleanup_events();
      return BN254S_E_INTERNAL;
    }
  }
  // query rounds
  std::vector<u32> qidx(P.num_queries);
  for (auto& q : qidx) q = (u32)(ch.challenge() % M2);
  CHK(hipMemcpyAsync(d_qidx, qidx.data(), qidx.size() * 4, hipMemcpyHostToDevice, st));
  if (stream) {  // leaf rows of the queries: one more pass of chunk LDEs, 84 rows taken out of each
    const int nq = (int)P.num_queries;
    for (int c0 = 0; c0 < W; c0 += CH) {
      const int cn = std::min(CH, W - c0);
      lde_chunk(d_tcoef + (size_t)c0 * N, cn);
      fri_gather_rows(d_ldechunk, M2, cn, d_qidx, nq, d_qrows + (size_t)c0 * nq, st);
    }
    for (int c0 = 0; c0 < A; c0 += CH) {
      const int cn = std::min(CH, A - c0);
      lde_chunk(d_acoef + (size_t)c0 * N, cn);
      fri_gather_rows(d_ldechunk, M2, cn, d_qidx, nq, d_qrows + (size_t)(W + c0) * nq, st);
    }
  }
  {
    QueryGatherArgs G;
    G.lde[0] = d_tlde; G.lde[1] = d_alde; G.lde[2] = d_qlde;
    if (stream) {
      G.lde[0] = d_qrows;
      G.lde[1] = d_qrows + (size_t)W * P.num_queries;
      G.lde_by_query[0] = G.lde_by_query[1] = 1;
    }
    G.tree[0] = d_ttree; G.tree[1] = d_atree; G.tree[2] = d_qtree;
    G.width[0] = W; G.width[1] = A; G.width[2] = NQ;
    for (int l = 0; l < L; l++) {
      G.layer_vals[l] = layer_vals[l];
      G.layer_tree[l] = layer_trees[l];
    }
    G.n_layers = L;
    G.log_m2 = log_m2;
    G.cap_height = P.cap_height;
    G.M2 = M2;
    G.indices = d_qidx;
    G.out = d_qout;
    G.words_per_query = wpq;
    fri_gather_queries(G, P.num_queries, st);
  }
  u64* h_q = (u64*)sl.pinned;  // (the opening partials in it were consumed above)
  CHK(hipMemcpyAsync(h_q, d_qout, n_q * 8, hipMemcpyDeviceToHost, st));
  CHK(hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, st));
  se(ST_FRI);
  se(ST_TOTAL);
  CHK(hipStreamSynchronize(st));
  CHK(hipGetLastError());
  if (h_err) {
    err = "device self-check failed: " + std::to_string(h_err);
    cleanup_events();
    return h_err;
  }

  // ---- assemble (layout: include/bn254_stark.h) ------------------------------------------------------------
  std::vector<u64>& o = pr->words;
  o.clear();
  o.reserve(3 * CAPW + 4 * (W + A) + n_ctlz + 2 * NQ + L * CAPW + n_q + 2 * final_poly.size() + 13);
  auto mark = [&](int id) { pr->sec_off[id] = o.size(); };
  for (int t = 0; t < 3; t++) { mark(BN254S_SEC_TRACE_CAP + t); o.insert(o.end(), caps[t], caps[t] + CAPW); }
  mark(BN254S_SEC_LOCAL_VALUES);
  for (int p = 0; p < W; p++) { o.push_back(op(p, 0)); o.push_back(op(p, 1)); }
  mark(BN254S_SEC_NEXT_VALUES);
  for (int p = 0; p < W; p++) { o.push_back(op(p, 2)); o.push_back(op(p, 3)); }
  mark(BN254S_SEC_AUX_POLYS);
  for (int p = W; p < W + A; p++) { o.push_back(op(p, 0)); o.push_back(op(p, 1)); }
  mark(BN254S_SEC_AUX_POLYS_NEXT);
  for (int p = W; p < W + A; p++) { o.push_back(op(p, 2)); o.push_back(op(p, 3)); }
  mark(BN254S_SEC_CTL_ZS_FIRST);
  for (int i = 0; i < n_ctlz; i++) o.push_back(op(W + num_lookup + i, 4));
  mark(BN254S_SEC_QUOTIENT_POLYS);
  for (int p = W + A; p < W + A + NQ; p++) { o.push_back(op(p, 0)); o.push_back(op(p, 1)); }
  mark(BN254S_SEC_FRI_CAPS);
  for (int l = 0; l < L; l++) o.insert(o.end(), layer_caps[l].begin(), layer_caps[l].end());
  mark(BN254S_SEC_QUERY_ROUNDS);
  o.insert(o.end(), h_q, h_q + n_q);
  mark(BN254S_SEC_FINAL_POLY);
  for (auto& cf : final_poly) { o.push_back(cf.c0); o.push_back(cf.c1); }
  mark(BN254S_SEC_POW_WITNESS);
  o.push_back(pow_witness);
  mark(BN254S_SEC_INIT_CHALLENGER_STATE);
  o.insert(o.end(), init_state, init_state + 12);
  mark(BN254S_SEC_COUNT);
  pr->degree_bits = log_n;
  for (int i = 0; i < ST_COUNT; i++) hipEventElapsedTime(&pr->stage_ms[i], ev[2 * i], ev[2 * i + 1]);
  cleanup_events();
  return BN254S_OK;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------
extern "C" {

// A batch in flight: proof i of the batch is one task of the context's worker pool (ctx.h WorkPool).
struct bn254s_batch {
  bn254s_ctx* c = nullptr;
  int kind = 0;
  bn254s_params params;
  const u64 *scalars = nullptr, *x = nullptr, *off = nullptr;
  size_t n_total = 0, per_proof = 0, n_proofs = 0;
  bn254s_proof** out = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  size_t remaining = 0;
  int first_rc = 0;
  std::string err;
};

int bn254s_prove_batch_begin(bn254s_ctx* c, int kind, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                             const uint64_t* off, size_t n_total, size_t per_proof, bn254s_proof** proofs_out,
                             bn254s_batch** handle) {
  if (!c || kind < 0 || kind > 2 || !params || !scalars || !x || (kind != KIND_FQ && !off) || !proofs_out || !handle || n_total == 0 ||
      per_proof == 0 || params->struct_size != sizeof(bn254s_params))
    return BN254S_E_INVALID_ARG;
  *handle = nullptr;
  const size_t n_proofs = (n_total + per_proof - 1) / per_proof;
  const size_t PW = point_words(kind);
  for (size_t i = 0; i < n_proofs; i++) proofs_out[i] = nullptr;
  HIP_TRY(c, hipSetDevice(c->device));
  // proofs in flight (one stream, host thread and workspace each; the streams share GPU_MAX_HW_QUEUES = 16 hardware queues - more
  // queues than that collapse the throughput: 24 / 32 queues gave 62 / 51 proofs/s).  Round 3, with arrival-order admission of the
  // wide sections: every queued proof should find a slot at once (32 proofs queued on 24 slots: 77.8 proofs/s; on 32 slots: 87.9),
  // and more proofs in flight keep a wide section ready at all times - bench.py with 8 / 16 / 24 / 32 proofs queued and as many
  // slots: 83.9 / 84.8 / 87.3 / 87.9 proofs/s (tools/gpu_depth_sweep.sh, gpu_depth_sweep2.sh); 48: 67.  A slot that is never used
  // costs a stream; its workspace (~4 GB for a 2^16-row proof) is allocated on first use.
  size_t n_slots = 32;
  if (const char* e = getenv("BN254S_SLOTS")) n_slots = std::min((size_t)bn254s_ctx::MAX_SLOTS, (size_t)std::max(1, atoi(e)));
  for (size_t s = 0; s < n_slots; s++)
    if (!c->slot(s)) return BN254S_E_HIP;
  c->workers.start(n_slots, c->device);
  bn254s_batch* B = new bn254s_batch();
  B->c = c;
  B->kind = kind;
  B->params = *params;
  B->scalars = scalars;
  B->x = x;
  B->off = off;
  B->n_total = n_total;
  B->per_proof = per_proof;
  B->n_proofs = n_proofs;
  B->out = proofs_out;
  B->remaining = n_proofs;
  for (size_t i = 0; i < n_proofs; i++)
    c->workers.push([B, i, PW](size_t s) {
      bool skip;
      {
        std::lock_guard<std::mutex> lk(B->mu);
        skip = B->first_rc != 0;  // an earlier proof of this batch failed: the call returns that error
      }
      if (!skip) {
        const size_t b = i * B->per_proof, cnt = std::min(B->per_proof, B->n_total - b);
        bn254s_proof* pr = new bn254s_proof();
        std::string err;
        bn254s_ctx* c = B->c;
        Slot& sl = *c->slot(s);
        auto attempt = [&](const std::function<void()>* after_alloc) {
          err.clear();
          return prove_on_slot(c, sl, B->kind, B->params, B->scalars + 4 * b, B->x + PW * b, B->off ? B->off + PW * b : nullptr, cnt,
                               pr, err, after_alloc);
        };
        int rc = attempt(nullptr);
        // Out of device memory: the workspaces the idle slots keep from earlier (smaller or differently shaped) proofs may be
        // what is in the way - give those back and try again; while other proofs are still running, wait for one of them to
        // finish and repeat (a batch of tall proofs then runs as many at a time as fit).  Alone and still too large: the error.
        // One task at a time does this (retry_mu, held until its workspace is complete or has been given back): two proofs
        // that both fit alone never fail on each other's partial workspaces, and "nobody else is running" is decided under
        // the same lock.  The completion counter is sampled before the attempt, so a proof that ends during it is not missed.
        WorkPool& wp = c->workers;
        while (rc == BN254S_E_OOM) {
          std::unique_lock<std::mutex> rl(wp.retry_mu);
          const size_t c0 = wp.completions_now();
          wp.for_idle_slots([&](size_t i) {
            if (i < c->n_slots_now()) c->slot(i)->mem.release();
          });
          const bool last = wp.active() <= 1;  // nobody else is running: nothing more will be given back
          const std::function<void()> unlock = [&]() { rl.unlock(); };
          rc = attempt(&unlock);
          if (rl.owns_lock()) rl.unlock();
          if (rc != BN254S_E_OOM || last) break;
          wp.wait_for_a_completion(c0);
        }
        if (rc != BN254S_OK) {
          hipStreamSynchronize(sl.st);
          delete pr;
          std::lock_guard<std::mutex> lk(B->mu);
          if (B->first_rc == 0) {
            B->first_rc = rc;
            B->err = err;
          }
        } else {
          B->out[i] = pr;
        }
      }
      std::lock_guard<std::mutex> lk(B->mu);  // (notify under the lock: the waiter frees the batch)
      if (--B->remaining == 0) B->cv.notify_all();
    });
  *handle = B;
  return BN254S_OK;
}

int bn254s_prove_batch_end(bn254s_batch* B) {
  if (!B) return BN254S_E_INVALID_ARG;
  {
    std::unique_lock<std::mutex> lk(B->mu);
    B->cv.wait(lk, [&] { return B->remaining == 0; });
  }
  const int rc = B->first_rc;
  if (rc) {
    B->c->set_err(B->err);
    for (size_t i = 0; i < B->n_proofs; i++) {
      delete B->out[i];
      B->out[i] = nullptr;
    }
  }
  delete B;
  return rc;
}

int bn254s_prove_batch(bn254s_ctx* c, int kind, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                       const uint64_t* off, size_t n_total, size_t per_proof, bn254s_proof** proofs_out) {
  bn254s_batch* B = nullptr;
  int rc = bn254s_prove_batch_begin(c, kind, params, scalars, x, off, n_total, per_proof, proofs_out, &B);
  return rc != BN254S_OK ? rc : bn254s_prove_batch_end(B);
}

int bn254s_prove_g1_batch(bn254s_ctx* c, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                          const uint64_t* off, size_t n_total, size_t per_proof, bn254s_proof** proofs_out) {
  return bn254s_prove_batch(c, KIND_G1, params, scalars, x, off, n_total, per_proof, proofs_out);
}

// The single-proof entry points queue ONE proof of n instances on the worker pool, like a batch of one: a call made while a
// batch is open (between bn254s_prove_batch_begin and _end) takes its turn behind the proofs already queued and runs on a free
// slot - it never shares a stream or workspace with a running proof.
static int prove_one(bn254s_ctx* c, int kind, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x,
                     const uint64_t* off, size_t n, bn254s_proof** out) {
  if (!c || !params || !scalars || !x || (kind != KIND_FQ && !off) || !out || n == 0 || params->struct_size != sizeof(bn254s_params))
    return BN254S_E_INVALID_ARG;
  *out = nullptr;
  return bn254s_prove_batch(c, kind, params, scalars, x, off, n, n, out);
}
int bn254s_prove_g1(bn254s_ctx* c, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x, const uint64_t* off,
                    size_t n, bn254s_proof** out) {
  return prove_one(c, KIND_G1, params, scalars, x, off, n, out);
}
// One process, several GPUs: proof i goes to context i mod n_ctx (proofs share no state), each context
// pipelines its share exactly like bn254s_prove_batch.  No inter-GPU traffic.
int bn254s_prove_batch_multi(bn254s_ctx** ctxs, size_t n_ctx, int kind, const bn254s_params* params, const uint64_t* scalars,
                             const uint64_t* x, const uint64_t* off, size_t n_total, size_t per_proof, bn254s_proof** proofs_out) {
  if (!ctxs || n_ctx == 0 || kind < 0 || kind > 2 || !params || !scalars || !x || (kind != KIND_FQ && !off) || !proofs_out ||
      n_total == 0 || per_proof == 0)
    return BN254S_E_INVALID_ARG;
  for (size_t g = 0; g < n_ctx; g++)
    if (!ctxs[g]) return BN254S_E_INVALID_ARG;
  const size_t n_proofs = (n_total + per_proof - 1) / per_proof;
  const size_t PW = point_words(kind);
  for (size_t i = 0; i < n_proofs; i++) proofs_out[i] = nullptr;
  std::vector<int> rcs(n_ctx, BN254S_OK);
  std::vector<std::thread> th;
  for (size_t g = 0; g < n_ctx; g++)
    th.emplace_back([&, g]() {
      // gather this context's share into contiguous job arrays (proof i of the share = global proof g + i * n_ctx)
      std::vector<u64> ls, lx, lo;
      std::vector<size_t> ids;
      for (size_t i = g; i < n_proofs; i += n_ctx) {
        const size_t b = i * per_proof, cnt = std::min(per_proof, n_total - b);
        if (cnt != per_proof && i + 1 != n_proofs) continue;  // (only the very last proof may be short)
        ids.push_back(i);
        ls.insert(ls.end(), scalars + 4 * b, scalars + 4 * (b + cnt));
        lx.insert(lx.end(), x + PW * b, x + PW * (b + cnt));
        if (off) lo.insert(lo.end(), off + PW * b, off + PW * (b + cnt));
      }
      if (ids.empty()) return;
      // a short last proof must stay last in its share: it is, since shares are in increasing global order
      std::vector<bn254s_proof*> out(ids.size(), nullptr);
      rcs[g] = bn254s_prove_batch(ctxs[g], kind, params, ls.data(), lx.data(), off ? lo.data() : nullptr, ls.size() / 4, per_proof,
                                  out.data());
      if (rcs[g] == BN254S_OK)
        for (size_t k = 0; k < ids.size(); k++) proofs_out[ids[k]] = out[k];
    });
  for (auto& t : th) t.join();
  int rc = BN254S_OK;
  for (size_t g = 0; g < n_ctx; g++)
    if (rcs[g] != BN254S_OK && rc == BN254S_OK) rc = rcs[g];
  if (rc != BN254S_OK)
    for (size_t i = 0; i < n_proofs; i++)
      if (proofs_out[i]) {
        delete proofs_out[i];
        proofs_out[i] = nullptr;
      }
  return rc;
}

int bn254s_prove_g2(bn254s_ctx* c, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x, const uint64_t* off,
                    size_t n, bn254s_proof** out) {
  return prove_one(c, KIND_G2, params, scalars, x, off, n, out);
}
int bn254s_prove_fq_exp(bn254s_ctx* c, const bn254s_params* params, const uint64_t* scalars, const uint64_t* x, size_t n,
                        bn254s_proof** out) {
  return prove_one(c, KIND_FQ, params, scalars, x, nullptr, n, out);
}

int bn254s_proof_words(const bn254s_proof* p, const uint64_t** data, size_t* len) {
  if (!p || !data || !len) return BN254S_E_INVALID_ARG;
  *data = p->words.data();
  *len = p->words.size();
  return BN254S_OK;
}
int bn254s_proof_section(const bn254s_proof* p, int id, const uint64_t** data, size_t* len) {
  if (!p || !data || !len || id < 0 || id >= BN254S_SEC_COUNT) return BN254S_E_INVALID_ARG;
  *data = p->words.data() + p->sec_off[id];
  *len = p->sec_off[id + 1] - p->sec_off[id];
  return BN254S_OK;
}
int bn254s_proof_degree_bits(const bn254s_proof* p) { return p ? p->degree_bits : 0; }
int bn254s_proof_outputs(const bn254s_proof* p, const uint64_t** data, size_t* len) {
  if (!p || !data || !len) return BN254S_E_INVALID_ARG;
  *data = p->outputs.data();
  *len = p->outputs.size();
  return BN254S_OK;
}
int bn254s_proof_stage_ms(const bn254s_proof* p, const float** ms, size_t* n_stages) {
  if (!p || !ms || !n_stages) return BN254S_E_INVALID_ARG;
  *ms = p->stage_ms;
  *n_stages = ST_COUNT;
  return BN254S_OK;
}
const char* bn254s_stage_name(size_t stage) { return stage < ST_COUNT ? STAGE_NAMES[stage] : ""; }
size_t bn254s_proof_serialize(const bn254s_proof* p, uint8_t* buf, size_t cap) {
  if (!p) return 0;
  size_t need = p->words.size() * 8;
  if (buf && cap >= need) memcpy(buf, p->words.data(), need);  // hosts are little-endian
  return need;
}
void bn254s_proof_free(bn254s_proof* p) { delete p; }

int bn254s_generate_trace(bn254s_ctx* c, int kind, const uint64_t* scalars, const uint64_t* x, const uint64_t* off, size_t n,
                          uint32_t min_rows_log2, uint64_t* trace_out, uint64_t* outputs) {
  if (!c || kind < 0 || kind > 2 || !scalars || !x || (kind != KIND_FQ && !off) || n == 0 || !trace_out) return BN254S_E_INVALID_ARG;
  if (min_rows_log2 < 16) {  // the range-check table needs all 2^16 values (scalar_mul_stark.rs:71-87)
    c->set_err("min_rows_log2 must be >= 16");
    return BN254S_E_INVALID_ARG;
  }
  HIP_TRY(c, hipSetDevice(c->device));
  const int PW = point_words(kind);
  const StarkShape sh = shape_for(kind);
  size_t N = rows_for(n, min_rows_log2);
  const size_t in_words = n * (4 + 2 * (size_t)PW);
  u64* d_in = c->words("gt.in", in_words);
  u64* d_trace = c->words("gt.trace", (size_t)sh.W * N);
  const size_t trace_scr = kind == KIND_G1 ? g1_trace_scratch_words(n) : kind == KIND_G2 ? g2_trace_scratch_words(n) : fq_trace_scratch_words(n);
  u64* d_scr = c->words("gt.scratch", trace_scr);
  u64* d_out = c->words("gt.out", n * PW + 8);
  if (!d_in || !d_trace || !d_scr || !d_out) return BN254S_E_OOM;
  int* d_err = (int*)(d_out + n * PW);
  u64 *d_sc = d_in, *d_x = d_in + 4 * n, *d_off = d_x + (size_t)PW * n;
  HIP_TRY(c, hipMemsetAsync(d_err, 0, 4, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_sc, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_x, x, n * (size_t)PW * 8, hipMemcpyHostToDevice, c->stream));
  if (kind != KIND_FQ) HIP_TRY(c, hipMemcpyAsync(d_off, off, n * (size_t)PW * 8, hipMemcpyHostToDevice, c->stream));
  int trc = kind == KIND_G1   ? g1_generate_trace_device(d_sc, d_x, d_off, n, d_trace, N, d_scr, d_out, d_err, c->stream)
            : kind == KIND_G2 ? g2_generate_trace_device(d_sc, d_x, d_off, n, d_trace, N, d_scr, d_out, d_err, c->stream)
                              : fq_generate_trace_device(d_sc, d_x, n, d_trace, N, d_scr, d_out, d_err, c->stream);
  if (trc) {
    c->set_err("trace generation launch failed");
    return BN254S_E_HIP;
  }
  int h_err = 0;
  HIP_TRY(c, hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(trace_out, d_trace, (size_t)sh.W * N * 8, hipMemcpyDeviceToHost, c->stream));
  if (outputs) HIP_TRY(c, hipMemcpyAsync(outputs, d_out, n * (size_t)PW * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (h_err) {
    c->set_err("trace generation reported device error " + std::to_string(h_err));
    return h_err;
  }
  return BN254S_OK;
}
int bn254s_g1_generate_trace(bn254s_ctx* c, const uint64_t* scalars, const uint64_t* x, const uint64_t* off, size_t n,
                             uint32_t min_rows_log2, uint64_t* trace_out, uint64_t* outputs) {
  return bn254s_generate_trace(c, KIND_G1, scalars, x, off, n, min_rows_log2, trace_out, outputs);
}

}  // extern "C"

// loads this translation unit's code object (the HIP runtime defers that to the first launch otherwise)
void prover_module_warm() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_quotient_chunks));
}
