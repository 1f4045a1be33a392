// ctx.hip -- device context and C ABI of libpirgpu (see include/pirgpu.h).
//
// Host-side orchestration of the reference's PIRServer::processQuery
// (server.cpp:173-195) and PIRDatabase (database.cpp) on one MI355X: the encoded
// database, the Galois keys and every intermediate live in HBM; the host only
// sequences kernel launches on the context's stream.
//
// Map of this file (search for "[n]"):
//   [1]  state: Worker (one query in flight), BatchLane (a group of 8), KeySet, BatchSet, pirgpu_ctx
//   [2]  helpers: errors, NTT / Encode tables, options, workspace geometry
//   [3]  oblivious expansion: expand_core (levels: ks_digit -> ks_mac_intt -> ks_mac_combine / ks_combine; last level
//        in the NTT domain), expand_query_to_sv
//   [4]  multiply: scan_group_mfma / scan_on_device, post_scan_stage (inverse NTT, upper levels, fold)
//   [5]  C ABI: create / destroy, options, database loading (db_encode, db_pack), finalize
//   [6]  C ABI: Galois keys, per-client key sets (handles with generations, pins, LRU)
//   [7]  C ABI: single query, join / fork, test hooks
//   [8]  batch pipeline: staging, lanes, expand_group_on_lane, batch_run_mfma (groups of 8 on two lanes)
//   [9]  multi-GPU entry points: u64 exchange, packed row shards, slot shards (pirgpu_slots_*), fix-up, pack40
//   [10] measurement
// The kernels are in kernels.hip, scan_mfma.hip, ntt_kernels.hip (+ ntt_core.h, arith.h); the wire level in wire.cpp.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pirgpu.h"
#include "device_params.h"
#include "host_math.h"
#include "kernels.h"
#include "wire.h"

using namespace pirgpu;

namespace {

thread_local std::string g_create_error;
// last failure of THIS thread (and the context it happened on): pirgpu_last_error stays meaningful when other
// threads use the same context in between
thread_local std::string t_last_error;
thread_local const pirgpu_ctx* t_last_error_ctx = nullptr;
thread_local uint64_t t_last_error_gen = 0;   // generation id of that context (a new context may reuse the address)
std::atomic<uint64_t> g_ctx_generation{0};

struct Fail {
  int code;
  std::string msg;
};

#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      throw Fail{PIRGPU_INTERNAL, std::string(#expr) + ": " + hipGetErrorString(e_)};     \
  } while (0)

enum Phase { PH_EXPAND = 0, PH_SVNTT, PH_SCAN, PH_UPPER, PH_FINAL, PH_COUNT };

}  // namespace

// ======================================================================================================================
// [1] STATE: workers, lanes, key sets, batch sets, the context
// ======================================================================================================================
// Per-query working set: one HIP stream plus every intermediate of one query in flight.
// A context owns one worker by default; pirgpu_set_concurrency adds more so that several
// queries overlap on the GPU (the latency-bound expansion of one hides under the
// bandwidth-bound scan of another).  Database, keys and tables are shared, read-only.
struct Worker {
  hipStream_t stream = nullptr;
  uint32_t keyset = 0;               // key set of the query this worker currently holds
  uint64_t *res_a = nullptr, *res_b = nullptr, *prod = nullptr, *dig = nullptr, *sv_ntt = nullptr;
  uint64_t* d_query = nullptr;
  uint32_t staged_nq = 0;
  const uint64_t* sv_cur = nullptr;  // selection vector the multiply reads: sv_ntt, or caller-owned memory
  const uint64_t* sv_rows = nullptr; // packed multi-GPU exchange: only this shard's dimension-0 selectors, local index
  std::vector<uint64_t*> lvl;  // per level results; lvl[0] = reply
  uint64_t* pt_buf = nullptr;
  uint64_t* scan_part = nullptr;
  uint64_t* up_scratch = nullptr;    // split upper level (large rings): transformed plaintexts of one block of children
  size_t up_scratch_words = 0;
  uint8_t* selp = nullptr;           // digit-packed selectors of the group this worker leads (MFMA scan)
  bool reply_valid = false;
  hipEvent_t ev_expanded = nullptr, ev_scanned = nullptr;  // batch mode: cross-stream hand-offs
  hipEvent_t ev_done = nullptr;      // batch mode: this worker's reply is complete (its buffers may be reused)
  hipEvent_t ev_join = nullptr;      // pirgpu_join: everything queued on this worker's stream so far
};

// Batch mode with the MFMA scan: a group of up to 8 queries is expanded TOGETHER on one lane (the
// expansion kernels run over nodes x queries, ciphertext index = node * B + query), scanned in one
// database pass, and finished per query on the workers' own streams.  Two lanes alternate so that the
// bandwidth-bound scan of one group overlaps the compute-bound expansion of the next.
// The first few levels of a group's expansion tree are a chain of small, latency-bound launches (8-128 tree ciphertexts:
// three dependent launches per level, 5-7 us each on their own, ~20 us each while the other lane's big kernels own the
// chip).  On the lane they cost 0.2-0.35 ms of a group's ~3 ms although they need almost no CU time.  Round 4: they run
// AHEAD on the context's head stream, in small buffers of their own (a ring of two per lane), and the lane picks the
// tree up at level `head_levels` behind an event: the chain is off the lanes' critical path, its few workgroups slip in
// beside the lanes' kernels.
struct HeadSlot {
  uint64_t *res_a = nullptr, *res_b = nullptr, *dig = nullptr, *prod = nullptr;
  hipEvent_t ev_ready = nullptr;    // head stream: the slot holds the tree of its group at level head_levels
  hipEvent_t ev_free = nullptr;     // lane stream: the lane's first level has consumed it
  bool in_use = false;              // ev_free has been recorded at least once
};

struct BatchLane {
  hipStream_t stream = nullptr;
  HeadSlot head[2];
  uint32_t head_next = 0;
  uint64_t *res_a = nullptr, *res_b = nullptr, *prod = nullptr, *dig = nullptr;
  uint8_t* selp = nullptr;
  hipEvent_t ev_scanned = nullptr;
  hipEvent_t ev_join = nullptr;      // pirgpu_join: everything queued on this lane so far
  // the multiply of a whole group runs on the lane's stream in these query-major buffers: one launch per kernel
  // for the (up to) 8 queries instead of one stream + one launch per query
  std::vector<uint64_t*> lvl;        // [level][query][lvl_cts[level]][2][k][N]
  uint64_t* pt_buf = nullptr;        // [query][pt_words]
  uint64_t* scan_part = nullptr;     // [query][column chunk][scan_rows][2][k][N] (matrices wider than one chunk)
  uint64_t* up_scratch = nullptr;    // split upper level (large rings)
  size_t up_scratch_words = 0;
};

// Where one multiply (scan results -> reply) runs: a worker's buffers (one query) or a lane's (a group).
struct Stage {
  hipStream_t stream;
  uint64_t* const* lvl;   // per level, query-major
  uint64_t* pt_buf;       // query-major, pt_words each
  uint32_t n;             // queries
  MfmaPtrs sel;           // per query: the selection vector (whole, NTT form) or, with local_rows, only this shard's
  bool local_rows;        // dimension-0 selectors at local index (packed multi-GPU exchange)
  uint64_t** up_scratch = nullptr;   // owner's scratch for the split upper level, grown on demand
  size_t* up_scratch_words = nullptr;
  bool sel_f64 = false;   // the selectors are exact doubles (a lane's own expansion, pirgpu_ctx::sel_f64)
  bool rows_inverted = false;   // the row sums in lvl[d - 1] are in coefficient form already (slot-sharded step: the
                                // inverse transform gathered them out of the exchange buffer)
};

// One client's Galois keys on the device.  The reference deserialises the keys of every request into a local
// (server.cpp:46-48), so they are per request, not server state; here up to keyset_cap sets stay resident (least
// recently used evicted), slot 0 being the set the direct pirgpu_set_galois_key API installs.
struct KeySet {
  std::map<uint32_t, uint64_t*> keys;   // Galois element -> [k][2][k+1][N], device order, the flavour's element type
  std::vector<uint8_t> blob;            // wire layer: the serialized GaloisKeys object these keys came from
  uint64_t fingerprint = 0;             // sampled hash of `blob` (candidate selection before the full compare)
  uint64_t last_use = 0;
  uint32_t gen = 0;                     // bumped whenever the slot is emptied: a handle of an earlier tenant goes stale
  uint32_t pins = 0;                    // pirgpu_keyset_pin: requests in flight that use this set (never evicted meanwhile)
};

// What the ABI hands out for a key set: (generation << 12) | slot index.  Slot 0 (pirgpu_set_galois_key's set) has the
// handle 0.  A handle whose generation no longer matches names a set that was evicted or released since -- every entry
// point that takes one fails with FailedPrecondition instead of silently switching with another client's keys.
constexpr uint32_t kSlotBits = 12;
constexpr uint32_t kSlotMask = (1u << kSlotBits) - 1;

// One set of batch state: the staged queries and the replies of one batch (device resident), where its replies go,
// its per-query key sets and its pinned host staging.  A context has kBatchSets of them so that two request windows
// can be in flight -- the next one parsed / staged / queued while the previous one's replies are still coming back
// (wire.cpp); lanes, workers and streams are shared, their use is ordered by the streams themselves.
constexpr int kBatchSets = 2;
struct BatchSet {
  uint64_t *d_bquery = nullptr, *d_breply = nullptr;
  uint64_t* ext_reply = nullptr;   // pirgpu_batch_set_reply_buffer: the caller's device buffer batches write replies to
  uint64_t ext_reply_cts = 0;      // its capacity in ciphertexts
  uint64_t* host_reply = nullptr;  // pirgpu_batch_set_host_replies: pinned host memory every group downloads its replies to
  uint64_t host_reply_cts = 0;     // ... its capacity in ciphertexts
  bool host_reply_done = false;    // the batch that just ran queued those downloads: batch_fetch into it only waits
  uint32_t batch_cap = 0, batch_count = 0;   // batch_count: replies the reply buffer holds
  uint32_t staged_count = 0;   // queries pirgpu_batch_stage left in d_bquery (0 again when the buffers are reallocated)
  bool batch_valid = false;
  std::vector<uint32_t> batch_keysets;      // key set INDEX per staged query (all 0 unless pirgpu_batch_set_keysets) ...
  std::vector<uint32_t> batch_keyset_gens;  // ... and the generation it had then: checked again when the batch is run
  // pirgpu_batch_stage_async: the queries arrive in pieces of kStagePiece queries on the main stream, one event each;
  // a group waits for the pieces that hold its queries instead of the whole upload
  std::vector<hipEvent_t> st_events;
  uint32_t st_pieces = 0;          // 0: staged synchronously, nothing to wait for
  // group-wise reply download (pirgpu_batch_set_host_replies): one event per group of the batch that just ran, recorded
  // behind the group's device-to-host copy; dl_end[i] = one past the last query of group i, dl_next = next to report
  std::vector<hipEvent_t> dl_events;
  std::vector<hipEvent_t> rd_events;   // "this group's replies are final" on the lane's stream: what the copy stream waits for
  std::vector<uint32_t> dl_end;
  size_t dl_next = 0;
  uint64_t *h_query = nullptr, *h_reply = nullptr;   // pinned host staging of the wire layer
  size_t h_query_words = 0, h_reply_words = 0;
};
constexpr uint32_t kStagePiece = 8;
static thread_local int t_batch_set = 0;   // pirgpu_batch_select

struct pirgpu_ctx {
  pirgpu_params prm{};
  uint32_t N = 0, logN = 0, k = 0, d = 0;
  size_t ctw = 0;  // words per ciphertext
  uint32_t dims[PIRGPU_MAX_DIMS]{};
  uint64_t stride[PIRGPU_MAX_DIMS + 1]{};  // plaintexts under one node of level l
  uint32_t sv_off[PIRGPU_MAX_DIMS + 1]{};  // selection-vector offset of dimension l
  uint32_t dim_sum = 0;
  uint32_t er = 0, E = 0;  // ExpansionRatio, 2*ER
  uint64_t reply_cts = 1;
  uint64_t P = 0;              // num_pt of the whole database
  uint32_t sb = 0, se = 0;     // shard of top-level indices
  // slot shard (multi-GPU): this context holds the NTT slots [slot0, slot0 + nslots) of EVERY
  // plaintext (device order) in the scan's operand layout; nslots == k N: the whole ring
  uint32_t slot0 = 0, nslots = 0;
  bool slot_sharded = false;
  hipEvent_t ev_after = nullptr;   // pirgpu_slots_*: the position of the caller's `after` stream
  uint64_t pt_begin = 0, pt_end = 0;  // plaintext range held by this context
  uint32_t bits = 0;           // bits per coefficient for item packing

  int device = 0;
  hipStream_t stream = nullptr;
  const NttOps* ops = nullptr;  // NTT kernels for this ring degree
  int mode = kNttInt;           // arithmetic flavour of the NTT kernels (NttMode)
  DevParams hp{};
  DevParams* dp = nullptr;
  std::vector<void*> allocs;

  uint64_t* d_db = nullptr;
  std::vector<uint8_t> loaded;  // per local plaintext
  uint64_t n_loaded = 0;
  std::vector<KeySet> keysets{1};           // resident key sets; [0] = pirgpu_set_galois_key's
  uint32_t keyset_cap = 64;                 // client slots (pirgpu_set_keyset_capacity); 4.7 MB each at N = 4096, k = 2
  uint32_t cur_keyset = 0;                  // slot the single-query entry points use (pirgpu_query_use_keyset) ...
  uint32_t cur_keyset_gen = 0;              // ... and its generation at that time
  uint64_t keyset_clock = 0, key_uploads = 0, keyset_evictions = 0;
  std::vector<uint64_t*> key_pool;          // device key buffers of emptied sets, reused by the next upload (no hipMalloc)
  uint64_t* d_key_stage = nullptr;          // one key in SEAL order on its way to device order (allocated once)
  std::map<uint32_t, uint64_t*> xpow;       // shift -> NTT_j(x^(-shift)), [k][N] doubles (NTT-domain last expansion level)
  std::map<uint32_t, uint16_t*> gperm;      // Galois element g -> [2][N] u16: sigma_g / sigma_(g^-1) on NTT positions (galois_perm_table)

  // workspace geometry (computed on first use) and the workers holding the buffers
  bool ws_ready = false;
  std::vector<Worker> workers;     // workers[0] uses `stream`
  std::vector<uint64_t> lvl_rows;  // nodes per level inside the shard
  std::vector<uint64_t> lvl_cts;   // ciphertexts per level result buffer
  uint64_t m_max = 1;              // expansion tree width (ciphertexts)
  uint64_t pt_words = 0;
  // batch mode (pirgpu_batch_*): kBatchSets independent sets of batch state (BatchSet above); the calling thread's
  // pirgpu_batch_select picks the one its pirgpu_batch_* calls operate on (default 0)
  BatchSet sets[kBatchSets];
  BatchSet& bs();
  uint32_t n_active = 1;
  uint32_t upper_blocks = 512;   // target workgroup count of upper_fused_kernel (PIRGPU_UPPER_BLOCKS)
  uint32_t upper_blocks_batch = 64;   // the same per query in batch mode, where other queries fill the chip too: fewer
                                      // chunks = fewer partial sums to write and fold (PIRGPU_UPPER_BLOCKS_BATCH)
  uint32_t scan_wgs_batch = 128; // persistent workgroups (= CUs) of the MFMA scan in the batch pipeline; 0 = all CUs
                                 // (PIRGPU_SCAN_MFMA_WGS_BATCH)
  bool in_batch = false;         // set while the batch pipeline enqueues work
  uint32_t scan_nsplit = 1, scan_cps = 0, scan_rows = 0, scan_cols = 0;
  uint32_t scan_rpt = 4, scan_block = 256;  // rows per thread / workgroup size of the scan kernel
  bool scan_limb = false;                   // 28-bit limb accumulators (all data moduli < 2^50)
  uint32_t mq_nq = 4, mq_rows = 1;          // batch mode: queries per database pass / rows per wave
  bool mq_single = true;                    // use the LDS-shared scan kernel for single queries too
  uint32_t mq_single_rows = 4;              // rows per wave of that kernel for a single query (2 or 4)
  bool mq_single_limb = false;              // single query: 128-bit accumulators need fewer registers (3 WG/CU)
  uint64_t scan_npt = 0;
  // digit-sliced int8-MFMA scan (scan_mfma.hip): d >= 2, database additionally held in operand layout
  bool mfma_on = false;
  MfmaGeom mg{};
  uint32_t mfma_nq = kMaxMfmaQueries;       // queries per database pass in batch mode
  bool mfma_single = true;                  // single queries use it too (off for matrices wider than one chunk)
  bool scan_f64_fold = false;               // the scan folds its digit diagonals in exact fp64 arithmetic (option
                                            // SCAN_F64_FOLD; every data modulus below 2^50)
  bool scan_f64_fold_batch = false;         // ... in the launches that serve a group of queries (option SCAN_F64_FOLD_BATCH)
  bool split_upper = false;                 // upper level as transform-to-scratch + elementwise MAC (N >= 16384 in the fp64
                                            // flavours, where the fused kernel spills; PIRGPU_SPLIT_UPPER=0/1 overrides)
  uint64_t split_upper_words = (3ull << 30) / 8;  // scratch budget per lane / worker (PIRGPU_SPLIT_UPPER_MB)
  // the digit kernel of the wide expansion levels and the transform kernel of the split upper level run all transforms
  // of one source polynomial in ONE workgroup: one load and one permutation per source, stores drain under the next
  // transform (option LOOP_TRANSFORMS; fp64 flavours)
  bool loop_transforms = true;
  uint32_t loop_min_sources = 1024;         // ... from this many (tree ciphertext, digit) sources per launch on
  uint32_t fuse_mac_nodes = 128;            // ... from this many tree ciphertexts per level on (narrower levels are latency-
                                            // bound: two dependent transform kernels cost more than mac + light combine)
  bool fuse_mac_combine = true;             // levels below the last: combine step in the data residues' MAC + inverse-NTT
                                            // kernel (PIRGPU_FUSE_MAC_COMBINE=0: separate ks_combine pass)
  bool fuse_last_level = true;              // last expansion level fused with the selector NTT (PIRGPU_FUSE_LAST=0: off)
  bool last_level_ntt = true;               // ... and carried out in the NTT domain (PIRGPU_LAST_NTT=0: coefficient form)
  bool tree40 = true;                       // expansion tree between fused wide levels as 5-byte polynomials (PIRGPU_TREE40=0: doubles)
  bool c0_ntt = false;                      // the fused levels keep the tree's c0 polynomials in NTT form (option C0_NTT = 1 / 2; measured slower)
  bool c0_ntt_split = false;                // ... with component 0 of a level as a launch of its own (C0_NTT = 2)
  bool want_sel_f64 = true;
  bool sel_f64 = false;                     // batch lanes keep their selectors as exact doubles between the last expansion
                                            // level and their two consumers (sel_pack, upper level): no u64 round trip
                                            // (fp64 flavours, MFMA scan, NTT-domain last level; PIRGPU_SEL_F64=0: off)
  bool pack40 = false;                      // key-switch digits stored packed (PIRGPU_PACK40) ...
  int pack_bytes = 5;                       // ... in 5 (all moduli < 2^39), 6 (< 2^47) or 7 (< 2^55) bytes per residue (PIRGPU_PACK_BYTES)
  uint8_t* d_dbp = nullptr;
  bool packed_valid = false;
  bool staging_released = false;            // pirgpu_db_finalize(release): only the operand-layout copy is left
  std::vector<BatchLane> lanes;             // created on the first batch
  uint64_t groups_run = 0;
  hipStream_t head_stream = nullptr;        // the narrow first levels of every group's expansion (HeadSlot)
  uint32_t head_levels = 5;                 // levels 0 .. head_levels - 1 run there (option HEAD_LEVELS; 0: all on the lane)
  uint32_t head_mode = 1;                   // option HEAD_MODE: 0 never, 1 batches staged with pirgpu_batch_stage_async, 2 always
  hipEvent_t ev_head_tail = nullptr;        // the head stream's position (an upload into a batch set waits for it)
  hipStream_t copy_stream = nullptr;        // device-to-host downloads of finished groups (pirgpu_batch_set_host_replies):
                                            // a lane that downloaded its own replies sat idle for 8 MB of PCIe per group
  hipEvent_t ev_fetch[2] = {nullptr, nullptr};   // pirgpu_query_fetch_begin: the two halves of the reply's download
  hipEvent_t ev_fork = nullptr;             // pirgpu_fork: the main stream's position
  hipEvent_t ev_main_join = nullptr;        // pirgpu_join_stream onto a caller's stream: the main stream's position

  bool prof = false;
  // batch pipeline under profiling: HIP events around the scan launches of the groups (the launch that serves the
  // headline step runs on `scan_wgs_batch` workgroups beside the other lane's kernels -- not the full-chip single-query
  // launch the roofline block quotes); read out by pirgpu_batch_scan_timings
  std::vector<hipEvent_t> bscan_ev;   // pairs (start, stop)
  uint32_t bscan_n = 0, bscan_wgs = 0, bscan_nq = 0;
  static constexpr int kMaxProfRuns = 256;
  std::vector<hipEvent_t> ev;  // kMaxProfRuns x (PH_COUNT + 1), created lazily
  int prof_runs = 0;           // runs recorded since the last read-out
  int prof_cur = -1;           // event set of the run being recorded (-1: not recording)
  float timings[6]{};

  // options by name (pirgpu_set_option); a name not set here falls back to the environment variable PIRGPU_<NAME>
  // (A/B scripts under tools/), then to the built-in default
  std::map<std::string, int64_t> opts;
  std::string err;
  uint64_t gen = 0;         // unique per context (thread-local error messages are keyed on it, not on the address)
  std::recursive_mutex mu;  // per ABI call; pirgpu_process_request holds it across its whole body
  uint64_t zero_pts = 0;    // all-zero database plaintexts in this shard (SEAL: "result ciphertext is transparent")
  uint64_t remote_zero_pts = 0;   // ... in the other shards of a row-sharded database (pirgpu_set_remote_zero_plaintexts)
  bool allow_transparent = false;

  template <typename T>
  T* dalloc(size_t count) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    allocs.push_back(p);
    return static_cast<T*>(p);
  }
  void use_device() { HIP_TRY(hipSetDevice(device)); }
};

BatchSet& pirgpu_ctx::bs() { return sets[t_batch_set]; }

namespace {

// ======================================================================================================================
// [2] HELPERS: errors, tables (twiddles, Encode order), options, workspace geometry (ensure_workspace)
// ======================================================================================================================
int fail(pirgpu_ctx* c, int code, const std::string& msg) {
  if (c) {
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    c->err = msg;
    t_last_error = msg;
    t_last_error_ctx = c;
    t_last_error_gen = c->gen;
  }
  return code;
}

template <typename F>
int guarded(pirgpu_ctx* c, F&& f) {
  if (!c) return PIRGPU_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  try {
    c->use_device();
    return f();
  } catch (const Fail& e) {
    return fail(c, e.code, e.msg);
  } catch (const std::exception& e) {
    return fail(c, PIRGPU_INTERNAL, e.what());
  }
}

void build_tables(pirgpu_ctx* c) {
  const uint32_t N = c->N, k = c->k;
  DevParams& hp = c->hp;
  hp.N = N;
  hp.logN = c->logN;
  hp.k = k;
  std::vector<Twiddle> tw(N), itw(N);
  uint64_t qmax = 0;
  for (uint32_t i = 0; i <= k; ++i) {
    const uint64_t q = i < k ? c->prm.coeff_modulus[i] : c->prm.special_prime;
    qmax = std::max(qmax, q);
    hp.mod[i].q = q;
    hm::u128 ratio = (~(hm::u128)0) / q;  // floor((2^128 - 1) / q) == floor(2^128 / q) for odd q > 1
    hp.mod[i].br_lo = (uint64_t)ratio;
    hp.mod[i].br_hi = (uint64_t)(ratio >> 64);
    const uint64_t psi = hm::minimal_primitive_root(2ull * N, q);
    if (!psi) throw Fail{PIRGPU_INVALID_ARGUMENT, "modulus has no primitive 2N-th root of unity"};
    const uint64_t ipsi = hm::invmod_prime(psi, q);
    uint64_t pw = 1, ipw = 1;
    for (uint32_t j = 0; j < N; ++j) {
      uint32_t r = hm::bitrev(j, c->logN);
      tw[r] = Twiddle{pw, hm::shoup(pw, q)};
      itw[r] = Twiddle{ipw, hm::shoup(ipw, q)};
      pw = hm::mulmod(pw, psi, q);
      ipw = hm::mulmod(ipw, ipsi, q);
    }
    Twiddle* dev = c->dalloc<Twiddle>((size_t)2 * N);
    HIP_TRY(hipMemcpy(dev, tw.data(), N * sizeof(Twiddle), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dev + N, itw.data(), N * sizeof(Twiddle), hipMemcpyHostToDevice));
    hp.tab[i].tw = dev;
    hp.tab[i].itw = dev + N;
    const uint64_t ninv = hm::invmod_prime(N % q, q);
    const uint64_t iw1n = hm::mulmod(itw[1].w, ninv, q);
    hp.tab[i].ninv = Twiddle{ninv, hm::shoup(ninv, q)};
    hp.tab[i].iw1n = Twiddle{iw1n, hm::shoup(iw1n, q)};
    // exact-fp64 flavour: the same constants as signed doubles in (-q/2, q/2]
    auto centered = [q](uint64_t v) { return v > q / 2 ? -(double)(q - v) : (double)v; };
    std::vector<double> twf(N), itwf(N);
    for (uint32_t j = 0; j < N; ++j) {
      twf[j] = centered(tw[j].w);
      itwf[j] = centered(itw[j].w);
    }
    double* devf = c->dalloc<double>((size_t)2 * N);
    HIP_TRY(hipMemcpy(devf, twf.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(devf + N, itwf.data(), N * sizeof(double), hipMemcpyHostToDevice));
    hp.tab[i].twf = devf;
    hp.tab[i].itwf = devf + N;
    hp.tab[i].ninv_f = centered(ninv);
    hp.tab[i].iw1n_f = centered(iw1n);
    hp.tab[i].qd = (double)q;
    hp.tab[i].qinvd = 1.0 / (double)q;
  }
  // NTT arithmetic flavour (ntt_core.h).  PIRGPU_NTT_MODE=0 forces the integer path.
  c->mode = qmax < (1ull << 46) ? kNttF64 : (qmax < (1ull << 49) ? kNttF64Wide : kNttInt);
  if (const char* v = pirgpu_env("PIRGPU_NTT_MODE")) {
    int want = atoi(v);
    if (want == kNttInt || (want == kNttF64Wide && c->mode != kNttInt) || want == c->mode) c->mode = want;
  }
  hp.ntt_mode = c->mode;
  hp.f64_lazy_inv = (64 - (uint32_t)__builtin_clzll(qmax)) + c->logN <= 52 ? 1u : 0u;
  if (const char* v = pirgpu_env("PIRGPU_F64_LAZY_INV")) hp.f64_lazy_inv = atoi(v) ? hp.f64_lazy_inv : 0u;
  const uint64_t p = c->prm.special_prime, t = c->prm.plain_modulus;
  hp.p_half = p >> 1;
  hp.p_f = (double)p;
  hp.p_half_f = (double)(p >> 1);
  hp.t = t;
  hp.plain_thr = (t + 1) >> 1;
  for (uint32_t j = 0; j < k; ++j) {
    const uint64_t q = hp.mod[j].q;
    hp.p_half_mod[j] = (p >> 1) % q;
    hp.p_inv[j] = hm::invmod_prime(p % q, q);
    hp.p_inv_s[j] = hm::shoup(hp.p_inv[j], q);
    hp.p_inv_f[j] = hp.p_inv[j] > q / 2 ? -(double)(q - hp.p_inv[j]) : (double)hp.p_inv[j];
    hp.lift_inc[j] = q - (t % q);
    uint64_t w = (1ull << 32) % q, acc = w;
    for (int e = 0; e < 3; ++e) {     // 2^(32 (e + 1)) mod q, centred
      hp.fold_w[j][e] = acc > q / 2 ? -(double)(q - acc) : (double)acc;
      acc = hm::mulmod(acc, w, q);
    }
  }
  // CiphertextReencoder::Encode order (reference ct_reencoder.cpp:49-69)
  const uint32_t b = hm::bits_per_coeff(t);
  hp.enc_bits = b;
  uint32_t e = 0;
  for (uint32_t poly = 0; poly < 2; ++poly)
    for (uint32_t j = 0; j < k; ++j) {
      uint32_t ler = hm::local_expansion_ratio(hp.mod[j].q, b);
      for (uint32_t i = 0; i < ler; ++i) {
        if (e >= (uint32_t)kMaxEnc) throw Fail{PIRGPU_INVALID_ARGUMENT, "expansion ratio too large"};
        hp.enc_poly[e] = (uint8_t)poly;
        hp.enc_res[e] = (uint8_t)j;
        hp.enc_shift[e] = (uint8_t)(i * b);
        ++e;
      }
    }
  hp.enc_count = e;
  c->E = e;
  c->er = e / 2;
  uint32_t qbits = 64 - __builtin_clzll(qmax);
  int room = 128 - 2 * (int)qbits;
  hp.lazy_limit = 1u << std::min(30, std::max(0, room));
  c->dp = c->dalloc<DevParams>(1);
  HIP_TRY(hipMemcpy(c->dp, &hp, sizeof(DevParams), hipMemcpyHostToDevice));
}

uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// Option `name` (upper case, without the PIRGPU_ prefix): pirgpu_set_option's value, else the environment variable
// PIRGPU_<name>, else dflt.  *present reports whether either was given.
int64_t option(const pirgpu_ctx* c, const char* name, int64_t dflt, bool* present = nullptr) {
  if (present) *present = true;
  auto it = c->opts.find(name);
  if (it != c->opts.end()) return it->second;
  const std::string env = std::string("PIRGPU_") + name;
  const char* v = pirgpu_env(env.c_str());
  if (v && *v) return strtoll(v, nullptr, 10);
  if (present) *present = false;
  return dflt;
}

// waits for everything the batch pipeline has enqueued: lane streams (grouped expansion + multiply) and worker streams
void sync_batch_streams(pirgpu_ctx* c) {
  for (BatchLane& ln : c->lanes)
    if (ln.stream) HIP_TRY(hipStreamSynchronize(ln.stream));
  if (c->head_stream) HIP_TRY(hipStreamSynchronize(c->head_stream));
  if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));   // reply downloads queued behind the lanes
  for (Worker& w : c->workers)
    if (w.stream) HIP_TRY(hipStreamSynchronize(w.stream));
}

bool all_zero_bytes(const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (p[i]) return false;
  return true;
}

// loaded[] per local plaintext: 0 = not loaded, 1 = loaded, 2 = loaded and identically zero
void note_plaintext(pirgpu_ctx* c, uint64_t local, bool zero) {
  uint8_t& st = c->loaded[local];
  if (!st) ++c->n_loaded;
  if (st == 2) --c->zero_pts;
  st = zero ? 2 : 1;
  if (zero) ++c->zero_pts;
}

// Evaluator::multiply_plain throws logic_error("result ciphertext is transparent") when its result has an
// all-zero c1 (SEAL's default build, SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT), which happens for every query as
// soon as one database plaintext is identically zero; PIRDatabase::multiply maps it to InternalError
// (reference database.cpp:308-315).  pirgpu_set_transparent_policy(ctx, 1) returns the mathematically defined
// reply instead.
void check_transparent(pirgpu_ctx* c) {
  // the reference fails every query as soon as ANY plaintext of the whole database is zero: a row shard also counts
  // the zero plaintexts the other shards reported, so that all ranks of a sharded server take the same decision
  if ((c->zero_pts || c->remote_zero_pts) && !c->allow_transparent)
    throw Fail{PIRGPU_INTERNAL, "result ciphertext is transparent"};
}

void alloc_worker(pirgpu_ctx* c, Worker& w);
void ensure_expansion_buffers(pirgpu_ctx* c, Worker& w);

void ensure_workspace(pirgpu_ctx* c) {
  if (c->ws_ready) return;
  const uint32_t N = c->N, k = c->k, d = c->d;

  const uint64_t m_max = std::min<uint64_t>(N, hm::next_power_two(std::max<uint32_t>(c->dim_sum, 1)));
  c->m_max = m_max;
  // per-level node counts inside this shard and result buffers
  const uint64_t shard_pts = c->pt_end - c->pt_begin;
  c->upper_blocks = (uint32_t)std::max<int64_t>(1, option(c, "UPPER_BLOCKS", c->upper_blocks));
  c->upper_blocks_batch = (uint32_t)std::max<int64_t>(1, option(c, "UPPER_BLOCKS_BATCH", c->upper_blocks_batch));
  c->scan_wgs_batch = (uint32_t)std::max<int64_t>(0, option(c, "SCAN_MFMA_WGS_BATCH", c->scan_wgs_batch));
  c->upper_blocks_batch = std::min(c->upper_blocks_batch, c->upper_blocks);  // the scratch is sized for upper_blocks
  c->lvl_rows.assign(d, 0);
  c->lvl_cts.assign(d, 0);
  uint64_t pt_words = 0;
  for (uint32_t l = 0; l < d; ++l) {
    uint64_t rows = l == 0 ? 1 : ceil_div(shard_pts, c->stride[l]);
    uint64_t C = 1;
    for (uint32_t x = l; x + 1 < d; ++x) C *= c->E;
    c->lvl_rows[l] = rows;
    c->lvl_cts[l] = std::max<uint64_t>(rows, 1) * C;
    if (l + 1 < d) {
      // chunk partial sums of upper_fused_kernel: n_chunks x (rows * C * E * 2 * k) polynomials
      const uint64_t per_chunk = std::max<uint64_t>(rows, 1) * (C / c->E) * c->E * k;
      const uint64_t n_chunks = std::min<uint64_t>(c->dims[l], std::max<uint64_t>(1, ceil_div(c->upper_blocks, per_chunk)));
      pt_words = std::max<uint64_t>(pt_words, (n_chunks + 1) * std::max<uint64_t>(rows, 1) * C * 2 * k * N);
    }
  }
  c->pt_words = pt_words;
  // base-level scan geometry
  if (d == 1) {
    c->scan_rows = 1;
    c->scan_cols = (uint32_t)shard_pts;
  } else {
    c->scan_cols = c->dims[d - 1];
    c->scan_rows = (uint32_t)ceil_div(shard_pts, c->scan_cols);
  }
  c->scan_npt = shard_pts;
  {
    // Scan launch geometry.  Tuning knobs (environment, read once per context):
    //   PIRGPU_SCAN_ROWS   rows accumulated per thread (1,2,3,4,6,8)
    //   PIRGPU_SCAN_BLOCK  workgroup size (64..256)
    //   PIRGPU_SCAN_NSPLIT column splits (partial sums reduced by reduce_splits_kernel)
    auto env_u32 = [c](const char* name, uint32_t dflt) {   // name = "PIRGPU_<OPTION>"
      return (uint32_t)option(c, name + 7, dflt);
    };
    //   PIRGPU_SCAN_LIMB   0 forces the generic 128-bit accumulators
    c->scan_limb = true;
    for (uint32_t j = 0; j < k; ++j) c->scan_limb = c->scan_limb && (c->hp.mod[j].q >> 50) == 0;
    if (!env_u32("PIRGPU_SCAN_LIMB", 1)) c->scan_limb = false;
    //   (settled in rounds 1 - 2, constants since round 5: 4 queries share one pass of the 64-bit kernels in batch mode,
    //   one row per wave then; a single query takes four rows per wave with limb accumulators off -- HISTORY section 9)
    //   PIRGPU_SCAN_MQ_SINGLE=0 routes single queries through scan_kernel instead of the LDS-shared
    //                      scan_mq_kernel
    c->mq_nq = 4;
    c->mq_rows = 1;
    c->mq_single = env_u32("PIRGPU_SCAN_MQ_SINGLE", 1) != 0;
    c->mq_single_rows = 4;
    c->mq_single_limb = false;
    c->scan_rpt = env_u32("PIRGPU_SCAN_ROWS", 4);
    c->scan_block = env_u32("PIRGPU_SCAN_BLOCK", 256);
    const uint32_t xblocks = (k * N / 2 + c->scan_block - 1) / c->scan_block;
    const uint32_t yblocks = (c->scan_rows + c->scan_rpt - 1) / c->scan_rpt;
    uint64_t have = (uint64_t)xblocks * std::max<uint32_t>(yblocks, 1) * c->scan_block / 256;
    // split the columns only when whole rows cannot fill the chip (d = 1 / few rows)
    uint32_t want = have >= 512 ? 1 : (uint32_t)ceil_div(1024, std::max<uint64_t>(have, 1));
    want = env_u32("PIRGPU_SCAN_NSPLIT", want);
    want = std::max<uint32_t>(1, std::min<uint32_t>(want, std::max<uint32_t>(c->scan_cols, 1)));
    c->scan_cps = (uint32_t)ceil_div(std::max<uint32_t>(c->scan_cols, 1), want);
    c->scan_nsplit = (uint32_t)ceil_div(std::max<uint32_t>(c->scan_cols, 1), c->scan_cps);
    //   PIRGPU_SCAN_MFMA=0 keeps the 64-bit multiply-accumulate kernels for d >= 2 as well
    //   PIRGPU_SCAN_MFMA_NQ queries per database pass of the MFMA scan in batch mode (1..8)
    c->fuse_last_level = env_u32("PIRGPU_FUSE_LAST", 1) != 0;
    c->fuse_mac_combine = env_u32("PIRGPU_FUSE_MAC_COMBINE", 1) != 0;
    c->last_level_ntt = env_u32("PIRGPU_LAST_NTT", 1) != 0;
    //   PIRGPU_C0_NTT      c0 of the expansion tree in NTT form from the first fused level on (needs the fused levels and
    //                      the NTT-domain last level; fp64 flavours)
    {
      // 0: off (the default: measured -4 % as two launches, -7.6 % as one kernel, profiles/r06_ab_c0_ntt_*.txt), 1: both
      // components of a level in one kernel, 2: component 0 as a launch of its own
      const uint32_t v = env_u32("PIRGPU_C0_NTT", 0);
      c->c0_ntt = v != 0;
      c->c0_ntt_split = v >= 2;
    }
    c->want_sel_f64 = env_u32("PIRGPU_SEL_F64", 1) != 0;
    c->tree40 = env_u32("PIRGPU_TREE40", 1) != 0;
    c->fuse_mac_nodes = 128;       // swept in round 2 (HISTORY section 4): a constant since round 5
    //   PIRGPU_HEAD_LEVELS  expansion levels of a batch group that run ahead on the head stream (0: none)
    c->head_levels = std::min<uint32_t>(env_u32("PIRGPU_HEAD_LEVELS", c->head_levels), 8);
    c->head_mode = std::min<uint32_t>(env_u32("PIRGPU_HEAD_MODE", c->head_mode), 2);
    c->split_upper = env_u32("PIRGPU_SPLIT_UPPER", c->logN >= 14 ? 1 : 0) != 0 && c->mode != kNttInt;
    c->split_upper_words = (uint64_t)env_u32("PIRGPU_SPLIT_UPPER_MB", 3072) * (1ull << 20) / 8;
    c->loop_transforms = env_u32("PIRGPU_LOOP_TRANSFORMS", 1) != 0 && c->mode != kNttInt;
    // (loop_min_sources: swept 512 - 4096 in round 4, 1 024 stays: a constant since round 5)
    c->pack40 = env_u32("PIRGPU_PACK40", 1) != 0;
    // packed storage of the key-switch intermediates: the fp64 flavours store x + q (|x| <= q) in the fewest whole bytes
    // -- 5 for q < 2^39, 6 for q < 2^47 (cfg 4: 43 / 44 bits), 7 for q < 2^55 (cfg 5: 48 / 49 bits) -- instead of doubles;
    // the integer flavour knows the 5-byte form only.  PIRGPU_PACK_BYTES=<5|6|7> forces a (sufficient) width, 8 = doubles.
    {
      uint32_t bits = 0;
      for (uint32_t j = 0; j <= k; ++j) bits = std::max<uint32_t>(bits, 64 - (uint32_t)__builtin_clzll(c->hp.mod[j].q));
      const int need = bits <= 39 ? 5 : (bits <= 47 ? 6 : (bits <= 55 ? 7 : 8));
      int want = (int)env_u32("PIRGPU_PACK_BYTES", (uint32_t)need);
      if (want < need) want = need;
      if (c->mode == kNttInt && want != 5) want = 8;
      if (want > 7) c->pack40 = false;
      c->pack_bytes = c->pack40 ? want : 5;
      // (N = 16384 with 7-byte residues: the combine kernel that also reads and writes the TREE packed needs 68 bytes of
      // scratch under the 128-register cap of a 1024-thread workgroup -- the tree stays in doubles there, digits and
      // products are packed)
      if (c->pack40 && c->pack_bytes == 7 && c->logN >= 14 && !env_u32("PIRGPU_TREE40_WIDE", 0)) c->tree40 = false;
      const NttOps* ops = ntt_ops_for(N, c->pack_bytes);
      if (ops != c->ops) {
        c->ops = ops;
        HIP_TRY(c->ops->configure(c->mode));
      }
    }
    //   PIRGPU_SCAN_MFMA_WIDE  0 / 1 forces the 8-wave / 4-wave (one wave per SIMD, up to 7 k-steps) scan kernel
    bool wide_given = false;
    const int64_t wide = option(c, "SCAN_MFMA_WIDE", 0, &wide_given);
    //   PIRGPU_SCAN_MFMA_TOP4=0 keeps the top digit of database and selectors as a full byte (scan_mfma.hip TOP4)
    c->mg = mfma_geometry(c->hp, c->scan_rows, c->scan_cols, wide_given ? (int)(wide != 0) : -1,
                          env_u32("PIRGPU_SCAN_MFMA_TOP4", 1) != 0);
    c->mfma_on = env_u32("PIRGPU_SCAN_MFMA", 1) != 0 && d >= 2 && c->mg.L != 0 && c->scan_rows >= 8 && shard_pts > 0;
    if (c->slot_sharded && (!c->mfma_on || c->mg.nchunks != 1 || c->nslots % c->mg.NW))
      throw Fail{PIRGPU_INVALID_ARGUMENT, "a slot shard needs the int8-MFMA scan in one column chunk (d = 2, >= 8 rows, "
                                          "moduli below 2^55, at most 28 column groups)"};
    c->mfma_nq = std::max<uint32_t>(1, std::min<uint32_t>(env_u32("PIRGPU_SCAN_MFMA_NQ", kMaxMfmaQueries),
                                                          kMaxMfmaQueries));
    //   PIRGPU_SCAN_MFMA_SINGLE: a single query on a matrix wider than one column chunk pays the partial-sum
    //   round trip for one query only; there the 64-bit kernels on the u64 copy are faster (cfg 4: 5.6 vs 6.4 ms)
    c->mfma_single = env_u32("PIRGPU_SCAN_MFMA_SINGLE", c->mg.nchunks == 1 ? 1 : 0) != 0;
    //   PIRGPU_SCAN_F64_FOLD  fp64 fold of the digit diagonals (moduli below 2^50).  Default: on from 6 digits per
    //   residue (cfg 4: scan 4.53 against 4.76 ms, cfg 5: 8.33 against 8.57, same box); at 5 digits (cfg 3) the
    //   single-query pass measured 3 % slower with it (0.212 against 0.206 ms) and the batch step 0.5 % faster: off
    c->scan_f64_fold = env_u32("PIRGPU_SCAN_F64_FOLD", c->mg.L >= 6 ? 1 : 0) != 0;
    //   PIRGPU_SCAN_F64_FOLD_BATCH  the same choice for the launches that serve a GROUP of queries (batch pipeline on part
    //   of the chip, slot-sharded step): on whenever the moduli allow -- with 16 result columns per tile instead of 2 the
    //   fold is 8 x the work per database byte, and there the cheaper fold pays at 5 digits too (round 6, same box, three
    //   alternating runs: 5 362 / 5 426 / 5 363 against 5 322 / 5 331 / 5 319 queries/s; profiles/r06_ab_scan_store_fold.txt)
    c->scan_f64_fold_batch = env_u32("PIRGPU_SCAN_F64_FOLD_BATCH", 1) != 0;
    for (uint32_t j = 0; j < k; ++j) {
      c->scan_f64_fold = c->scan_f64_fold && (c->hp.mod[j].q >> 50) == 0;
      c->scan_f64_fold_batch = c->scan_f64_fold_batch && (c->hp.mod[j].q >> 50) == 0;
    }
    // selectors as doubles inside a lane: every query ciphertext must go through ks_last_ntt_kernel (>= 2 items each)
    const uint64_t rem = c->dim_sum % N;
    c->sel_f64 = c->want_sel_f64 && c->mfma_on && c->mode != kNttInt && c->fuse_last_level && c->last_level_ntt &&
                 !c->split_upper && c->dim_sum >= 2 && rem != 1;
  }
  c->ws_ready = true;
  if (c->workers.empty()) c->workers.emplace_back();
  c->workers[0].stream = c->stream;
  for (Worker& w : c->workers) alloc_worker(c, w);
  ensure_expansion_buffers(c, c->workers[0]);  // worker 0 serves the single-query entry points and test hooks
}

// Expansion tree (ping/pong) and key-switch scratch of one worker, on first use.
void ensure_expansion_buffers(pirgpu_ctx* c, Worker& w) {
  if (w.res_a) return;
  const uint32_t N = c->N, k = c->k;
  w.res_a = c->dalloc<uint64_t>(c->m_max * c->ctw);
  w.res_b = c->dalloc<uint64_t>(c->m_max * c->ctw);
  w.prod = c->dalloc<uint64_t>(std::max<uint64_t>(c->m_max / 2, 1) * 2 * (k + 1) * N);
  w.dig = c->dalloc<uint64_t>(std::max<uint64_t>(c->m_max / 2, 1) * (k + 1) * k * N);
}

// Allocates one worker's buffers (and its stream unless it is the context's main stream).
void alloc_worker(pirgpu_ctx* c, Worker& w) {
  if (w.sv_ntt) return;
  const size_t ctw = c->ctw;
  if (!w.stream) HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
  // the expansion tree / key-switch scratch is allocated on first use (ensure_expansion_buffers): in batch
  // mode the lanes expand, so only a worker that runs single queries needs its own
  w.sv_ntt = c->dalloc<uint64_t>((size_t)std::max<uint32_t>(c->dim_sum, 1) * ctw);
  w.d_query = c->dalloc<uint64_t>((size_t)(c->dim_sum / c->N + 1) * ctw);
  w.lvl.assign(c->d, nullptr);
  for (uint32_t l = 0; l < c->d; ++l) w.lvl[l] = c->dalloc<uint64_t>(c->lvl_cts[l] * ctw);
  if (c->pt_words) w.pt_buf = c->dalloc<uint64_t>(c->pt_words);
  const uint32_t parts = std::max<uint32_t>(c->scan_nsplit, c->mfma_on ? c->mg.nchunks : 1);
  if (parts > 1) w.scan_part = c->dalloc<uint64_t>((size_t)parts * std::max<uint32_t>(c->scan_rows, 1) * ctw);
  HIP_TRY(hipEventCreateWithFlags(&w.ev_expanded, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&w.ev_scanned, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
}

// ======================================================================================================================
// [3] OBLIVIOUS EXPANSION (server.cpp:105-171): keys per query, expand_core, expand_query_to_sv
// ======================================================================================================================
uint32_t galois_inverse(uint32_t g, uint32_t N) {
  // (Z/2N)^* has exponent dividing N, so g^-1 = g^(N-1) mod 2N
  uint64_t mod = 2ull * N, r = 1, b = g % mod;
  uint32_t e = N - 1;
  while (e) {
    if (e & 1) r = (r * b) % mod;
    b = (b * b) % mod;
    e >>= 1;
  }
  return (uint32_t)r;
}

// The key sets a staged batch / the single-query selection name must still be the ones that were named: a set that was
// evicted or released since (its generation moved on) fails the run with FailedPrecondition instead of switching the
// query with the next tenant's keys.
uint32_t current_keyset(pirgpu_ctx* c) {
  const uint32_t i = c->cur_keyset;
  if (i >= c->keysets.size() || (i && c->keysets[i].gen != c->cur_keyset_gen))
    throw Fail{PIRGPU_FAILED_PRECONDITION, "stale key set handle: the selected set was evicted or released (claim it again)"};
  return i;
}
void check_staged_keysets(pirgpu_ctx* c) {
  BatchSet& b = c->bs();
  for (size_t q = 0; q < b.batch_keysets.size(); ++q) {
    const uint32_t i = b.batch_keysets[q];
    if (i >= c->keysets.size() || (i && q < b.batch_keyset_gens.size() && c->keysets[i].gen != b.batch_keyset_gens[q]))
      throw Fail{PIRGPU_FAILED_PRECONDITION,
                 "stale key set handle: a key set of the staged batch was evicted or released (pirgpu_batch_set_keysets again)"};
  }
}

const uint64_t* find_key(pirgpu_ctx* c, uint32_t slot, uint32_t g) {
  if (slot >= c->keysets.size()) throw Fail{PIRGPU_INVALID_ARGUMENT, "key set slot out of range"};
  auto& keys = c->keysets[slot].keys;
  auto it = keys.find(g);
  if (it == keys.end())
    throw Fail{PIRGPU_INTERNAL, "Galois key not present"};  // SEAL throws -> InternalError (server.cpp:72-74)
  return it->second;
}

// Keys for Galois element g of the B queries of a group (ksets[q] = key set slot of query q; nullptr: the context's
// current slot for all).  One client -> B = 1, the kernels then index a single pointer.
KeyPtrs keys_for(pirgpu_ctx* c, uint32_t g, const uint32_t* ksets, uint32_t B) {
  KeyPtrs kp{};
  bool same = true;
  for (uint32_t q = 0; q < B; ++q) {
    kp.p[q] = find_key(c, ksets ? ksets[q] : current_keyset(c), g);
    same = same && kp.p[q] == kp.p[0];
  }
  kp.B = same ? 1 : B;
  return kp;
}

void record(pirgpu_ctx* c, Worker& w, int idx) {
  if (c->prof_cur >= 0) HIP_TRY(hipEventRecord(c->ev[(size_t)c->prof_cur * (PH_COUNT + 1) + idx], w.stream));
}

void begin_profiled_run(pirgpu_ctx* c) {
  c->prof_cur = -1;
  if (!c->prof || c->prof_runs >= pirgpu_ctx::kMaxProfRuns) return;
  if (c->ev.empty()) {
    c->ev.resize((size_t)pirgpu_ctx::kMaxProfRuns * (PH_COUNT + 1));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
  }
  c->prof_cur = c->prof_runs++;
}

// NTT_j(x^(-shift)) for the k data moduli as doubles in device order (ks_last_ntt_kernel's X); built once per shift.
const uint64_t* xpow_table(pirgpu_ctx* c, hipStream_t st, uint32_t shift) {
  auto it = c->xpow.find(shift);
  if (it != c->xpow.end()) return it->second;
  const uint32_t N = c->N, k = c->k;
  std::vector<uint64_t> h((size_t)k * N, 0);
  for (uint32_t j = 0; j < k; ++j) h[(size_t)j * N + (N - shift)] = c->hp.mod[j].q - 1;   // x^(-s) = -x^(N-s)
  uint64_t* buf = c->dalloc<uint64_t>((size_t)k * N);
  HIP_TRY(hipMemcpy(buf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(c->ops->ntt_batch(st, c->mode, c->dp, buf, k, k, 0, false));
  HIP_TRY(launch_tree_convert(st, c->dp, c->mode, buf, buf, (uint64_t)k * N, true));
  HIP_TRY(hipStreamSynchronize(st));   // other streams use the table without an event
  c->xpow[shift] = buf;
  return buf;
}

// sigma_g acting on NTT-form data (ks_last_ntt_kernel, ks_combine_c0_ntt): SEAL position P holds the evaluation at
// psi^(2 br(P) + 1), so NTT(sigma_g(a)) = NTT(a) o pi_g with pi_g(P) = br(((2 br(P) + 1) g mod 2N - 1) / 2).  The table
// holds, for the thread that owns positions EPT t .. EPT t + EPT - 1, the PADDED LDS word index of pi_g(P)'s device slot
// (ntt_core.h lds_idx), first for g, then for g^-1; built once per Galois element.
const uint16_t* galois_perm_table(pirgpu_ctx* c, hipStream_t st, uint32_t g) {
  auto it = c->gperm.find(g);
  if (it != c->gperm.end()) return it->second;
  const uint32_t N = c->N, logN = c->logN, R = (uint32_t)ntt_log_ept((int)logN), EPT = 1u << R, NT = N >> R;
  auto brev = [&](uint32_t v) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < logN; ++b) r |= ((v >> b) & 1u) << (logN - 1 - b);
    return r;
  };
  std::vector<uint16_t> h((size_t)2 * N);
  const uint32_t gs[2] = {g, galois_inverse(g, N)};
  for (int w = 0; w < 2; ++w)
    for (uint32_t P = 0; P < N; ++P) {
      const uint32_t ex = (uint32_t)(((uint64_t)(2 * brev(P) + 1) * gs[w]) & (2 * N - 1));
      const uint32_t Pin = brev(ex >> 1);
      const uint32_t slot = (Pin & (EPT - 1)) * NT + (Pin >> R);
      h[(size_t)w * N + P] = (uint16_t)(slot + (slot >> R));
    }
  uint16_t* buf = c->dalloc<uint16_t>((size_t)2 * N);
  HIP_TRY(hipMemcpy(buf, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(st));   // (hipMemcpy of pageable memory is synchronous; other streams use the table without an event)
  c->gperm[g] = buf;
  return buf;
}

// oblivious_expansion(ct, n) on the device (reference server.cpp:105-146).
// Input ciphertext must already be in res_a[0]; returns the buffer holding the
// next_power_two(n) results.
// B queries at once: ciphertext index = node * B + query, so every level is the same three launches
// over B times the nodes (the per-level Galois element and monomial shift are uniform).
// sel_dst (optional, fp64 flavours): per query, where selector 0 of this ciphertext's expansion goes (NTT form).  The
// last level then produces the selectors directly -- in the NTT domain (ks_last_ntt_kernel; the default) or fused with
// their forward transform (ks_last_level_kernel) -- and nullptr is returned: there is no coefficient-form result to
// transform any more.  sel_f64: selectors written as exact doubles (a batch lane's own consumers only).
// Levels: narrow ones digit -> products of all moduli -> combine; wide ones digit -> special-prime product -> data
// products + combine in one kernel, the tree between two such levels in 5-byte polynomials (cur40).
// A tree in the middle of its expansion: handed from the head stream's levels to the lane's (HeadSlot).
struct TreeState {
  uint64_t* cur = nullptr;   // buffer holding the current level's ciphertexts
  bool cur40 = false;        // ... as 5-byte polynomials (between fused levels) instead of doubles
  bool c0ntt = false;        // ... with their c0 polynomials in NTT form (fused levels, ks_combine_c0_ntt)
};

// Levels [j_begin, j_end) of the tree (j_end clamped to the tree's depth).  `from` (j_begin > 0): where the earlier levels
// left the tree -- that buffer is only READ; the first level here writes res_a, later ones ping-pong res_a / res_b.
// `to` (j_end < depth): receives the state after level j_end - 1 instead of the function running to the leaves.
// after_first: called when the first level's launches are queued (the `from` buffer may be recycled behind them).
uint64_t* expand_core(pirgpu_ctx* c, hipStream_t st, uint64_t* res_a, uint64_t* res_b, uint64_t* dig, uint64_t* prod,
                      uint32_t n, uint32_t B, const MfmaPtrs* sel_dst = nullptr, bool sel_f64 = false,
                      const uint32_t* ksets = nullptr, uint32_t j_begin = 0, uint32_t j_end = UINT32_MAX,
                      const TreeState* from = nullptr, TreeState* to = nullptr,
                      const std::function<void()>& after_first = nullptr) {
  const uint32_t N = c->N, k = c->k;
  if (n > N) throw Fail{PIRGPU_INVALID_ARGUMENT, "Cannot expand more items from a CT than poly modulus degree"};
  const uint32_t logm = hm::ceil_log2(n);
  const bool fuse_last = sel_dst && c->mode != kNttInt && c->fuse_last_level;
  if (sel_f64 && !(fuse_last && c->last_level_ntt && n >= 2)) throw Fail{PIRGPU_INTERNAL, "double-form selectors need the NTT-domain last level"};
  uint64_t *cur = res_a, *nxt = res_b;
  bool cur40 = false;   // `cur` holds 5-byte polynomials (between fused levels) instead of doubles
  bool c0ntt = false;   // `cur` holds its c0 polynomials in NTT form
  if (from) {
    cur = from->cur;
    cur40 = from->cur40;
    c0ntt = from->c0ntt;
    nxt = res_a;
  }
  const uint32_t j_stop = std::min(j_end, logm);
  // c0 in NTT form through the fused levels (ks_combine_c0_ntt): only for a tree whose leaves are consumed in NTT form by
  // ks_last_ntt_kernel -- pirgpu_expand's coefficient-form results keep the coefficient-form tree
  const bool c0_path = c->c0_ntt && fuse_last && c->last_level_ntt && c->mode != kNttInt && c->fuse_mac_combine;
  const uint32_t fuse_from = B > 1 ? std::max<uint32_t>(c->fuse_mac_nodes / 2, 1) : c->fuse_mac_nodes;
  // level j runs the fused form (special-prime product, then data products + combine in one kernel)
  auto level_fused = [&](uint32_t j) {
    return c->mode != kNttInt && c->fuse_mac_combine && ((1u << j) * B) >= fuse_from && !(fuse_last && j + 1 == logm);
  };
  for (uint32_t j = j_begin; j < j_stop; ++j) {
    // (picking a tree up from another buffer: after its first level here the ping-pong is res_a <-> res_b)
    auto level_done = [&]() {
      if (from && j == j_begin) {
        nxt = res_b;
        if (after_first) after_first();
      }
    };
    const uint32_t g = (N >> j) + 1;
    const KeyPtrs key = keys_for(c, g, ksets, B);   // per query of the group: its own client's key
    const uint32_t nodes = (1u << j) * B;
    const bool last_ntt = fuse_last && j + 1 == logm && c->last_level_ntt;
    const bool c0_in_digit = last_ntt && !c0ntt && ks_digit_takes_c0(nodes);
    if (cur40 && !(last_ntt ? (c0ntt || c0_in_digit) : true)) throw Fail{PIRGPU_INTERNAL, "5-byte tree reached a level that cannot read it"};
    // the root enters a fused level directly (a group wide enough at level 0): its c0 goes to NTT form in place --
    // never in a buffer handed over by another stream (`from` is read-only; its owner converted it after ITS last level)
    if (c0_path && !c0ntt && level_fused(j) && !cur40 && !(from && j == j_begin)) {
      HIP_TRY(c->ops->tree_c0_fwd(st, c->mode, c->dp, k, cur, nodes));
      c0ntt = true;
    }
    HIP_TRY(c->ops->ks_digit(st, c->mode, c->dp, k, cur, g, nodes, dig, c->pack40, c0_in_digit ? prod : nullptr, cur40,
                             c->loop_transforms && (uint64_t)nodes * k >= c->loop_min_sources));
    // (a group of B queries reaches the width at which the fused form pays one level earlier than a single query:
    // measured +0.6 % batched with the threshold at 64 tree ciphertexts, while a single query loses latency below 128)
    if (level_fused(j)) {
      // special-prime product first (the only one that goes through HBM), then the data residues with the combine
      // step in their epilogue: no data products in HBM, no separate combine pass.  Between two fused levels (and into
      // the NTT-domain last level) the tree is written as 5-byte polynomials: these launches are HBM-bound
      const uint32_t next_nodes = nodes * 2;
      const bool next_last = j + 2 == logm;
      const bool next_reads40 = next_last ? (fuse_last && c->last_level_ntt && (c0ntt || ks_digit_takes_c0(next_nodes)))
                                          : (next_nodes >= fuse_from);
      const bool out40 = c->tree40 && c->pack40 && j + 1 < logm && next_reads40 && (1u << j) < (N >> ntt_log_ept((int)c->logN));
      HIP_TRY(c->ops->ks_mac_intt(st, c->mode, c->dp, k, dig, key, nodes, prod, c->pack40, k, 1));
      HIP_TRY(c->ops->ks_mac_combine(st, c->mode, c->dp, k, dig, key, prod, cur, g, nodes, 1u << j, nxt, c->pack40, cur40,
                                     out40, c0ntt ? xpow_table(c, st, 1u << j) : nullptr,
                                     c0ntt ? galois_perm_table(c, st, g) : nullptr, c->c0_ntt_split));
      cur40 = out40;
      std::swap(cur, nxt);
      level_done();
      continue;
    }
    if (cur40 && !last_ntt) throw Fail{PIRGPU_INTERNAL, "5-byte tree reached an unfused level"};
    if (c0ntt && !last_ntt) throw Fail{PIRGPU_INTERNAL, "NTT-form c0 reached an unfused level"};
    if (fuse_last && j + 1 == logm && c->last_level_ntt) {
      // last level in the NTT domain: only the special-prime product is inverse-transformed
      const uint64_t* X = xpow_table(c, st, 1u << j);
      HIP_TRY(c->ops->ks_mac_intt(st, c->mode, c->dp, k, dig, key, nodes, prod, c->pack40, k, 1));
      HIP_TRY(c->ops->ks_last_ntt(st, c->mode, c->dp, k, cur, dig, key, prod, X, g, galois_inverse(g, N), 1u << j, n, B,
                                  *sel_dst, nodes, c->pack40, sel_f64, c0_in_digit, c0ntt ? (cur40 ? 2 : 1) : 0,
                                  galois_perm_table(c, st, g)));
      return nullptr;
    }
    HIP_TRY(c->ops->ks_mac_intt(st, c->mode, c->dp, k, dig, key, nodes, prod, c->pack40, 0, k + 1));
    if (fuse_last && j + 1 == logm) {
      HIP_TRY(c->ops->ks_last_level(st, c->mode, c->dp, k, cur, prod, g, 1u << j, n, B, *sel_dst, nodes, c->pack40));
      return nullptr;
    }
    // outputs n*B.. of the last level are never read (only the first n results per query are used)
    const uint32_t hi_limit = j + 1 == logm ? n * B : UINT32_MAX;
    HIP_TRY(launch_ks_combine(st, c->dp, c->mode, N, k, cur, prod, galois_inverse(g, N), nodes, 1u << j, true, hi_limit,
                              c->pack40, nxt, c->pack_bytes));
    std::swap(cur, nxt);
    // the next level is a fused one: this (narrow) level's outputs, in this stream's own buffer, get their c0 transformed
    if (c0_path && j + 1 < logm && level_fused(j + 1)) {
      HIP_TRY(c->ops->tree_c0_fwd(st, c->mode, c->dp, k, cur, nodes * 2));
      c0ntt = true;
    }
    level_done();
  }
  if (to) {
    to->cur = cur;
    to->cur40 = cur40;
    to->c0ntt = c0ntt;
  }
  return cur;
}

uint64_t* expand_on_device(pirgpu_ctx* c, Worker& w, uint32_t n) {
  ensure_expansion_buffers(c, w);
  return expand_core(c, w.stream, w.res_a, w.res_b, w.dig, w.prod, n, 1, nullptr, false, &w.keyset);
}

// expansion of all staged query ciphertexts into sv_ntt (NTT form) -- reference
// server.cpp:148-171 followed by the lazy transform_to_ntt_inplace of
// database.cpp:190,222 applied to every selector.
void expand_query_to_sv(pirgpu_ctx* c, Worker& w, const uint64_t* d_query, uint32_t nq) {
  ensure_expansion_buffers(c, w);
  w.sv_cur = nullptr;
  w.sv_rows = nullptr;
  const uint32_t N = c->N, k = c->k;
  const size_t ctw = c->ctw;
  uint64_t remaining = c->dim_sum;
  uint64_t produced = 0;
  for (uint32_t q = 0; q < nq; ++q) {
    uint32_t n = (uint32_t)std::min<uint64_t>(remaining, N);
    if (n > 0) {
      // the query ciphertext (canonical u64) becomes the root of the expansion tree (the tree's element type)
      HIP_TRY(launch_tree_convert(w.stream, c->dp, c->mode, d_query + (size_t)q * ctw, w.res_a, ctw, true));
      MfmaPtrs dst{};
      dst.p[0] = w.sv_ntt + produced * ctw;
      uint64_t* res = expand_core(c, w.stream, w.res_a, w.res_b, w.dig, w.prod, n, 1, &dst, false, &w.keyset);
      if (res) HIP_TRY(c->ops->ct_ntt_fwd_oop(w.stream, c->mode, c->dp, k, res, w.sv_ntt + produced * ctw, n, true));
    }
    produced += n;
    remaining -= n;
    if (remaining == 0) break;
  }
}

// ======================================================================================================================
// [4] MULTIPLY (database.cpp:170-258): the scan (scan_group_mfma / scan_on_device) and everything after it (post_scan_stage)
// ======================================================================================================================
// PIRDatabase::multiply on the device (reference database.cpp:170-258), with the
// selection vector already in NTT form in sv_ntt.  Leaves the reply in lvl[0].
// selectors of the scanned (last) dimension for this worker's query
const uint64_t* scan_selectors(pirgpu_ctx* c, Worker& w) {
  const uint64_t* sv = w.sv_cur ? w.sv_cur : w.sv_ntt;
  return sv + (size_t)c->sv_off[c->d - 1] * c->ctw + (c->d == 1 ? (size_t)c->sb * c->ctw : 0);
}

bool mq_usable(pirgpu_ctx* c) { return c->scan_nsplit == 1 && c->scan_rows >= 1 && c->scan_cols >= 1; }

// bytes of the operand-layout copy this context holds (its slots of every plaintext)
size_t dbp_bytes(const pirgpu_ctx* c) { return (size_t)c->nslots * c->mg.RT * c->mg.KG * c->mg.tile_bytes; }

// A slot shard holds 1 / G of every plaintext: it serves the pirgpu_slots_* step only.
void refuse_slot_shard(const pirgpu_ctx* c) {
  if (c->slot_sharded)
    throw Fail{PIRGPU_FAILED_PRECONDITION, "this context is a slot shard: it serves the pirgpu_slots_* entry points only"};
}

// Brings the operand-layout copy of the database up to date (after loads); one-time cost per load.
void ensure_packed(pirgpu_ctx* c) {
  if (!c->mfma_on || c->packed_valid) return;
  if (!c->d_dbp) c->d_dbp = c->dalloc<uint8_t>(dbp_bytes(c));
  HIP_TRY(launch_db_pack(c->stream, c->dp, c->mg, c->d_db, c->d_dbp, c->scan_rows, c->scan_cols, c->k * c->N, c->slot0,
                         c->nslots));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->packed_valid = true;
}

// One pass of the MFMA scan for up to 8 queries (the caller has ordered stream `st` after their expansions): pack
// the column selectors (unless they arrive packed), scan, fold column chunks.  Row sums of query q go to
// out + q * scan_rows ciphertexts; `part` (query-major as well) holds the per-chunk sums of matrices wider than one chunk.
void scan_group_mfma(pirgpu_ctx* c, hipStream_t st, uint8_t*& selp, const MfmaPtrs& col_sel, uint32_t n, uint64_t* out_base,
                     uint64_t* part, Worker* profiled, const uint8_t* packed = nullptr, bool sel_f64 = false,
                     bool share_chip = false) {
  const uint32_t kN = c->k * c->N;
  const uint64_t words = (uint64_t)c->scan_rows * c->ctw;
  ensure_packed(c);
  // query q of the group writes at out + q * qstride: the lane's row sums, or -- matrices wider than one chunk -- its
  // per-chunk partial sums
  uint64_t* const out = c->mg.nchunks > 1 ? part : out_base;
  const uint64_t out_qstride = c->mg.nchunks > 1 ? (uint64_t)c->mg.nchunks * words : words;
  if (!packed) {  // with `packed` the group's digit-packed column selectors already exist (multi-GPU exchange)
    if (!selp) selp = c->dalloc<uint8_t>(c->mg.sel_bytes);
    HIP_TRY(launch_sel_pack(st, c->dp, c->mg, col_sel, n, selp, c->scan_cols, kN, sel_f64));
    packed = selp;
  }
  if (profiled) record(c, *profiled, PH_SCAN);  // selector packing counts as selector preparation, not as the scan
  // batch pipeline: the pass runs on part of the chip and overlaps the other lane's transform kernels whenever the
  // call that queued it has a second group for the other lane (share_chip: decided from the batch's shape, not from
  // the streams' state at enqueue time); a lone group gets the whole chip
  const bool share = share_chip && !profiled && c->scan_wgs_batch;
  const uint32_t wgs = share ? c->scan_wgs_batch : 0;
  constexpr uint32_t kMaxBscan = 64;
  const bool timed = c->prof && !profiled && c->bscan_n < kMaxBscan;
  if (timed) {
    while (c->bscan_ev.size() < 2 * (size_t)(c->bscan_n + 1)) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      c->bscan_ev.push_back(e);
    }
    HIP_TRY(hipEventRecord(c->bscan_ev[2 * c->bscan_n], st));
  }
  HIP_TRY(launch_scan_mfma(st, c->dp, c->mg, c->d_dbp, packed, out, out_qstride, n, c->scan_rows, kN, words, wgs,
                           n > 1 ? (c->scan_f64_fold || c->scan_f64_fold_batch) : c->scan_f64_fold));
  if (timed) {
    HIP_TRY(hipEventRecord(c->bscan_ev[2 * c->bscan_n + 1], st));
    ++c->bscan_n;
    c->bscan_wgs = wgs;
    c->bscan_nq = n;
  }
  if (c->mg.nchunks > 1)
    HIP_TRY(launch_reduce_splits(st, c->dp, part, c->mg.nchunks, words, out_base, n, (uint64_t)c->mg.nchunks * words, words));
}

// Base case of PIRDatabase::multiply (reference database.cpp:185-194,238-247): one fused
// multiply_plain + add_inplace pass over the database.  Leaves NTT-form row sums in lvl[d-1].
void scan_on_device(pirgpu_ctx* c, Worker& w) {
  const uint32_t N = c->N, k = c->k, d = c->d;
  const size_t ctw = c->ctw;
  refuse_slot_shard(c);
  if (c->n_loaded != c->pt_end - c->pt_begin)
    throw Fail{PIRGPU_FAILED_PRECONDITION, "database not fully loaded"};
  check_transparent(c);
  if (c->pt_end == c->pt_begin) return;
  if (c->mfma_on && c->mfma_single) {
    MfmaPtrs col{};
    col.p[0] = scan_selectors(c, w);
    scan_group_mfma(c, w.stream, w.selp, col, 1, w.lvl[d - 1], w.scan_part, &w);
    return;
  }
  const uint64_t* sv_base = scan_selectors(c, w);
  uint64_t* base_out = w.lvl[d - 1];
  if (c->mq_single && mq_usable(c)) {
    HIP_TRY(launch_scan_mq(w.stream, c->dp, N, k, c->d_db, &sv_base, &base_out, 1, c->scan_rows, c->scan_cols,
                           c->mq_single_rows, c->mq_single_limb && c->scan_limb));
    return;
  }
  uint64_t* scan_out = c->scan_nsplit > 1 ? w.scan_part : base_out;
  HIP_TRY(launch_scan(w.stream, c->dp, N, k, c->d_db, sv_base, scan_out, c->scan_rows, c->scan_cols, c->scan_npt,
                      c->scan_nsplit, c->scan_cps, c->scan_rpt, c->scan_block, c->scan_limb));
  if (c->scan_nsplit > 1)
    HIP_TRY(launch_reduce_splits(w.stream, c->dp, w.scan_part, c->scan_nsplit, (uint64_t)c->scan_rows * ctw,
                                 base_out));
}

// Everything after the scan: inverse NTT of the row sums and the upper recursion levels
// (reference database.cpp:196-254).  Leaves the reply in lvl[0].
void post_scan_stage(pirgpu_ctx* c, const Stage& sg, Worker* profiled) {
  const uint32_t N = c->N, k = c->k, d = c->d;
  const size_t ctw = c->ctw;
  const uint64_t shard_pts = c->pt_end - c->pt_begin;
  hipStream_t st = sg.stream;
  if (shard_pts == 0) {
    HIP_TRY(hipMemsetAsync(sg.lvl[0], 0, (size_t)sg.n * c->reply_cts * ctw * 8, st));
    return;
  }
  if (profiled) record(c, *profiled, PH_UPPER);  // end of scan phase
  if (!sg.rows_inverted)
    HIP_TRY(c->ops->ntt_batch(st, c->mode, c->dp, sg.lvl[d - 1], (uint64_t)sg.n * c->scan_rows * 2 * k, k, 0, true));
  // upper levels: fused re-encode + lift + NTT + multiply-accumulate over chunks of children,
  // then one kernel folds the chunk sums and applies the inverse NTT
  uint64_t C = 1;  // ciphertexts per child
  for (int l = (int)d - 2; l >= 0; --l) {
    const uint64_t nch = ceil_div(shard_pts, c->stride[l + 1]);
    const uint64_t rows = c->lvl_rows[l];
    // dimension-0 selectors: the query's whole selection vector at index sv_off[0] + shard_begin + i, or (packed
    // multi-GPU exchange) a buffer holding just this shard's rows at local index i
    const bool local_rows = l == 0 && sg.local_rows;
    const uint32_t sv_first = local_rows ? 0 : c->sv_off[l] + (l == 0 ? c->sb : 0);
    // enough workgroups to fill the chip: ~1024 over (queries * rows * C * chunks * E * k)
    const uint64_t per_chunk = rows * C * c->E * k;
    const uint32_t target = sg.n > 1 || c->in_batch ? c->upper_blocks_batch : c->upper_blocks;
    // children one output row actually has in this shard: dims[l], or -- at the top level of a row shard -- only the
    // shard's rows (a 20-row shard of 162 used to leave three of its four chunks empty)
    const uint64_t kids = rows == 1 ? std::min<uint64_t>(c->dims[l], std::max<uint64_t>(nch, 1)) : c->dims[l];
    uint32_t n_chunks = (uint32_t)std::min<uint64_t>(kids, std::max<uint64_t>(1, ceil_div(target, per_chunk)));
    const uint32_t chunk_len = (uint32_t)ceil_div(kids, n_chunks);
    n_chunks = (uint32_t)ceil_div(kids, chunk_len);
    const uint64_t out_polys = rows * C * c->E * 2 * k;
    if (out_polys * n_chunks * N > c->pt_words)
      throw Fail{PIRGPU_INTERNAL, "upper-level scratch undersized"};
    if (local_rows && l != 0) throw Fail{PIRGPU_INTERNAL, "local row selectors are a d = 2 feature"};
    if (c->split_upper && sg.up_scratch) {
      // large rings: transform the re-encoded plaintexts of a block of children into scratch, multiply-accumulate
      // them elementwise, next block (upper_ntt_kernel / upper_mac_kernel)
      const uint64_t unit = (uint64_t)sg.n * rows * C * c->E * k * N;   // scratch words per child of the block
      // children per output row that exist in this shard: at the top level of a row shard only its own rows (the
      // selector buffer of the packed multi-GPU exchange holds no more than those -- reading on to dims[l] ran past it)
      const uint32_t nd = (uint32_t)kids;
      uint32_t blk = (uint32_t)std::min<uint64_t>(nd, std::max<uint64_t>(1, c->split_upper_words / unit));
      if (*sg.up_scratch_words < unit * blk) {
        HIP_TRY(hipStreamSynchronize(st));
        if (*sg.up_scratch) {
          auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)*sg.up_scratch);
          if (it != c->allocs.end()) c->allocs.erase(it);
          HIP_TRY(hipFree(*sg.up_scratch));
          *sg.up_scratch = nullptr;
        }
        *sg.up_scratch = c->dalloc<uint64_t>(unit * blk);
        *sg.up_scratch_words = unit * blk;
      }
      for (uint32_t b0 = 0; b0 < nd; b0 += blk) {
        HIP_TRY(c->ops->upper_ntt(st, c->mode, c->dp, k, c->E, sg.lvl[l + 1], *sg.up_scratch, (uint32_t)rows, c->dims[l],
                                  (uint32_t)nch, (uint32_t)C, b0, blk, sg.n, c->lvl_cts[l + 1] * ctw, c->loop_transforms));
        HIP_TRY(launch_upper_mac(st, c->dp, *sg.up_scratch, sg.sel, sg.pt_buf, sg.lvl[l], sg.n, (uint32_t)rows, (uint32_t)C,
                                 c->E, k, N, sv_first, b0, blk, nd, b0 == 0, b0 + blk >= nd, c->pt_words,
                                 c->lvl_cts[l] * ctw));
      }
      if (l == 0 && profiled) record(c, *profiled, PH_FINAL);
    } else {
      HIP_TRY(c->ops->upper_fused(st, c->mode, c->dp, k, c->E, sg.lvl[l + 1], sg.sel, sg.pt_buf, (uint32_t)rows,
                                  c->dims[l], (uint32_t)nch, sv_first, (uint32_t)C, chunk_len, n_chunks, sg.n,
                                  c->lvl_cts[l + 1] * ctw, c->pt_words, sg.sel_f64));
      if (l == 0 && profiled) record(c, *profiled, PH_FINAL);
      // fold the chunk sums (wide, elementwise) and return to coefficient form (database.cpp:250-254)
      HIP_TRY(launch_reduce_splits(st, c->dp, sg.pt_buf, n_chunks, out_polys * N, sg.lvl[l], sg.n, c->pt_words,
                                   c->lvl_cts[l] * ctw));
    }
    HIP_TRY(c->ops->ntt_batch(st, c->mode, c->dp, sg.lvl[l], (uint64_t)sg.n * out_polys, k, 0, true));
    C *= c->E;
  }
  if (d == 1 && profiled) record(c, *profiled, PH_FINAL);
}

// Everything after the scan for one worker's query: inverse NTT of the row sums and the upper recursion levels
// (reference database.cpp:196-254).  Leaves the reply in lvl[0].
void post_scan_on_device(pirgpu_ctx* c, Worker& w) {
  Stage sg{w.stream, w.lvl.data(), w.pt_buf, 1, MfmaPtrs{}, w.sv_rows != nullptr, &w.up_scratch, &w.up_scratch_words};
  sg.sel.p[0] = w.sv_rows ? w.sv_rows : (w.sv_cur ? w.sv_cur : w.sv_ntt);
  post_scan_stage(c, sg, &w);
}

void multiply_on_device(pirgpu_ctx* c, Worker& w) {
  scan_on_device(c, w);
  post_scan_on_device(c, w);
}

void run_staged(pirgpu_ctx* c, Worker& w, bool profile) {
  ensure_workspace(c);
  if (w.staged_nq != c->dim_sum / c->N + 1)
    throw Fail{PIRGPU_INVALID_ARGUMENT,
               "Number of ciphertexts doesn't match number of items for oblivious expansion."};
  c->prof_cur = -1;
  w.keyset = current_keyset(c);
  // a batch group that borrowed this worker's selection vector (its multiply runs on a lane stream) must be done
  HIP_TRY(hipStreamWaitEvent(w.stream, w.ev_done, 0));
  if (profile) begin_profiled_run(c);
  record(c, w, PH_EXPAND);
  // expansion and selection-vector NTT are interleaved per query ciphertext; the
  // PH_SVNTT mark is taken after the last expansion level of the last ciphertext.
  expand_query_to_sv(c, w, w.d_query, w.staged_nq);
  record(c, w, PH_SVNTT);
  record(c, w, PH_SCAN);
  multiply_on_device(c, w);
  record(c, w, PH_COUNT);
  c->prof_cur = -1;
  w.reply_valid = true;
  // a batch group that borrows this worker's selection vector next (on a lane's stream, possibly queued by another
  // thread before this query has been fetched) waits for this point
  HIP_TRY(hipEventRecord(w.ev_done, w.stream));
}

}  // namespace

// =============================================================================
// C ABI
// =============================================================================

extern "C" {

// ======================================================================================================================
// [5] C ABI: create / destroy, options, database loading
// ======================================================================================================================
const char* pirgpu_create_error(void) { return g_create_error.c_str(); }

int pirgpu_create(const pirgpu_params* p, pirgpu_ctx** out) {
  if (!p || !out) return PIRGPU_INVALID_ARGUMENT;
  *out = nullptr;
  pirgpu_ctx* c = new pirgpu_ctx();
  c->gen = ++g_ctx_generation;
  auto bail = [&](int code, const std::string& msg) {
    g_create_error = msg;
    pirgpu_destroy(c);
    return code;
  };
  try {
    c->prm = *p;
    const uint32_t N = p->poly_modulus_degree, k = p->num_data_primes;
    if (N < 2048 || N > 16384 || (N & (N - 1)))
      return bail(PIRGPU_INVALID_ARGUMENT, "poly_modulus_degree must be 2048, 4096, 8192 or 16384");
    if (k < 1 || k > PIRGPU_MAX_PRIMES) return bail(PIRGPU_INVALID_ARGUMENT, "invalid number of data primes");
    if (p->use_ciphertext_multiplication)
      return bail(PIRGPU_UNIMPLEMENTED,
                  "use_ciphertext_multiplication is not supported by the MI355X path (decomposition mode only)");
    if (p->special_prime == 0)
      return bail(PIRGPU_INVALID_ARGUMENT, "a key-switching special prime is required (SEAL: keyswitching unsupported)");
    for (uint32_t i = 0; i <= k; ++i) {
      uint64_t q = i < k ? p->coeff_modulus[i] : p->special_prime;
      if (q >> 61 || q < 2 || !hm::is_prime(q) || (q - 1) % (2ull * N))
        return bail(PIRGPU_INVALID_ARGUMENT, "coeff modulus must be a prime < 2^61 congruent to 1 mod 2N");
      for (uint32_t j = 0; j < i; ++j)
        if (p->coeff_modulus[j] == q) return bail(PIRGPU_INVALID_ARGUMENT, "coeff moduli must be distinct");
    }
    if (p->plain_modulus < 2 || p->plain_modulus >> 60)
      return bail(PIRGPU_INVALID_ARGUMENT, "invalid plain modulus");
    if (p->num_dimensions < 1 || p->num_dimensions > PIRGPU_MAX_DIMS)
      return bail(PIRGPU_INVALID_ARGUMENT, "invalid number of dimensions");
    c->N = N;
    c->k = k;
    while ((1u << c->logN) < N) ++c->logN;
    c->ctw = (size_t)2 * k * N;
    c->d = p->num_dimensions;
    c->P = p->num_pt;
    uint64_t ds = 0;
    for (uint32_t l = 0; l < c->d; ++l) {
      c->dims[l] = p->dimensions[l];
      if (c->dims[l] == 0) return bail(PIRGPU_INVALID_ARGUMENT, "dimension of size 0");
      c->sv_off[l] = (uint32_t)ds;
      ds += c->dims[l];
    }
    c->sv_off[c->d] = (uint32_t)ds;
    if (ds > (1ull << 24)) return bail(PIRGPU_INVALID_ARGUMENT, "dimension sum too large");
    c->dim_sum = (uint32_t)ds;
    c->stride[c->d] = 1;
    for (int l = (int)c->d - 1; l >= 0; --l) c->stride[l] = c->stride[l + 1] * c->dims[l];
    c->sb = p->shard_begin;
    c->se = p->shard_end;
    if (c->sb == 0 && c->se == 0) c->se = c->dims[0];
    if (c->sb > c->se || c->se > c->dims[0]) return bail(PIRGPU_INVALID_ARGUMENT, "invalid shard range");
    {
      const uint32_t kN = k * N;
      c->slot0 = p->slot_begin;
      c->nslots = (p->slot_begin == 0 && p->slot_end == 0) ? kN : p->slot_end - p->slot_begin;
      if (p->slot_end > kN || (p->slot_end && p->slot_end <= p->slot_begin) || c->slot0 % 16 || c->nslots % 16)
        return bail(PIRGPU_INVALID_ARGUMENT, "invalid slot range (multiples of 16 inside [0, k N))");
      c->slot_sharded = c->nslots != kN;
      if (c->slot_sharded && (c->d != 2 || c->sb != 0 || c->se != c->dims[0]))
        return bail(PIRGPU_INVALID_ARGUMENT, "a slot shard needs d = 2 and all rows (no row shard)");
    }
    c->pt_begin = std::min<uint64_t>((uint64_t)c->sb * c->stride[1], c->P);
    c->pt_end = std::min<uint64_t>((uint64_t)c->se * c->stride[1], c->P);
    const uint32_t bdef = hm::bits_per_coeff(p->plain_modulus);
    if (p->bits_per_coeff > bdef) return bail(PIRGPU_INVALID_ARGUMENT, "Bits per coefficient greater than max");
    c->bits = p->bits_per_coeff ? p->bits_per_coeff : bdef;
    c->device = p->device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
      return bail(PIRGPU_INTERNAL, "no HIP device available (libpirgpu has no CPU fallback)");
    if (c->device < 0 || c->device >= ndev) return bail(PIRGPU_INVALID_ARGUMENT, "invalid device ordinal");
    c->use_device();
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->ops = ntt_ops_for(N);
    if (!c->ops) return bail(PIRGPU_INVALID_ARGUMENT, "poly_modulus_degree must be 2048, 4096, 8192 or 16384");
    build_tables(c);
    HIP_TRY(c->ops->configure(c->mode));
    c->reply_cts = 1;
    for (uint32_t l = 1; l < c->d; ++l) c->reply_cts *= c->E;
    const uint64_t shard_pts = c->pt_end - c->pt_begin;
    // rows are padded with zero plaintexts to full length so the scan kernels are branch-free
    const uint64_t cols_last = c->dims[c->d - 1];
    const uint64_t padded = c->d == 1 ? shard_pts : ceil_div(shard_pts, cols_last) * cols_last;
    c->d_db = c->dalloc<uint64_t>(padded * k * N);
    if (padded > shard_pts)
      HIP_TRY(hipMemset(c->d_db + shard_pts * k * N, 0, (padded - shard_pts) * k * N * 8));
    c->loaded.assign(shard_pts, 0);
  } catch (const Fail& e) {
    return bail(e.code, e.msg);
  } catch (const std::exception& e) {
    return bail(PIRGPU_INTERNAL, e.what());
  }
  *out = c;
  return PIRGPU_OK;
}

void pirgpu_destroy(pirgpu_ctx* c) {
  if (!c) return;
  pirgpu_wire_forget(c);
  if (c->stream) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (BatchLane& ln : c->lanes)
      if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    for (size_t i = 1; i < c->workers.size(); ++i)
      if (c->workers[i].stream) (void)hipStreamSynchronize(c->workers[i].stream);
  }
  for (KeySet& ks : c->keysets)
    for (auto& kv : ks.keys) (void)hipFree(kv.second);
  for (uint64_t* p : c->key_pool) (void)hipFree(p);
  for (void* p : c->allocs) (void)hipFree(p);
  for (BatchSet& b : c->sets) {
    if (b.h_query) (void)hipHostFree(b.h_query);
    if (b.h_reply) (void)hipHostFree(b.h_reply);
    for (hipEvent_t e : b.dl_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : b.rd_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : b.st_events) (void)hipEventDestroy(e);
  }
  for (auto& e : c->ev) (void)hipEventDestroy(e);
  for (auto& e : c->bscan_ev) (void)hipEventDestroy(e);
  for (Worker& w : c->workers) {
    if (w.ev_expanded) (void)hipEventDestroy(w.ev_expanded);
    if (w.ev_scanned) (void)hipEventDestroy(w.ev_scanned);
    if (w.ev_done) (void)hipEventDestroy(w.ev_done);
    if (w.ev_join) (void)hipEventDestroy(w.ev_join);
  }
  for (BatchLane& ln : c->lanes) {
    if (ln.ev_scanned) (void)hipEventDestroy(ln.ev_scanned);
    if (ln.ev_join) (void)hipEventDestroy(ln.ev_join);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
  }
  if (c->head_stream) {
    (void)hipStreamSynchronize(c->head_stream);
    (void)hipStreamDestroy(c->head_stream);
  }
  if (c->ev_head_tail) (void)hipEventDestroy(c->ev_head_tail);
  for (BatchLane& ln : c->lanes)
    for (HeadSlot& hs : ln.head) {
      if (hs.ev_ready) (void)hipEventDestroy(hs.ev_ready);
      if (hs.ev_free) (void)hipEventDestroy(hs.ev_free);
    }
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
  }
  for (hipEvent_t e : c->ev_fetch)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_after) (void)hipEventDestroy(c->ev_after);
  if (c->ev_main_join) (void)hipEventDestroy(c->ev_main_join);
  for (size_t i = 1; i < c->workers.size(); ++i)
    if (c->workers[i].stream) (void)hipStreamDestroy(c->workers[i].stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* pirgpu_last_error(const pirgpu_ctx* c) {
  if (!c) return "null context";
  if (t_last_error_ctx == c && t_last_error_gen == c->gen)
    return t_last_error.c_str();  // this thread's own last failure on this context
  return c->err.c_str();
}

void pirgpu_set_error(pirgpu_ctx* c, const char* message) { (void)fail(c, 0, message ? message : ""); }

void pirgpu_request_lock(pirgpu_ctx* c) {
  if (c) c->mu.lock();
}
void pirgpu_request_unlock(pirgpu_ctx* c) {
  if (c) c->mu.unlock();
}

int pirgpu_set_transparent_policy(pirgpu_ctx* c, int allow) {
  return guarded(c, [&]() -> int {
    c->allow_transparent = allow != 0;
    return PIRGPU_OK;
  });
}

// Options the library understands (upper case in the environment: PIRGPU_<NAME>); `early` ones shape the workspace
// and must be set before the context is first used.
static const struct { const char* name; bool early; } kOptions[] = {
    {"UPPER_BLOCKS", true}, {"UPPER_BLOCKS_BATCH", true}, {"SCAN_MFMA_WGS_BATCH", false}, {"SCAN_LIMB", true},
    {"SCAN_MQ_SINGLE", true}, {"SCAN_ROWS", true}, {"SCAN_BLOCK", true}, {"SCAN_NSPLIT", true},
    {"FUSE_LAST", true}, {"FUSE_MAC_COMBINE", true}, {"LAST_NTT", true}, {"C0_NTT", true}, {"SEL_F64", true}, {"TREE40", true},
    {"SPLIT_UPPER", true}, {"SPLIT_UPPER_MB", true}, {"PACK40", true}, {"PACK_BYTES", true}, {"TREE40_WIDE", true},
    {"SCAN_MFMA", true}, {"SCAN_MFMA_WIDE", true}, {"SCAN_MFMA_TOP4", true}, {"SCAN_MFMA_NQ", true}, {"SCAN_MFMA_SINGLE", true},
    {"HEAD_LEVELS", true}, {"HEAD_MODE", true}, {"SCAN_F64_FOLD", true}, {"SCAN_F64_FOLD_BATCH", true}, {"LOOP_TRANSFORMS", true},
    {"SLOTS_SCAN_WGS", false}, {"SLOTS_GATHER_NTT", false}, {"SLOTS_SCAN_BLK_MAJOR", false},
};

int pirgpu_set_option(pirgpu_ctx* c, const char* name, int64_t value) {
  return guarded(c, [&]() -> int {
    if (!name) return fail(c, PIRGPU_INVALID_ARGUMENT, "null option name");
    std::string up(name);
    for (char& ch : up) ch = (char)toupper((unsigned char)ch);
    for (const auto& o : kOptions) {
      if (up != o.name) continue;
      if (o.early && c->ws_ready)
        return fail(c, PIRGPU_FAILED_PRECONDITION, "option " + up + " shapes the workspace: set it before the context is first used");
      c->opts[up] = value;
      if (up == "SCAN_MFMA_WGS_BATCH") c->scan_wgs_batch = (uint32_t)std::max<int64_t>(0, value);
      return PIRGPU_OK;
    }
    return fail(c, PIRGPU_INVALID_ARGUMENT, "unknown option " + up);
  });
}

int pirgpu_get_option(pirgpu_ctx* c, const char* name, int64_t* value) {
  return guarded(c, [&]() -> int {
    if (!name || !value) return fail(c, PIRGPU_INVALID_ARGUMENT, "null argument");
    std::string up(name);
    for (char& ch : up) ch = (char)toupper((unsigned char)ch);
    for (const auto& o : kOptions)
      if (up == o.name) {
        bool present = false;
        *value = option(c, o.name, -1, &present);   // -1: not set anywhere, the built-in default applies
        return PIRGPU_OK;
      }
    return fail(c, PIRGPU_INVALID_ARGUMENT, "unknown option " + up);
  });
}

uint64_t pirgpu_zero_plaintexts(const pirgpu_ctx* c) { return c ? c->zero_pts : 0; }

int pirgpu_set_remote_zero_plaintexts(pirgpu_ctx* c, uint64_t n) {
  return guarded(c, [&]() -> int {
    c->remote_zero_pts = n;
    return PIRGPU_OK;
  });
}

int pirgpu_check_ready(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    if (c->n_loaded != c->pt_end - c->pt_begin) return fail(c, PIRGPU_FAILED_PRECONDITION, "database not fully loaded");
    check_transparent(c);
    return PIRGPU_OK;
  });
}

int pirgpu_get_params(const pirgpu_ctx* c, pirgpu_params* out) {
  if (!c || !out) return PIRGPU_INVALID_ARGUMENT;
  *out = c->prm;
  out->shard_begin = c->sb;
  out->shard_end = c->se;
  return PIRGPU_OK;
}

uint64_t pirgpu_db_size(const pirgpu_ctx* c) { return c ? c->n_loaded : 0; }
uint64_t pirgpu_reply_ct_count(const pirgpu_ctx* c) { return c ? c->reply_cts : 0; }
uint32_t pirgpu_expansion_ratio(const pirgpu_ctx* c) { return c ? c->er : 0; }
uint64_t pirgpu_scan_bytes(const pirgpu_ctx* cc) {
  pirgpu_ctx* c = const_cast<pirgpu_ctx*>(cc);
  if (!c) return 0;
  try {
    c->use_device();
    ensure_workspace(c);
  } catch (...) {
    return 0;
  }
  // bytes a single-query pass over the database must read: the operand-layout copy when that pass is the MFMA scan
  return c->mfma_on && c->mfma_single ? (uint64_t)dbp_bytes(c) : (c->pt_end - c->pt_begin) * c->k * c->N * 8;
}

int pirgpu_ntt_mode(const pirgpu_ctx* c) { return c ? c->mode : -1; }

int pirgpu_scan_info(pirgpu_ctx* c, uint32_t info[8]) {
  return guarded(c, [&]() -> int {
    if (!info) return fail(c, PIRGPU_INVALID_ARGUMENT, "null info");
    ensure_workspace(c);
    info[0] = c->mfma_on ? 1 : 0;
    info[1] = c->mfma_on ? c->mg.L : 0;
    info[2] = c->mfma_on ? c->mg.nchunks : c->scan_nsplit;
    info[3] = c->mfma_on ? c->mg.KS : 0;
    info[4] = c->mfma_on ? c->mfma_nq : (mq_usable(c) ? std::min<uint32_t>(c->mq_nq, kMaxScanQueries) : 1);
    info[5] = c->scan_rows;
    info[6] = c->scan_cols;
    info[7] = (c->mfma_on && c->mfma_single ? 1u : 0u) | (c->mfma_on && c->mg.top4 ? 2u : 0u);
    return PIRGPU_OK;
  });
}

int pirgpu_db_load_items(pirgpu_ctx* c, const uint8_t* items, uint64_t num_items, uint32_t bytes_per_item) {
  return guarded(c, [&]() -> int {
    const pirgpu_params& p = c->prm;
    if (num_items != p.num_items)  // reference database.cpp:85-90
      return fail(c, PIRGPU_INVALID_ARGUMENT,
                  "Database size " + std::to_string(num_items) + " does not match params value " +
                      std::to_string(p.num_items));
    if (bytes_per_item != p.bytes_per_item || p.items_per_plaintext == 0 || (!items && num_items))
      return fail(c, PIRGPU_INVALID_ARGUMENT, "item size does not match parameters");
    const uint64_t ipp = p.items_per_plaintext;
    const uint64_t bytes_per_pt = ipp * bytes_per_item;
    if (c->staging_released)
      return fail(c, PIRGPU_FAILED_PRECONDITION, "database staging was released by pirgpu_db_finalize; it cannot be reloaded");
    c->packed_valid = false;
    // StringEncoder::calc_num_coeff (reference string_encoder.cpp:88-95)
    if ((uint64_t)std::ceil((double)(bytes_per_pt * 8) / c->bits) > c->N)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "Number of coefficients needed greater than poly modulus degree");
    if (ceil_div(num_items, ipp) > c->P) return fail(c, PIRGPU_INVALID_ARGUMENT, "more items than plaintexts");
    const uint64_t total_bytes = num_items * bytes_per_item;
    const uint64_t chunk_pts = std::max<uint64_t>(1, (64ull << 20) / std::max<uint64_t>(bytes_per_pt, 1));
    uint8_t* d_bytes = nullptr;
    HIP_TRY(hipMalloc((void**)&d_bytes, chunk_pts * bytes_per_pt));
    try {
      for (uint64_t pt = c->pt_begin; pt < c->pt_end; pt += chunk_pts) {
        const uint64_t n = std::min<uint64_t>(chunk_pts, c->pt_end - pt);
        const uint64_t b0 = std::min<uint64_t>(pt * bytes_per_pt, total_bytes);
        const uint64_t b1 = std::min<uint64_t>((pt + n) * bytes_per_pt, total_bytes);
        if (b1 > b0) HIP_TRY(hipMemcpyAsync(d_bytes, items + b0, b1 - b0, hipMemcpyHostToDevice, c->stream));
        for (uint64_t i = 0; i < n; ++i) {  // all-zero plaintexts (host scan while the copy is in flight)
          const uint64_t p0 = std::min<uint64_t>((pt + i) * bytes_per_pt, total_bytes);
          const uint64_t p1 = std::min<uint64_t>((pt + i + 1) * bytes_per_pt, total_bytes);
          note_plaintext(c, pt - c->pt_begin + i, all_zero_bytes(items + p0, p1 - p0));
        }
        HIP_TRY(c->ops->db_encode(c->stream, c->mode, c->dp, c->k, nullptr, d_bytes, bytes_per_pt, b1 - b0, c->bits,
                                  n, c->d_db + (pt - c->pt_begin) * c->k * c->N));
        HIP_TRY(hipStreamSynchronize(c->stream));
      }
    } catch (...) {
      (void)hipFree(d_bytes);
      throw;
    }
    HIP_TRY(hipFree(d_bytes));
    return PIRGPU_OK;
  });
}

int pirgpu_db_load_coeffs(pirgpu_ctx* c, uint64_t first_pt, uint64_t n_pt, const uint64_t* coeffs) {
  return guarded(c, [&]() -> int {
    if (first_pt + n_pt > c->P || (!coeffs && n_pt)) return fail(c, PIRGPU_INVALID_ARGUMENT, "plaintext range out of bounds");
    const uint64_t lo = std::max(first_pt, c->pt_begin), hi = std::min(first_pt + n_pt, c->pt_end);
    if (lo >= hi) return PIRGPU_OK;
    if (c->staging_released)
      return fail(c, PIRGPU_FAILED_PRECONDITION, "database staging was released by pirgpu_db_finalize; it cannot be reloaded");
    c->packed_valid = false;
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / (c->N * 8));
    uint64_t* d_coeffs = nullptr;
    HIP_TRY(hipMalloc((void**)&d_coeffs, chunk * c->N * 8));
    try {
      for (uint64_t pt = lo; pt < hi; pt += chunk) {
        const uint64_t n = std::min<uint64_t>(chunk, hi - pt);
        HIP_TRY(hipMemcpyAsync(d_coeffs, coeffs + (pt - first_pt) * c->N, n * c->N * 8, hipMemcpyHostToDevice,
                               c->stream));
        for (uint64_t i = 0; i < n; ++i)
          note_plaintext(c, pt - c->pt_begin + i,
                         all_zero_bytes(reinterpret_cast<const uint8_t*>(coeffs + (pt - first_pt + i) * c->N),
                                        (size_t)c->N * 8));
        HIP_TRY(c->ops->db_encode(c->stream, c->mode, c->dp, c->k, d_coeffs, nullptr, 0, 0, c->bits, n,
                                  c->d_db + (pt - c->pt_begin) * c->k * c->N));
        HIP_TRY(hipStreamSynchronize(c->stream));
      }
    } catch (...) {
      (void)hipFree(d_coeffs);
      throw;
    }
    HIP_TRY(hipFree(d_coeffs));
    return PIRGPU_OK;
  });
}

int pirgpu_db_finalize(pirgpu_ctx* c, int release_staging) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    if (c->n_loaded != c->pt_end - c->pt_begin) return fail(c, PIRGPU_FAILED_PRECONDITION, "database not fully loaded");
    ensure_packed(c);
    if (release_staging && c->mfma_on && c->d_db) {
      sync_batch_streams(c);
      HIP_TRY(hipStreamSynchronize(c->stream));
      auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)c->d_db);
      if (it != c->allocs.end()) c->allocs.erase(it);
      HIP_TRY(hipFree(c->d_db));
      c->d_db = nullptr;
      c->staging_released = true;
      c->mfma_single = true;  // the 64-bit kernels read the staging copy
    }
    return PIRGPU_OK;
  });
}

int pirgpu_db_read_plaintext(pirgpu_ctx* c, uint64_t pt_index, uint64_t* out) {
  return guarded(c, [&]() -> int {
    if (pt_index < c->pt_begin || pt_index >= c->pt_end || !out)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "plaintext index outside this shard");
    uint64_t* stage = nullptr;
    HIP_TRY(hipMalloc((void**)&stage, (size_t)2 * c->k * c->N * 8));
    try {
      const uint64_t* src = c->d_db ? c->d_db + (pt_index - c->pt_begin) * c->k * c->N : nullptr;
      if (!src) {  // staging released: gather the digits of this plaintext from the operand layout
        refuse_slot_shard(c);   // ... which a slot shard holds only 1 / G of
        const uint64_t local = pt_index - c->pt_begin;
        uint64_t* tmp = stage + (size_t)c->k * c->N;
        HIP_TRY(launch_db_unpack(c->stream, c->dp, c->mg, c->d_dbp, tmp, (uint32_t)(local / c->scan_cols),
                                 (uint32_t)(local % c->scan_cols), c->k * c->N));
        src = tmp;
      }
      HIP_TRY(launch_ntt_reorder(c->stream, c->N, src, stage, c->k, false, false));
      HIP_TRY(hipMemcpyAsync(out, stage, (size_t)c->k * c->N * 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (...) {
      (void)hipFree(stage);
      throw;
    }
    HIP_TRY(hipFree(stage));
    return PIRGPU_OK;
  });
}

// ======================================================================================================================
// [6] C ABI: Galois keys and per-client key sets
// ======================================================================================================================
// Uploads one Galois key into a key set slot (SEAL's NTT order at the boundary, device order in HBM).  A new client's
// first request pays for 12 of these (N = 4096): the staging buffer is allocated once per context and the key buffers
// of emptied sets are recycled, so the steady state of a server whose clients come and go allocates nothing.
static void upload_key(pirgpu_ctx* c, uint32_t slot, uint32_t g, const uint64_t* key, bool wait = true) {
  if (slot >= c->keysets.size()) throw Fail{PIRGPU_INVALID_ARGUMENT, "key set slot out of range"};
  if (!key || !(g & 1) || g >= 2 * c->N) throw Fail{PIRGPU_INVALID_ARGUMENT, "invalid Galois element"};
  const size_t words = (size_t)c->k * 2 * (c->k + 1) * c->N;
  auto& keys = c->keysets[slot].keys;
  uint64_t* dev = nullptr;
  auto it = keys.find(g);
  if (it != keys.end()) {
    // overwriting a key that queued work may still read: wait for it (never happens on the request path, where a
    // slot is emptied -- after a drain -- before it is refilled)
    HIP_TRY(hipStreamSynchronize(c->stream));
    sync_batch_streams(c);
    dev = it->second;
  } else {
    if (!c->key_pool.empty()) {
      dev = c->key_pool.back();
      c->key_pool.pop_back();
    } else {
      HIP_TRY(hipMalloc((void**)&dev, words * 8));
    }
    keys[g] = dev;
  }
  if (!c->d_key_stage) c->d_key_stage = c->dalloc<uint64_t>(words);
  // stream order on the main stream covers the reuse of the staging buffer by the next key
  HIP_TRY(hipMemcpyAsync(c->d_key_stage, key, words * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(launch_ntt_reorder(c->stream, c->N, c->d_key_stage, dev, (uint64_t)c->k * 2 * (c->k + 1), true,
                             c->mode != kNttInt));
  // the lanes / workers that will read the key do not follow this stream: the caller (or this call) waits once
  if (wait) HIP_TRY(hipStreamSynchronize(c->stream));
  ++c->key_uploads;
}

// Empties a slot; waits for everything in flight first (queued kernels may still read its keys).
static void clear_keyset(pirgpu_ctx* c, uint32_t slot) {
  KeySet& ks = c->keysets[slot];
  if (!ks.keys.empty()) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    sync_batch_streams(c);
    for (auto& kv : ks.keys) c->key_pool.push_back(kv.second);   // recycled by the next upload
    ks.keys.clear();
  }
  ks.blob.clear();
  ks.fingerprint = 0;
  if (slot) ks.gen = (ks.gen + 1) & ((1u << (32 - kSlotBits)) - 1);   // handles of the previous tenant go stale
}

static uint32_t slot_handle(const pirgpu_ctx* c, uint32_t index) { return index ? (c->keysets[index].gen << kSlotBits) | index : 0; }

// Handle -> slot index; FailedPrecondition for a handle whose set was evicted or released since it was handed out.
static uint32_t resolve_slot(pirgpu_ctx* c, uint32_t handle) {
  const uint32_t index = handle & kSlotMask;
  if (index >= c->keysets.size()) throw Fail{PIRGPU_INVALID_ARGUMENT, "key set slot out of range"};
  if (index && c->keysets[index].gen != handle >> kSlotBits)
    throw Fail{PIRGPU_FAILED_PRECONDITION, "stale key set handle: the set was evicted or released (claim it again)"};
  if (!index && handle) throw Fail{PIRGPU_INVALID_ARGUMENT, "key set slot out of range"};
  return index;
}

// cheap candidate filter before the byte-for-byte compare: length + 64 words sampled across the blob
static uint64_t blob_fingerprint(const uint8_t* blob, size_t len) {
  uint64_t h = 0xcbf29ce484222325ull ^ len;
  const size_t words = len / 8, step = std::max<size_t>(1, words / 64);
  for (size_t i = 0; i < words; i += step) {
    uint64_t w;
    memcpy(&w, blob + i * 8, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  return h | 1;   // never 0 (0 = no blob)
}

int pirgpu_set_galois_key(pirgpu_ctx* c, uint32_t g, const uint64_t* key) {
  return guarded(c, [&]() -> int {
    c->keysets[0].blob.clear();
    c->keysets[0].fingerprint = 0;
    upload_key(c, 0, g, key);
    return PIRGPU_OK;
  });
}

int pirgpu_clear_galois_keys(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    clear_keyset(c, 0);
    return PIRGPU_OK;
  });
}

int pirgpu_set_keyset_capacity(pirgpu_ctx* c, uint32_t slots) {
  return guarded(c, [&]() -> int {
    if (slots < 1 || slots > 1024) return fail(c, PIRGPU_INVALID_ARGUMENT, "key set capacity must be in [1, 1024]");
    for (size_t i = (size_t)slots + 1; i < c->keysets.size(); ++i)
      if (c->keysets[i].pins) return fail(c, PIRGPU_FAILED_PRECONDITION, "a key set beyond the new capacity is in use by requests in flight");
    while (c->keysets.size() > (size_t)slots + 1) {   // shrinking drops the highest slots
      clear_keyset(c, (uint32_t)c->keysets.size() - 1);
      c->keysets.pop_back();
    }
    c->keyset_cap = slots;   // (a staged batch / selection that named a dropped slot fails its next run: stale handle)
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_lookup(pirgpu_ctx* c, const uint8_t* blob, size_t len, int verify, uint32_t* slot) {
  return guarded(c, [&]() -> int {
    if (!slot || (!blob && len)) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    *slot = 0;
    if (!len) return PIRGPU_OK;
    const uint64_t fp = blob_fingerprint(blob, len);
    for (uint32_t i = 1; i < c->keysets.size(); ++i) {
      KeySet& ks = c->keysets[i];
      if (ks.fingerprint == fp && ks.blob.size() == len && (!verify || memcmp(ks.blob.data(), blob, len) == 0)) {
        ks.last_use = ++c->keyset_clock;
        *slot = slot_handle(c, i);
        break;
      }
    }
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_verify(pirgpu_ctx* c, uint32_t slot, const uint8_t* blob, size_t len) {
  if (!c || !blob) return 0;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const uint32_t index = slot & kSlotMask;
  if (index == 0 || index >= c->keysets.size() || c->keysets[index].gen != slot >> kSlotBits) return 0;
  const KeySet& ks = c->keysets[index];
  return ks.blob.size() == len && memcmp(ks.blob.data(), blob, len) == 0 ? 1 : 0;
}

size_t pirgpu_keyset_blob(pirgpu_ctx* c, uint32_t slot, const uint8_t** blob) {
  if (blob) *blob = nullptr;
  if (!c || !blob) return 0;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const uint32_t index = slot & kSlotMask;
  if (index == 0 || index >= c->keysets.size() || c->keysets[index].gen != slot >> kSlotBits) return 0;
  *blob = c->keysets[index].blob.data();
  return c->keysets[index].blob.size();
}

int pirgpu_keyset_claim(pirgpu_ctx* c, const uint8_t* blob, size_t len, uint32_t* slot) {
  return guarded(c, [&]() -> int {
    if (!slot || (!blob && len)) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    uint32_t pick = 0;
    for (uint32_t i = 1; i < c->keysets.size() && !pick; ++i)
      if (c->keysets[i].keys.empty() && c->keysets[i].blob.empty() && !c->keysets[i].pins) pick = i;
    if (!pick && c->keysets.size() < (size_t)c->keyset_cap + 1) {
      c->keysets.emplace_back();
      pick = (uint32_t)c->keysets.size() - 1;
    }
    if (!pick) {   // least recently used -- never a set that requests in flight are pinned to.  (A staged batch or the
                   // single-query selection that still names the evicted set notices at its next run: its handle is stale.)
      uint64_t best = UINT64_MAX;
      for (uint32_t i = 1; i < c->keysets.size(); ++i)
        if (c->keysets[i].last_use < best && !c->keysets[i].pins) {
          best = c->keysets[i].last_use;
          pick = i;
        }
      if (!pick)
        return fail(c, PIRGPU_FAILED_PRECONDITION,
                    "every key set slot is in use by the requests being processed (pirgpu_set_keyset_capacity)");
      clear_keyset(c, pick);
      ++c->keyset_evictions;
    }
    KeySet& ks = c->keysets[pick];
    if (len) {
      ks.blob.assign(blob, blob + len);
      ks.fingerprint = blob_fingerprint(blob, len);
    }
    ks.last_use = ++c->keyset_clock;
    *slot = slot_handle(c, pick);
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_release(pirgpu_ctx* c, uint32_t slot) {
  return guarded(c, [&]() -> int {
    const uint32_t index = resolve_slot(c, slot);
    if (index == 0) return fail(c, PIRGPU_INVALID_ARGUMENT, "key set slot out of range");
    if (c->keysets[index].pins) return fail(c, PIRGPU_FAILED_PRECONDITION, "the key set is in use by requests in flight");
    clear_keyset(c, index);
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_set_key(pirgpu_ctx* c, uint32_t slot, uint32_t g, const uint64_t* key) {
  return guarded(c, [&]() -> int {
    upload_key(c, resolve_slot(c, slot), g, key);
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_set_keys(pirgpu_ctx* c, uint32_t slot, uint32_t n, const uint32_t* galois_elts, const uint64_t* const* keys) {
  return guarded(c, [&]() -> int {
    if (n && (!galois_elts || !keys)) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    const uint32_t index = resolve_slot(c, slot);
    // one wait for the whole set (a new client's 12 keys: 12 stream synchronisations used to be a quarter of its
    // first request); the staging buffer's reuse from key to key is covered by stream order
    for (uint32_t i = 0; i < n; ++i) upload_key(c, index, galois_elts[i], keys[i], false);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_pin(pirgpu_ctx* c, uint32_t slot) {
  return guarded(c, [&]() -> int {
    const uint32_t index = resolve_slot(c, slot);
    if (index) ++c->keysets[index].pins;
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_unpin(pirgpu_ctx* c, uint32_t slot) {
  if (!c) return PIRGPU_INVALID_ARGUMENT;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  const uint32_t index = slot & kSlotMask;   // a pinned set cannot have changed generation
  if (index && index < c->keysets.size() && c->keysets[index].pins) --c->keysets[index].pins;
  return PIRGPU_OK;
}

// The single-query selection as a handle: the generation it was SELECTED with (not the slot's current one -- a selection
// that went stale must stay stale), 0 for the default set.
uint32_t pirgpu_current_keyset(pirgpu_ctx* c) {
  if (!c) return 0;
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  return c->cur_keyset ? (c->cur_keyset_gen << kSlotBits) | (c->cur_keyset & kSlotMask) : 0;
}

// wire layer: the selection saved and put back exactly as it was (index, generation), valid or not
void pirgpu_keyset_selection_get(pirgpu_ctx* c, uint32_t sel[2]) {
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  sel[0] = c->cur_keyset;
  sel[1] = c->cur_keyset_gen;
}
void pirgpu_keyset_selection_set(pirgpu_ctx* c, const uint32_t sel[2]) {
  std::lock_guard<std::recursive_mutex> lock(c->mu);
  c->cur_keyset = sel[0];
  c->cur_keyset_gen = sel[1];
}

int pirgpu_query_use_keyset(pirgpu_ctx* c, uint32_t slot) {
  return guarded(c, [&]() -> int {
    const uint32_t index = resolve_slot(c, slot);
    c->cur_keyset = index;
    c->cur_keyset_gen = c->keysets[index].gen;
    c->keysets[index].last_use = ++c->keyset_clock;
    return PIRGPU_OK;
  });
}

int pirgpu_batch_set_keysets(pirgpu_ctx* c, const uint32_t* slots, uint32_t count) {
  return guarded(c, [&]() -> int {
    if (!slots || count != c->bs().staged_count) return fail(c, PIRGPU_INVALID_ARGUMENT, "one key set slot per staged query");
    std::vector<uint32_t> idx(count);
    for (uint32_t i = 0; i < count; ++i) idx[i] = resolve_slot(c, slots[i]);
    ++c->keyset_clock;
    std::vector<uint32_t> gens(count);
    for (uint32_t i = 0; i < count; ++i) {
      c->keysets[idx[i]].last_use = c->keyset_clock;
      gens[i] = c->keysets[idx[i]].gen;
    }
    c->bs().batch_keysets = std::move(idx);
    c->bs().batch_keyset_gens = std::move(gens);
    return PIRGPU_OK;
  });
}

int pirgpu_keyset_stats(pirgpu_ctx* c, uint64_t stats[4]) {
  return guarded(c, [&]() -> int {
    if (!stats) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    uint64_t resident = 0;
    for (uint32_t i = 1; i < c->keysets.size(); ++i) resident += c->keysets[i].keys.empty() ? 0 : 1;
    stats[0] = resident;
    stats[1] = c->key_uploads;
    stats[2] = c->keyset_evictions;
    stats[3] = c->keyset_cap;
    return PIRGPU_OK;
  });
}

int pirgpu_batch_select(pirgpu_ctx* c, uint32_t which) {
  if (!c || which >= (uint32_t)kBatchSets) return PIRGPU_INVALID_ARGUMENT;
  t_batch_set = (int)which;
  return PIRGPU_OK;
}

// Pinned host staging owned by the context (the wire layer parses queries straight into it and serialises replies
// straight out of it: H2D / D2H then run at link speed and asynchronously instead of through pageable bounce buffers).
static uint64_t* host_buffer(pirgpu_ctx* c, uint64_t*& buf, size_t& cap, size_t words) {
  if (words > cap) {
    if (buf) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipHostFree(buf));
      buf = nullptr;
      cap = 0;
    }
    HIP_TRY(hipHostMalloc((void**)&buf, words * 8, hipHostMallocDefault));
    cap = words;
  }
  return buf;
}

uint64_t* pirgpu_host_query_buffer(pirgpu_ctx* c, uint32_t count) {
  uint64_t* out = nullptr;
  (void)guarded(c, [&]() -> int {
    out = host_buffer(c, c->bs().h_query, c->bs().h_query_words, (size_t)std::max<uint32_t>(count, 1) * (c->dim_sum / c->N + 1) * c->ctw);
    return PIRGPU_OK;
  });
  return out;
}

uint64_t* pirgpu_host_reply_buffer(pirgpu_ctx* c, uint32_t count) {
  uint64_t* out = nullptr;
  (void)guarded(c, [&]() -> int {
    out = host_buffer(c, c->bs().h_reply, c->bs().h_reply_words, (size_t)std::max<uint32_t>(count, 1) * c->reply_cts * c->ctw);
    return PIRGPU_OK;
  });
  return out;
}

// ======================================================================================================================
// [7] C ABI: single query (stage / run / fetch), stream ordering hooks (join / fork), test hooks (expand, substitute, multiply)
// ======================================================================================================================
static int query_stage_impl(pirgpu_ctx* c, const uint64_t* query, uint32_t nq, bool wait) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    Worker& w = c->workers[0];
    if (!query || nq != c->dim_sum / c->N + 1)  // reference server.cpp:154-158
      return fail(c, PIRGPU_INVALID_ARGUMENT,
                  "Number of ciphertexts doesn't match number of items for oblivious expansion.");
    HIP_TRY(hipStreamWaitEvent(c->stream, w.ev_done, 0));
    HIP_TRY(hipMemcpyAsync(w.d_query, query, (size_t)nq * c->ctw * 8, hipMemcpyHostToDevice, c->stream));
    if (wait) HIP_TRY(hipStreamSynchronize(c->stream));   // the caller may reuse `query` at once
    w.staged_nq = nq;
    return PIRGPU_OK;
  });
}

int pirgpu_query_stage(pirgpu_ctx* c, const uint64_t* query, uint32_t nq) { return query_stage_impl(c, query, nq, true); }

int pirgpu_query_stage_async(pirgpu_ctx* c, const uint64_t* pinned_query, uint32_t nq) {
  return query_stage_impl(c, pinned_query, nq, false);
}

int pirgpu_query_run(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    run_staged(c, c->workers[0], true);
    return PIRGPU_OK;
  });
}

int pirgpu_sync(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (BatchLane& ln : c->lanes)
      if (ln.stream) HIP_TRY(hipStreamSynchronize(ln.stream));
    for (Worker& w : c->workers)
      if (w.stream) HIP_TRY(hipStreamSynchronize(w.stream));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
    return PIRGPU_OK;
  });
}

// ---- device-side ordering with the caller's streams (pipelined multi-GPU step) ----
//
// The library queues its work on its own HIP streams (one main stream, the batch lanes, the workers).  A caller that
// runs collectives on ITS streams orders them against that work without stopping the host: pirgpu_join makes the main
// stream wait -- on the device -- for everything queued on the lanes and workers so far, so the main stream (handed
// out by pirgpu_stream_handle, e.g. for torch.cuda.ExternalStream) becomes the completion point an event can be
// recorded on; pirgpu_fork makes the lanes and workers wait for everything the main stream has been made to wait for
// (e.g. an event of the caller's collective), so work queued afterwards starts behind it.

void* pirgpu_stream_handle(pirgpu_ctx* c) { return c ? (void*)c->stream : nullptr; }

int pirgpu_join(pirgpu_ctx* c) { return pirgpu_join_stream(c, nullptr); }

int pirgpu_device_synchronize(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    c->use_device();
    HIP_TRY(hipDeviceSynchronize());
    return PIRGPU_OK;
  });
}

int pirgpu_join_stream(pirgpu_ctx* c, void* stream) {
  return guarded(c, [&]() -> int {
    hipStream_t target = stream ? (hipStream_t)stream : c->stream;
    for (BatchLane& ln : c->lanes) {
      if (!ln.stream) continue;
      if (!ln.ev_join) HIP_TRY(hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(ln.ev_join, ln.stream));
      HIP_TRY(hipStreamWaitEvent(target, ln.ev_join, 0));
    }
    for (Worker& w : c->workers) {
      if (!w.stream || w.stream == c->stream) continue;
      if (!w.ev_join) HIP_TRY(hipEventCreateWithFlags(&w.ev_join, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(w.ev_join, w.stream));
      HIP_TRY(hipStreamWaitEvent(target, w.ev_join, 0));
    }
    if (target != c->stream) {   // the main stream carries work too (worker 0, copies): the target follows it as well
      if (!c->ev_main_join) HIP_TRY(hipEventCreateWithFlags(&c->ev_main_join, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(c->ev_main_join, c->stream));
      HIP_TRY(hipStreamWaitEvent(target, c->ev_main_join, 0));
    }
    return PIRGPU_OK;
  });
}

int pirgpu_batch_set_reply_buffer(pirgpu_ctx* c, uint64_t* device_buf, uint64_t cap) {
  return guarded(c, [&]() -> int {
    if (c->in_batch) return fail(c, PIRGPU_FAILED_PRECONDITION, "a batch is being queued");
    if (device_buf && cap == 0) return fail(c, PIRGPU_INVALID_ARGUMENT, "empty reply buffer");
    c->bs().ext_reply = device_buf;
    c->bs().ext_reply_cts = device_buf ? cap : 0;
    c->bs().batch_valid = false;   // replies of an earlier batch live in the other buffer
    return PIRGPU_OK;
  });
}

int pirgpu_batch_set_host_replies(pirgpu_ctx* c, uint64_t* pinned_host, uint64_t cap) {
  return guarded(c, [&]() -> int {
    if (c->in_batch) return fail(c, PIRGPU_FAILED_PRECONDITION, "a batch is being queued");
    c->bs().host_reply = pinned_host;
    c->bs().host_reply_cts = pinned_host ? cap : 0;
    c->bs().host_reply_done = false;
    return PIRGPU_OK;
  });
}

int pirgpu_batch_next_host_replies(pirgpu_ctx* c, uint32_t* ready) {
  if (!c || !ready) return PIRGPU_INVALID_ARGUMENT;
  // the wait itself happens WITHOUT the context's lock held by this call (the caller may hold it recursively anyway)
  hipEvent_t ev = nullptr;
  uint32_t end = 0;
  int rc = guarded(c, [&]() -> int {
    if (!c->bs().host_reply_done) return fail(c, PIRGPU_FAILED_PRECONDITION, "the last batch did not download its replies group by group");
    if (c->bs().dl_next >= c->bs().dl_end.size()) {
      *ready = c->bs().dl_end.empty() ? 0 : c->bs().dl_end.back();
      return PIRGPU_OK;
    }
    ev = c->bs().dl_events[c->bs().dl_next];
    end = c->bs().dl_end[c->bs().dl_next];
    ++c->bs().dl_next;
    return PIRGPU_OK;
  });
  if (rc || !ev) return rc;
  if (hipEventSynchronize(ev) != hipSuccess) return fail(c, PIRGPU_INTERNAL, "waiting for a group's replies failed");
  *ready = end;
  return PIRGPU_OK;
}

int pirgpu_fork(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
    for (BatchLane& ln : c->lanes)
      if (ln.stream) HIP_TRY(hipStreamWaitEvent(ln.stream, c->ev_fork, 0));
    if (c->head_stream) HIP_TRY(hipStreamWaitEvent(c->head_stream, c->ev_fork, 0));
    for (Worker& w : c->workers)
      if (w.stream && w.stream != c->stream) HIP_TRY(hipStreamWaitEvent(w.stream, c->ev_fork, 0));
    return PIRGPU_OK;
  });
}

int pirgpu_query_fetch(pirgpu_ctx* c, uint64_t* reply, uint64_t cap, uint64_t* count) {
  return guarded(c, [&]() -> int {
    if (c->workers.empty()) return fail(c, PIRGPU_FAILED_PRECONDITION, "no query has been run");
    Worker& w = c->workers[0];
    if (!w.reply_valid) return fail(c, PIRGPU_FAILED_PRECONDITION, "no query has been run");
    if (!reply || cap < c->reply_cts) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    HIP_TRY(hipMemcpyAsync(reply, w.lvl[0], c->reply_cts * c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (count) *count = c->reply_cts;
    return PIRGPU_OK;
  });
}

// pirgpu_query_fetch in two halves: both downloads are queued at once, the caller serialises the first half of the
// reply while the second is still crossing PCIe (a megabyte of memcpy against 20 us of transfer: the tail of a lone
// request).  fetch_begin returns the number of ciphertexts in the first part; fetch_wait(part) blocks until that part has
// landed in `reply` (pinned host memory).
int pirgpu_query_fetch_begin(pirgpu_ctx* c, uint64_t* reply, uint64_t cap, uint64_t* count, uint64_t* first_part) {
  return guarded(c, [&]() -> int {
    if (c->workers.empty()) return fail(c, PIRGPU_FAILED_PRECONDITION, "no query has been run");
    Worker& w = c->workers[0];
    if (!w.reply_valid) return fail(c, PIRGPU_FAILED_PRECONDITION, "no query has been run");
    if (!reply || cap < c->reply_cts) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    for (hipEvent_t& e : c->ev_fetch)
      if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const uint64_t a = (c->reply_cts + 1) / 2;
    HIP_TRY(hipMemcpyAsync(reply, w.lvl[0], a * c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->ev_fetch[0], c->stream));
    if (c->reply_cts > a)
      HIP_TRY(hipMemcpyAsync(reply + a * c->ctw, w.lvl[0] + a * c->ctw, (c->reply_cts - a) * c->ctw * 8, hipMemcpyDeviceToHost,
                             c->stream));
    HIP_TRY(hipEventRecord(c->ev_fetch[1], c->stream));
    if (count) *count = c->reply_cts;
    if (first_part) *first_part = a;
    return PIRGPU_OK;
  });
}

int pirgpu_query_fetch_wait(pirgpu_ctx* c, int part) {
  if (!c || part < 0 || part > 1 || !c->ev_fetch[part]) return PIRGPU_INVALID_ARGUMENT;
  if (hipEventSynchronize(c->ev_fetch[part]) != hipSuccess) return fail(c, PIRGPU_INTERNAL, "waiting for the reply failed");
  return PIRGPU_OK;
}

int pirgpu_process_query(pirgpu_ctx* c, const uint64_t* query, uint32_t nq, uint64_t* reply, uint64_t cap,
                         uint64_t* count) {
  int rc = pirgpu_query_stage(c, query, nq);
  if (rc) return rc;
  rc = pirgpu_query_run(c);
  if (rc) return rc;
  return pirgpu_query_fetch(c, reply, cap, count);
}

uint64_t* pirgpu_reply_device_ptr(pirgpu_ctx* c) { return (c && c->ws_ready) ? c->workers[0].lvl[0] : nullptr; }

int pirgpu_reply_copy_to_device(pirgpu_ctx* c, uint64_t* dst, uint64_t cap) {
  return guarded(c, [&]() -> int {
    if (c->workers.empty()) return fail(c, PIRGPU_FAILED_PRECONDITION, "no query has been run");
    Worker& w = c->workers[0];
    if (!w.reply_valid) return fail(c, PIRGPU_FAILED_PRECONDITION, "no query has been run");
    if (!dst || cap < c->reply_cts) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    HIP_TRY(hipMemcpyAsync(dst, w.lvl[0], c->reply_cts * c->ctw * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_expand(pirgpu_ctx* c, const uint64_t* ct, uint32_t num_items, uint64_t* out) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    Worker& w = c->workers[0];
    if (!ct || (!out && num_items)) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    if (num_items > c->N)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "Cannot expand more items from a CT than poly modulus degree");
    // the shared workspace is sized for this context's dim_sum
    const uint64_t m_max = std::min<uint64_t>(c->N, hm::next_power_two(std::max<uint32_t>(c->dim_sum, 1)));
    if (hm::next_power_two(num_items) > m_max)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "num_items exceeds this context's expansion workspace");
    w.keyset = current_keyset(c);
    HIP_TRY(hipMemcpyAsync(w.res_b, ct, c->ctw * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_tree_convert(c->stream, c->dp, c->mode, w.res_b, w.res_a, c->ctw, true));
    uint64_t* res = expand_on_device(c, w, num_items);
    if (num_items) {  // tree element type -> canonical residues (in place), then out
      HIP_TRY(launch_tree_convert(c->stream, c->dp, c->mode, res, res, (uint64_t)num_items * c->ctw, false));
      HIP_TRY(hipMemcpyAsync(out, res, (size_t)num_items * c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_expand_multi(pirgpu_ctx* c, const uint64_t* cts, uint32_t num_cts, uint64_t total_items, uint64_t* out) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    Worker& w = c->workers[0];
    if (!cts || !out) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    if (num_cts != total_items / c->N + 1)  // reference server.cpp:154-158
      return fail(c, PIRGPU_INVALID_ARGUMENT,
                  "Number of ciphertexts doesn't match number of items for oblivious expansion.");
    const uint64_t m_max = std::min<uint64_t>(c->N, hm::next_power_two(std::max<uint32_t>(c->dim_sum, 1)));
    uint64_t remaining = total_items, produced = 0;
    w.keyset = current_keyset(c);
    for (uint32_t q = 0; q < num_cts && remaining; ++q) {
      uint32_t n = (uint32_t)std::min<uint64_t>(remaining, c->N);
      if (hm::next_power_two(n) > m_max)
        return fail(c, PIRGPU_INVALID_ARGUMENT, "total_items exceeds this context's expansion workspace");
      HIP_TRY(hipMemcpyAsync(w.res_b, cts + (size_t)q * c->ctw, c->ctw * 8, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(launch_tree_convert(c->stream, c->dp, c->mode, w.res_b, w.res_a, c->ctw, true));
      uint64_t* res = expand_on_device(c, w, n);
      HIP_TRY(launch_tree_convert(c->stream, c->dp, c->mode, res, res, (uint64_t)n * c->ctw, false));
      HIP_TRY(hipMemcpyAsync(out + produced * c->ctw, res, (size_t)n * c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      produced += n;
      remaining -= n;
    }
    return PIRGPU_OK;
  });
}

int pirgpu_substitute_power_x(pirgpu_ctx* c, uint64_t* ct, uint32_t power) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    Worker& w = c->workers[0];
    if (!ct) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    if (!(power & 1) || power >= 2 * c->N)  // SEAL: "Galois element is not valid" -> InternalError
      return fail(c, PIRGPU_INTERNAL, "Galois element is not valid");
    KeyPtrs key{};
    key.p[0] = find_key(c, current_keyset(c), power);
    key.B = 1;
    HIP_TRY(hipMemcpyAsync(w.res_b, ct, c->ctw * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_tree_convert(c->stream, c->dp, c->mode, w.res_b, w.res_a, c->ctw, true));
    HIP_TRY(c->ops->ks_digit(c->stream, c->mode, c->dp, c->k, w.res_a, power, 1, w.dig, c->pack40, nullptr, false, false));
    HIP_TRY(c->ops->ks_mac_intt(c->stream, c->mode, c->dp, c->k, w.dig, key, 1, w.prod, c->pack40, 0, c->k + 1));
    HIP_TRY(launch_ks_combine(c->stream, c->dp, c->mode, c->N, c->k, w.res_a, w.prod, galois_inverse(power, c->N), 1, 0,
                              false, /*hi_limit: unused without the expand step*/ 0, c->pack40, w.res_b, c->pack_bytes));
    HIP_TRY(launch_tree_convert(c->stream, c->dp, c->mode, w.res_b, w.res_b, c->ctw, false));
    HIP_TRY(hipMemcpyAsync(ct, w.res_b, c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_multiply_inverse_power_of_x(pirgpu_ctx* c, const uint64_t* ct, uint32_t kpow, uint64_t* out) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    Worker& w = c->workers[0];
    if (!ct || !out) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    const uint32_t twoN = 2 * c->N;
    const uint32_t index = (twoN - (kpow % twoN)) % twoN;  // reference server.cpp:87-88
    HIP_TRY(hipMemcpyAsync(w.res_a, ct, c->ctw * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_monomial_shift(c->stream, c->dp, c->N, c->k, w.res_a, index, 1, w.res_b));
    HIP_TRY(hipMemcpyAsync(out, w.res_b, c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_multiply(pirgpu_ctx* c, const uint64_t* sv, uint64_t sv_count, uint64_t* reply, uint64_t cap,
                    uint64_t* count) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    Worker& w = c->workers[0];
    if (sv_count != c->dim_sum)  // reference database.cpp:297-300
      return fail(c, PIRGPU_INVALID_ARGUMENT, "Selection vector size does not match dimensions");
    if (!sv || !reply || cap < c->reply_cts) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    // stage the coefficient-form selectors through res_a in workspace-sized pieces
    const uint64_t m_max = std::min<uint64_t>(c->N, hm::next_power_two(std::max<uint32_t>(c->dim_sum, 1)));
    for (uint64_t s = 0; s < sv_count; s += m_max) {
      const uint64_t n = std::min<uint64_t>(m_max, sv_count - s);
      HIP_TRY(hipMemcpyAsync(w.res_a, sv + s * c->ctw, n * c->ctw * 8, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c->ops->ct_ntt_fwd_oop(c->stream, c->mode, c->dp, c->k, w.res_a, w.sv_ntt + s * c->ctw, n, false));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->prof_cur = -1;
    w.sv_cur = nullptr;
    w.sv_rows = nullptr;
    multiply_on_device(c, w);
    HIP_TRY(hipMemcpyAsync(reply, w.lvl[0], c->reply_cts * c->ctw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (count) *count = c->reply_cts;
    return PIRGPU_OK;
  });
}

// ======================================================================================================================
// [8] BATCH PIPELINE (server.cpp:60-63): staging, lanes, grouped expansion, batch_run_mfma
// ======================================================================================================================
// ---- batch mode: `count` independent queries, spread round-robin over the workers ----

int pirgpu_set_concurrency(pirgpu_ctx* c, uint32_t n_workers) {
  return guarded(c, [&]() -> int {
    if (n_workers < 1 || n_workers > 32) return fail(c, PIRGPU_INVALID_ARGUMENT, "workers must be in [1, 32]");
    ensure_workspace(c);
    while (c->workers.size() < n_workers) {
      c->workers.emplace_back();
      alloc_worker(c, c->workers.back());
    }
    c->n_active = n_workers;
    return PIRGPU_OK;
  });
}

static void ensure_batch_capacity(pirgpu_ctx* c, uint32_t count);
// Where a batch's replies are written and read: the context's own buffer, or the caller's (pirgpu_batch_set_reply_buffer).
static inline uint64_t* reply_base(pirgpu_ctx* c) { return c->bs().ext_reply ? c->bs().ext_reply : c->bs().d_breply; }
static void check_reply_target(pirgpu_ctx* c, uint64_t count) {
  if (c->bs().ext_reply && count * c->reply_cts > c->bs().ext_reply_cts)
    throw Fail{PIRGPU_INVALID_ARGUMENT, "the caller's reply buffer (pirgpu_batch_set_reply_buffer) is too small for this batch"};
}

static int batch_stage_impl(pirgpu_ctx* c, const uint64_t* queries, uint32_t nq, uint32_t count, bool async) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    if (!queries || nq != c->dim_sum / c->N + 1)  // reference server.cpp:154-158
      return fail(c, PIRGPU_INVALID_ARGUMENT,
                  "Number of ciphertexts doesn't match number of items for oblivious expansion.");
    if (count == 0 || count > 4096) return fail(c, PIRGPU_INVALID_ARGUMENT, "batch size must be in [1, 4096]");
    ensure_batch_capacity(c, count);
    BatchSet& b = c->bs();
    const size_t qwords = (size_t)nq * c->ctw;
    // head-stream work queued earlier may still import queries of this set's previous batch: uploads go behind it
    if (c->ev_head_tail) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_head_tail, 0));
    if (async) {
      // pieces of kStagePiece queries, one event each: the first group starts as soon as ITS queries are on the device
      const uint32_t pieces = (count + kStagePiece - 1) / kStagePiece;
      while (b.st_events.size() < pieces) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b.st_events.push_back(e);
      }
      for (uint32_t p = 0; p < pieces; ++p) {
        const uint32_t q0 = p * kStagePiece, n = std::min<uint32_t>(kStagePiece, count - q0);
        HIP_TRY(hipMemcpyAsync(b.d_bquery + (size_t)q0 * qwords, queries + (size_t)q0 * qwords, (size_t)n * qwords * 8,
                               hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(b.st_events[p], c->stream));
      }
      b.st_pieces = pieces;
    } else {
      HIP_TRY(hipMemcpyAsync(b.d_bquery, queries, (size_t)count * qwords * 8, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      b.st_pieces = 0;
    }
    b.batch_count = count;
    b.staged_count = count;
    b.batch_keysets.assign(count, 0);   // pirgpu_batch_set_keysets assigns other clients' key sets
    b.batch_keyset_gens.assign(count, 0);
    b.batch_valid = false;
    return PIRGPU_OK;
  });
}

int pirgpu_batch_stage(pirgpu_ctx* c, const uint64_t* queries, uint32_t nq, uint32_t count) {
  return batch_stage_impl(c, queries, nq, count, false);
}

int pirgpu_batch_stage_async(pirgpu_ctx* c, const uint64_t* pinned_queries, uint32_t nq, uint32_t count) {
  return batch_stage_impl(c, pinned_queries, nq, count, true);
}

int pirgpu_batch_unstage(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    BatchSet& b = c->bs();
    b.staged_count = 0;
    b.batch_keysets.clear();
    b.batch_keyset_gens.clear();
    b.st_pieces = 0;
    return PIRGPU_OK;
  });
}

// `stream` waits for the upload of the staged queries [first, first + n) (asynchronously staged batches only)
static void wait_staged(pirgpu_ctx* c, hipStream_t stream, uint32_t first, uint32_t n) {
  BatchSet& b = c->bs();
  if (!b.st_pieces || !n) return;
  const uint32_t p0 = first / kStagePiece, p1 = std::min<uint32_t>(b.st_pieces - 1, (first + n - 1) / kStagePiece);
  for (uint32_t p = p0; p <= p1; ++p) HIP_TRY(hipStreamWaitEvent(stream, b.st_events[p], 0));
}

// Batch staging (queries + replies, device resident) for at least `count` queries.  Growth frees the old pair
// (after every stream that may still touch it has drained) and at least doubles, so a peer that sends ever
// larger requests cannot accumulate stale buffers.
static void ensure_batch_capacity(pirgpu_ctx* c, uint32_t count) {
  if (count <= c->bs().batch_cap) return;
  const uint32_t nq = c->dim_sum / c->N + 1;
  const uint32_t old_cap = c->bs().batch_cap;
  if (c->bs().d_bquery || c->bs().d_breply) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (BatchLane& ln : c->lanes)
      if (ln.stream) HIP_TRY(hipStreamSynchronize(ln.stream));
    for (Worker& w : c->workers)
      if (w.stream) HIP_TRY(hipStreamSynchronize(w.stream));
    for (uint64_t* p : {c->bs().d_bquery, c->bs().d_breply}) {
      if (!p) continue;
      auto it = std::find(c->allocs.begin(), c->allocs.end(), (void*)p);
      if (it != c->allocs.end()) c->allocs.erase(it);
      HIP_TRY(hipFree(p));
    }
    c->bs().d_bquery = c->bs().d_breply = nullptr;
    c->bs().batch_cap = 0;
    c->bs().batch_valid = false;
    c->bs().staged_count = 0;   // the staged queries went with the old buffer
  }
  const uint32_t cap = std::max<uint32_t>(count, std::min<uint32_t>(4096, 2 * old_cap));
  c->bs().d_bquery = c->dalloc<uint64_t>((size_t)cap * nq * c->ctw);
  c->bs().d_breply = c->dalloc<uint64_t>((size_t)cap * c->reply_cts * c->ctw);
  c->bs().batch_cap = cap;
}

// Lanes (stream + expansion buffers for up to 8 interleaved queries), created on the first batch.
static void ensure_lanes(pirgpu_ctx* c, bool with_expansion_buffers) {
  const uint32_t N = c->N, k = c->k;
  if (c->lanes.empty()) {
    // groups in flight (each lane = one stream + one set of group buffers): two -- one, three and four lanes were measured
    // in rounds 2 and 4 (3 857 / 4 114 / 4 090 against 4 165; 5 304 against 5 427 queries/s), a constant since round 5
    const uint32_t n_lanes = 2;
    c->lanes.resize(n_lanes);
    for (BatchLane& ln : c->lanes) {
      HIP_TRY(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&ln.ev_scanned, hipEventDisableTiming));
    }
  }
  if (with_expansion_buffers && !c->lanes[0].res_a) {
    const uint64_t half = std::max<uint64_t>(c->m_max / 2, 1);
    for (BatchLane& ln : c->lanes) {
      ln.res_a = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * c->m_max * c->ctw);
      ln.res_b = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * c->m_max * c->ctw);
      ln.prod = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * half * 2 * (k + 1) * N);
      ln.dig = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * half * (k + 1) * k * N);
    }
  }
  if (c->mfma_on && c->lanes[0].lvl.empty()) {  // the group's multiply runs in lane-owned, query-major buffers
    for (BatchLane& ln : c->lanes) {
      ln.lvl.assign(c->d, nullptr);
      for (uint32_t l = 0; l < c->d; ++l) ln.lvl[l] = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * c->lvl_cts[l] * c->ctw);
      if (c->pt_words) ln.pt_buf = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * c->pt_words);
      if (c->mg.nchunks > 1)
        ln.scan_part = c->dalloc<uint64_t>((size_t)kMaxMfmaQueries * c->mg.nchunks * std::max<uint32_t>(c->scan_rows, 1) * c->ctw);
    }
  }
}

// Batched oblivious expansion of the staged queries first .. first+B-1 on lane `ln`, B queries interleaved
// (ciphertext index = node * B + query), into the members' own selection vectors (NTT form).
static void ensure_head_slot(pirgpu_ctx* c, HeadSlot& hs) {
  if (hs.res_a) return;
  const uint32_t N = c->N, k = c->k;
  const uint64_t cts = ((uint64_t)1 << c->head_levels) * kMaxMfmaQueries;   // tree ciphertexts after the last head level
  hs.res_a = c->dalloc<uint64_t>(cts * c->ctw);
  hs.res_b = c->dalloc<uint64_t>(cts * c->ctw);
  hs.prod = c->dalloc<uint64_t>(std::max<uint64_t>(cts / 2, 1) * 2 * (k + 1) * N);
  hs.dig = c->dalloc<uint64_t>(std::max<uint64_t>(cts / 2, 1) * (k + 1) * k * N);
  HIP_TRY(hipEventCreateWithFlags(&hs.ev_ready, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&hs.ev_free, hipEventDisableTiming));
  if (!c->head_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->head_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_head_tail, hipEventDisableTiming));
  }
}

// sv_dst (optional): the selection vector of query q goes to sv_dst + q * dim_sum ciphertexts (caller-owned memory that
// outlives the members' buffers: the slot-sharded step keeps its row selectors there for two steps) instead of the
// members' own buffers.
static void expand_group_on_lane(pirgpu_ctx* c, BatchLane& ln, Worker* const* members, uint32_t B, uint32_t first,
                                 bool sel_f64 = false, uint64_t* sv_dst = nullptr) {
  const uint32_t N = c->N, k = c->k;
  const uint32_t nq = c->dim_sum / N + 1;
  const size_t ctw = c->ctw, qwords = (size_t)nq * ctw;
  uint64_t remaining = c->dim_sum, produced = 0;
  // the narrow first levels on the head stream (HeadSlot): one query ciphertext per query (the usual case), a real group,
  // and at least two levels left for the lane
  // -- for batches that were staged asynchronously (request windows queued behind one another: +6 % through the wire,
  // 12.0 against 12.9 ms per window of 64 with two callers) or always (head_mode 2); a back-to-back loop over one staged
  // batch gains nothing from it (5 325-5 408 against 5 379-5 433 queries/s, same box) and keeps the plain path
  const bool use_head = nq == 1 && B > 1 && c->head_levels > 0 && (c->head_mode == 2 || (c->head_mode == 1 && c->bs().st_pieces > 0)) &&
                        hm::ceil_log2((uint32_t)std::min<uint64_t>(c->dim_sum, N)) >= c->head_levels + 2;
  if (!use_head) wait_staged(c, ln.stream, first, B);
  for (uint32_t qc = 0; qc < nq && remaining; ++qc) {
    const uint32_t slots = (uint32_t)std::min<uint64_t>(remaining, N);
    MfmaPtrs dst{};
    uint32_t ksets[kMaxMfmaQueries];   // every query of the group is switched with its own client's keys
    for (uint32_t q = 0; q < B; ++q) {
      dst.p[q] = (sv_dst ? sv_dst + (size_t)q * c->dim_sum * ctw : members[q]->sv_ntt) + produced * ctw;
      ksets[q] = first + q < c->bs().batch_keysets.size() ? c->bs().batch_keysets[first + q] : 0;
    }
    const uint64_t* roots = c->bs().d_bquery + (size_t)first * qwords + (size_t)qc * ctw;
    uint64_t* res;
    if (use_head) {
      HeadSlot& hs = ln.head[ln.head_next++ & 1];
      ensure_head_slot(c, hs);
      hipStream_t hst = c->head_stream;
      if (hs.in_use) HIP_TRY(hipStreamWaitEvent(hst, hs.ev_free, 0));   // the lane is done with the slot's previous tree
      wait_staged(c, hst, first, B);
      HIP_TRY(launch_tree_convert(hst, c->dp, c->mode, roots, hs.res_a, (uint64_t)B * ctw, true, ctw, qwords));
      TreeState at;
      (void)expand_core(c, hst, hs.res_a, hs.res_b, hs.dig, hs.prod, slots, B, &dst, sel_f64, ksets, 0, c->head_levels, nullptr, &at);
      HIP_TRY(hipEventRecord(hs.ev_ready, hst));
      HIP_TRY(hipEventRecord(c->ev_head_tail, hst));
      HIP_TRY(hipStreamWaitEvent(ln.stream, hs.ev_ready, 0));
      res = expand_core(c, ln.stream, ln.res_a, ln.res_b, ln.dig, ln.prod, slots, B, &dst, sel_f64, ksets, c->head_levels,
                        UINT32_MAX, &at, nullptr, [&]() {
                          HIP_TRY(hipEventRecord(hs.ev_free, ln.stream));
                          hs.in_use = true;
                        });
    } else {
      // the B query ciphertexts, gathered side by side out of the staged batch, become the roots of the B interleaved
      // trees (one strided import launch: no separate 2-D copy)
      HIP_TRY(launch_tree_convert(ln.stream, c->dp, c->mode, roots, ln.res_a, (uint64_t)B * ctw, true, ctw, qwords));
      res = expand_core(c, ln.stream, ln.res_a, ln.res_b, ln.dig, ln.prod, slots, B, &dst, sel_f64, ksets);
    }
    if (res) HIP_TRY(c->ops->ct_ntt_fwd_split(ln.stream, c->mode, c->dp, k, res, dst, B, (uint64_t)slots * B));
    produced += slots;
    remaining -= slots;
  }
  for (uint32_t q = 0; q < B; ++q) {
    members[q]->sv_cur = nullptr;
    members[q]->sv_rows = nullptr;
  }
}

// Batch mode with the MFMA scan (see BatchLane): groups of up to mfma_nq queries share one batched
// expansion and one database pass; consecutive groups alternate between two lanes.
struct PackedInput {            // multi-GPU packed exchange (pirgpu_batch_run_packed)
  const uint8_t* packed;       // [src rank][group] digit-packed column selectors, mg.sel_bytes each
  const uint64_t* rows;        // [query][this shard's rows][2][k][N] dimension-0 selectors
  uint32_t per_rank;           // queries per source rank (groups never span two source ranks)
};

// Members (selection-vector buffers) of the group that runs on lane `li`: lane-bound, so that stream order on the
// lane covers their reuse; with fewer than 2 x G workers every group uses lane 0.
static uint32_t lanes_in_use(pirgpu_ctx* c, uint32_t G) {
  const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
  return std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)c->lanes.size(), W / std::max<uint32_t>(G, 1)));
}

static void batch_run_mfma(pirgpu_ctx* c, uint32_t count, const uint64_t* ext_sv, const PackedInput* pk = nullptr) {
  const size_t rwords = (size_t)c->reply_cts * c->ctw, svwords = (size_t)c->dim_sum * c->ctw;
  const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
  const uint32_t G = std::min<uint32_t>(c->mfma_nq, W);
  const uint32_t my_rows = c->se - c->sb;
  ensure_lanes(c, !ext_sv && !pk);
  const uint32_t nl = lanes_in_use(c, G);
  // groups: runs of up to G consecutive queries; with packed input a group is what ONE source rank packed together
  // (its queries in runs of kMaxMfmaQueries), so the walk restarts at every source-rank boundary
  const uint32_t span = pk ? pk->per_rank : count;
  // groups this call queues: with two or more (and two lanes) every pass runs on half the chip
  uint32_t n_groups = 0;
  for (uint32_t rank0 = 0; rank0 < count; rank0 += span) {
    const uint32_t in_span = std::min<uint32_t>(span, count - rank0);
    n_groups += (in_span + (pk ? (uint32_t)kMaxMfmaQueries : G) - 1) / (pk ? (uint32_t)kMaxMfmaQueries : G);
  }
  const bool share_chip = n_groups >= 2 && nl >= 2;
  // pirgpu_batch_set_host_replies: every group downloads its replies as soon as they exist (on its lane's stream)
  const bool host_dl = c->bs().host_reply && (uint64_t)count * c->reply_cts <= c->bs().host_reply_cts;
  c->bs().host_reply_done = host_dl;
  c->bs().dl_end.clear();
  c->bs().dl_next = 0;
  for (uint32_t rank0 = 0; rank0 < count; rank0 += span) {
    const uint32_t step = pk ? (uint32_t)kMaxMfmaQueries : G;
    const uint32_t in_span = std::min<uint32_t>(span, count - rank0);
    for (uint32_t j0 = 0; j0 < in_span; j0 += step) {
      const uint32_t B = std::min<uint32_t>(step, in_span - j0);
      if (B > W) throw Fail{PIRGPU_FAILED_PRECONDITION, "packed groups need min(queries per rank, 8) workers (pirgpu_set_concurrency)"};
      const uint32_t li = (uint32_t)(c->groups_run++ % nl);
      BatchLane& ln = c->lanes[li];
      Worker* members[kMaxMfmaQueries];
      for (uint32_t q = 0; q < B; ++q) {
        members[q] = &c->workers[(li * G + q) % W];
        HIP_TRY(hipStreamWaitEvent(ln.stream, members[q]->ev_done, 0));  // whoever used its buffers last is done
      }
      const uint32_t first = rank0 + j0;  // global index of the group's first query
      const uint8_t* packed = nullptr;
      // level 0 of the group = its replies, query-major with the reply buffer's own stride: the last fold and the final
      // inverse transform write them where pirgpu_batch_fetch reads them (no device-to-device copy per group)
      uint64_t* lvl_ptrs[PIRGPU_MAX_DIMS];
      for (uint32_t l = 0; l < c->d; ++l) lvl_ptrs[l] = ln.lvl[l];
      const bool direct_reply = c->d >= 2;   // d = 1 would make the scan itself write there: keep the lane buffer
      if (direct_reply) lvl_ptrs[0] = reply_base(c) + (size_t)first * rwords;
      Stage sg{ln.stream, lvl_ptrs, ln.pt_buf, B, MfmaPtrs{}, pk != nullptr, &ln.up_scratch, &ln.up_scratch_words};
      MfmaPtrs col{};
      if (pk) {
        const uint32_t groups_per_rank = (pk->per_rank + kMaxMfmaQueries - 1) / kMaxMfmaQueries;
        packed = pk->packed + ((size_t)(rank0 / pk->per_rank) * groups_per_rank + j0 / kMaxMfmaQueries) * c->mg.sel_bytes;
        for (uint32_t q = 0; q < B; ++q) sg.sel.p[q] = pk->rows + (size_t)(first + q) * my_rows * c->ctw;
      } else {
        if (!ext_sv) {
          expand_group_on_lane(c, ln, members, B, first, c->sel_f64);
          sg.sel_f64 = c->sel_f64;   // written and read on this lane only: exact doubles instead of u64
        }
        for (uint32_t q = 0; q < B; ++q) {
          const uint64_t* sv = ext_sv ? ext_sv + (size_t)(first + q) * svwords : members[q]->sv_ntt;
          sg.sel.p[q] = sv;
          col.p[q] = sv + (size_t)c->sv_off[c->d - 1] * c->ctw;
        }
      }
      scan_group_mfma(c, ln.stream, ln.selp, col, B, ln.lvl[c->d - 1], ln.scan_part, nullptr, packed, sg.sel_f64,
                      share_chip);
      post_scan_stage(c, sg, nullptr);
      if (!direct_reply)
        HIP_TRY(hipMemcpyAsync(reply_base(c) + (size_t)first * rwords, ln.lvl[0], (size_t)B * rwords * 8,
                               hipMemcpyDeviceToDevice, ln.stream));
      if (host_dl) {
        // the group's replies start their way to the host while the next groups are computed -- on the context's COPY
        // stream, behind an event of the lane: the lane itself goes straight on to its next group (round 3 queued the
        // copy on the lane, which then idled for 8 MB of PCIe per group: 5-6 % of a lane's time at cfg 3)
        BatchSet& b = c->bs();
        const size_t gi = b.dl_end.size();
        while (gi >= b.dl_events.size()) {
          hipEvent_t e;
          HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          b.dl_events.push_back(e);
          HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          b.rd_events.push_back(e);
        }
        if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventRecord(b.rd_events[gi], ln.stream));
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, b.rd_events[gi], 0));
        HIP_TRY(hipMemcpyAsync(b.host_reply + (size_t)first * rwords, reply_base(c) + (size_t)first * rwords,
                               (size_t)B * rwords * 8, hipMemcpyDeviceToHost, c->copy_stream));
        HIP_TRY(hipEventRecord(b.dl_events[gi], c->copy_stream));
        b.dl_end.push_back(first + B);
      }
      for (uint32_t q = 0; q < B; ++q) {
        HIP_TRY(hipEventRecord(members[q]->ev_done, ln.stream));
        members[q]->reply_valid = false;  // the group's replies live in the lane / batch buffers, not in the worker
      }
    }
  }
}

// Shared body of pirgpu_batch_run / pirgpu_batch_run_selectors: `count` queries in rounds of W (one worker
// per query).  Expansion is batched per group of up to 8 queries on a lane; with the MFMA scan the group
// also shares the database pass (batch_run_mfma), otherwise groups of up to 4 workers share one pass of
// scan_mq_kernel (d = 1 / few rows), hand-offs between streams through events.  With ext_sv the expansion
// is skipped and query i reads its NTT-form selection vector at ext_sv + i*dim_sum.
static void batch_run_impl_body(pirgpu_ctx* c, uint32_t count, const uint64_t* ext_sv);
static void batch_run_impl(pirgpu_ctx* c, uint32_t count, const uint64_t* ext_sv) {
  struct Flag {
    bool& f;
    explicit Flag(bool& x) : f(x) { f = true; }
    ~Flag() { f = false; }
  } flag(c->in_batch);
  batch_run_impl_body(c, count, ext_sv);
}
static void batch_run_impl_body(pirgpu_ctx* c, uint32_t count, const uint64_t* ext_sv) {
  refuse_slot_shard(c);
  c->bs().host_reply_done = false;   // set again by the path that queues per-group downloads (batch_run_mfma)
  const size_t rwords = (size_t)c->reply_cts * c->ctw;
  const size_t svwords = (size_t)c->dim_sum * c->ctw;
  const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
  const uint32_t G = mq_usable(c) && c->pt_end > c->pt_begin ? std::min<uint32_t>(c->mq_nq, kMaxScanQueries) : 1;
  if (c->n_loaded != c->pt_end - c->pt_begin)
    throw Fail{PIRGPU_FAILED_PRECONDITION, "database not fully loaded"};
  check_transparent(c);
  check_reply_target(c, count);
  if (!ext_sv) check_staged_keysets(c);
  ensure_packed(c);
  c->prof_cur = -1;
  if (c->mfma_on) {
    batch_run_mfma(c, count, ext_sv);
    c->bs().batch_valid = true;
    return;
  }
  if (!ext_sv) ensure_lanes(c, true);
  for (uint32_t base = 0; base < count; base += W) {
    const uint32_t n = std::min<uint32_t>(W, count - base);
    if (ext_sv) {
      for (uint32_t j = 0; j < n; ++j) {
        Worker& w = c->workers[j];
        w.sv_cur = ext_sv + (size_t)(base + j) * svwords;
        HIP_TRY(hipEventRecord(w.ev_expanded, w.stream));
      }
    } else {
      for (uint32_t j0 = 0; j0 < n; j0 += kMaxMfmaQueries) {
        const uint32_t B = std::min<uint32_t>(kMaxMfmaQueries, n - j0);
        BatchLane& ln = c->lanes[c->groups_run++ % c->lanes.size()];
        Worker* members[kMaxMfmaQueries];
        for (uint32_t q = 0; q < B; ++q) {
          members[q] = &c->workers[j0 + q];
          HIP_TRY(hipStreamWaitEvent(ln.stream, members[q]->ev_done, 0));  // its selection vector is free again
        }
        expand_group_on_lane(c, ln, members, B, base + j0);
        for (uint32_t q = 0; q < B; ++q) {
          HIP_TRY(hipEventRecord(members[q]->ev_expanded, ln.stream));
          HIP_TRY(hipStreamWaitEvent(members[q]->stream, members[q]->ev_expanded, 0));
        }
      }
    }
    for (uint32_t j0 = 0; j0 < n; j0 += G) {
      const uint32_t g = std::min<uint32_t>(G, n - j0);
      uint32_t done = 0;
      while (done < g) {  // group sizes the kernel is instantiated for: 4, 2, 1
        const uint32_t take = g - done >= 4 && G >= 4 ? 4 : (g - done >= 2 && G >= 2 ? 2 : 1);
        Worker& lead = c->workers[j0 + done];
        if (take == 1) {
          scan_on_device(c, lead);
        } else {
          const uint64_t* svp[kMaxScanQueries];
          uint64_t* outp[kMaxScanQueries];
          for (uint32_t q = 0; q < take; ++q) {
            Worker& m = c->workers[j0 + done + q];
            if (q) HIP_TRY(hipStreamWaitEvent(lead.stream, m.ev_expanded, 0));
            svp[q] = scan_selectors(c, m);
            outp[q] = m.lvl[c->d - 1];
          }
          const uint32_t rpw = take == 4 ? (c->mq_rows > 2 ? 1 : c->mq_rows) : (c->mq_rows > 2 ? 2 : c->mq_rows);
          HIP_TRY(launch_scan_mq(lead.stream, c->dp, c->N, c->k, c->d_db, svp, outp, take, c->scan_rows,
                                 c->scan_cols, rpw, c->scan_limb));
          HIP_TRY(hipEventRecord(lead.ev_scanned, lead.stream));
          for (uint32_t q = 1; q < take; ++q)
            HIP_TRY(hipStreamWaitEvent(c->workers[j0 + done + q].stream, lead.ev_scanned, 0));
        }
        done += take;
      }
    }
    for (uint32_t j = 0; j < n; ++j) {
      Worker& w = c->workers[j];
      post_scan_on_device(c, w);
      HIP_TRY(hipMemcpyAsync(reply_base(c) + (base + j) * rwords, w.lvl[0], rwords * 8, hipMemcpyDeviceToDevice,
                             w.stream));
      HIP_TRY(hipEventRecord(w.ev_done, w.stream));
      w.reply_valid = true;
    }
  }
  c->bs().batch_valid = true;
}

int pirgpu_batch_run(pirgpu_ctx* c) {
  return guarded(c, [&]() -> int {
    if (!c->bs().staged_count) return fail(c, PIRGPU_FAILED_PRECONDITION, "no batch has been staged");
    batch_run_impl(c, c->bs().staged_count, nullptr);
    c->bs().batch_count = c->bs().staged_count;
    return PIRGPU_OK;
  });
}

// ======================================================================================================================
// [9] MULTI-GPU entry points: u64 selector exchange, packed row-shard exchange, slot shards, reply copies, fix-up, pack40
// ======================================================================================================================
// ---- query-parallel expansion for multi-GPU ----

int pirgpu_batch_expand(pirgpu_ctx* c, uint32_t first, uint32_t count, uint64_t* device_dst) {
  return guarded(c, [&]() -> int {
    if (!c->bs().staged_count || (uint64_t)first + count > c->bs().staged_count)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "query range outside the staged batch");
    if (!device_dst && count) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    const uint32_t nq = c->dim_sum / c->N + 1;
    const size_t qwords = (size_t)nq * c->ctw, svwords = (size_t)c->dim_sum * c->ctw;
    const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
    c->prof_cur = -1;
    check_staged_keysets(c);
    for (uint32_t i = 0; i < count; ++i) {
      Worker& w = c->workers[i % W];
      HIP_TRY(hipStreamWaitEvent(w.stream, w.ev_done, 0));   // a lane's group may still be reading its buffers
      wait_staged(c, w.stream, first + i, 1);
      HIP_TRY(hipMemcpyAsync(w.d_query, c->bs().d_bquery + (size_t)(first + i) * qwords, qwords * 8,
                             hipMemcpyDeviceToDevice, w.stream));
      w.staged_nq = nq;
      w.keyset = first + i < c->bs().batch_keysets.size() ? c->bs().batch_keysets[first + i] : 0;
      expand_query_to_sv(c, w, w.d_query, nq);
      HIP_TRY(hipMemcpyAsync(device_dst + (size_t)i * svwords, w.sv_ntt, svwords * 8, hipMemcpyDeviceToDevice,
                             w.stream));
    }
    for (Worker& w : c->workers) HIP_TRY(hipStreamSynchronize(w.stream));
    return PIRGPU_OK;
  });
}

int pirgpu_batch_run_selectors(pirgpu_ctx* c, const uint64_t* device_sv, uint32_t count) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    if (!device_sv || count == 0 || count > 4096) return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid batch");
    ensure_batch_capacity(c, count);
    batch_run_impl(c, count, device_sv);
    c->bs().batch_count = count;
    return PIRGPU_OK;
  });
}

// ---- packed selector exchange for row-sharded multi-GPU runs ----

uint64_t pirgpu_packed_selector_bytes(pirgpu_ctx* c) {
  uint64_t bytes = 0;
  (void)guarded(c, [&]() -> int {
    ensure_workspace(c);
    // d = 2 only (dimension 0 = rows, dimension 1 = the scanned columns) and the shard scanned by the MFMA kernel
    if (c->d == 2 && c->mfma_on) bytes = c->mg.sel_bytes;
    return PIRGPU_OK;
  });
  return bytes;
}

static int batch_expand_packed_impl(pirgpu_ctx* c, uint32_t first_query, uint32_t count, uint8_t* device_packed,
                                    uint64_t* device_rows, const uint32_t* row_cuts, uint32_t n_ranks, bool wait) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    if (c->d != 2 || !c->mfma_on)
      return fail(c, PIRGPU_FAILED_PRECONDITION, "packed selector exchange needs d = 2 and the int8-MFMA scan");
    if (!c->bs().staged_count || (uint64_t)first_query + count > c->bs().staged_count)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "query range outside the staged batch");
    if (!device_packed || !device_rows || !row_cuts || n_ranks == 0 || row_cuts[0] != 0 || row_cuts[n_ranks] != c->dims[0])
      return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid packed-exchange buffers or row cuts");
    for (uint32_t s = 0; s < n_ranks; ++s)
      if (row_cuts[s] > row_cuts[s + 1]) return fail(c, PIRGPU_INVALID_ARGUMENT, "row cuts must be non-decreasing");
    const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
    if (W < std::min<uint32_t>(count, kMaxMfmaQueries))
      return fail(c, PIRGPU_FAILED_PRECONDITION, "packed groups need min(count, 8) workers (pirgpu_set_concurrency)");
    ensure_lanes(c, true);
    c->prof_cur = -1;
    check_staged_keysets(c);
    const size_t ctw = c->ctw;
    const uint32_t kN = c->k * c->N;
    const uint32_t first = first_query;
    const uint32_t G = std::min<uint32_t>(kMaxMfmaQueries, W), nl = lanes_in_use(c, G);
    uint32_t g = 0;
    for (uint32_t j0 = 0; j0 < count; j0 += kMaxMfmaQueries, ++g) {
      const uint32_t B = std::min<uint32_t>(kMaxMfmaQueries, count - j0);
      const uint32_t li = (uint32_t)(c->groups_run++ % nl);
      BatchLane& ln = c->lanes[li];
      Worker* members[kMaxMfmaQueries];
      for (uint32_t q = 0; q < B; ++q) {
        members[q] = &c->workers[(li * G + q) % W];
        HIP_TRY(hipStreamWaitEvent(ln.stream, members[q]->ev_done, 0));
      }
      expand_group_on_lane(c, ln, members, B, first + j0);
      // column selectors (dimension 1) of the whole group -> one B-operand buffer
      MfmaPtrs sv{};
      for (uint32_t q = 0; q < B; ++q) sv.p[q] = members[q]->sv_ntt + (size_t)c->sv_off[1] * ctw;
      HIP_TRY(launch_sel_pack(ln.stream, c->dp, c->mg, sv, B, device_packed + (size_t)g * c->mg.sel_bytes, c->scan_cols, kN));
      // row selectors (dimension 0): block of destination rank s = [query][rows of s][2][k][N]
      for (uint32_t s2 = 0; s2 < n_ranks; ++s2) {
        const uint32_t r0 = row_cuts[s2], nr = row_cuts[s2 + 1] - r0;
        if (!nr) continue;
        for (uint32_t q = 0; q < B; ++q)
          HIP_TRY(hipMemcpyAsync(device_rows + ((size_t)count * r0 + (size_t)(j0 + q) * nr) * ctw,
                                 members[q]->sv_ntt + (size_t)(c->sv_off[0] + r0) * ctw, (size_t)nr * ctw * 8,
                                 hipMemcpyDeviceToDevice, ln.stream));
      }
      for (uint32_t q = 0; q < B; ++q) {  // the members' selection vectors are free once this lane got here
        HIP_TRY(hipEventRecord(members[q]->ev_done, ln.stream));
      }
    }
    if (wait)
      for (BatchLane& ln : c->lanes) HIP_TRY(hipStreamSynchronize(ln.stream));
    return PIRGPU_OK;
  });
}

int pirgpu_batch_expand_packed(pirgpu_ctx* c, uint32_t first_query, uint32_t count, uint8_t* device_packed,
                               uint64_t* device_rows, const uint32_t* row_cuts, uint32_t n_ranks) {
  return batch_expand_packed_impl(c, first_query, count, device_packed, device_rows, row_cuts, n_ranks, true);
}

int pirgpu_batch_expand_packed_async(pirgpu_ctx* c, uint32_t first_query, uint32_t count, uint8_t* device_packed,
                                     uint64_t* device_rows, const uint32_t* row_cuts, uint32_t n_ranks) {
  return batch_expand_packed_impl(c, first_query, count, device_packed, device_rows, row_cuts, n_ranks, false);
}

int pirgpu_batch_run_packed(pirgpu_ctx* c, const uint8_t* device_packed, uint32_t n_ranks, uint32_t per_rank,
                            const uint64_t* device_rows) {
  return guarded(c, [&]() -> int {
    ensure_workspace(c);
    if (c->d != 2 || !c->mfma_on)
      return fail(c, PIRGPU_FAILED_PRECONDITION, "packed selector exchange needs d = 2 and the int8-MFMA scan");
    refuse_slot_shard(c);
    const uint64_t count = (uint64_t)n_ranks * per_rank;
    if (!device_packed || !device_rows || count == 0 || count > 4096) return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid batch");
    if (c->n_loaded != c->pt_end - c->pt_begin) return fail(c, PIRGPU_FAILED_PRECONDITION, "database not fully loaded");
    check_transparent(c);
    check_reply_target(c, count);
    ensure_packed(c);
    ensure_batch_capacity(c, (uint32_t)count);
    c->prof_cur = -1;
    PackedInput pk{device_packed, device_rows, per_rank};
    c->in_batch = true;
    try {
      batch_run_mfma(c, (uint32_t)count, nullptr, &pk);
    } catch (...) {
      c->in_batch = false;
      throw;
    }
    c->in_batch = false;
    c->bs().batch_count = (uint32_t)count;
    c->bs().batch_valid = true;
    return PIRGPU_OK;
  });
}

// ---- slot-sharded multi-GPU step ----
//
// The base case of PIRDatabase::multiply (reference database.cpp:185-194) is a dyadic product in NTT form: independent
// per NTT slot.  G ranks therefore each hold the slots [cut[g], cut[g+1]) of EVERY plaintext (1 / G of the bytes; the
// operand layout has the slot outermost) and a step of B queries runs as
//   E  rank g expands its own B / G queries (groups of 8) and packs their column selectors, cut by destination rank;
//   X1 all-to-all: rank h receives ITS slots of every group's packed column selectors (1 / G of each);
//   S  rank h scans its slots for all B queries in one launch: full row tiles, row sums [query][row, comp][its slots];
//   X2 all-to-all: the row sums go back to the rank that expanded the query;
//   U  that rank puts the slot pieces together and runs inverse NTT + upper level for its own queries on ALL rows with
//      the row selectors it kept: reply i ends on the rank that owns query i.  No row-selector exchange, no reduce.
// The three entry points queue E, S and U on the lanes without waiting for the device.  `after` / `then` (hipStream_t
// of the caller, may be NULL) order them against the caller's collectives: every lane the call uses first waits for what
// is queued on `after` so far, and `then` waits for what the call queued.

namespace {

void check_cuts(pirgpu_ctx* c, const uint32_t* cuts, uint32_t n_ranks, SliceMap& map) {
  const uint32_t kN = c->k * c->N;
  if (!cuts || n_ranks == 0 || n_ranks > (uint32_t)kMaxSlices || cuts[0] != 0 || cuts[n_ranks] != kN)
    throw Fail{PIRGPU_INVALID_ARGUMENT, "slot cuts must run from 0 to k N over at most 16 ranks"};
  map = SliceMap{};
  map.n = n_ranks;
  for (uint32_t r = 0; r <= n_ranks; ++r) {
    if (cuts[r] % 16 || (r && cuts[r] < cuts[r - 1])) throw Fail{PIRGPU_INVALID_ARGUMENT, "slot cuts must be non-decreasing multiples of 16"};
    map.cut[r] = cuts[r];
  }
}

void lane_after(pirgpu_ctx* c, hipStream_t lane, void* after) {
  if (!after) return;
  if (!c->ev_after) HIP_TRY(hipEventCreateWithFlags(&c->ev_after, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->ev_after, (hipStream_t)after));
  HIP_TRY(hipStreamWaitEvent(lane, c->ev_after, 0));
}

void lane_then(BatchLane& ln, void* then) {
  if (!then) return;
  if (!ln.ev_join) HIP_TRY(hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(ln.ev_join, ln.stream));
  HIP_TRY(hipStreamWaitEvent((hipStream_t)then, ln.ev_join, 0));
}

void check_slots_ctx(pirgpu_ctx* c) {
  ensure_workspace(c);
  if (c->d != 2 || !c->mfma_on || c->mg.nchunks != 1 || c->sb != 0 || c->se != c->dims[0])
    throw Fail{PIRGPU_FAILED_PRECONDITION, "the slot-sharded step needs d = 2, all rows and the int8-MFMA scan in one column chunk"};
}

}  // namespace

uint64_t pirgpu_slots_packed_bytes(pirgpu_ctx* c, uint32_t slots) {
  uint64_t bytes = 0;
  (void)guarded(c, [&]() -> int {
    ensure_workspace(c);
    if (c->d == 2 && c->mfma_on && c->mg.nchunks == 1) bytes = (uint64_t)slots * c->mg.KG * c->mg.tile_bytes;
    return PIRGPU_OK;
  });
  return bytes;
}

int pirgpu_slots_expand_async(pirgpu_ctx* c, uint32_t first, uint32_t count, uint8_t* device_packed, uint64_t* device_sv,
                              const uint32_t* slot_cuts, uint32_t n_ranks, void* after, void* then) {
  return guarded(c, [&]() -> int {
    check_slots_ctx(c);
    SliceMap map;
    check_cuts(c, slot_cuts, n_ranks, map);
    if (!c->bs().staged_count || (uint64_t)first + count > c->bs().staged_count)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "query range outside the staged batch");
    if (!device_packed || !device_sv || count == 0) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
    if (W < std::min<uint32_t>(count, kMaxMfmaQueries))
      return fail(c, PIRGPU_FAILED_PRECONDITION, "groups need min(count, 8) workers (pirgpu_set_concurrency)");
    ensure_lanes(c, true);
    c->prof_cur = -1;
    check_staged_keysets(c);
    const size_t ctw = c->ctw, svwords = (size_t)c->dim_sum * ctw;
    const uint32_t kN = c->k * c->N;
    const uint32_t G = std::min<uint32_t>(kMaxMfmaQueries, W), nl = lanes_in_use(c, G);
    const uint32_t groups = (count + kMaxMfmaQueries - 1) / kMaxMfmaQueries;
    const uint64_t slot_bytes = (uint64_t)c->mg.KG * c->mg.tile_bytes;
    if (after && c->head_stream) lane_after(c, c->head_stream, after);
    uint32_t g = 0;
    for (uint32_t j0 = 0; j0 < count; j0 += kMaxMfmaQueries, ++g) {
      const uint32_t B = std::min<uint32_t>(kMaxMfmaQueries, count - j0);
      BatchLane& ln = c->lanes[c->groups_run++ % nl];
      const uint32_t li = (uint32_t)(&ln - c->lanes.data());
      lane_after(c, ln.stream, after);
      Worker* members[kMaxMfmaQueries];
      for (uint32_t q = 0; q < B; ++q) {
        members[q] = &c->workers[(li * G + q) % W];
        HIP_TRY(hipStreamWaitEvent(ln.stream, members[q]->ev_done, 0));
      }
      uint64_t* sv_dst = device_sv + (size_t)j0 * svwords;
      expand_group_on_lane(c, ln, members, B, first + j0, c->sel_f64, sv_dst);
      // the group's column selectors -> B-operand tiles, rank r's slots as piece [r][group g] of the send buffer
      MfmaPtrs sv{};
      for (uint32_t q = 0; q < B; ++q) sv.p[q] = sv_dst + (size_t)q * svwords + (size_t)c->sv_off[1] * ctw;
      for (uint32_t r = 0; r < n_ranks; ++r)
        map.off[r] = ((uint64_t)groups * map.cut[r] + (uint64_t)g * (map.cut[r + 1] - map.cut[r])) * slot_bytes;
      HIP_TRY(launch_sel_pack(ln.stream, c->dp, c->mg, sv, B, device_packed, c->scan_cols, kN, c->sel_f64, &map));
      for (uint32_t q = 0; q < B; ++q) HIP_TRY(hipEventRecord(members[q]->ev_done, ln.stream));
      lane_then(ln, then);
    }
    return PIRGPU_OK;
  });
}

int pirgpu_slots_scan_async(pirgpu_ctx* c, const uint8_t* device_packed, uint32_t n_ranks, uint32_t per_rank,
                            uint64_t* device_rowsums, void* after, void* then) {
  return guarded(c, [&]() -> int {
    check_slots_ctx(c);
    const uint64_t count = (uint64_t)n_ranks * per_rank;
    if (!device_packed || !device_rowsums || count == 0 || count > 4096) return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid batch");
    if (c->n_loaded != c->pt_end - c->pt_begin) return fail(c, PIRGPU_FAILED_PRECONDITION, "database not fully loaded");
    check_transparent(c);
    ensure_packed(c);
    ensure_lanes(c, true);
    c->prof_cur = -1;
    const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
    const uint32_t nl = lanes_in_use(c, std::min<uint32_t>(kMaxMfmaQueries, W));
    BatchLane& ln = c->lanes[c->groups_run++ % nl];
    lane_after(c, ln.stream, after);
    const uint32_t groups = (per_rank + kMaxMfmaQueries - 1) / kMaxMfmaQueries;
    const uint64_t piece = (uint64_t)c->nslots * c->mg.KG * c->mg.tile_bytes;      // my slots of one packed group
    const uint64_t qwords = (uint64_t)c->scan_rows * 2 * c->nslots;                // row sums of one query on my slots
    // workgroups: the pass shares the chip with the other lane's expansion like the single-GPU batch pass (option
    // SLOTS_SCAN_WGS; 0 = one per CU).  Units in (slot block, group) order (option SLOTS_SCAN_BLK_MAJOR, default on):
    // the workgroups running side by side then read the same database tiles for different groups -- the launch was
    // 6 % (cfg 3) / 11 % (cfg 4) shorter than with one sweep of the slots per group, the step 1.2 - 1.5 %
    const uint32_t wgs = (uint32_t)std::max<int64_t>(0, option(c, "SLOTS_SCAN_WGS", c->scan_wgs_batch));
    ScanGroups grp{};
    auto flush = [&]() {
      if (!grp.n) return;
      HIP_TRY(launch_scan_mfma_groups(ln.stream, c->dp, c->mg, c->d_dbp, grp, c->scan_rows, 0, wgs,
                                      c->scan_f64_fold || c->scan_f64_fold_batch, c->slot0,
                                      c->nslots, qwords, c->nslots, option(c, "SLOTS_SCAN_BLK_MAJOR", 1) != 0));
      grp = ScanGroups{};
    };
    for (uint32_t r = 0; r < n_ranks; ++r)
      for (uint32_t g = 0; g < groups; ++g) {
        grp.sel[grp.n] = device_packed + ((uint64_t)r * groups + g) * piece;
        grp.out[grp.n] = device_rowsums + ((uint64_t)r * per_rank + (uint64_t)g * kMaxMfmaQueries) * qwords;
        grp.nq[grp.n] = (uint8_t)std::min<uint32_t>(kMaxMfmaQueries, per_rank - g * kMaxMfmaQueries);
        if (++grp.n == (uint32_t)kMaxScanGroups) flush();
      }
    flush();
    lane_then(ln, then);
    return PIRGPU_OK;
  });
}

int pirgpu_slots_finish_async(pirgpu_ctx* c, const uint64_t* device_rowsums, uint32_t count, const uint64_t* device_sv,
                              const uint32_t* slot_cuts, uint32_t n_ranks, uint64_t* device_replies, void* after,
                              void* then) {
  return guarded(c, [&]() -> int {
    check_slots_ctx(c);
    SliceMap map;
    check_cuts(c, slot_cuts, n_ranks, map);
    if (!device_rowsums || !device_sv || !device_replies || count == 0 || count > 4096)
      return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid batch");
    ensure_lanes(c, true);
    c->prof_cur = -1;
    const uint32_t W = std::max<uint32_t>(1, std::min<uint32_t>(c->n_active, (uint32_t)c->workers.size()));
    const uint32_t nl = lanes_in_use(c, std::min<uint32_t>(kMaxMfmaQueries, W));
    const size_t ctw = c->ctw, svwords = (size_t)c->dim_sum * ctw, rwords = (size_t)c->reply_cts * ctw;
    const uint32_t kN = c->k * c->N, RC = c->scan_rows * 2;
    const uint64_t words = (uint64_t)c->scan_rows * ctw;
    struct Flag {
      bool& f;
      explicit Flag(bool& x) : f(x) { f = true; }
      ~Flag() { f = false; }
    } flag(c->in_batch);
    for (uint32_t j0 = 0; j0 < count; j0 += kMaxMfmaQueries) {
      const uint32_t B = std::min<uint32_t>(kMaxMfmaQueries, count - j0);
      BatchLane& ln = c->lanes[c->groups_run++ % nl];
      lane_after(c, ln.stream, after);
      // rank h's block of the receive buffer holds [count][RC][slots of h]: the group's queries are rows j0 .. j0 + B of
      // it; the inverse transform of the row sums (database.cpp:250-254) gathers them from there (option
      // SLOTS_GATHER_NTT = 0: a separate assembly pass, then the plain in-place transform)
      bool gather = option(c, "SLOTS_GATHER_NTT", 1) != 0;
      const uint32_t nt = c->N >> ntt_log_ept((int)c->logN);    // threads of a transform workgroup
      for (uint32_t r = 0; r <= n_ranks; ++r) gather = gather && map.cut[r] % nt == 0;
      if (gather)
        HIP_TRY(c->ops->ntt_inv_gather(ln.stream, c->mode, c->dp, c->k, device_rowsums, ln.lvl[c->d - 1], map, RC, B, count, j0));
      else
        HIP_TRY(launch_slots_assemble(ln.stream, device_rowsums, ln.lvl[c->d - 1], map, RC, kN, B, words, count, j0));
      uint64_t* lvl_ptrs[PIRGPU_MAX_DIMS];
      for (uint32_t l = 0; l < c->d; ++l) lvl_ptrs[l] = ln.lvl[l];
      lvl_ptrs[0] = device_replies + (size_t)j0 * rwords;
      Stage sg{ln.stream, lvl_ptrs, ln.pt_buf, B, MfmaPtrs{}, false, &ln.up_scratch, &ln.up_scratch_words};
      for (uint32_t q = 0; q < B; ++q) sg.sel.p[q] = device_sv + (size_t)(j0 + q) * svwords;
      sg.sel_f64 = c->sel_f64;
      sg.rows_inverted = gather;
      post_scan_stage(c, sg, nullptr);
      lane_then(ln, then);
    }
    return PIRGPU_OK;
  });
}

int pirgpu_batch_reply_copy_to_device(pirgpu_ctx* c, uint64_t* dst, uint64_t cap) {
  return guarded(c, [&]() -> int {
    if (!c->bs().batch_valid) return fail(c, PIRGPU_FAILED_PRECONDITION, "no batch has been run");
    const uint64_t total = (uint64_t)c->bs().batch_count * c->reply_cts;
    if (!dst || cap < total) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    sync_batch_streams(c);
    // a device-to-device hipMemcpy on the null stream may return before the copy has run, and the context's
    // streams are non-blocking (not ordered with the null stream): copy on the context's stream and wait
    HIP_TRY(hipMemcpyAsync(dst, reply_base(c), total * c->ctw * 8, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_batch_reply_copy_to_device_async(pirgpu_ctx* c, uint64_t* dst, uint64_t cap) {
  int rc = pirgpu_join(c);   // the main stream now follows every lane: the copy below sees the finished batch
  if (rc) return rc;
  return guarded(c, [&]() -> int {
    if (!c->bs().batch_valid) return fail(c, PIRGPU_FAILED_PRECONDITION, "no batch has been run");
    const uint64_t total = (uint64_t)c->bs().batch_count * c->reply_cts;
    if (!dst || cap < total) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    HIP_TRY(hipMemcpyAsync(dst, reply_base(c), total * c->ctw * 8, hipMemcpyDeviceToDevice, c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_batch_fetch(pirgpu_ctx* c, uint64_t* replies, uint64_t cap, uint64_t* count) {
  return guarded(c, [&]() -> int {
    if (!c->bs().batch_valid) return fail(c, PIRGPU_FAILED_PRECONDITION, "no batch has been run");
    const uint64_t total = (uint64_t)c->bs().batch_count * c->reply_cts;
    if (!replies || cap < total) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
    sync_batch_streams(c);
    if (!(c->bs().host_reply_done && replies == c->bs().host_reply))   // else: the groups downloaded their replies themselves
      HIP_TRY(hipMemcpy(replies, reply_base(c), total * c->ctw * 8, hipMemcpyDeviceToHost));
    if (count) *count = total;
    return PIRGPU_OK;
  });
}

static int ntt_hook(pirgpu_ctx* c, uint64_t* polys, uint64_t count, int key_level, bool inverse) {
  return guarded(c, [&]() -> int {
    if (!polys && count) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    const uint32_t per = key_level ? c->k + 1 : c->k;
    const uint64_t npoly = key_level ? count * per : count * 2 * per;
    // the hook speaks SEAL's NTT order at the boundary; the kernels use device order
    uint64_t* dev = nullptr;
    HIP_TRY(hipMalloc((void**)&dev, 2 * std::max<uint64_t>(npoly, 1) * c->N * 8));
    uint64_t* tmp = dev + std::max<uint64_t>(npoly, 1) * c->N;
    try {
      if (inverse) {
        HIP_TRY(hipMemcpyAsync(tmp, polys, npoly * c->N * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(launch_ntt_reorder(c->stream, c->N, tmp, dev, npoly, true, false));
        HIP_TRY(c->ops->ntt_batch(c->stream, c->mode, c->dp, dev, npoly, per, 0, true));
        HIP_TRY(hipMemcpyAsync(polys, dev, npoly * c->N * 8, hipMemcpyDeviceToHost, c->stream));
      } else {
        HIP_TRY(hipMemcpyAsync(dev, polys, npoly * c->N * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c->ops->ntt_batch(c->stream, c->mode, c->dp, dev, npoly, per, 0, false));
        HIP_TRY(launch_ntt_reorder(c->stream, c->N, dev, tmp, npoly, false, false));
        HIP_TRY(hipMemcpyAsync(polys, tmp, npoly * c->N * 8, hipMemcpyDeviceToHost, c->stream));
      }
      HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (...) {
      (void)hipFree(dev);
      throw;
    }
    HIP_TRY(hipFree(dev));
    return PIRGPU_OK;
  });
}

int pirgpu_ntt_forward(pirgpu_ctx* c, uint64_t* polys, uint64_t count, int key_level) {
  return ntt_hook(c, polys, count, key_level, false);
}
int pirgpu_ntt_inverse(pirgpu_ctx* c, uint64_t* polys, uint64_t count, int key_level) {
  return ntt_hook(c, polys, count, key_level, true);
}

int pirgpu_reduce_fixup_device(pirgpu_ctx* c, uint64_t* device_ptr, uint64_t count) {
  return guarded(c, [&]() -> int {
    if (!device_ptr) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    HIP_TRY(launch_reduce_splits(c->stream, c->dp, device_ptr, 1, count * c->ctw, device_ptr));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PIRGPU_OK;
  });
}

int pirgpu_reduce_fixup_device_async(pirgpu_ctx* c, uint64_t* device_ptr, uint64_t count, void* stream) {
  return guarded(c, [&]() -> int {
    if (!device_ptr) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    HIP_TRY(launch_reduce_splits(st, c->dp, device_ptr, 1, count * c->ctw, device_ptr));
    return PIRGPU_OK;
  });
}

// Row selectors of the multi-GPU exchange in 5 bytes per residue (moduli below 2^40): pack before the all-to-all,
// unpack behind it.  words: multiple of 4; `packed` holds words * 5 / 4 dwords.  stream NULL: the main stream.
static bool moduli_fit_40_bits(const pirgpu_ctx* c) {
  for (uint32_t j = 0; j < c->k; ++j)
    if (c->hp.mod[j].q >> 40) return false;
  return true;
}
int pirgpu_pack40_supported(pirgpu_ctx* c) { return c && moduli_fit_40_bits(c) ? 1 : 0; }

int pirgpu_pack40_device_async(pirgpu_ctx* c, const uint64_t* words_in, uint32_t* packed, uint64_t words, void* stream) {
  return guarded(c, [&]() -> int {
    if (!words_in || !packed || words % 4) return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid buffers for the 5-byte form");
    if (!moduli_fit_40_bits(c)) return fail(c, PIRGPU_FAILED_PRECONDITION, "the 5-byte form needs moduli below 2^40");
    HIP_TRY(launch_pack40x4(stream ? (hipStream_t)stream : c->stream, words_in, packed, words));
    return PIRGPU_OK;
  });
}

int pirgpu_unpack40_device_async(pirgpu_ctx* c, const uint32_t* packed, uint64_t* words_out, uint64_t words, void* stream) {
  return guarded(c, [&]() -> int {
    if (!words_out || !packed || words % 4) return fail(c, PIRGPU_INVALID_ARGUMENT, "invalid buffers for the 5-byte form");
    if (!moduli_fit_40_bits(c)) return fail(c, PIRGPU_FAILED_PRECONDITION, "the 5-byte form needs moduli below 2^40");
    HIP_TRY(launch_unpack40x4(stream ? (hipStream_t)stream : c->stream, packed, words_out, words));
    return PIRGPU_OK;
  });
}

// ======================================================================================================================
// [10] MEASUREMENT: phase timings, batch scan timings
// ======================================================================================================================
int pirgpu_set_profiling(pirgpu_ctx* c, int enabled) {
  return guarded(c, [&]() -> int {
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->prof = enabled != 0;
    c->prof_runs = 0;
    c->bscan_n = 0;
    return PIRGPU_OK;
  });
}

int pirgpu_last_timings(pirgpu_ctx* c, float ms[6], uint32_t* runs) {
  return guarded(c, [&]() -> int {
    if (!ms) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    if (c->prof_runs > 0) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      // events per run: [EXPAND] start, [SVNTT] end of expansion + selector NTT, [SCAN] scan start,
      //                 [UPPER] scan end, [FINAL] last multiply-accumulate end, [COUNT] end of run
      double acc[6] = {0, 0, 0, 0, 0, 0};
      for (int r = 0; r < c->prof_runs; ++r) {
        hipEvent_t* e = &c->ev[(size_t)r * (PH_COUNT + 1)];
        float t;
        HIP_TRY(hipEventElapsedTime(&t, e[PH_EXPAND], e[PH_SCAN]));
        acc[0] += t;
        HIP_TRY(hipEventElapsedTime(&t, e[PH_SCAN], e[PH_UPPER]));
        acc[2] += t;
        HIP_TRY(hipEventElapsedTime(&t, e[PH_UPPER], e[PH_FINAL]));
        acc[3] += t;
        HIP_TRY(hipEventElapsedTime(&t, e[PH_FINAL], e[PH_COUNT]));
        acc[4] += t;
        HIP_TRY(hipEventElapsedTime(&t, e[PH_EXPAND], e[PH_COUNT]));
        acc[5] += t;
      }
      for (int i = 0; i < 6; ++i) c->timings[i] = (float)(acc[i] / c->prof_runs);
      if (runs) *runs = (uint32_t)c->prof_runs;
      c->prof_runs = 0;
    } else if (runs) {
      *runs = 0;
    }
    memcpy(ms, c->timings, sizeof(c->timings));
    return PIRGPU_OK;
  });
}

int pirgpu_batch_scan_timings(pirgpu_ctx* c, float* mean_ms, float* min_ms, uint32_t* launches, uint32_t* workgroups,
                              uint32_t* queries) {
  return guarded(c, [&]() -> int {
    if (!mean_ms || !min_ms || !launches) return fail(c, PIRGPU_INVALID_ARGUMENT, "null buffer");
    sync_batch_streams(c);
    double sum = 0, mn = 1e30;
    for (uint32_t i = 0; i < c->bscan_n; ++i) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, c->bscan_ev[2 * i], c->bscan_ev[2 * i + 1]));
      sum += t;
      mn = std::min<double>(mn, t);
    }
    *launches = c->bscan_n;
    *mean_ms = c->bscan_n ? (float)(sum / c->bscan_n) : 0.f;
    *min_ms = c->bscan_n ? (float)mn : 0.f;
    if (workgroups) *workgroups = c->bscan_wgs;
    if (queries) *queries = c->bscan_nq;
    c->bscan_n = 0;
    return PIRGPU_OK;
  });
}

uint32_t pirgpu_get_concurrency(pirgpu_ctx* c) { return c ? c->n_active : 0; }

// pirgpu_free: wire.cpp (response buffers are recycled)

}  // extern "C"
