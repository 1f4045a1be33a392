// wire.cpp -- PIRServer::ProcessRequest at wire level (reference server.cpp:44-65):
// pir/proto/payload.proto framing (proto3, hand-rolled: libprotobuf is not in this
// image) around SEAL 3.5.6 binary objects (serialization.h:81-139 -> Ciphertext /
// GaloisKeys ::save/load, format restated in SURVEY.md App. A.6).
//
// Status of the SEAL object codec: written from the published 3.5.6 layout and NOT verifiable against
// a real SEAL build in this image (tools/check_external_pair.py replays a SEAL-produced Request/Response
// pair when a SEAL machine provides one).  Both object forms are accepted: fully expanded
// (galois_keys_local + SaveRequest, as server_test.cpp builds them) and seed-compressed Serializable<>
// objects (what PIRClient::initialize sends for its keys, client.cpp:47-54; the c1 halves are re-sampled
// with SEAL's BlakePRNG + sample_poly_uniform restated in wire_codec.cpp).
#include "wire.h"
#include "wire_codec.h"
#include "env_gate.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pirgpu.h"

namespace pirgpu {
namespace wire {

struct ParsedRequest {
  std::vector<std::pair<const uint8_t*, size_t>> queries;  // Ciphertexts sub-messages
  const uint8_t* galois_keys = nullptr;
  size_t galois_keys_len = 0;
  const uint8_t* relin_keys = nullptr;
  size_t relin_keys_len = 0;
};

// pir.Request (payload.proto:27-36)
static ParsedRequest parse_request(const uint8_t* request, size_t request_len) {
  ParsedRequest pr;
  Reader r{request, request + request_len};
  while (r.p < r.end) {
    uint64_t tag;
    if (!r.varint(tag)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Request"};
    const uint32_t field = (uint32_t)(tag >> 3), wt = (uint32_t)(tag & 7);
    const uint8_t* d;
    size_t l;
    if (field == 1 && wt == 2) {
      if (!r.bytes(d, l)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Request.query"};
      pr.queries.emplace_back(d, l);
    } else if (field == 2 && wt == 2) {
      if (!r.bytes(pr.galois_keys, pr.galois_keys_len))
        throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Request.galois_keys"};
    } else if (field == 3 && wt == 2) {
      if (!r.bytes(pr.relin_keys, pr.relin_keys_len))
        throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Request.relin_keys"};
    } else if (!r.skip(wt)) {
      throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Request"};
    }
  }
  return pr;
}

}  // namespace wire
}  // namespace pirgpu

using namespace pirgpu::wire;

extern "C" {

// Exposed for tests of the codec (declared in wire.h, not part of pirgpu.h).
void pirgpu_wire_parms_id(uint32_t N, const uint64_t* moduli, size_t n_moduli, uint64_t t, uint64_t out[4]) {
  parms_id(N, moduli, n_moduli, t, out);
}
void pirgpu_wire_blake2b(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen) {
  blake2b(out, outlen, in, inlen);
}

void pirgpu_wire_blake2xb(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen, const uint8_t* key,
                          size_t keylen) {
  blake2xb(out, outlen, in, inlen, key, keylen);
}
// SEAL's seeded uniform sampler (BlakePRNG + sample_poly_uniform): seed[64] -> out[n_moduli][N]
void pirgpu_wire_sample_poly_uniform(const uint8_t* seed, const uint64_t* moduli, uint32_t n_moduli, uint32_t N,
                                     uint64_t* out) {
  SealPrng rng(seed);
  sample_poly_uniform(rng, moduli, n_moduli, N, out);
}
// KSwitchKeys::load of a serialized GaloisKeys / RelinKeys object (expanded or seed-compressed) without a device:
// copies the key stored at `index` (Galois element 2*index+1; RelinKeys: 0) to out [k][2][k+1][N].
// 0, or the status the server would fail with; NotFound (5) if the object holds no key at that index.
int pirgpu_wire_load_kswitch_key(const pirgpu_params* params, const uint8_t* blob, size_t len, uint64_t index,
                                 uint64_t* out) {
  if (!params || !blob || !out) return PIRGPU_INVALID_ARGUMENT;
  try {
    const Shape sh = make_shape(*params);
    const size_t words = (size_t)sh.k * 2 * (sh.k + 1) * sh.N;
    bool found = false;
    load_kswitch_keys(sh, blob, len, [&](uint64_t i, const uint64_t* key) {
      if (i == index) {
        memcpy(out, key, words * 8);
        found = true;
      }
    });
    return found ? PIRGPU_OK : 5;
  } catch (const Err& e) {
    return e.code;
  } catch (const std::exception&) {
    return PIRGPU_INTERNAL;
  }
}

// Parses and validates a serialized pir.Request against `params` exactly as pirgpu_process_request
// does (framing, SEAL headers, parms_id, shapes, coefficient ranges) without touching a device:
// 0 if it would be accepted, else the status code it would fail with.  *n_queries = number of queries.
int pirgpu_wire_validate_request(const pirgpu_params* params, const uint8_t* request, size_t request_len,
                                 uint32_t* n_queries) {
  if (!params || (!request && request_len)) return PIRGPU_INVALID_ARGUMENT;
  try {
    const Shape sh = make_shape(*params);
    ParsedRequest pr = parse_request(request, request_len);
    load_kswitch_keys(sh, pr.galois_keys, pr.galois_keys_len, nullptr);
    if (pr.relin_keys_len) load_kswitch_keys(sh, pr.relin_keys, pr.relin_keys_len, nullptr);
    std::vector<uint64_t> qbuf;
    for (auto& qm : pr.queries) (void)load_query(sh, qm.first, qm.second, qbuf);
    if (n_queries) *n_queries = (uint32_t)pr.queries.size();
    return PIRGPU_OK;
  } catch (const Err& e) {
    return e.code;
  } catch (const std::exception&) {
    return PIRGPU_INTERNAL;
  }
}

// Largest number of queries handed to the batch pipeline at once: bounds the device memory one (untrusted)
// request can claim (query + reply staging grow with the batch); longer requests run as several batches.
static const uint32_t kMaxRequestBatch = 64;

}  // extern "C"

namespace {

// One pir.Request being served.  Requests of different clients that are in flight at the same time (threads calling
// pirgpu_process_request concurrently, or one pirgpu_process_requests call) are served TOGETHER: their queries go through
// the batch pipeline as one batch -- grouped expansion with every query switched by its own client's Galois keys, one
// database pass per group of 8 -- instead of one after the other at the single-query rate.
// Response buffers are RECYCLED: a buffer handed back with pirgpu_free goes to a process-wide pool (up to 256 MB) and
// the next response of about that size is serialised into it.  A megabyte per reply from malloc is an mmap, 256 page
// faults while it is filled and a munmap (with TLB shoot-downs across the process' threads) when the caller frees it:
// 0.1 ms per response for the free alone, 6.4 ms of a 64-client window.  Every buffer carries a 64-byte header
// (magic, capacity) in front of what the caller sees.
struct PoolHdr {
  uint64_t magic;
  size_t cap;     // usable bytes behind the header
};
constexpr size_t kPoolHdr = 64;
constexpr uint64_t kPoolMagic = 0x7069726770756221ull;
constexpr size_t kPoolLimit = 256u << 20;
std::mutex g_pool_mu;
std::vector<uint8_t*> g_pool;     // headers of free buffers
size_t g_pool_bytes = 0;

uint8_t* pool_alloc(size_t want, size_t* cap) {   // -> user pointer
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); ++i) {
      const size_t c = reinterpret_cast<PoolHdr*>(g_pool[i])->cap;
      if (c >= want && c <= 2 * want + 4096 && (best == g_pool.size() || c < reinterpret_cast<PoolHdr*>(g_pool[best])->cap))
        best = i;
    }
    if (best != g_pool.size()) {
      uint8_t* base = g_pool[best];
      g_pool[best] = g_pool.back();
      g_pool.pop_back();
      g_pool_bytes -= reinterpret_cast<PoolHdr*>(base)->cap;
      *cap = reinterpret_cast<PoolHdr*>(base)->cap;
      return base + kPoolHdr;
    }
  }
  uint8_t* base = (uint8_t*)malloc(want + kPoolHdr);
  if (!base) throw std::bad_alloc();
  PoolHdr* h = reinterpret_cast<PoolHdr*>(base);
  h->magic = kPoolMagic;
  h->cap = want;
  *cap = want;
  return base + kPoolHdr;
}

void pool_release(uint8_t* user) {
  if (!user) return;
  uint8_t* base = user - kPoolHdr;
  PoolHdr* h = reinterpret_cast<PoolHdr*>(base);
  if (h->magic != kPoolMagic) {   // not one of ours (cannot happen through the documented API)
    free(user);
    return;
  }
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    if (h->cap >= 4096 && g_pool_bytes + h->cap <= kPoolLimit && g_pool.size() < 4096) {
      g_pool.push_back(base);
      g_pool_bytes += h->cap;
      return;
    }
  }
  h->magic = 0;
  free(base);
}

// A response being assembled: ONE buffer, sized up front, that finish() hands to the caller as it is (round 3 first built
// a std::string and copied it -- two passes over a megabyte of fresh pages per reply).
struct OutBuf {
  uint8_t* p = nullptr;   // user pointer (pool_alloc)
  size_t n = 0, cap = 0;
  OutBuf() = default;
  OutBuf(const OutBuf&) = delete;
  OutBuf& operator=(const OutBuf&) = delete;
  ~OutBuf() { pool_release(p); }
  void reserve(size_t want) {
    if (want <= cap) return;
    size_t got = 0;
    uint8_t* q = pool_alloc(std::max(want, cap + cap / 2), &got);
    if (n) memcpy(q, p, n);
    pool_release(p);
    p = q;
    cap = got;
  }
  void append(const void* src, size_t len) {
    reserve(n + len);
    memcpy(p + n, src, len);
    n += len;
  }
  void clear() { n = 0; }
  uint8_t* release(size_t* len) {   // never null: an empty response is a valid (zero-length) message
    if (!p) reserve(1);
    uint8_t* r = p;
    *len = n;
    p = nullptr;
    n = cap = 0;
    return r;
  }
};

struct RelinCache;
struct Job {
  const uint8_t* request = nullptr;
  size_t request_len = 0;
  int rc = 0;
  std::string err;
  OutBuf out;                 // serialized pir.Response
  bool done = false;
  // filled while serving
  ParsedRequest pr;
  uint32_t slot = 0;          // resident key set of this client
  bool uniform = true;        // every query has the ciphertext count the dimensions call for (server.cpp:154)
  bool unverified = false;    // the slot was taken on its fingerprint alone: the key bytes are compared under the GPU work
  bool mismatch = false;      // ... and differed: install this client's keys and serve the request again
  bool relin_checked = false; // the RelinKeys field (if any) has been validated
  int pins = 0;               // times `slot` is pinned for this request (pirgpu_keyset_pin; once per window in flight
                              // that holds queries of it): unpinned as the windows finish
};

struct Server {               // what serving needs to know about a context
  pirgpu_ctx* ctx;
  pirgpu_params prm;
  Shape sh;
  size_t ctw;
  uint64_t n_reply;
  uint32_t nq_expected;
  struct RelinCache* relin = nullptr;   // per context (Combiner)
  size_t relin_keep = 256;
};

// Response.reply (payload.proto:39-42) for one query: n ciphertexts, written straight into the response buffer
// Upper bound of what append_reply adds for a reply of n ciphertexts (tags and varint lengths: < 32 bytes each)
size_t reply_bytes_bound(const Shape& sh, uint64_t n) { return 32 + n * (32 + saved_ciphertext_size(sh)); }

// `ready(i)` (optional) is called before ciphertext i is copied: the caller can wait there for the part of the download
// that holds it.
void append_reply(OutBuf& out, const Shape& sh, const uint64_t* cts_words, uint64_t n, size_t ctw,
                  const std::function<void(uint64_t)>& ready = nullptr) {
  const size_t ct_size = saved_ciphertext_size(sh);
  std::string per_ct_head;                                    // Ciphertexts.ct = 1: tag + length + the object's prefix
  per_ct_head.push_back((char)((1 << 3) | 2));
  put_varint(per_ct_head, ct_size);
  const size_t per_ct = per_ct_head.size() + ct_size;         // tag + length + object
  append_ciphertext_prefix(per_ct_head, sh);
  std::string reply_head;
  reply_head.push_back((char)((1 << 3) | 2));                 // Response.reply = 1, length-delimited
  put_varint(reply_head, n * per_ct);
  out.reserve(out.n + reply_head.size() + n * per_ct);
  out.append(reply_head.data(), reply_head.size());
  for (uint64_t i = 0; i < n; ++i) {
    if (ready) ready(i);
    out.append(per_ct_head.data(), per_ct_head.size());
    out.append(cts_words + i * ctw, ctw * 8);
  }
}

// ---------------------------------------------------------------------------------------------- worker pool
//
// Host threads the wire layer uses next to the serving thread (key compare, query parsing, seed expansion, response
// serialisation): created once per process, on first use, and kept -- round 3 started fresh std::async threads in every
// window (three places), a thread creation + join per task on the request path.  The singleton is never destroyed
// (its threads sleep on a condition variable; the process ends under them).
class Pool {
 public:
  static Pool& get() {
    static Pool* p = new Pool();
    return *p;
  }
  size_t size() const { return threads_.size(); }

  // A set of tasks somebody waits for.
  struct Group {
    std::mutex m;
    std::condition_variable cv;
    size_t pending = 0;
    void wait() {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return pending == 0; });
    }
  };

  void submit(Group& g, std::function<void()> fn) {
    {
      std::lock_guard<std::mutex> lk(g.m);
      ++g.pending;
    }
    try {
      std::lock_guard<std::mutex> lk(m_);
      q_.push_back(Task{&g, std::move(fn)});
    } catch (...) {   // the push failed (allocation): nothing was queued, so nothing will ever count this task down
      std::lock_guard<std::mutex> lk(g.m);
      --g.pending;
      throw;
    }
    cv_.notify_one();
  }

  // fn(i) for every i < n on up to max_threads threads, the caller among them; returns when all are done.
  // fn must not throw.
  void parallel_for(size_t n, size_t max_threads, const std::function<void(size_t)>& fn) {
    if (n == 0) return;
    const size_t helpers = std::min(std::min(max_threads, n) - 1, size());
    std::atomic<size_t> next{0};
    auto loop = [&] {
      for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
    };
    Group g;
    struct Waiter {   // whatever unwinds past here, the queued helpers are done with `g`, `next` and `fn` first
      Group& g;
      ~Waiter() { g.wait(); }
    } waiter{g};
    for (size_t t = 0; t < helpers; ++t) submit(g, loop);
    loop();
  }

 private:
  struct Task {
    Group* g;
    std::function<void()> fn;
  };
  Pool() {
    const size_t hw = std::thread::hardware_concurrency();
    size_t n = std::max<size_t>(1, std::min<size_t>(16, hw ? hw / 2 : 1));   // never more than half the machine
    if (const char* v = pirgpu_env("PIRGPU_WIRE_THREADS")) n = std::max<size_t>(1, std::min<size_t>(64, (size_t)atoi(v)));
    for (size_t i = 0; i < n; ++i) {
      threads_.emplace_back([this] { run(); });
      threads_.back().detach();
    }
  }
  void run() {
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        t = std::move(q_.front());
        q_.pop_front();
      }
      try {
        t.fn();
      } catch (...) {   // tasks report through their own state; nothing may unwind a pool thread
      }
      {
        std::lock_guard<std::mutex> lk(t.g->m);
        if (--t.g->pending == 0) t.g->cv.notify_all();
      }
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::deque<Task> q_;
  std::vector<std::thread> threads_;
};

// SEALDeserialize<RelinKeys> (server.cpp:53-58): only CT-multiplication mode uses the keys, but a malformed non-empty
// field is InvalidArgument in the reference, so the object is parsed and validated here too.  The reference client sends
// its RelinKeys with EVERY request (client.cpp:80-90) -- seed-compressed, i.e. 0.2-0.4 ms of BLAKE2Xb re-sampling per
// request at N = 4096 -- so the verdict is remembered per resident key set: a request whose relin bytes equal the ones
// already validated for its client's set skips the parse (a memcmp of 0.2 MB instead).  Keyed by the key set HANDLE
// (slot + generation): an evicted set's entry can never be hit by the next tenant.
struct RelinCache {
  std::mutex m;
  std::map<uint32_t, std::shared_ptr<const std::vector<uint8_t>>> ok;   // handle -> validated bytes
  bool known_good(uint32_t slot, const uint8_t* p, size_t n) {
    std::shared_ptr<const std::vector<uint8_t>> have;
    {
      std::lock_guard<std::mutex> lk(m);
      auto it = ok.find(slot);
      if (it == ok.end()) return false;
      have = it->second;
    }
    return have->size() == n && memcmp(have->data(), p, n) == 0;
  }
  void remember(uint32_t slot, const uint8_t* p, size_t n, size_t keep) {
    auto copy = std::make_shared<const std::vector<uint8_t>>(p, p + n);
    std::lock_guard<std::mutex> lk(m);
    if (ok.size() >= keep) ok.clear();   // bounded: at most a few entries per key set slot
    ok[slot] = std::move(copy);
  }
};

struct Trace {   // PIRGPU_WIRE_TRACE=1: host-side phase times of a window, to stderr
  bool on = pirgpu_env("PIRGPU_WIRE_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what, int window = -1) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[wire w%-2d] %-22s %8.3f ms\n", window, what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

void fail_job(Job& job, int code, const std::string& msg) {
  job.rc = code;
  job.err = msg;
  job.out.clear();
}

// RAII for the request-level critical section of the context (recursive with the per-call lock of the ABI entry points)
struct CtxLock {
  pirgpu_ctx* c;
  explicit CtxLock(pirgpu_ctx* ctx) : c(ctx) { pirgpu_request_lock(c); }
  ~CtxLock() { pirgpu_request_unlock(c); }
  CtxLock(const CtxLock&) = delete;
  CtxLock& operator=(const CtxLock&) = delete;
};

// SEALDeserialize<GaloisKeys> (server.cpp:46-48) with the device-resident key cache: a client that
// repeats its (multi-MB) key object byte for byte finds its keys resident; a new client's keys are parsed -- the
// objects of a seed-compressed Serializable<GaloisKeys>, what the reference client sends (client.cpp:47-54), are
// re-expanded on the pool's threads --, validated as a whole (a malformed object must not leave half-installed keys
// behind) and uploaded into a free or the least recently used slot.  `speculative`: accept a resident set on its
// fingerprint alone -- the caller verifies the bytes while the GPU already works and discards the result on a
// mismatch.  The slot comes back PINNED (pirgpu_keyset_pin): it cannot be evicted until the caller unpins it.
void resolve_keys(const Server& sv, Job& job, bool speculative, bool* unverified) {
  if (unverified) *unverified = false;
  uint32_t slot = 0;
  int rc = 0;
  CtxLock lock(sv.ctx);   // lookup + pin (or claim + upload + pin) are one step with respect to other windows
  if (job.pr.galois_keys_len) {
    rc = pirgpu_keyset_lookup(sv.ctx, job.pr.galois_keys, job.pr.galois_keys_len, speculative ? 0 : 1, &slot);
    if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  }
  if (slot) {
    rc = pirgpu_keyset_pin(sv.ctx, slot);
    if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
    if (unverified) *unverified = speculative;
    job.slot = slot;
    job.pins = 1;
    return;
  }
  // empty bytes -> load throws -> InvalidArgument, like the reference
  std::vector<std::pair<uint64_t, std::vector<uint64_t>>> parsed;
  load_kswitch_keys_parallel(sv.sh, job.pr.galois_keys, job.pr.galois_keys_len,
                             [](size_t n, const std::function<void(size_t)>& fn) { Pool::get().parallel_for(n, 16, fn); },
                             parsed);
  rc = pirgpu_keyset_claim(sv.ctx, job.pr.galois_keys, job.pr.galois_keys_len, &slot);
  if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  {
    std::vector<uint32_t> elts;
    std::vector<const uint64_t*> ptrs;
    for (auto& kv : parsed) {
      elts.push_back((uint32_t)(2 * kv.first + 1));
      ptrs.push_back(kv.second.data());
    }
    rc = pirgpu_keyset_set_keys(sv.ctx, slot, (uint32_t)elts.size(), elts.data(), ptrs.data());   // one wait for the set
    if (rc) {
      const std::string msg = pirgpu_last_error(sv.ctx);
      (void)pirgpu_keyset_release(sv.ctx, slot);
      throw Err{rc, msg};
    }
  }
  rc = pirgpu_keyset_pin(sv.ctx, slot);
  if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  job.slot = slot;
  job.pins = 1;
}

// The request's RelinKeys field, if any: validated (or recognised as already validated for this client's key set).
// Throws Err like the loader.  Runs on pool threads / under the GPU's work: only the job's status depends on it.
void check_relin_keys(const Server& sv, const Job& job) {
  if (!job.pr.relin_keys_len) return;   // empty = absent (server.cpp:53)
  if (sv.relin && job.slot && sv.relin->known_good(job.slot, job.pr.relin_keys, job.pr.relin_keys_len)) return;
  load_kswitch_keys(sv.sh, job.pr.relin_keys, job.pr.relin_keys_len, nullptr);
  if (sv.relin && job.slot) sv.relin->remember(job.slot, job.pr.relin_keys, job.pr.relin_keys_len, sv.relin_keep);
}

void unpin_job(const Server& sv, Job& job) {
  if (job.pins > 0) {
    (void)pirgpu_keyset_unpin(sv.ctx, job.slot);
    --job.pins;
  }
}

// One query through the single-query path (lowest latency; also reports a wrong ciphertext count at the offending
// query like the reference, server.cpp:154-158).  Query parsed into pinned staging, reply serialised out of it.
// `while_running` (optional) is host work done between enqueueing the kernels and waiting for the reply.
// The caller holds the context's request lock: worker 0 and the single-query selection are context state.
void run_single(const Server& sv, Job& job, const std::pair<const uint8_t*, size_t>& qm,
                const std::function<void()>& while_running = nullptr) {
  uint64_t* hq = pirgpu_host_query_buffer(sv.ctx, 1);
  uint64_t* hr = pirgpu_host_reply_buffer(sv.ctx, 1);
  if (!hq || !hr) throw Err{PIRGPU_INTERNAL, pirgpu_last_error(sv.ctx)};
  const uint32_t nq = load_query_into(sv.sh, qm.first, qm.second, hq, sv.nq_expected);
  // the direct API's selection survives a request exactly as it was -- a selection that had gone stale stays stale
  // (round-tripping it through a handle would have re-validated it against the slot's next tenant)
  uint32_t callers_sel[2];
  pirgpu_keyset_selection_get(sv.ctx, callers_sel);
  int rc = pirgpu_query_use_keyset(sv.ctx, job.slot);
  uint64_t got = 0;
  if (!rc) rc = pirgpu_query_stage_async(sv.ctx, hq, nq);   // pinned staging, untouched until the fetch below returns
  if (!rc) rc = pirgpu_query_run(sv.ctx);      // asynchronous: every kernel of the path is queued
  const std::string msg = rc ? pirgpu_last_error(sv.ctx) : "";
  pirgpu_keyset_selection_set(sv.ctx, callers_sel);
  if (rc) throw Err{rc, msg};
  if (while_running) {
    try {
      while_running();
    } catch (...) {
      // the kernels are queued and read this request's key set and pinned staging: nobody may unpin / reuse either
      // before they are done
      (void)pirgpu_sync(sv.ctx);
      throw;
    }
  }
  // the reply comes back in two halves queued together: the first is serialised while the second crosses PCIe
  uint64_t first_part = 0;
  rc = pirgpu_query_fetch_begin(sv.ctx, hr, sv.n_reply, &got, &first_part);
  if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  int wrc = 0;
  append_reply(job.out, sv.sh, hr, got, sv.ctw, [&](uint64_t i) {
    if (i == 0 && !wrc) wrc = pirgpu_query_fetch_wait(sv.ctx, 0);
    if (i == first_part && !wrc) wrc = pirgpu_query_fetch_wait(sv.ctx, 1);
  });
  if (!wrc && first_part >= got) wrc = pirgpu_query_fetch_wait(sv.ctx, 1);
  if (wrc) throw Err{wrc, pirgpu_last_error(sv.ctx)};
}

// One WINDOW of requests on its way through the batch pipeline: up to kMaxRequestBatch queries of at most
// `max_clients` requests, staged in one of the context's batch sets.  begin() does everything up to "the window's
// kernels are queued"; finish() everything from there to "its responses are serialised".  Two windows can be in flight
// on one context -- the next one parsed, staged and queued (by the same serving thread or by another one) while the
// previous one's groups are still being computed and its replies are on their way back -- so the GPU's queues do not
// run dry between windows: a window of 64 clients was 15.3 ms of which the GPU was busy 12 (round 3).
struct Item {
  Job* job;
  uint32_t qi;
};

struct Window {
  int set = 0;                       // batch set of the context (pirgpu_batch_select)
  int home = 0;                      // the serving thread's own set, selected again when begin / finish return
  int id = 0;                        // for the trace
  std::vector<Job*> jobs;            // requests with queries in this window (a request's queries never span windows
                                     // unless it has more than kMaxRequestBatch of them)
  std::vector<Item> chunk;           // the queries queued, in reply order
  std::vector<std::pair<uint32_t, uint32_t>> runs;   // [first, end) items of one request
  std::vector<uint32_t> slots;
  uint64_t *hq = nullptr, *hr = nullptr;
  uint32_t room = 0;
  bool queued = false;               // the batch pipeline holds this window's work
  int rc = 0;
  std::string err;
};

template <typename F>
struct Finally {
  F f;
  ~Finally() { f(); }
};
template <typename F>
Finally<F> finally(F f) { return Finally<F>{std::move(f)}; }

// Byte-for-byte compare of the key objects of the window's requests with the resident sets they were matched to by
// fingerprint -- once per request, on the pool's threads (4.7 MB per client: 0.3-0.4 ms each on one thread, which was
// most of a request's host time), while the GPU runs the window just queued.  The resident copies are read without the
// context's lock: the slots are pinned (no eviction, no reinstall).
void verify_keys_of(const Server& sv, Window& w) {
  std::vector<Job*> todo;
  for (const Item& it : w.chunk)
    if ((it.job->unverified || (it.job->pr.relin_keys_len && !it.job->relin_checked)) &&
        std::find(todo.begin(), todo.end(), it.job) == todo.end())
      todo.push_back(it.job);
  if (todo.empty()) return;
  std::vector<std::pair<const uint8_t*, size_t>> resident(todo.size());
  for (size_t i = 0; i < todo.size(); ++i)
    if (todo[i]->unverified) resident[i].second = pirgpu_keyset_blob(sv.ctx, todo[i]->slot, &resident[i].first);
  // (64 clients' objects are 600 MB to read, more than the host's last-level cache: memory-bound, so many threads)
  std::vector<Err> relin_err(todo.size(), Err{0, ""});
  Pool::get().parallel_for(todo.size(), 16, [&](size_t i) {
    Job* job = todo[i];
    if (job->unverified) {
      const uint8_t* r = resident[i].first;
      const size_t len = resident[i].second;
      job->mismatch = !(r && len == job->pr.galois_keys_len && memcmp(r, job->pr.galois_keys, len) == 0);
      job->unverified = false;
    }
    if (job->pr.relin_keys_len && !job->relin_checked && !job->mismatch) {
      try {
        check_relin_keys(sv, *job);
        job->relin_checked = true;
      } catch (const Err& e) {
        relin_err[i] = e;
      } catch (const std::exception& e) {
        relin_err[i] = Err{PIRGPU_INTERNAL, e.what()};
      }
    }
  });
  for (size_t i = 0; i < todo.size(); ++i)
    if (relin_err[i].code && !todo[i]->rc) fail_job(*todo[i], relin_err[i].code, relin_err[i].msg);   // its replies are dropped
}

// Parses the queries of `items` into the window's pinned staging (on the pool's threads: 128 KiB + a range check per
// ciphertext), drops what failed, maps the response buffers, stages the queries asynchronously and queues the window.
void begin_window(const Server& sv, Window& w, const std::vector<Item>& items, Trace& trace) {
  const size_t qwords = (size_t)sv.nq_expected * sv.ctw;
  (void)pirgpu_batch_select(sv.ctx, (uint32_t)w.set);
  auto unselect = finally([&] { (void)pirgpu_batch_select(sv.ctx, (uint32_t)w.home); });
  w.room = (uint32_t)items.size();
  {
    CtxLock lock(sv.ctx);      // (re)allocation of pinned memory is context state
    w.hq = pirgpu_host_query_buffer(sv.ctx, w.room);
    w.hr = pirgpu_host_reply_buffer(sv.ctx, w.room);
  }
  if (!w.hq || !w.hr) {
    const std::string msg = pirgpu_last_error(sv.ctx);
    for (const Item& it : items)
      if (!it.job->rc) fail_job(*it.job, PIRGPU_INTERNAL, msg);
    return;
  }
  // (1) queries -> pinned staging, in parallel; per-item outcome applied afterwards on this thread
  struct Outcome {
    uint32_t nq = 0;
    int code = 0;
    std::string msg;
  };
  std::vector<Outcome> res(items.size());
  Pool::get().parallel_for(items.size(), 8, [&](size_t i) {
    const Job& job = *items[i].job;
    if (job.rc) return;
    try {
      const auto& qm = job.pr.queries[items[i].qi];
      res[i].nq = load_query_into(sv.sh, qm.first, qm.second, w.hq + i * qwords, sv.nq_expected);
    } catch (const Err& e) {
      res[i].code = e.code;
      res[i].msg = e.msg;
    } catch (const std::exception& e) {
      res[i].code = PIRGPU_INTERNAL;
      res[i].msg = e.what();
    }
  });
  for (size_t i = 0; i < items.size(); ++i) {
    Job& job = *items[i].job;
    if (job.rc) continue;
    if (res[i].code) fail_job(job, res[i].code, res[i].msg);          // the first failing query of the request, in order
    else if (res[i].nq != sv.nq_expected) job.uniform = false;         // sequential path (reports it like the reference)
  }
  // queries of requests that failed or turned non-uniform are dropped from the window
  w.chunk.clear();
  w.slots.clear();
  for (size_t i = 0; i < items.size(); ++i) {
    const Job& job = *items[i].job;
    if (job.rc || !job.uniform || job.mismatch) continue;
    const size_t keep = w.chunk.size();
    if (keep != i) memmove(w.hq + keep * qwords, w.hq + i * qwords, qwords * 8);
    w.chunk.push_back(items[i]);
    w.slots.push_back(job.slot);
  }
  trace.mark("load queries", w.id);
  if (w.chunk.empty()) return;
  const uint32_t count = (uint32_t)w.chunk.size();
  // the response buffers are mapped NOW, before the GPU phase: the threads that fill them while the GPU runs then only
  // touch pages -- an mmap / munmap in the middle of the GPU's work goes through the driver's MMU notifier and was
  // measured to stretch a 64-client window from 15 to 25 ms.  Replies in request order: a request's queries are
  // consecutive items, so appending in item order keeps reply[i] answering query[i] (server.cpp:60-63)
  w.runs.clear();
  for (uint32_t i = 0; i < count;) {
    uint32_t e = i + 1;
    while (e < count && w.chunk[e].job == w.chunk[i].job) ++e;
    w.runs.emplace_back(i, e);
    i = e;
  }
  for (auto& run : w.runs) {
    Job& job = *w.chunk[run.first].job;
    try {
      job.out.reserve(job.out.n + (run.second - run.first) * reply_bytes_bound(sv.sh, sv.n_reply));
    } catch (const std::exception& e) {
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
  }
  // (2) stage + queue: the only part of a window that holds the context (lanes, workers and streams are shared between
  // the windows in flight; what is queued here runs behind the previous window's work in stream order)
  {
    CtxLock lock(sv.ctx);
    const uint32_t before = pirgpu_get_concurrency(sv.ctx);
    int rc = pirgpu_set_concurrency(sv.ctx, std::max<uint32_t>(before, 16));
    if (!rc) rc = pirgpu_batch_stage_async(sv.ctx, w.hq, sv.nq_expected, count);   // piecewise: group 0 starts after 1 MB
    if (!rc) rc = pirgpu_batch_set_keysets(sv.ctx, w.slots.data(), count);
    // every group sends its replies to the pinned buffer as soon as they exist: finish_window only waits
    if (!rc) rc = pirgpu_batch_set_host_replies(sv.ctx, w.hr, (uint64_t)w.room * sv.n_reply);
    if (!rc) rc = pirgpu_batch_run(sv.ctx);      // asynchronous: the window's kernels are queued
    if (rc) w.err = pirgpu_last_error(sv.ctx);
    w.rc = rc;
    (void)pirgpu_set_concurrency(sv.ctx, before);
  }
  w.queued = w.rc == 0;
  trace.mark("stage + enqueue", w.id);
}

// Everything after "queued": key compare under the GPU's work, the replies group by group as they land in pinned
// memory, serialisation on the pool's threads while the later groups are still being computed.
void finish_window(const Server& sv, Window& w, Trace& trace) {
  (void)pirgpu_batch_select(sv.ctx, (uint32_t)w.set);
  auto cleanup = finally([&] {
    (void)pirgpu_batch_set_host_replies(sv.ctx, nullptr, 0);
    (void)pirgpu_batch_unstage(sv.ctx);            // nothing staged keeps referring to the window's key sets
    (void)pirgpu_batch_select(sv.ctx, (uint32_t)w.home);
    for (Job* job : w.jobs) unpin_job(sv, *job);
  });
  if (w.chunk.empty()) return;
  const uint32_t count = (uint32_t)w.chunk.size();
  const size_t rwords = (size_t)sv.n_reply * sv.ctw;
  if (w.rc) {
    for (const Item& it : w.chunk)
      if (!it.job->rc) fail_job(*it.job, w.rc, w.err);
    return;
  }
  verify_keys_of(sv, w);   // host work under the GPU's
  trace.mark("verify keys", w.id);
  auto serialise = [&](size_t r) {
    Job& job = *w.chunk[w.runs[r].first].job;
    if (job.rc || !job.uniform || job.mismatch) return;
    try {
      for (uint32_t i = w.runs[r].first; i < w.runs[r].second; ++i)
        append_reply(job.out, sv.sh, w.hr + (size_t)i * rwords, sv.n_reply, sv.ctw);
    } catch (const std::exception& e) {
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
  };
  Pool::Group g;
  // the tasks capture this frame (g, serialise, w): whatever unwinds past here -- a failed submit included -- waits for
  // the tasks already queued first
  auto drain = finally([&g] { g.wait(); });
  size_t next_run = 0;
  uint32_t ready = 0;
  int rc = 0;
  while (ready < count) {
    uint32_t upto = 0;
    rc = pirgpu_batch_next_host_replies(sv.ctx, &upto);   // waits for the next group's download, not for the device
    if (rc || upto <= ready) break;                        // not downloaded group-wise: everything below
    ready = upto;
    while (next_run < w.runs.size() && w.runs[next_run].second <= ready) {
      const size_t r = next_run++;
      Pool::get().submit(g, [&serialise, r] { serialise(r); });   // one request per task: a megabyte per reply
    }
  }
  if (ready < count) {
    // d = 1 / 64-bit scan contexts do not download group by group: wait for the batch (this also waits for whatever
    // another window has queued behind it) and serialise everything here
    uint64_t got = 0;
    rc = pirgpu_batch_fetch(sv.ctx, w.hr, (uint64_t)count * sv.n_reply, &got);
    if (!rc)
      while (next_run < w.runs.size()) {
        const size_t r = next_run++;
        Pool::get().submit(g, [&serialise, r] { serialise(r); });
      }
  }
  g.wait();
  trace.mark("wait + serialise", w.id);
  if (rc) {
    const std::string msg = pirgpu_last_error(sv.ctx);
    for (const Item& it : w.chunk)
      if (!it.job->rc) fail_job(*it.job, rc, msg);
  }
}

// Which batch sets of a context are taken by windows in flight (any thread); a serving thread owns the sets it took.
struct Combiner {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job*> pending;
  bool set_busy[2] = {false, false};
  int sets_allowed = 2;   // 1 while the context has a single key-set slot: two windows could not both pin a client's set
  RelinCache relin;
  int take_set() {   // call with m held; -1: none free
    if ((int)set_busy[0] + (int)set_busy[1] >= sets_allowed) return -1;
    for (int i = 0; i < 2; ++i)
      if (!set_busy[i]) {
        set_busy[i] = true;
        return i;
      }
    return -1;
  }
};

// Windows pin their clients' key sets, half the slots per window; with ONE slot a second leader's window could never claim
// a set while the first one's is pinned (a legitimate request would fail with "every key set slot is in use"): such a
// context serves one window at a time.
int sets_allowed_for(pirgpu_ctx* ctx) {
  uint64_t stats[4] = {0, 0, 0, 16};
  (void)pirgpu_keyset_stats(ctx, stats);
  return stats[3] >= 2 ? 2 : 1;
}

// Serves `n` requests on batch set `first_set` (which the caller owns) and, when it is free, the other one: windows of
// at most kMaxRequestBatch queries / half the key-set capacity in clients, two in flight.
void serve(pirgpu_ctx* ctx, Combiner& cb, int first_set, Job* const* jobs, size_t n) {
  Server sv{};
  sv.ctx = ctx;
  if (pirgpu_get_params(ctx, &sv.prm)) {
    for (size_t i = 0; i < n; ++i) fail_job(*jobs[i], PIRGPU_INVALID_ARGUMENT, "invalid context");
    return;
  }
  sv.sh = make_shape(sv.prm);
  sv.ctw = (size_t)2 * sv.sh.k * sv.sh.N;
  sv.n_reply = pirgpu_reply_ct_count(ctx);
  uint64_t dim_sum = 0;
  for (uint32_t l = 0; l < sv.prm.num_dimensions; ++l) dim_sum += sv.prm.dimensions[l];
  sv.nq_expected = (uint32_t)(dim_sum / sv.sh.N + 1);  // server.cpp:154
  sv.relin = &cb.relin;
  Trace trace;
  // this thread's pinned staging and batch state are those of the set it owns (the lone-query and the sequential paths
  // use them too); the calling thread's default selection (0) is restored when the call returns
  (void)pirgpu_batch_select(ctx, (uint32_t)first_set);
  auto reselect = finally([&] { (void)pirgpu_batch_select(ctx, 0); });
  uint64_t stats[4] = {0, 0, 0, 16};
  (void)pirgpu_keyset_stats(ctx, stats);
  // every window pins its clients' key sets until it is finished, and two windows can be in flight (this thread's or
  // another serving thread's): half the slots each, so that a new client's claim always finds an unpinned set
  const size_t max_clients = std::max<size_t>(1, std::min<size_t>(stats[3] >= 2 ? stats[3] / 2 : 1, kMaxRequestBatch));
  const bool may_overlap = stats[3] >= 2;
  sv.relin_keep = 4 * (size_t)stats[3] + 16;

  // (1) parse every request; RelinKeys validated when present (server.cpp:53-58: only CT-multiplication mode uses them,
  // but a malformed non-empty field is InvalidArgument in the reference)
  uint64_t total_queries = 0;
  for (size_t i = 0; i < n; ++i) {
    Job& job = *jobs[i];
    if (job.rc) continue;
    try {
      job.pr = parse_request(job.request, job.request_len);
      total_queries += job.pr.queries.size();
    } catch (const Err& e) {
      fail_job(job, e.code, e.msg);
    } catch (const std::exception& e) {
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
  }
  trace.mark("parse");

  // keys of one request: a fingerprint match first (the 4.7 MB compare runs under the GPU's work), else parse + upload
  auto keys_for_job = [&](Job& job) {
    if (job.rc) return;
    if (job.pins > 0) {     // a request cut over several windows: one more pin on the set it already resolved to
      if (pirgpu_keyset_pin(sv.ctx, job.slot) == 0) ++job.pins;
      else fail_job(job, PIRGPU_INTERNAL, pirgpu_last_error(sv.ctx));
      return;
    }
    try {
      resolve_keys(sv, job, true, &job.unverified);
    } catch (const Err& e) {
      unpin_job(sv, job);
      fail_job(job, e.code, e.msg);
    } catch (const std::exception& e) {
      unpin_job(sv, job);
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
  };

  // (2) one query in the whole call: the single-query path (lowest latency), under the context's request lock
  if (total_queries == 1) {
    CtxLock lock(ctx);
    for (size_t i = 0; i < n; ++i) {
      Job& job = *jobs[i];
      if (job.rc) continue;
      keys_for_job(job);
      if (!job.rc && job.pr.queries.empty()) {
        try {
          check_relin_keys(sv, job);
        } catch (const Err& e) {
          fail_job(job, e.code, e.msg);
        }
      }
      if (job.rc || job.pr.queries.empty()) {
        unpin_job(sv, job);
        continue;
      }
      try {
        bool verified = true;
        run_single(sv, job, job.pr.queries[0], [&]() {   // the byte-for-byte key compare runs under the GPU work
          if (job.unverified) verified = pirgpu_keyset_verify(sv.ctx, job.slot, job.pr.galois_keys, job.pr.galois_keys_len) != 0;
          if (verified) check_relin_keys(sv, job);       // ... and so does the RelinKeys validation (throws -> the job fails)
        });
        if (!verified) {
          // same fingerprint, different bytes: not this client's keys after all -- install them and run again
          job.out.clear();
          unpin_job(sv, job);
          resolve_keys(sv, job, false, nullptr);
          check_relin_keys(sv, job);
          run_single(sv, job, job.pr.queries[0]);
        }
      } catch (const Err& e) {
        fail_job(job, e.code, e.msg);
      } catch (const std::exception& e) {
        fail_job(job, PIRGPU_INTERNAL, e.what());
      }
      unpin_job(sv, job);
    }
    return;
  }

  // (3) several queries: windows through the batch pipeline, every query with its client's key set.  Requests whose
  // queries do not all have the expected ciphertext count, and those whose key bytes turn out to differ from the
  // resident set their fingerprint matched, take the sequential path at the end.
  int sets[2] = {first_set, -1};
  auto release_second = finally([&] {
    if (sets[1] >= 0) {
      std::lock_guard<std::mutex> lk(cb.m);
      cb.set_busy[sets[1]] = false;
      cb.cv.notify_all();
    }
  });
  Window win[2];
  bool open[2] = {false, false};
  std::deque<int> order;     // open windows, oldest first
  int wid = 0;
  auto finish_oldest = [&] {
    const int slot = order.front();
    order.pop_front();
    finish_window(sv, win[slot], trace);
    open[slot] = false;
    win[slot] = Window();
    return slot;
  };
  // (Round 4 tried cutting a lone call that fits one window into two half windows in flight together: no gain for a
  // lone caller -- windows of 32 run the GPU less efficiently than they overlap -- and 5 % lost with two callers, whose
  // calls then grab both batch sets from each other.  One window per <= 64 queries it stays.)
  const size_t window_cap = kMaxRequestBatch;
  size_t ji = 0;        // next request
  uint32_t qi = 0;      // its next query
  while (ji < n) {
    // the next window's items: whole requests while they fit (a request with more than window_cap queries is cut)
    std::vector<Item> items;
    std::vector<Job*> wjobs;
    while (ji < n && items.size() < window_cap && wjobs.size() < max_clients) {
      Job& job = *jobs[ji];
      const uint32_t left = job.rc ? 0 : (uint32_t)job.pr.queries.size() - qi;
      if (!left) {
        if (!job.rc && job.pr.queries.empty()) {   // no query: the keys are still deserialised (server.cpp:46-58)
          keys_for_job(job);
          try {
            if (!job.rc) check_relin_keys(sv, job);
          } catch (const Err& e) {
            fail_job(job, e.code, e.msg);
          }
          unpin_job(sv, job);
        }
        ++ji;
        qi = 0;
        continue;
      }
      if (left > window_cap - items.size() && !items.empty() && left <= window_cap) break;   // next window
      wjobs.push_back(&job);
      const uint32_t take = (uint32_t)std::min<size_t>(left, window_cap - items.size());
      for (uint32_t q = 0; q < take; ++q) items.push_back({&job, qi + q});
      qi += take;
      if (qi == job.pr.queries.size()) {
        ++ji;
        qi = 0;
      }
    }
    if (items.empty()) break;
    // a free slot for it: the second batch set is taken when nobody else has it; otherwise finish the window in flight
    int slot = -1;
    for (int s2 = 0; s2 < 2 && slot < 0; ++s2)
      if (!open[s2] && sets[s2] >= 0) slot = s2;
    if (slot < 0 && sets[1] < 0 && may_overlap) {
      std::lock_guard<std::mutex> lk(cb.m);
      const int got = cb.take_set();
      if (got >= 0) {
        sets[1] = got;
        slot = 1;
      }
    }
    if (slot < 0) slot = finish_oldest();
    Window& w = win[slot];
    w.set = sets[slot];
    w.home = first_set;
    w.id = wid++;
    w.jobs = wjobs;
    for (Job* job : wjobs) keys_for_job(*job);      // pins the window's key sets
    trace.mark("resolve keys", w.id);
    begin_window(sv, w, items, trace);
    open[slot] = true;
    order.push_back(slot);
  }
  while (!order.empty()) (void)finish_oldest();

  // sequential path for the requests with a wrong ciphertext count somewhere, and for those whose key bytes turned out
  // to differ from the resident set their fingerprint matched (another client's object: install theirs, serve again)
  for (size_t i = 0; i < n; ++i) {
    Job& job = *jobs[i];
    if (job.rc || (job.uniform && !job.mismatch)) continue;
    job.out.clear();
    CtxLock lock(ctx);
    try {
      resolve_keys(sv, job, false, nullptr);      // verified lookup, or this client's own keys installed
      job.unverified = job.mismatch = false;
      if (!job.relin_checked) check_relin_keys(sv, job);
      for (auto& qm : job.pr.queries) run_single(sv, job, qm);
    } catch (const Err& e) {
      fail_job(job, e.code, e.msg);
    } catch (const std::exception& e) {
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
    unpin_job(sv, job);
  }
}

// serve() catches what a request can cause per request; anything that still escapes (out of memory while queueing, ...)
// must not leave the combiner without its batch set or cross the C ABI: every request of the call fails with Internal.
void serve_guarded(pirgpu_ctx* ctx, Combiner& cb, int set, Job* const* jobs, size_t n) noexcept {
  const char* what = nullptr;
  std::string msg;
  int code = PIRGPU_INTERNAL;
  try {
    serve(ctx, cb, set, jobs, n);
    return;
  } catch (const Err& e) {
    code = e.code;
    msg = e.msg;
    what = msg.c_str();
  } catch (const std::exception& e) {
    msg = e.what();
    what = msg.c_str();
  } catch (...) {
    what = "unexpected failure while serving";
  }
  for (size_t i = 0; i < n; ++i) {
    for (; jobs[i]->pins > 0; --jobs[i]->pins) (void)pirgpu_keyset_unpin(ctx, jobs[i]->slot);
    try {
      fail_job(*jobs[i], code, what);
    } catch (...) {
      jobs[i]->rc = code;
    }
  }
}

int finish(pirgpu_ctx* ctx, Job& job, uint8_t** response, size_t* response_len) {
  if (job.rc) {
    pirgpu_set_error(ctx, job.err.c_str());   // on the CALLING thread: pirgpu_last_error is per thread
    return job.rc;
  }
  try {
    *response = job.out.release(response_len);   // the buffer the reply was serialised into: no copy
  } catch (const std::bad_alloc&) {
    return PIRGPU_INTERNAL;
  }
  return PIRGPU_OK;
}

// Requests that arrive while every batch set is taken are queued and served together by whichever thread gets a set
// next (flat combining): no extra latency when the server is idle, cross-client batching under load -- and with two
// batch sets per context the next window is parsed, staged and queued while the previous one is still on the GPU.
thread_local std::vector<std::string> t_request_errors;   // per request of this thread's last pirgpu_process_requests
std::mutex g_combiners_mu;
std::map<pirgpu_ctx*, std::shared_ptr<Combiner>> g_combiners;

std::shared_ptr<Combiner> combiner_for(pirgpu_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_combiners_mu);
  auto& p = g_combiners[ctx];
  if (!p) p = std::make_shared<Combiner>();
  return p;
}

}  // namespace

extern "C" {

void pirgpu_free(void* p) { pool_release(static_cast<uint8_t*>(p)); }

void pirgpu_wire_forget(pirgpu_ctx* ctx) {
  std::lock_guard<std::mutex> lock(g_combiners_mu);
  g_combiners.erase(ctx);
}

int pirgpu_process_request(pirgpu_ctx* ctx, const uint8_t* request, size_t request_len, uint8_t** response,
                           size_t* response_len) {
  if (!ctx || (!request && request_len) || !response || !response_len) return PIRGPU_INVALID_ARGUMENT;
  *response = nullptr;
  *response_len = 0;
  Job job;
  job.request = request;
  job.request_len = request_len;
  std::shared_ptr<Combiner> cb = combiner_for(ctx);
  const int allowed = sets_allowed_for(ctx);   // (asks the context: before the combiner's lock is taken)
  {
    std::unique_lock<std::mutex> lk(cb->m);
    cb->sets_allowed = allowed;
    cb->pending.push_back(&job);
    while (!job.done) {
      // (nothing pending: this thread's request is being served by another leader -- wait for it)
      const int set = cb->pending.empty() ? -1 : cb->take_set();
      if (set < 0) {
        cb->cv.wait(lk);
        continue;
      }
      // lead: serve what is queued (this thread's own request is in there unless another leader took it) and hand over
      std::vector<Job*> batch;
      while (!cb->pending.empty() && batch.size() < kMaxRequestBatch) {
        batch.push_back(cb->pending.front());
        cb->pending.pop_front();
      }
      lk.unlock();
      if (!batch.empty()) serve_guarded(ctx, *cb, set, batch.data(), batch.size());
      lk.lock();
      for (Job* j : batch) j->done = true;
      cb->set_busy[set] = false;
      cb->cv.notify_all();
    }
  }
  return finish(ctx, job, response, response_len);
}

int pirgpu_process_requests(pirgpu_ctx* ctx, uint32_t n, const uint8_t* const* requests, const size_t* request_lens,
                            uint8_t** responses, size_t* response_lens, int* status) {
  if (!ctx || (n && (!requests || !request_lens || !responses || !response_lens || !status))) return PIRGPU_INVALID_ARGUMENT;
  std::vector<Job> jobs(n);
  std::vector<Job*> ptrs(n);
  for (uint32_t i = 0; i < n; ++i) {
    responses[i] = nullptr;
    response_lens[i] = 0;
    jobs[i].request = requests[i];
    jobs[i].request_len = request_lens[i];
    ptrs[i] = &jobs[i];
    if (!requests[i] && request_lens[i]) fail_job(jobs[i], PIRGPU_INVALID_ARGUMENT, "null request");
  }
  Trace trace;
  std::shared_ptr<Combiner> cb = combiner_for(ctx);
  int set;
  const int allowed = sets_allowed_for(ctx);
  {
    std::unique_lock<std::mutex> lk(cb->m);
    cb->sets_allowed = allowed;
    while ((set = cb->take_set()) < 0) cb->cv.wait(lk);
  }
  serve_guarded(ctx, *cb, set, ptrs.data(), n);
  {
    std::lock_guard<std::mutex> lk(cb->m);
    cb->set_busy[set] = false;
    cb->cv.notify_all();
  }
  trace.mark("serve (whole call)");
  int worst = PIRGPU_OK;
  t_request_errors.assign(n, std::string());
  for (uint32_t i = 0; i < n; ++i) {
    status[i] = finish(ctx, jobs[i], &responses[i], &response_lens[i]);
    if (status[i]) t_request_errors[i] = jobs[i].err;
    if (status[i] && !worst) worst = status[i];
  }
  trace.mark("finish");
  return worst;
}

const char* pirgpu_request_error(uint32_t i) {
  return i < t_request_errors.size() ? t_request_errors[i].c_str() : "";
}

// pirgpu_process_requests in two halves, so that ONE calling thread can have two calls in flight: begin() hands the call
// to a serving thread of the library and returns; end() waits for it.  The second call's parsing, staging and queueing
// then run under the first one's tail (its last groups on the GPU, their download and serialisation) exactly as they do
// for two calling threads -- a synchronous caller fills and drains the two-lane pipeline alone with every call.
struct PendingCall {
  std::thread th;
  int rc = PIRGPU_INTERNAL;
  std::vector<std::string> errors;
};

int pirgpu_process_requests_begin(pirgpu_ctx* ctx, uint32_t n, const uint8_t* const* requests, const size_t* request_lens,
                                  uint8_t** responses, size_t* response_lens, int* status, void** call) {
  if (!call) return PIRGPU_INVALID_ARGUMENT;
  *call = nullptr;
  if (!ctx || (n && (!requests || !request_lens || !responses || !response_lens || !status))) return PIRGPU_INVALID_ARGUMENT;
  PendingCall* pc = nullptr;
  try {
    pc = new PendingCall();
    pc->th = std::thread([=] {
      pc->rc = pirgpu_process_requests(ctx, n, requests, request_lens, responses, response_lens, status);
      pc->errors = t_request_errors;     // (this serving thread's: handed to the thread that calls end())
    });
  } catch (...) {
    delete pc;
    return PIRGPU_INTERNAL;
  }
  *call = pc;
  return PIRGPU_OK;
}

int pirgpu_process_requests_end(void* call) {
  PendingCall* pc = static_cast<PendingCall*>(call);
  if (!pc) return PIRGPU_INVALID_ARGUMENT;
  if (pc->th.joinable()) pc->th.join();
  const int rc = pc->rc;
  t_request_errors = std::move(pc->errors);   // pirgpu_request_error(i) on the calling thread answers for this call
  delete pc;
  return rc;
}

}  // extern "C"
