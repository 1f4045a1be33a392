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

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
};

struct Server {               // what serving needs to know about a context
  pirgpu_ctx* ctx;
  pirgpu_params prm;
  Shape sh;
  size_t ctw;
  uint64_t n_reply;
  uint32_t nq_expected;
};

// Response.reply (payload.proto:39-42) for one query: n ciphertexts, written straight into the response buffer
// Upper bound of what append_reply adds for a reply of n ciphertexts (tags and varint lengths: < 32 bytes each)
size_t reply_bytes_bound(const Shape& sh, uint64_t n) { return 32 + n * (32 + saved_ciphertext_size(sh)); }

void append_reply(OutBuf& out, const Shape& sh, const uint64_t* cts_words, uint64_t n, size_t ctw) {
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
    out.append(per_ct_head.data(), per_ct_head.size());
    out.append(cts_words + i * ctw, ctw * 8);
  }
}

// SEALDeserialize<GaloisKeys> (server.cpp:46-48) with the device-resident key cache: a client that
// repeats its (multi-MB) key object byte for byte finds its keys resident; a new client's keys are parsed, validated
// as a whole (a malformed object must not leave half-installed keys behind) and uploaded into a free or the least
// recently used slot.  `speculative`: accept a resident set on its fingerprint alone -- the caller verifies the bytes
// (pirgpu_keyset_verify) while the GPU already works and discards the result on a mismatch.
void resolve_keys(const Server& sv, Job& job, bool speculative, bool* unverified) {
  if (unverified) *unverified = false;
  uint32_t slot = 0;
  int rc = 0;
  if (job.pr.galois_keys_len) {
    rc = pirgpu_keyset_lookup(sv.ctx, job.pr.galois_keys, job.pr.galois_keys_len, speculative ? 0 : 1, &slot);
    if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  }
  if (slot) {
    if (unverified) *unverified = speculative;
    job.slot = slot;
    return;
  }
  // empty bytes -> load throws -> InvalidArgument, like the reference
  std::vector<std::pair<uint32_t, std::vector<uint64_t>>> parsed;
  const size_t key_words = (size_t)sv.sh.k * 2 * (sv.sh.k + 1) * sv.sh.N;
  load_kswitch_keys(sv.sh, job.pr.galois_keys, job.pr.galois_keys_len, [&](uint64_t index, const uint64_t* key) {
    parsed.emplace_back((uint32_t)(2 * index + 1), std::vector<uint64_t>(key, key + key_words));
  });
  rc = pirgpu_keyset_claim(sv.ctx, job.pr.galois_keys, job.pr.galois_keys_len, &slot);
  if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  for (auto& kv : parsed) {
    rc = pirgpu_keyset_set_key(sv.ctx, slot, kv.first, kv.second.data());
    if (rc) {
      const std::string msg = pirgpu_last_error(sv.ctx);
      (void)pirgpu_keyset_release(sv.ctx, slot);
      throw Err{rc, msg};
    }
  }
  job.slot = slot;
}

// One query through the single-query path (lowest latency; also reports a wrong ciphertext count at the offending
// query like the reference, server.cpp:154-158).  Query parsed into pinned staging, reply serialised out of it.
// `while_running` (optional) is host work done between enqueueing the kernels and waiting for the reply.
void run_single(const Server& sv, Job& job, const std::pair<const uint8_t*, size_t>& qm,
                const std::function<void()>& while_running = nullptr) {
  uint64_t* hq = pirgpu_host_query_buffer(sv.ctx, 1);
  uint64_t* hr = pirgpu_host_reply_buffer(sv.ctx, 1);
  if (!hq || !hr) throw Err{PIRGPU_INTERNAL, pirgpu_last_error(sv.ctx)};
  const uint32_t nq = load_query_into(sv.sh, qm.first, qm.second, hq, sv.nq_expected);
  const uint32_t callers_slot = pirgpu_current_keyset(sv.ctx);   // the direct API's selection survives a request
  int rc = pirgpu_query_use_keyset(sv.ctx, job.slot);
  uint64_t got = 0;
  if (!rc) rc = pirgpu_query_stage(sv.ctx, hq, nq);
  if (!rc) rc = pirgpu_query_run(sv.ctx);      // asynchronous: every kernel of the path is queued
  const std::string msg = rc ? pirgpu_last_error(sv.ctx) : "";
  (void)pirgpu_query_use_keyset(sv.ctx, callers_slot);
  if (rc) throw Err{rc, msg};
  if (while_running) while_running();
  rc = pirgpu_query_fetch(sv.ctx, hr, sv.n_reply, &got);
  if (rc) throw Err{rc, pirgpu_last_error(sv.ctx)};
  append_reply(job.out, sv.sh, hr, got, sv.ctw);
}

void fail_job(Job& job, int code, const std::string& msg) {
  job.rc = code;
  job.err = msg;
  job.out.clear();
}

// Host threads a window may use next to the serving thread: at most `most`, never more than half the machine.
size_t worker_threads(size_t most) {
  const size_t hw = std::thread::hardware_concurrency();
  return std::max<size_t>(1, std::min<size_t>(most, hw ? hw / 2 : 1));
}

// Byte-for-byte compare of the key objects of the requests behind `items` with the resident sets they were matched
// to by fingerprint -- once per request, on up to 16 worker threads (4.7 MB per client: 0.3-0.4 ms each on one
// thread, which was most of a request's host time), while the GPU runs the chunk just queued.  The resident copies are
// read without the context's lock: the request lock is held and the window's slots are pinned.
template <typename Item>
void verify_keys_of(const Server& sv, Item* items, uint32_t count) {
  std::vector<Job*> todo;
  for (uint32_t i = 0; i < count; ++i) {
    Job* job = items[i].job;
    if (job->unverified && std::find(todo.begin(), todo.end(), job) == todo.end()) todo.push_back(job);
  }
  if (todo.empty()) return;
  // the resident copies are looked up HERE, on the serving thread: it holds the context's (recursive) lock for the whole
  // window, a worker thread asking for it would wait for ever
  std::vector<std::pair<const uint8_t*, size_t>> resident(todo.size());
  for (size_t i = 0; i < todo.size(); ++i) resident[i].second = pirgpu_keyset_blob(sv.ctx, todo[i]->slot, &resident[i].first);
  auto check_at = [&](size_t i) {
    Job* job = todo[i];
    const uint8_t* r = resident[i].first;
    const size_t len = resident[i].second;
    job->mismatch = !(r && len == job->pr.galois_keys_len && memcmp(r, job->pr.galois_keys, len) == 0);
    job->unverified = false;
  };
  // (64 clients' objects are 600 MB to read, more than the host's last-level cache: memory-bound, so many threads)
  const size_t n_threads = std::min<size_t>(worker_threads(16), todo.size());
  std::vector<std::future<void>> workers;
  for (size_t t = 1; t < n_threads; ++t)
    workers.push_back(std::async(std::launch::async, [&, t] {
      for (size_t i = t; i < todo.size(); i += n_threads) check_at(i);
    }));
  for (size_t i = 0; i < todo.size(); i += n_threads) check_at(i);
  for (auto& w : workers) w.get();
}

struct Trace {   // PIRGPU_WIRE_TRACE=1: host-side phase times of a window, to stderr
  bool on = getenv("PIRGPU_WIRE_TRACE") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void mark(const char* what) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[wire] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

// Serves a window of requests under the context's request lock.
void serve_window(const Server& sv, Job* const* jobs, size_t n) {
  Trace trace;
  // (1) parse, resolve keys, validate relin keys -- per request; a failing request does not affect the others
  uint32_t total_queries = 0;
  bool unverified = false;
  const bool lone = n == 1;
  for (size_t i = 0; i < n; ++i) {
    Job& job = *jobs[i];
    if (job.rc) continue;
    try {
      job.pr = parse_request(job.request, job.request_len);
      // every request starts on a fingerprint match; the key bytes (4.7 MB per client) are compared while the GPU works:
      // by this thread for a lone single-query request, on worker threads for a window of requests
      (void)lone;
      resolve_keys(sv, job, true, &job.unverified);
      unverified = unverified || job.unverified;
      // SEALDeserialize<RelinKeys> when present (server.cpp:53-58): only CT-multiplication mode uses them, but a
      // malformed non-empty field is InvalidArgument in the reference, so it is parsed and validated here too
      if (job.pr.relin_keys_len) load_kswitch_keys(sv.sh, job.pr.relin_keys, job.pr.relin_keys_len, nullptr);
      total_queries += (uint32_t)job.pr.queries.size();
    } catch (const Err& e) {
      fail_job(job, e.code, e.msg);
    } catch (const std::exception& e) {
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
  }
  trace.mark("parse + resolve keys");
  // (2) one query in the whole window: the single-query path
  if (total_queries == 1) {
    for (size_t i = 0; i < n; ++i) {
      Job& job = *jobs[i];
      if (job.rc || job.pr.queries.empty()) continue;
      try {
        bool verified = true;
        run_single(sv, job, job.pr.queries[0], [&]() {   // the byte-for-byte key compare runs under the GPU work
          if (job.unverified) verified = pirgpu_keyset_verify(sv.ctx, job.slot, job.pr.galois_keys, job.pr.galois_keys_len) != 0;
        });
        if (!verified) {
          // same fingerprint, different bytes: not this client's keys after all -- install them and run again
          job.out.clear();
          resolve_keys(sv, job, false, nullptr);
          run_single(sv, job, job.pr.queries[0]);
        }
      } catch (const Err& e) {
        fail_job(job, e.code, e.msg);
      } catch (const std::exception& e) {
        fail_job(job, PIRGPU_INTERNAL, e.what());
      }
    }
    return;
  }
  // (3) several queries: the batch pipeline in chunks of <= kMaxRequestBatch, every query with its client's key set.
  // Requests whose queries do not all have the expected ciphertext count take the sequential path, which reports the
  // error at the offending query like the reference does (nothing of theirs has run on the device at that point).
  struct Item { Job* job; uint32_t qi; };
  std::vector<Item> items;
  for (size_t i = 0; i < n; ++i) {
    Job& job = *jobs[i];
    if (job.rc) continue;
    for (uint32_t q = 0; q < job.pr.queries.size(); ++q) items.push_back({&job, q});
  }
  const uint32_t before = pirgpu_get_concurrency(sv.ctx);
  int rc = pirgpu_set_concurrency(sv.ctx, std::max<uint32_t>(before, std::min<uint32_t>((uint32_t)items.size(), 16)));
  if (rc) {
    const std::string msg = pirgpu_last_error(sv.ctx);
    for (size_t i = 0; i < n; ++i)
      if (!jobs[i]->rc) fail_job(*jobs[i], rc, msg);
    return;
  }
  const size_t qwords = (size_t)sv.nq_expected * sv.ctw, rwords = (size_t)sv.n_reply * sv.ctw;
  size_t pos = 0;
  std::vector<uint32_t> slots;
  while (pos < items.size()) {
    // pinned staging for this chunk only (a reply is (2 ER)^(d-1) ciphertexts: 24 MiB per query at N = 16384, k = 4)
    const uint32_t room = (uint32_t)std::min<size_t>(kMaxRequestBatch, items.size() - pos);
    uint64_t* hq = pirgpu_host_query_buffer(sv.ctx, room);
    uint64_t* hr = pirgpu_host_reply_buffer(sv.ctx, room);
    std::vector<Item> chunk;
    slots.clear();
    while (pos < items.size() && chunk.size() < room) {
      Item it = items[pos++];
      Job& job = *it.job;
      if (job.rc || !job.uniform || job.mismatch) continue;
      try {
        if (!hq || !hr) throw Err{PIRGPU_INTERNAL, pirgpu_last_error(sv.ctx)};
        const auto& qm = job.pr.queries[it.qi];
        const uint32_t nq = load_query_into(sv.sh, qm.first, qm.second, hq + chunk.size() * qwords, sv.nq_expected);
        if (nq != sv.nq_expected) {
          job.uniform = false;     // sequential path below (its earlier queries are recomputed there: error path only)
          continue;
        }
        chunk.push_back(it);
        slots.push_back(job.slot);
      } catch (const Err& e) {
        fail_job(job, e.code, e.msg);
      } catch (const std::exception& e) {
        fail_job(job, PIRGPU_INTERNAL, e.what());
      }
    }
    // queries of requests that failed or turned non-uniform while the chunk was being filled are dropped from it
    size_t keep = 0;
    for (size_t i = 0; i < chunk.size(); ++i) {
      if (chunk[i].job->rc || !chunk[i].job->uniform || chunk[i].job->mismatch) continue;
      if (keep != i) {
        memmove(hq + keep * qwords, hq + i * qwords, qwords * 8);
        chunk[keep] = chunk[i];
        slots[keep] = slots[i];
      }
      ++keep;
    }
    chunk.resize(keep);
    slots.resize(keep);
    if (chunk.empty()) continue;
    const uint32_t count = (uint32_t)chunk.size();
    // the response buffers are mapped NOW, before the GPU phase: the worker threads that fill them while the GPU runs
    // then only touch pages -- an mmap / munmap in the middle of the GPU's work goes through the driver's MMU notifier
    // and was measured to stretch a 64-client window from 15 to 25 ms
    // replies in request order: a request's queries are consecutive items, so appending in item order keeps
    // reply[i] answering query[i] (server.cpp:60-63)
    std::vector<std::pair<uint32_t, uint32_t>> runs;          // [first, end) items of one request inside the chunk
    for (uint32_t i = 0; i < count;) {
      uint32_t e = i + 1;
      while (e < count && chunk[e].job == chunk[i].job) ++e;
      runs.emplace_back(i, e);
      i = e;
    }
    for (auto& run : runs) {
      Job& job = *chunk[run.first].job;
      try {
        job.out.reserve(job.out.n + (run.second - run.first) * reply_bytes_bound(sv.sh, sv.n_reply));
      } catch (const std::exception& e) {
        fail_job(job, PIRGPU_INTERNAL, e.what());
      }
    }
    trace.mark("load queries");
    rc = pirgpu_batch_stage(sv.ctx, hq, sv.nq_expected, count);
    if (!rc) rc = pirgpu_batch_set_keysets(sv.ctx, slots.data(), count);
    // every group sends its replies to the pinned buffer as soon as they exist: the fetch below only waits
    if (!rc) rc = pirgpu_batch_set_host_replies(sv.ctx, hr, (uint64_t)room * sv.n_reply);
    if (!rc) rc = pirgpu_batch_run(sv.ctx);      // asynchronous: the chunk's kernels are queued
    trace.mark("stage + enqueue");
    if (!rc) verify_keys_of(sv, chunk.data(), count);   // host work under the GPU's
    trace.mark("verify keys");
    auto serialise = [&](size_t r) {
      Job& job = *chunk[runs[r].first].job;
      if (job.rc || !job.uniform || job.mismatch) return;
      try {
        for (uint32_t i = runs[r].first; i < runs[r].second; ++i)
          append_reply(job.out, sv.sh, hr + (size_t)i * rwords, sv.n_reply, sv.ctw);
      } catch (const std::exception& e) {
        fail_job(job, PIRGPU_INTERNAL, e.what());
      }
    };
    // While the GPU is still computing the later groups: as soon as a group's replies have landed in the pinned buffer,
    // the requests they complete are serialised on a worker thread (a megabyte per reply, into freshly mapped pages)
    std::vector<std::future<void>> workers;
    size_t next_run = 0;
    static const bool stream_replies = !(getenv("PIRGPU_WIRE_STREAM") && getenv("PIRGPU_WIRE_STREAM")[0] == '0');
    if (!rc && stream_replies) {
      uint32_t ready = 0;
      while (ready < count) {
        uint32_t upto = 0;
        if (pirgpu_batch_next_host_replies(sv.ctx, &upto) != 0 || upto <= ready) break;   // not group-wise: all below
        ready = upto;
        size_t e = next_run;
        while (e < runs.size() && runs[e].second <= ready) ++e;
        if (e > next_run) {
          workers.push_back(std::async(std::launch::async, [&serialise, next_run, e] {
            for (size_t r = next_run; r < e; ++r) serialise(r);
          }));
          next_run = e;
        }
      }
    }
    uint64_t got = 0;
    if (!rc) rc = pirgpu_batch_fetch(sv.ctx, hr, (uint64_t)count * sv.n_reply, &got);   // everything has arrived
    (void)pirgpu_batch_set_host_replies(sv.ctx, nullptr, 0);
    for (auto& w : workers) w.get();
    workers.clear();
    trace.mark("wait + fetch");
    if (rc) {
      const std::string msg = pirgpu_last_error(sv.ctx);
      for (auto& it : chunk)
        if (!it.job->rc) fail_job(*it.job, rc, msg);
      continue;
    }
    // what has not been serialised on the way (all of it without group-wise download): up to eight threads
    const size_t left = runs.size() - next_run;
    const size_t n_threads = std::min<size_t>(worker_threads(8), left);
    for (size_t t = 1; t < n_threads; ++t)
      workers.push_back(std::async(std::launch::async, [&, t] {
        for (size_t r = next_run + t; r < runs.size(); r += n_threads) serialise(r);
      }));
    for (size_t r = next_run; r < runs.size(); r += std::max<size_t>(n_threads, 1)) serialise(r);
    for (auto& w : workers) w.get();
    trace.mark("serialise");
  }
  (void)pirgpu_set_concurrency(sv.ctx, before);
  // sequential path for the requests with a wrong ciphertext count somewhere, and for those whose key bytes turned out
  // to differ from the resident set their fingerprint matched (another client's object: install theirs, serve again)
  for (size_t i = 0; i < n; ++i) {
    Job& job = *jobs[i];
    if (job.rc || (job.uniform && !job.mismatch)) continue;
    job.out.clear();
    try {
      if (job.unverified && !job.mismatch &&
          !pirgpu_keyset_verify(sv.ctx, job.slot, job.pr.galois_keys, job.pr.galois_keys_len))
        job.mismatch = true;                       // (a request that never reached a batch chunk)
      if (job.mismatch) {
        resolve_keys(sv, job, false, nullptr);
        job.unverified = job.mismatch = false;
      }
      for (auto& qm : job.pr.queries) run_single(sv, job, qm);
    } catch (const Err& e) {
      fail_job(job, e.code, e.msg);
    } catch (const std::exception& e) {
      fail_job(job, PIRGPU_INTERNAL, e.what());
    }
  }
}

// Serves `n` requests: windows of at most `capacity` clients (so that every client's key set of a window can be
// resident at once) and kMaxRequestBatch queries' worth of requests.
void serve(pirgpu_ctx* ctx, Job* const* jobs, size_t n) {
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
  // One window = one critical section on the context (the batch staging, the lanes and the key set slots are
  // context state); PIRServer::ProcessRequest is const and re-entrant in the reference because everything is local.
  pirgpu_request_lock(ctx);
  struct Unlock {
    pirgpu_ctx* c;
    ~Unlock() {
      pirgpu_keyset_pin_end(c);
      pirgpu_request_unlock(c);
    }
  } unlock{ctx};
  uint64_t stats[4] = {0, 0, 0, 16};
  (void)pirgpu_keyset_stats(ctx, stats);
  const size_t window = std::max<size_t>(1, std::min<size_t>(stats[3], kMaxRequestBatch));
  for (size_t first = 0; first < n; first += window) {
    pirgpu_keyset_pin_begin(ctx);   // key sets touched from here on are not evicted until the window is done
    serve_window(sv, jobs + first, std::min(window, n - first));
  }
}

// serve() catches what a request can cause per request; anything that still escapes (out of memory while queueing, ...)
// must not leave the combiner without a leader or cross the C ABI: every request of the call fails with Internal.
void serve_guarded(pirgpu_ctx* ctx, Job* const* jobs, size_t n) noexcept {
  const char* what = nullptr;
  std::string msg;
  int code = PIRGPU_INTERNAL;
  try {
    serve(ctx, jobs, n);
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

// Requests that arrive while another thread is serving are queued and served together by whichever thread gets the
// context next (flat combining): no extra latency when the server is idle, cross-client batching under load.
struct Combiner {
  std::mutex m;
  std::condition_variable cv;
  std::deque<Job*> pending;
  bool leader = false;
};
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
  {
    std::unique_lock<std::mutex> lk(cb->m);
    cb->pending.push_back(&job);
    while (!job.done) {
      if (cb->leader) {
        cb->cv.wait(lk);
        continue;
      }
      // lead: serve what is queued (this thread's own request is in there) and hand over
      cb->leader = true;
      std::vector<Job*> batch;
      while (!cb->pending.empty() && batch.size() < kMaxRequestBatch) {
        batch.push_back(cb->pending.front());
        cb->pending.pop_front();
      }
      lk.unlock();
      serve_guarded(ctx, batch.data(), batch.size());
      lk.lock();
      for (Job* j : batch) j->done = true;
      cb->leader = false;
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
  serve_guarded(ctx, ptrs.data(), n);
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

}  // extern "C"
