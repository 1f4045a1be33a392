// wire_windows_test.cpp -- host-only stress test of the wire layer's request windows (pir_amd/csrc/wire.cpp): flat
// combining over two batch sets, windows in flight, key-set pins / handles / evictions, the worker pool, request
// order inside a response.  wire.cpp + wire_codec.cpp are linked against the MOCK device backend below instead of
// ctx.hip: the same C ABI (include/pirgpu.h, csrc/wire.h), replies computed on an "executor" thread that plays the
// GPU's in-order streams with random delays -- so uploads really are asynchronous (a staging buffer reused too early
// gives wrong replies) and groups really complete one after the other.  Built with -fsanitize=thread or
// address,undefined by tests/test_wire_windows.py.  Test infrastructure: nothing here ships.
//
// Mock reply: reply ciphertext r of a query, word w  =  (query word w + keysum(client's keys) + r) mod q_j, with
// keysum = sum of every uploaded key word -- any mix-up of queries, clients, key sets or reply order changes it.
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pirgpu.h"
#include "../../pir_amd/csrc/wire.h"
#include "../../pir_amd/csrc/wire_codec.h"

using namespace pirgpu::wire;

// ------------------------------------------------------------------------------------------------ mock backend
namespace {
constexpr uint32_t kN = 2048, kK = 1, kReplyCts = 2, kGroup = 8;
constexpr uint64_t kQ0 = 0x7e00001ull, kSpecial = 0x7ffe001ull, kT = 12289;   // any values: nothing is transformed here
constexpr size_t kCtw = 2 * kK * kN;

struct MockKeySet {
  std::vector<uint8_t> blob;
  uint64_t sum = 0;
  uint32_t n_keys = 0, gen = 0, pins = 0;
  uint64_t last_use = 0;
};

struct Event {
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
  void set() {
    std::lock_guard<std::mutex> lk(m);
    done = true;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return done; });
  }
};

struct MockSet {
  std::vector<uint64_t> dq;            // "device" copy of the staged queries
  uint32_t staged = 0;
  std::vector<uint32_t> slots;         // key set index per staged query
  uint64_t* host_reply = nullptr;
  uint64_t host_cap = 0;
  bool host_done = false;
  std::vector<std::shared_ptr<Event>> dl;
  std::vector<uint32_t> dl_end;
  size_t dl_next = 0;
  std::vector<uint64_t> hq, hr;
};
thread_local int t_set = 0;
}  // namespace

struct pirgpu_ctx {
  std::recursive_mutex mu;
  std::vector<MockKeySet> keysets{1};
  uint32_t cap = 64, cur = 0, n_active = 1;
  uint64_t clock = 0, uploads = 0, evictions = 0;
  MockSet sets[2];
  std::string err;
  // executor: one in-order queue, like the lanes' streams
  std::mutex qm;
  std::condition_variable qcv;
  std::deque<std::function<void()>> q;
  bool stop = false;
  std::thread exec;
  std::atomic<uint64_t> windows{0}, max_in_flight{0}, in_flight{0};
  // single-query path
  std::vector<uint64_t> w0_query, w0_reply;
  uint32_t w0_slot = 0;
  bool fail_next_run = false;
  std::shared_ptr<Event> fetch_ev[2];
  std::atomic<uint32_t> delay_us{150};   // upper bound of the executor's random delay per task

  pirgpu_ctx() {
    exec = std::thread([this] {
      std::mt19937 rng(7);
      for (;;) {
        std::function<void()> f;
        {
          std::unique_lock<std::mutex> lk(qm);
          qcv.wait(lk, [&] { return stop || !q.empty(); });
          if (q.empty()) return;
          f = std::move(q.front());
          q.pop_front();
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20 + rng() % delay_us.load()));
        f();
      }
    });
  }
  ~pirgpu_ctx() {
    {
      std::lock_guard<std::mutex> lk(qm);
      stop = true;
    }
    qcv.notify_all();
    exec.join();
  }
  void push(std::function<void()> f) {
    {
      std::lock_guard<std::mutex> lk(qm);
      q.push_back(std::move(f));
    }
    qcv.notify_one();
  }
  void drain() {
    auto ev = std::make_shared<Event>();
    push([ev] { ev->set(); });
    ev->wait();
  }
};

namespace {
constexpr uint32_t kSlotBits = 12, kSlotMask = (1u << kSlotBits) - 1;
uint32_t handle_of(pirgpu_ctx* c, uint32_t i) { return i ? (c->keysets[i].gen << kSlotBits) | i : 0; }
int fail(pirgpu_ctx* c, int code, const char* msg) {
  c->err = msg;
  return code;
}
bool resolve(pirgpu_ctx* c, uint32_t handle, uint32_t* index) {
  const uint32_t i = handle & kSlotMask;
  if (i >= c->keysets.size() || (i && c->keysets[i].gen != handle >> kSlotBits) || (!i && handle)) return false;
  *index = i;
  return true;
}
void mock_reply(const uint64_t* query, uint64_t keysum, uint64_t* out) {
  for (uint32_t r = 0; r < kReplyCts; ++r)
    for (size_t w = 0; w < kCtw; ++w) out[r * kCtw + w] = (query[w] + keysum % kQ0 + r) % kQ0;
}
}  // namespace

extern "C" {
int pirgpu_get_params(const pirgpu_ctx*, pirgpu_params* p) {
  memset(p, 0, sizeof(*p));
  p->poly_modulus_degree = kN;
  p->num_data_primes = kK;
  p->coeff_modulus[0] = kQ0;
  p->special_prime = kSpecial;
  p->plain_modulus = kT;
  p->num_dimensions = 2;
  p->dimensions[0] = p->dimensions[1] = 10;
  p->num_pt = 100;
  return 0;
}
uint64_t pirgpu_reply_ct_count(const pirgpu_ctx*) { return kReplyCts; }
const char* pirgpu_last_error(const pirgpu_ctx* c) { return c->err.c_str(); }
void pirgpu_set_error(pirgpu_ctx* c, const char* m) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->err = m ? m : "";
}
void pirgpu_request_lock(pirgpu_ctx* c) { c->mu.lock(); }
void pirgpu_request_unlock(pirgpu_ctx* c) { c->mu.unlock(); }
uint32_t pirgpu_get_concurrency(pirgpu_ctx* c) { return c->n_active; }
int pirgpu_set_concurrency(pirgpu_ctx* c, uint32_t n) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->n_active = n;
  return 0;
}
int pirgpu_clear_galois_keys(pirgpu_ctx*) { return 13; }
int pirgpu_set_galois_key(pirgpu_ctx*, uint32_t, const uint64_t*) { return 13; }
int pirgpu_process_query(pirgpu_ctx*, const uint64_t*, uint32_t, uint64_t*, uint64_t, uint64_t*) { return 13; }

int pirgpu_keyset_lookup(pirgpu_ctx* c, const uint8_t* blob, size_t len, int verify, uint32_t* slot) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  *slot = 0;
  for (uint32_t i = 1; i < c->keysets.size(); ++i) {
    MockKeySet& ks = c->keysets[i];
    // the "fingerprint": length + first 64 bytes (so that the test can build a colliding object)
    if (ks.blob.size() == len && len && memcmp(ks.blob.data(), blob, std::min<size_t>(len, 64)) == 0 &&
        (!verify || memcmp(ks.blob.data(), blob, len) == 0)) {
      ks.last_use = ++c->clock;
      *slot = handle_of(c, i);
      break;
    }
  }
  return 0;
}
int pirgpu_keyset_verify(pirgpu_ctx* c, uint32_t slot, const uint8_t* blob, size_t len) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t i;
  if (!resolve(c, slot, &i) || !i) return 0;
  return c->keysets[i].blob.size() == len && memcmp(c->keysets[i].blob.data(), blob, len) == 0;
}
size_t pirgpu_keyset_blob(pirgpu_ctx* c, uint32_t slot, const uint8_t** b) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t i;
  *b = nullptr;
  if (!resolve(c, slot, &i) || !i) return 0;
  *b = c->keysets[i].blob.data();
  return c->keysets[i].blob.size();
}
int pirgpu_keyset_claim(pirgpu_ctx* c, const uint8_t* blob, size_t len, uint32_t* slot) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t pick = 0;
  for (uint32_t i = 1; i < c->keysets.size() && !pick; ++i)
    if (c->keysets[i].blob.empty() && !c->keysets[i].n_keys && !c->keysets[i].pins) pick = i;
  if (!pick && c->keysets.size() < (size_t)c->cap + 1) {
    c->keysets.emplace_back();
    pick = (uint32_t)c->keysets.size() - 1;
  }
  if (!pick) {
    uint64_t best = UINT64_MAX;
    for (uint32_t i = 1; i < c->keysets.size(); ++i)
      if (c->keysets[i].last_use < best && !c->keysets[i].pins) {
        best = c->keysets[i].last_use;
        pick = i;
      }
    if (!pick) return fail(c, PIRGPU_FAILED_PRECONDITION, "every key set slot is in use by the requests being processed");
    c->drain();   // like the real one: queued work may still read the evicted keys
    MockKeySet& ks = c->keysets[pick];
    ks.blob.clear();
    ks.sum = 0;
    ks.n_keys = 0;
    ++ks.gen;
    ++c->evictions;
  }
  c->keysets[pick].blob.assign(blob, blob + len);
  c->keysets[pick].last_use = ++c->clock;
  *slot = handle_of(c, pick);
  return 0;
}
int pirgpu_keyset_release(pirgpu_ctx* c, uint32_t slot) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t i;
  if (!resolve(c, slot, &i) || !i) return fail(c, PIRGPU_INVALID_ARGUMENT, "bad slot");
  if (c->keysets[i].pins) return fail(c, PIRGPU_FAILED_PRECONDITION, "pinned");
  c->keysets[i] = MockKeySet{{}, 0, 0, c->keysets[i].gen + 1, 0, 0};
  return 0;
}
int pirgpu_keyset_set_key(pirgpu_ctx* c, uint32_t slot, uint32_t, const uint64_t* key) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t i;
  if (!resolve(c, slot, &i)) return fail(c, PIRGPU_FAILED_PRECONDITION, "stale key set handle");
  for (size_t w = 0; w < (size_t)kK * 2 * (kK + 1) * kN; ++w) c->keysets[i].sum += key[w];
  ++c->keysets[i].n_keys;
  ++c->uploads;
  return 0;
}
int pirgpu_keyset_set_keys(pirgpu_ctx* c, uint32_t slot, uint32_t n, const uint32_t* elts, const uint64_t* const* keys) {
  for (uint32_t i = 0; i < n; ++i) {
    const int rc = pirgpu_keyset_set_key(c, slot, elts[i], keys[i]);
    if (rc) return rc;
  }
  return 0;
}
int pirgpu_keyset_pin(pirgpu_ctx* c, uint32_t slot) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t i;
  if (!resolve(c, slot, &i)) return fail(c, PIRGPU_FAILED_PRECONDITION, "stale key set handle");
  if (i) ++c->keysets[i].pins;
  return 0;
}
int pirgpu_keyset_unpin(pirgpu_ctx* c, uint32_t slot) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  const uint32_t i = slot & kSlotMask;
  if (i && i < c->keysets.size()) {
    assert(c->keysets[i].pins > 0);
    --c->keysets[i].pins;
  }
  return 0;
}
int pirgpu_keyset_stats(pirgpu_ctx* c, uint64_t st[4]) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  st[0] = 0;
  for (uint32_t i = 1; i < c->keysets.size(); ++i) st[0] += c->keysets[i].n_keys ? 1 : 0;
  st[1] = c->uploads;
  st[2] = c->evictions;
  st[3] = c->cap;
  return 0;
}
uint32_t pirgpu_current_keyset(pirgpu_ctx* c) { return handle_of(c, c->cur); }
void pirgpu_keyset_selection_get(pirgpu_ctx* c, uint32_t sel[2]) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  sel[0] = c->cur;
  sel[1] = 0;
}
void pirgpu_keyset_selection_set(pirgpu_ctx* c, const uint32_t sel[2]) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->cur = sel[0];
}
int pirgpu_query_use_keyset(pirgpu_ctx* c, uint32_t slot) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  uint32_t i;
  if (!resolve(c, slot, &i)) return fail(c, PIRGPU_FAILED_PRECONDITION, "stale key set handle");
  c->cur = i;
  return 0;
}

int pirgpu_batch_select(pirgpu_ctx*, uint32_t which) {
  t_set = (int)which;
  return 0;
}
uint64_t* pirgpu_host_query_buffer(pirgpu_ctx* c, uint32_t n) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  MockSet& s = c->sets[t_set];
  if (s.hq.size() < (size_t)n * kCtw) s.hq.resize((size_t)n * kCtw);
  return s.hq.data();
}
uint64_t* pirgpu_host_reply_buffer(pirgpu_ctx* c, uint32_t n) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  MockSet& s = c->sets[t_set];
  if (s.hr.size() < (size_t)n * kReplyCts * kCtw) s.hr.resize((size_t)n * kReplyCts * kCtw);
  return s.hr.data();
}
int pirgpu_batch_stage_async(pirgpu_ctx* c, const uint64_t* q, uint32_t nq, uint32_t count) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (nq != 1) return fail(c, PIRGPU_INVALID_ARGUMENT, "Number of ciphertexts doesn't match number of items for oblivious expansion.");
  MockSet& s = c->sets[t_set];
  if (s.dq.size() < (size_t)count * kCtw) s.dq.resize((size_t)count * kCtw);
  uint64_t* dst = s.dq.data();
  for (uint32_t p = 0; p < count; p += kGroup) {   // the upload happens LATER, piece by piece, on the executor
    const uint32_t n = std::min(kGroup, count - p);
    c->push([dst, q, p, n] { memcpy(dst + (size_t)p * kCtw, q + (size_t)p * kCtw, (size_t)n * kCtw * 8); });
  }
  s.staged = count;
  s.slots.assign(count, 0);
  return 0;
}
int pirgpu_batch_stage(pirgpu_ctx* c, const uint64_t* q, uint32_t nq, uint32_t count) {
  int rc = pirgpu_batch_stage_async(c, q, nq, count);
  if (!rc) c->drain();
  return rc;
}
int pirgpu_batch_unstage(pirgpu_ctx* c) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->sets[t_set].staged = 0;
  c->sets[t_set].slots.clear();
  return 0;
}
int pirgpu_batch_set_keysets(pirgpu_ctx* c, const uint32_t* slots, uint32_t count) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  MockSet& s = c->sets[t_set];
  if (count != s.staged) return fail(c, PIRGPU_INVALID_ARGUMENT, "one key set slot per staged query");
  for (uint32_t i = 0; i < count; ++i)
    if (!resolve(c, slots[i], &s.slots[i])) return fail(c, PIRGPU_FAILED_PRECONDITION, "stale key set handle");
  return 0;
}
int pirgpu_batch_set_host_replies(pirgpu_ctx* c, uint64_t* host, uint64_t cap) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->sets[t_set].host_reply = host;
  c->sets[t_set].host_cap = host ? cap : 0;
  c->sets[t_set].host_done = false;
  return 0;
}
int pirgpu_batch_run(pirgpu_ctx* c) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  MockSet& s = c->sets[t_set];
  if (!s.staged) return fail(c, PIRGPU_FAILED_PRECONDITION, "no batch has been staged");
  if (c->fail_next_run) {
    c->fail_next_run = false;
    c->drain();   // (the uploads already queued are harmless in reality; here they would race with the next window's parser)
    return fail(c, PIRGPU_INTERNAL, "injected failure");
  }
  assert(s.host_reply && (uint64_t)s.staged * kReplyCts <= s.host_cap);
  s.host_done = true;
  s.dl.clear();
  s.dl_end.clear();
  s.dl_next = 0;
  ++c->windows;
  const uint64_t now = ++c->in_flight;
  uint64_t seen = c->max_in_flight.load();
  while (now > seen && !c->max_in_flight.compare_exchange_weak(seen, now)) {
  }
  for (uint32_t g0 = 0; g0 < s.staged; g0 += kGroup) {
    const uint32_t n = std::min(kGroup, s.staged - g0);
    auto ev = std::make_shared<Event>();
    s.dl.push_back(ev);
    s.dl_end.push_back(g0 + n);
    std::vector<uint64_t> sums(n);
    for (uint32_t q = 0; q < n; ++q) sums[q] = c->keysets[s.slots[g0 + q]].sum;   // keys are read when the group RUNS in
                                                                                  // reality; pinned sets cannot change
    const uint64_t* dq = s.dq.data();
    uint64_t* hr = s.host_reply;
    const bool last = g0 + n >= s.staged;
    c->push([c, dq, hr, g0, n, sums, ev, last] {
      for (uint32_t q = 0; q < n; ++q) mock_reply(dq + (size_t)(g0 + q) * kCtw, sums[q], hr + (size_t)(g0 + q) * kReplyCts * kCtw);
      if (last) --c->in_flight;
      ev->set();
    });
  }
  return 0;
}
int pirgpu_batch_next_host_replies(pirgpu_ctx* c, uint32_t* ready) {
  std::shared_ptr<Event> ev;
  uint32_t end = 0;
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    MockSet& s = c->sets[t_set];
    if (!s.host_done) return fail(c, PIRGPU_FAILED_PRECONDITION, "not group-wise");
    if (s.dl_next >= s.dl.size()) {
      *ready = s.dl_end.empty() ? 0 : s.dl_end.back();
      return 0;
    }
    ev = s.dl[s.dl_next];
    end = s.dl_end[s.dl_next++];
  }
  ev->wait();
  *ready = end;
  return 0;
}
int pirgpu_batch_fetch(pirgpu_ctx* c, uint64_t*, uint64_t, uint64_t*) { return fail(c, PIRGPU_INTERNAL, "mock: fetch not expected"); }
int pirgpu_sync(pirgpu_ctx* c) {
  c->drain();
  return 0;
}

int pirgpu_query_stage(pirgpu_ctx* c, const uint64_t* q, uint32_t nq) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (nq != 1) return fail(c, PIRGPU_INVALID_ARGUMENT, "Number of ciphertexts doesn't match number of items for oblivious expansion.");
  c->w0_query.assign(q, q + kCtw);
  return 0;
}
int pirgpu_query_stage_async(pirgpu_ctx* c, const uint64_t* q, uint32_t nq) { return pirgpu_query_stage(c, q, nq); }
int pirgpu_query_run(pirgpu_ctx* c) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->w0_slot = c->cur;
  return 0;
}
int pirgpu_query_fetch(pirgpu_ctx* c, uint64_t* reply, uint64_t cap, uint64_t* count) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (cap < kReplyCts) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
  c->drain();
  mock_reply(c->w0_query.data(), c->keysets[c->w0_slot].sum, reply);
  *count = kReplyCts;
  return 0;
}
// the lone query's reply in two halves: the second half is written LATER (on the executor), so a serialiser that did
// not wait for it would read stale bytes
int pirgpu_query_fetch_begin(pirgpu_ctx* c, uint64_t* reply, uint64_t cap, uint64_t* count, uint64_t* first_part) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (cap < kReplyCts) return fail(c, PIRGPU_INVALID_ARGUMENT, "reply buffer too small");
  auto full = std::make_shared<std::vector<uint64_t>>(kReplyCts * kCtw);
  mock_reply(c->w0_query.data(), c->keysets[c->w0_slot].sum, full->data());
  const uint64_t a = (kReplyCts + 1) / 2;
  for (int part = 0; part < 2; ++part) {
    c->fetch_ev[part] = std::make_shared<Event>();
    auto ev = c->fetch_ev[part];
    const uint64_t lo = part ? a : 0, hi = part ? kReplyCts : a;
    memset(reply + lo * kCtw, 0xEE, (hi - lo) * kCtw * 8);
    c->push([=] {
      memcpy(reply + lo * kCtw, full->data() + lo * kCtw, (hi - lo) * kCtw * 8);
      ev->set();
    });
  }
  *count = kReplyCts;
  *first_part = a;
  return 0;
}
int pirgpu_query_fetch_wait(pirgpu_ctx* c, int part) {
  std::shared_ptr<Event> ev;
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    ev = c->fetch_ev[part];
  }
  ev->wait();
  return 0;
}
}  // extern "C"

// ------------------------------------------------------------------------------------------------ test driver
namespace {

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

struct Client {
  std::string keys_blob;
  std::string relin_blob;   // RelinKeys: validated by the server (server.cpp:53-58), otherwise unused in this mode
  uint64_t keysum = 0;
};

Shape g_sh;

Client make_client(uint32_t seed, bool seeded) {
  std::mt19937_64 rng(seed);
  const size_t pk_words = (size_t)2 * (kK + 1) * kN, key_words = kK * pk_words;
  const uint32_t n_entries = 6;   // Galois indices 0..5, index 2 absent
  std::vector<std::vector<uint64_t>> keys(n_entries);
  std::vector<std::vector<uint8_t>> seeds(n_entries);
  std::vector<const uint64_t*> entries(n_entries, nullptr);
  std::vector<const uint8_t*> seed_ptrs(n_entries, nullptr);
  Client c;
  for (uint32_t e = 0; e < n_entries; ++e) {
    if (e == 2) continue;
    keys[e].resize(key_words);
    seeds[e].resize(kK * kSeedBytes);
    for (auto& b : seeds[e]) b = (uint8_t)rng();
    for (uint32_t j = 0; j < kK; ++j) {
      uint64_t* pk = keys[e].data() + j * pk_words;
      for (uint32_t poly = 0; poly < 2; ++poly)
        for (uint32_t r = 0; r <= kK; ++r) {
          const uint64_t q = g_sh.q[r];
          uint64_t* dst = pk + ((size_t)poly * (kK + 1) + r) * kN;
          for (uint32_t i = 0; i < kN; ++i) dst[i] = rng() % q;
        }
      if (seeded) {   // the c1 half is what the server will re-sample from the seed
        SealPrng prng(seeds[e].data() + j * kSeedBytes);
        uint64_t mods[kK + 1];
        for (uint32_t r = 0; r <= kK; ++r) mods[r] = g_sh.q[r];
        sample_poly_uniform(prng, mods, kK + 1, kN, pk + (size_t)(kK + 1) * kN);
      }
    }
    entries[e] = keys[e].data();
    seed_ptrs[e] = seeds[e].data();
    for (uint64_t w : keys[e]) c.keysum += w;
  }
  c.keys_blob = save_kswitch_keys(g_sh, entries, seeded ? &seed_ptrs : nullptr);
  // RelinKeys = a KSwitchKeys object with one entry; every second client sends one (seeded like its Galois keys)
  if (seed % 2 == 0) c.relin_blob = save_kswitch_keys(g_sh, {entries[0]}, seeded ? new std::vector<const uint8_t*>{seed_ptrs[0]} : nullptr);
  return c;
}

std::vector<uint64_t> make_query(uint64_t seed) {
  std::mt19937_64 rng(seed);
  std::vector<uint64_t> q(kCtw);
  for (auto& w : q) w = rng() % kQ0;
  return q;
}

std::string make_request(const Client& c, const std::vector<std::vector<uint64_t>>& queries, uint32_t cts_per_query = 1) {
  std::string req;
  for (const auto& q : queries) {
    std::string cts;
    for (uint32_t i = 0; i < cts_per_query; ++i) put_bytes_field(cts, 1, save_ciphertext(g_sh, q.data()));
    put_bytes_field(req, 1, cts);
  }
  put_bytes_field(req, 2, c.keys_blob);
  if (!c.relin_blob.empty()) put_bytes_field(req, 3, c.relin_blob);
  return req;
}

// pir.Response -> replies[query][kReplyCts * kCtw]
std::vector<std::vector<uint64_t>> parse_response(const uint8_t* p, size_t len) {
  std::vector<std::vector<uint64_t>> out;
  Reader r{p, p + len};
  while (r.p < r.end) {
    uint64_t tag;
    CHECK(r.varint(tag) && tag == ((1 << 3) | 2));
    const uint8_t* d;
    size_t l;
    CHECK(r.bytes(d, l));
    Reader cr{d, d + l};
    std::vector<uint64_t> reply, one;
    while (cr.p < cr.end) {
      CHECK(cr.varint(tag) && tag == ((1 << 3) | 2));
      const uint8_t* cd;
      size_t cl;
      CHECK(cr.bytes(cd, cl));
      Cursor c{cd, cd + cl};
      load_ciphertext(c, g_sh, false, one);
      reply.insert(reply.end(), one.begin(), one.end());
    }
    out.push_back(std::move(reply));
  }
  return out;
}

void check_response(const Client& c, const std::vector<std::vector<uint64_t>>& queries, const uint8_t* resp, size_t len) {
  auto replies = parse_response(resp, len);
  CHECK(replies.size() == queries.size());
  std::vector<uint64_t> want(kReplyCts * kCtw);
  for (size_t i = 0; i < queries.size(); ++i) {
    mock_reply(queries[i].data(), c.keysum, want.data());
    CHECK(replies[i] == want);
  }
}

void one_request(pirgpu_ctx* ctx, const Client& c, const std::vector<std::vector<uint64_t>>& queries) {
  const std::string req = make_request(c, queries);
  uint8_t* resp = nullptr;
  size_t len = 0;
  const int rc = pirgpu_process_request(ctx, (const uint8_t*)req.data(), req.size(), &resp, &len);
  if (rc) fprintf(stderr, "process_request failed: %d %s\n", rc, pirgpu_last_error(ctx));
  CHECK(rc == 0);
  check_response(c, queries, resp, len);
  pirgpu_free(resp);
}

void many_requests(pirgpu_ctx* ctx, const std::vector<const Client*>& cl, const std::vector<std::vector<std::vector<uint64_t>>>& qs,
                   const std::vector<int>& expect = {}) {
  const uint32_t n = (uint32_t)cl.size();
  std::vector<std::string> reqs(n);
  for (uint32_t i = 0; i < n; ++i) reqs[i] = make_request(*cl[i], qs[i]);
  std::vector<const uint8_t*> ptrs(n);
  std::vector<size_t> lens(n), rlens(n);
  std::vector<uint8_t*> resps(n);
  std::vector<int> status(n);
  for (uint32_t i = 0; i < n; ++i) {
    ptrs[i] = (const uint8_t*)reqs[i].data();
    lens[i] = reqs[i].size();
  }
  (void)pirgpu_process_requests(ctx, n, ptrs.data(), lens.data(), resps.data(), rlens.data(), status.data());
  for (uint32_t i = 0; i < n; ++i) {
    const int want = expect.empty() ? 0 : expect[i];
    if (status[i] != want) fprintf(stderr, "request %u: status %d (%s), expected %d\n", i, status[i], pirgpu_request_error(i), want);
    CHECK(status[i] == want);
    if (!status[i]) {
      check_response(*cl[i], qs[i], resps[i], rlens[i]);
      pirgpu_free(resps[i]);
    }
  }
}

// One call of pirgpu_process_requests in two halves (pirgpu_process_requests_begin / _end): everything the call touches
// lives here until end() has returned.
struct AsyncCall {
  std::vector<const Client*> cl;
  std::vector<std::vector<std::vector<uint64_t>>> qs;
  std::vector<std::string> reqs;
  std::vector<const uint8_t*> ptrs;
  std::vector<size_t> lens, rlens;
  std::vector<uint8_t*> resps;
  std::vector<int> status;
  void* call = nullptr;
  void prepare() {
    const uint32_t n = (uint32_t)cl.size();
    reqs.resize(n);
    ptrs.resize(n);
    lens.resize(n);
    rlens.assign(n, 0);
    resps.assign(n, nullptr);
    status.assign(n, -1);
    for (uint32_t i = 0; i < n; ++i) {
      reqs[i] = make_request(*cl[i], qs[i]);
      ptrs[i] = (const uint8_t*)reqs[i].data();
      lens[i] = reqs[i].size();
    }
  }
  void begin(pirgpu_ctx* ctx) {
    const uint32_t n = (uint32_t)cl.size();
    CHECK(pirgpu_process_requests_begin(ctx, n, ptrs.data(), lens.data(), resps.data(), rlens.data(), status.data(), &call) == 0);
    CHECK(call != nullptr);
  }
  void end(int want_rc = 0) {
    CHECK(pirgpu_process_requests_end(call) == want_rc);
    call = nullptr;
    for (size_t i = 0; i < cl.size(); ++i) {
      if (status[i]) fprintf(stderr, "async request %zu: status %d (%s)\n", i, status[i], pirgpu_request_error((uint32_t)i));
      CHECK(status[i] == 0);
      check_response(*cl[i], qs[i], resps[i], rlens[i]);
      pirgpu_free(resps[i]);
    }
  }
};

void no_pins_left(pirgpu_ctx* ctx) {
  for (auto& ks : ctx->keysets) CHECK(ks.pins == 0);
  CHECK(ctx->in_flight == 0);
}

}  // namespace

int main() {
  pirgpu_params prm;
  pirgpu_get_params(nullptr, &prm);
  g_sh = make_shape(prm);
  std::vector<Client> clients;
  for (uint32_t i = 0; i < 12; ++i) clients.push_back(make_client(100 + i, i % 3 == 1));   // every third: seed-compressed

  {  // (a) lone requests: single-query path, repeat client, multi-query request
    pirgpu_ctx ctx;
    one_request(&ctx, clients[0], {make_query(1)});
    one_request(&ctx, clients[0], {make_query(2)});
    one_request(&ctx, clients[1], {make_query(3)});                 // seeded keys
    one_request(&ctx, clients[1], {make_query(4), make_query(5), make_query(6)});
    CHECK(ctx.uploads == 2 * 5);
    no_pins_left(&ctx);
    printf("(a) lone requests OK\n");
  }
  {  // (b) 100 requests of 12 clients in one call: windows of 32 clients, two in flight
    pirgpu_ctx ctx;
    ctx.delay_us = 4000;   // a slow "GPU": the second window is certainly queued while the first one still runs
    std::vector<const Client*> cl;
    std::vector<std::vector<std::vector<uint64_t>>> qs;
    for (uint32_t i = 0; i < 100; ++i) {
      cl.push_back(&clients[(i * 7) % 12]);
      qs.push_back({make_query(1000 + i)});
      if (i % 9 == 0) qs.back().push_back(make_query(5000 + i));
    }
    many_requests(&ctx, cl, qs);
    CHECK(ctx.windows >= 3);
    CHECK(ctx.max_in_flight == 2);
    CHECK(ctx.uploads == 12 * 5);
    no_pins_left(&ctx);
    printf("(b) 100 requests in one call OK (%llu windows, %llu in flight at most)\n", (unsigned long long)ctx.windows.load(),
           (unsigned long long)ctx.max_in_flight.load());
  }
  {  // (c) one request with 150 queries: cut over three windows, replies in query order
    pirgpu_ctx ctx;
    std::vector<std::vector<uint64_t>> q;
    for (uint32_t i = 0; i < 150; ++i) q.push_back(make_query(9000 + i));
    one_request(&ctx, clients[4], q);
    CHECK(ctx.windows == 3);
    no_pins_left(&ctx);
    printf("(c) 150-query request OK\n");
  }
  {  // (d) failures stay local: malformed request, wrong ciphertext count, failing batch run
    pirgpu_ctx ctx;
    std::vector<const Client*> cl = {&clients[0], &clients[1], &clients[2], &clients[3]};
    std::vector<std::vector<std::vector<uint64_t>>> qs = {{make_query(1)}, {make_query(2)}, {make_query(3), make_query(4)}, {make_query(5)}};
    std::vector<std::string> reqs;
    for (size_t i = 0; i < 4; ++i) reqs.push_back(make_request(*cl[i], qs[i], i == 2 ? 2 : 1));   // request 2: 2 cts per query
    reqs[1].resize(reqs[1].size() - 9);                                                            // request 1: truncated
    std::vector<const uint8_t*> ptrs;
    std::vector<size_t> lens, rlens(4);
    std::vector<uint8_t*> resps(4);
    std::vector<int> status(4);
    for (auto& r : reqs) {
      ptrs.push_back((const uint8_t*)r.data());
      lens.push_back(r.size());
    }
    int rc = pirgpu_process_requests(&ctx, 4, ptrs.data(), lens.data(), resps.data(), rlens.data(), status.data());
    CHECK(rc == PIRGPU_INVALID_ARGUMENT);
    CHECK(status[0] == 0 && status[1] == PIRGPU_INVALID_ARGUMENT && status[2] == PIRGPU_INVALID_ARGUMENT && status[3] == 0);
    CHECK(std::string(pirgpu_request_error(2)).find("Number of ciphertexts") != std::string::npos);
    check_response(*cl[0], qs[0], resps[0], rlens[0]);
    check_response(*cl[3], qs[3], resps[3], rlens[3]);
    pirgpu_free(resps[0]);
    pirgpu_free(resps[3]);
    no_pins_left(&ctx);
    {   // a malformed RelinKeys object fails ITS request with InvalidArgument -- in a window and alone -- nobody else's
      Client bad = clients[2];
      bad.relin_blob = clients[0].relin_blob.substr(0, clients[0].relin_blob.size() - 11);
      CHECK(!clients[0].relin_blob.empty());
      many_requests(&ctx, {&clients[0], &bad, &clients[3]}, {{make_query(21)}, {make_query(22)}, {make_query(23)}},
                    {0, PIRGPU_INVALID_ARGUMENT, 0});
      const std::string req = make_request(bad, {make_query(24)});
      uint8_t* resp = nullptr;
      size_t len = 0;
      CHECK(pirgpu_process_request(&ctx, (const uint8_t*)req.data(), req.size(), &resp, &len) == PIRGPU_INVALID_ARGUMENT);
      one_request(&ctx, clients[2], {make_query(25)});
      no_pins_left(&ctx);
    }
    ctx.fail_next_run = true;   // the whole window fails with the backend's status, nothing is left pinned or in flight
    many_requests(&ctx, {&clients[0], &clients[3]}, {{make_query(7)}, {make_query(8)}}, {PIRGPU_INTERNAL, PIRGPU_INTERNAL});
    no_pins_left(&ctx);
    many_requests(&ctx, {&clients[0], &clients[3]}, {{make_query(7)}, {make_query(8)}});
    printf("(d) failures stay local OK\n");
  }
  {  // (e) same fingerprint, different bytes: the key compare under the window catches it, the request is served again
    pirgpu_ctx ctx;
    Client twin = clients[5];
    {   // the last residue of the object changed by one: still a valid object, same length, same first bytes
      uint64_t v;
      memcpy(&v, &twin.keys_blob[twin.keys_blob.size() - 8], 8);
      v = v ? v - 1 : 1;
      memcpy(&twin.keys_blob[twin.keys_blob.size() - 8], &v, 8);
    }
    // its keysum: reload through the codec
    {
      uint64_t sum = 0;
      load_kswitch_keys(g_sh, (const uint8_t*)twin.keys_blob.data(), twin.keys_blob.size(), [&](uint64_t, const uint64_t* key) {
        for (size_t w = 0; w < (size_t)kK * 2 * (kK + 1) * kN; ++w) sum += key[w];
      });
      twin.keysum = sum;
    }
    CHECK(twin.keysum != clients[5].keysum);
    many_requests(&ctx, {&clients[5], &clients[6]}, {{make_query(1)}, {make_query(2)}});
    many_requests(&ctx, {&twin, &clients[6]}, {{make_query(3)}, {make_query(4)}});     // inside a window
    one_request(&ctx, clients[5], {make_query(5)});
    one_request(&ctx, twin, {make_query(6)});                                            // lone request
    no_pins_left(&ctx);
    printf("(e) fingerprint collision OK\n");
  }
  {  // (f) many threads, more clients than key set slots (capacity 4: windows of 2 clients, evictions all the time)
    pirgpu_ctx ctx;
    ctx.cap = 4;
    std::atomic<int> errors{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
      th.emplace_back([&, t] {
        std::mt19937 rng(t);
        for (int it = 0; it < 25; ++it) {
          const Client& c = clients[rng() % clients.size()];
          std::vector<std::vector<uint64_t>> q;
          const int nq = 1 + (int)(rng() % 3);
          for (int i = 0; i < nq; ++i) q.push_back(make_query(rng()));
          if (t % 4 == 3) {
            many_requests(&ctx, {&c, &clients[(t + it) % clients.size()]}, {q, {make_query(it)}});
          } else {
            one_request(&ctx, c, q);
          }
        }
      });
    for (auto& x : th) x.join();
    CHECK(errors == 0);
    CHECK(ctx.evictions > 0);
    no_pins_left(&ctx);
    printf("(f) 8 threads, 12 clients on 4 slots OK (%llu windows, %llu evictions, %llu in flight at most)\n",
           (unsigned long long)ctx.windows.load(), (unsigned long long)ctx.evictions, (unsigned long long)ctx.max_in_flight.load());
  }
  {  // (g) capacity 1: one window in flight, one client per window
    pirgpu_ctx ctx;
    ctx.cap = 1;
    many_requests(&ctx, {&clients[0], &clients[1], &clients[0]}, {{make_query(1), make_query(2)}, {make_query(3), make_query(4)}, {make_query(5), make_query(6)}});
    CHECK(ctx.max_in_flight == 1);
    no_pins_left(&ctx);
    // ... also with several CALLING threads, each with its own client (two queries per request: the batch path, whose
    // windows pin their client's set): with one slot only one leader may have a window open -- a second one's claim would
    // find the only slot pinned and fail a legitimate request with FailedPrecondition
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
      th.emplace_back([&, t] {
        for (int it = 0; it < 10; ++it) {
          if (t & 1) many_requests(&ctx, {&clients[t]}, {{make_query(100 * t + it), make_query(7 * it + t)}});
          else one_request(&ctx, clients[t], {make_query(100 * t + it), make_query(7 * it + t)});
        }
      });
    for (auto& x : th) x.join();
    CHECK(ctx.max_in_flight == 1);
    no_pins_left(&ctx);
    printf("(g) capacity 1 OK\n");
  }
  {  // (h) ONE calling thread, two calls in flight (begin / begin / end / begin / end ...): the second call's window is
     // parsed, staged and queued while the first one's is still on the "GPU"
    pirgpu_ctx ctx;
    ctx.delay_us = 6000;
    const int n_calls = 6;
    std::vector<AsyncCall> calls(n_calls);
    for (int c = 0; c < n_calls; ++c)
      for (uint32_t i = 0; i < 10; ++i) {
        calls[c].cl.push_back(&clients[(c * 5 + i) % clients.size()]);
        calls[c].qs.push_back({make_query(70000 + 100 * c + i)});
        if (i % 4 == 0) calls[c].qs.back().push_back(make_query(80000 + 100 * c + i));
      }
    for (auto& c : calls) c.prepare();     // (building the requests is slow under the sanitizers: not between the calls)
    calls[0].begin(&ctx);
    for (int c = 1; c < n_calls; ++c) {
      calls[c].begin(&ctx);
      calls[c - 1].end();
    }
    calls[n_calls - 1].end();
    CHECK(ctx.max_in_flight == 2);
    no_pins_left(&ctx);
    CHECK(pirgpu_process_requests_end(nullptr) == PIRGPU_INVALID_ARGUMENT);
    void* none = (void*)1;
    CHECK(pirgpu_process_requests_begin(&ctx, 1, nullptr, nullptr, nullptr, nullptr, nullptr, &none) == PIRGPU_INVALID_ARGUMENT && none == nullptr);
    printf("(h) one caller, two calls in flight OK (%llu windows)\n", (unsigned long long)ctx.windows.load());
  }
  printf("wire_windows_test OK\n");
  return 0;
}
