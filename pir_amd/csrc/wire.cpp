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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
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

int pirgpu_process_request(pirgpu_ctx* ctx, const uint8_t* request, size_t request_len, uint8_t** response,
                           size_t* response_len) {
  if (!ctx || (!request && request_len) || !response || !response_len) return PIRGPU_INVALID_ARGUMENT;
  *response = nullptr;
  *response_len = 0;
  pirgpu_params prm;
  if (pirgpu_get_params(ctx, &prm)) return PIRGPU_INVALID_ARGUMENT;
  const Shape sh = make_shape(prm);
  const size_t ctw = (size_t)2 * sh.k * sh.N;
  // One request = one critical section on the context: the installed Galois keys are context state, so
  // two threads serving different clients must not interleave "install keys" and "run queries"
  // (PIRServer::ProcessRequest is const and re-entrant in the reference because its keys are locals).
  pirgpu_request_lock(ctx);
  struct Unlock {
    pirgpu_ctx* c;
    ~Unlock() { pirgpu_request_unlock(c); }
  } unlock{ctx};
  try {
    ParsedRequest pr = parse_request(request, request_len);
    // --- SEALDeserialize<GaloisKeys> (server.cpp:46-48): empty bytes -> load throws -> InvalidArgument.
    // The reference re-parses the keys on every request; here a client that repeats its (multi-MB)
    // key blob byte for byte keeps the device-resident keys of its previous request.
    int rc = 0;
    if (!pr.galois_keys_len || !pirgpu_keys_blob_matches(ctx, pr.galois_keys, pr.galois_keys_len)) {
      // parse and validate the whole object first: a malformed blob must not leave half-installed keys behind
      std::vector<std::pair<uint32_t, std::vector<uint64_t>>> parsed;
      const size_t key_words = (size_t)sh.k * 2 * (sh.k + 1) * sh.N;
      load_kswitch_keys(sh, pr.galois_keys, pr.galois_keys_len, [&](uint64_t index, const uint64_t* key) {
        parsed.emplace_back((uint32_t)(2 * index + 1), std::vector<uint64_t>(key, key + key_words));
      });
      rc = pirgpu_clear_galois_keys(ctx);
      if (rc) throw Err{rc, pirgpu_last_error(ctx)};
      for (auto& kv : parsed) {
        rc = pirgpu_set_galois_key(ctx, kv.first, kv.second.data());
        if (rc) throw Err{rc, pirgpu_last_error(ctx)};
      }
      pirgpu_keys_blob_set(ctx, pr.galois_keys, pr.galois_keys_len);
    }
    // --- SEALDeserialize<RelinKeys> when present (server.cpp:53-58): only CT-multiplication mode uses them,
    // but a malformed non-empty field is InvalidArgument in the reference, so it is parsed and validated here too.
    if (pr.relin_keys_len) load_kswitch_keys(sh, pr.relin_keys, pr.relin_keys_len, nullptr);
    // --- per query: LoadCiphertexts -> processQuery -> SaveCiphertexts (server.cpp:60-63,173-195)
    const uint64_t n_reply = pirgpu_reply_ct_count(ctx);
    uint64_t dim_sum = 0;
    for (uint32_t l = 0; l < prm.num_dimensions; ++l) dim_sum += prm.dimensions[l];
    const uint32_t nq_expected = (uint32_t)(dim_sum / sh.N + 1);  // server.cpp:154
    std::vector<uint64_t> reply, qbuf;
    std::string out;
    auto append_reply = [&](const uint64_t* cts_words, uint64_t n) {
      std::string cts;
      for (uint64_t i = 0; i < n; ++i) put_bytes_field(cts, 1, save_ciphertext(sh, cts_words + i * ctw));
      put_bytes_field(out, 1, cts);  // Response.reply (payload.proto:39-42)
    };
    // Several queries in one request (the loop of server.cpp:60-63) run through the batch pipeline (grouped
    // expansion, shared database passes) when every query has the ciphertext count the dimensions call for;
    // otherwise the sequential path reports the error at the offending query like the reference does
    // (nothing has run on the device at that point: parsing is host work).
    bool batched = false;
    if (pr.queries.size() > 1) {
      std::vector<uint64_t> all;
      bool uniform = true;
      for (size_t i = 0; i < pr.queries.size() && uniform; ++i) {
        const uint32_t nq = load_query(sh, pr.queries[i].first, pr.queries[i].second, qbuf);
        uniform = nq == nq_expected;
        if (uniform) all.insert(all.end(), qbuf.begin(), qbuf.end());
      }
      if (uniform) {
        const uint32_t total = (uint32_t)pr.queries.size();
        const uint32_t before = pirgpu_get_concurrency(ctx);
        rc = pirgpu_set_concurrency(ctx, std::max<uint32_t>(before, std::min<uint32_t>(total, 16)));
        for (uint32_t first = 0; first < total && !rc; first += kMaxRequestBatch) {
          const uint32_t count = std::min<uint32_t>(kMaxRequestBatch, total - first);
          rc = pirgpu_batch_stage(ctx, all.data() + (size_t)first * nq_expected * ctw, nq_expected, count);
          if (!rc) rc = pirgpu_batch_run(ctx);
          reply.resize((size_t)count * n_reply * ctw);
          uint64_t got = 0;
          if (!rc) rc = pirgpu_batch_fetch(ctx, reply.data(), (uint64_t)count * n_reply, &got);
          if (rc) break;
          for (uint32_t i = 0; i < count; ++i) append_reply(reply.data() + (size_t)i * n_reply * ctw, n_reply);
        }
        const std::string msg = rc ? pirgpu_last_error(ctx) : "";
        (void)pirgpu_set_concurrency(ctx, before);
        if (rc) throw Err{rc, msg};
        batched = true;
      }
    }
    if (!batched) {
      reply.resize(n_reply * ctw);
      for (auto& qm : pr.queries) {
        const uint32_t nq = load_query(sh, qm.first, qm.second, qbuf);
        uint64_t got = 0;
        rc = pirgpu_process_query(ctx, qbuf.data(), nq, reply.data(), n_reply, &got);
        if (rc) throw Err{rc, pirgpu_last_error(ctx)};
        append_reply(reply.data(), got);
      }
    }
    uint8_t* buf = (uint8_t*)malloc(out.size() ? out.size() : 1);
    if (!buf) return PIRGPU_INTERNAL;
    memcpy(buf, out.data(), out.size());
    *response = buf;
    *response_len = out.size();
    return PIRGPU_OK;
  } catch (const Err& e) {
    pirgpu_set_error(ctx, e.msg.c_str());
    return e.code;
  } catch (const std::exception& e) {
    pirgpu_set_error(ctx, e.what());
    return PIRGPU_INTERNAL;
  }
}

}  // extern "C"
