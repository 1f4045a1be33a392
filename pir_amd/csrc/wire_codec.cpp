// wire_codec.cpp -- pir/proto framing primitives + SEAL 3.5.6 binary object codec (host only, no
// device dependency: shared by the server library (wire.cpp) and the CPU client library (client.cpp)).
// Format restated in SURVEY.md App. A.6 from serialization.h:81-139 -> Ciphertext / PublicKey /
// KSwitchKeys ::save/load.  Written from the published 3.5.6 layout; not verifiable against a real
// SEAL build in this image.
#include "wire_codec.h"

#include <math.h>
#include <string.h>

namespace pirgpu {
namespace wire {

// ------------------------------------------------------------------ BLAKE2b (RFC 7693)

static const uint64_t kIV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
static const uint8_t kSigma[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

static inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t0, bool last) {
  uint64_t m[16], v[16];
  for (int i = 0; i < 16; ++i) memcpy(&m[i], block + 8 * i, 8);
  for (int i = 0; i < 8; ++i) {
    v[i] = h[i];
    v[i + 8] = kIV[i];
  }
  v[12] ^= t0;
  if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y)    \
  v[a] = v[a] + v[b] + (x);      \
  v[d] = rotr64(v[d] ^ v[a], 32); \
  v[c] = v[c] + v[d];            \
  v[b] = rotr64(v[b] ^ v[c], 24); \
  v[a] = v[a] + v[b] + (y);      \
  v[d] = rotr64(v[d] ^ v[a], 16); \
  v[c] = v[c] + v[d];            \
  v[b] = rotr64(v[b] ^ v[c], 63);
  for (int r = 0; r < 12; ++r) {
    const uint8_t* s = kSigma[r];
    B2G(0, 4, 8, 12, m[s[0]], m[s[1]])
    B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
    B2G(2, 6, 10, 14, m[s[4]], m[s[5]])
    B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
    B2G(0, 5, 10, 15, m[s[8]], m[s[9]])
    B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
    B2G(2, 7, 8, 13, m[s[12]], m[s[13]])
    B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
#undef B2G
  for (int i = 0; i < 8; ++i) h[i] ^= v[i] ^ v[i + 8];
}

// BLAKE2b with an explicit 64-byte parameter block (RFC 7693 section 2.5 / BLAKE2X) and optional key.
namespace {
struct B2State {
  uint64_t h[8];
  uint64_t t = 0;
  uint8_t buf[128];
  size_t buflen = 0;
  void init(const uint8_t param[64]) {
    for (int i = 0; i < 8; ++i) {
      uint64_t w;
      memcpy(&w, param + 8 * i, 8);
      h[i] = kIV[i] ^ w;
    }
    t = 0;
    buflen = 0;
  }
  void update(const uint8_t* in, size_t inlen) {
    while (inlen) {
      if (buflen == 128) {  // the buffered block is not the last one
        t += 128;
        b2_compress(h, buf, t, false);
        buflen = 0;
      }
      const size_t take = inlen < 128 - buflen ? inlen : 128 - buflen;
      memcpy(buf + buflen, in, take);
      buflen += take;
      in += take;
      inlen -= take;
    }
  }
  void final(uint8_t* out, size_t outlen) {
    t += buflen;
    memset(buf + buflen, 0, 128 - buflen);
    b2_compress(h, buf, t, true);
    uint8_t full[64];
    memcpy(full, h, 64);
    memcpy(out, full, outlen);
  }
};
}  // namespace

// unkeyed BLAKE2b with outlen <= 64
void blake2b(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen) {
  uint8_t param[64] = {0};
  param[0] = (uint8_t)outlen;  // digest_length
  param[2] = 1;                // fanout
  param[3] = 1;                // depth
  B2State st;
  st.init(param);
  st.update(in, inlen);
  st.final(out, outlen);
}

// BLAKE2Xb (the BLAKE2 team's blake2xb.c, which SEAL 3.5.6 vendors for its BlakePRNG): root hash
// H0 = BLAKE2b-512(key block || in) with xof_length = outlen in the parameter block, then output block i =
// BLAKE2b(H0) with node_offset = i, leaf_length = inner_length = 64, fanout = depth = 0, digest = block size.
void blake2xb(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen, const uint8_t* key, size_t keylen) {
  uint8_t param[64] = {0};
  param[0] = 64;
  param[1] = (uint8_t)keylen;
  param[2] = 1;
  param[3] = 1;
  const uint32_t xof = (uint32_t)outlen;
  memcpy(param + 12, &xof, 4);  // xof_length (leaf_length at 4, node_offset at 8)
  B2State root;
  root.init(param);
  if (keylen) {
    uint8_t block[128] = {0};
    memcpy(block, key, keylen);
    root.update(block, 128);
  }
  root.update(in, inlen);
  uint8_t h0[64];
  root.final(h0, 64);
  param[1] = 0;  // key_length
  param[2] = 0;  // fanout
  param[3] = 0;  // depth
  const uint32_t leaf = 64;
  memcpy(param + 4, &leaf, 4);
  param[17] = 64;  // inner_length (node_depth at 16 stays 0)
  for (uint32_t i = 0; outlen > 0; ++i) {
    const size_t block = outlen < 64 ? outlen : 64;
    param[0] = (uint8_t)block;
    memcpy(param + 8, &i, 4);  // node_offset
    B2State c;
    c.init(param);
    c.update(h0, 64);
    c.final(out + (size_t)i * 64, block);
    outlen -= block;
  }
}

// SEAL 3.5.6 BlakePRNG (randomgen.h/.cpp): a 4096-byte buffer refilled with
// blake2xb(buffer, 4096, &counter (u64, starting at 0), 8, seed, 64), counter++ per refill; consumers
// take bytes in order.  UNVERIFIED against a SEAL build (none in this image) -- see DESIGN.md section 8b.
SealPrng::SealPrng(const uint8_t seed_bytes[kSeedBytes]) { memcpy(seed, seed_bytes, kSeedBytes); }

void SealPrng::generate(uint8_t* dst, size_t n) {
  while (n) {
    if (head == sizeof(buf)) {
      blake2xb(buf, sizeof(buf), reinterpret_cast<const uint8_t*>(&counter), 8, seed, kSeedBytes);
      ++counter;
      head = 0;
    }
    const size_t take = n < sizeof(buf) - head ? n : sizeof(buf) - head;
    memcpy(dst, buf + head, take);
    head += take;
    dst += take;
    n -= take;
  }
}

// SEAL 3.5.6 sample_poly_uniform (util/rlwe.cpp): per modulus, per coefficient, rejection-sample
// rand = (u32 << 31) | (u32 >> 1) (a 63-bit value from two successive 32-bit outputs) below the largest
// multiple of the modulus, then reduce.  out: [n_moduli][N].
void sample_poly_uniform(SealPrng& rng, const uint64_t* moduli, uint32_t n_moduli, uint32_t N, uint64_t* out) {
  const uint64_t max_random = 0x7FFFFFFFFFFFFFFFULL;
  for (uint32_t j = 0; j < n_moduli; ++j) {
    const uint64_t q = moduli[j];
    const uint64_t max_multiple = max_random - max_random % q - 1;
    for (uint32_t i = 0; i < N; ++i) {
      uint64_t rand;
      do {
        const uint64_t a = rng.u32();
        const uint64_t b = rng.u32();
        rand = (a << 31) | (b >> 1);
      } while (rand >= max_multiple);
      out[(size_t)j * N + i] = rand % q;
    }
  }
}

// EncryptionParameters::compute_parms_id: BLAKE2b-256 over
// [scheme = BFV(1), N, q..., t] as little-endian u64.
void parms_id(uint32_t N, const uint64_t* moduli, size_t n_moduli, uint64_t t, uint64_t out[4]) {
  std::vector<uint64_t> data;
  data.push_back(1);
  data.push_back(N);
  for (size_t i = 0; i < n_moduli; ++i) data.push_back(moduli[i]);
  data.push_back(t);
  blake2b(reinterpret_cast<uint8_t*>(out), 32, reinterpret_cast<const uint8_t*>(data.data()), data.size() * 8);
}

// ------------------------------------------------------------------ proto3 primitives

void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) {
    s.push_back((char)(v | 0x80));
    v >>= 7;
  }
  s.push_back((char)v);
}

void put_bytes_field(std::string& s, uint32_t field, const std::string& payload) {
  put_varint(s, (field << 3) | 2);
  put_varint(s, payload.size());
  s.append(payload);
}

// ------------------------------------------------------------------ SEAL 3.5.6 objects

void put_u64(std::string& s, uint64_t v) { s.append(reinterpret_cast<const char*>(&v), 8); }

void put_header(std::string& s, uint64_t total_size) {
  uint8_t h[kHeader] = {0};
  h[0] = (uint8_t)(kSealMagic & 0xff);
  h[1] = (uint8_t)(kSealMagic >> 8);
  h[2] = 0x10;  // header size
  h[3] = 3;     // version major
  h[4] = 5;     // version minor
  h[5] = 0;     // compr_mode_type::none (seal.BUILD:15: zlib off)
  memcpy(h + 8, &total_size, 8);
  s.append(reinterpret_cast<const char*>(h), kHeader);
}

// Ciphertext::load (+ is_valid_for): returns residues [2][nres][N].
void load_ciphertext(Cursor& c, const Shape& sh, bool key_level, std::vector<uint64_t>& out) {
  out.resize(2ull * (key_level ? sh.k + 1 : sh.k) * sh.N);
  load_ciphertext_into(c, sh, key_level, out.data());
}

void load_ciphertext_into(Cursor& c, const Shape& sh, bool key_level, uint64_t* out) {
  const uint8_t* obj_end = c.header();
  Cursor o{c.p, obj_end};
  uint64_t id[4];
  for (int i = 0; i < 4; ++i) id[i] = o.u64();
  const uint64_t* want = key_level ? sh.key_id : sh.data_id;
  if (memcmp(id, want, 32) != 0) throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext data is invalid (parms_id mismatch)"};
  uint8_t is_ntt = o.u8();
  uint64_t size = o.u64(), N = o.u64(), cm = o.u64();
  (void)o.u64();  // scale (double); BFV requires 1.0, SEAL does not reject others on load
  const uint32_t nres = key_level ? sh.k + 1 : sh.k;
  if (size != 2) throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext size must be 2"};
  if (N != sh.N || cm != nres) throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext data is invalid (shape mismatch)"};
  if ((is_ntt != 0) != key_level) throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext NTT form does not match its level"};
  // IntArray<ct_coeff_type>
  const uint8_t* arr_end = Cursor{o.p, o.end}.header();
  Cursor a{o.p + kHeader, arr_end};
  uint64_t count = a.u64();
  const uint64_t full = 2ull * nres * sh.N;
  if (count != full && count != full / 2)
    throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext data is invalid (coefficient count)"};
  a.need(count * 8);
  memcpy(out, a.p, count * 8);
  if (count == full / 2) {
    // Seed-compressed object (Serializable<>, what PIRClient::initialize sends for its keys, client.cpp:47-54):
    // only c0 was saved; the 64-byte seed follows the IntArray and c1 is re-sampled from it
    // (Ciphertext::load_members -> expand_seed -> sample_poly_uniform over this object's moduli).
    Cursor sc{arr_end, o.end};
    sc.need(kSeedBytes);
    SealPrng rng(sc.p);
    uint64_t mods[PIRGPU_MAX_PRIMES + 1];
    for (uint32_t j = 0; j < nres; ++j) mods[j] = sh.q[j];
    sample_poly_uniform(rng, mods, nres, sh.N, out + full / 2);
  }
  // is_data_valid_for: every coefficient below its modulus
  for (uint32_t poly = 0; poly < 2; ++poly)
    for (uint32_t j = 0; j < nres; ++j) {
      const uint64_t q = (key_level && j == sh.k) ? sh.q[sh.k] : sh.q[j];
      const uint64_t* v = out + ((size_t)poly * nres + j) * sh.N;
      uint64_t bad = 0;   // branch-free scan (vectorises): any coefficient >= q
      for (uint32_t i = 0; i < sh.N; ++i) bad |= (uint64_t)(v[i] >= q);
      if (bad) throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext data is invalid (coefficient out of range)"};
    }
  c.p = obj_end;
}

// seed != nullptr: the Serializable<> form -- only c0 is written, followed by the 64-byte seed c1 was sampled from.
std::string save_ciphertext(const Shape& sh, const uint64_t* ct, bool key_level, const uint8_t* seed) {
  const uint32_t nres = key_level ? sh.k + 1 : sh.k;
  const uint64_t count = (seed ? 1ull : 2ull) * nres * sh.N;
  std::string arr;
  put_header(arr, kHeader + 8 + count * 8);
  put_u64(arr, count);
  arr.append(reinterpret_cast<const char*>(ct), count * 8);
  std::string body;
  const uint64_t* id = key_level ? sh.key_id : sh.data_id;
  for (int i = 0; i < 4; ++i) put_u64(body, id[i]);
  body.push_back(key_level ? 1 : 0);  // is_ntt_form
  put_u64(body, 2);
  put_u64(body, sh.N);
  put_u64(body, nres);
  double scale = 1.0;
  uint64_t sbits;
  memcpy(&sbits, &scale, 8);
  put_u64(body, sbits);
  body.append(arr);
  if (seed) body.append(reinterpret_cast<const char*>(seed), kSeedBytes);
  std::string out;
  put_header(out, kHeader + body.size());
  out.append(body);
  return out;
}

// Data-level ciphertexts are saved into one pre-sized buffer (responses are MBs: no per-ciphertext temporaries).
size_t saved_ciphertext_size(const Shape& sh) {
  return kHeader + 32 + 1 + 4 * 8 + kHeader + 8 + 2ull * sh.k * sh.N * 8;
}

void append_ciphertext(std::string& out, const Shape& sh, const uint64_t* ct) {
  append_ciphertext_prefix(out, sh);
  out.append(reinterpret_cast<const char*>(ct), 2ull * sh.k * sh.N * 8);
}

void append_ciphertext_prefix(std::string& out, const Shape& sh) {
  const uint64_t count = 2ull * sh.k * sh.N;
  put_header(out, saved_ciphertext_size(sh));
  for (int i = 0; i < 4; ++i) put_u64(out, sh.data_id[i]);
  out.push_back(0);  // is_ntt_form
  put_u64(out, 2);
  put_u64(out, sh.N);
  put_u64(out, sh.k);
  double scale = 1.0;
  uint64_t sbits;
  memcpy(&sbits, &scale, 8);
  put_u64(out, sbits);
  put_header(out, kHeader + 8 + count * 8);
  put_u64(out, count);
}

std::string save_public_key(const Shape& sh, const uint64_t* pk, const uint8_t* seed) {
  const std::string ct = save_ciphertext(sh, pk, true, seed);
  std::string out;
  put_header(out, kHeader + ct.size());
  out.append(ct);
  return out;
}

// KSwitchKeys::load (GaloisKeys: entry index i holds the key of Galois element 2i+1; RelinKeys: index 0).
void load_kswitch_keys(const Shape& sh, const uint8_t* data, size_t len,
                       const std::function<void(uint64_t, const uint64_t*)>& sink) {
  Cursor c{data, data + len};
  const uint8_t* obj_end = c.header();
  Cursor o{c.p, obj_end};
  uint64_t id[4];
  for (int i = 0; i < 4; ++i) id[i] = o.u64();
  if (memcmp(id, sh.key_id, 32) != 0) throw Err{PIRGPU_INVALID_ARGUMENT, "GaloisKeys data is invalid (parms_id mismatch)"};
  uint64_t dim1 = o.u64();
  if (dim1 > sh.N) throw Err{PIRGPU_INVALID_ARGUMENT, "GaloisKeys data is invalid"};
  const uint32_t km = sh.k + 1;
  std::vector<uint64_t> key((size_t)sh.k * 2 * km * sh.N), tmp;
  for (uint64_t index = 0; index < dim1; ++index) {
    uint64_t dim2 = o.u64();
    if (dim2 == 0) continue;
    if (dim2 != sh.k) throw Err{PIRGPU_INVALID_ARGUMENT, "GaloisKeys data is invalid (decomposition count)"};
    for (uint64_t j = 0; j < dim2; ++j) {
      const uint8_t* pk_end = o.header();  // PublicKey wrapper
      Cursor pk{o.p, pk_end};
      load_ciphertext(pk, sh, true, tmp);  // [2][k+1][N]
      memcpy(key.data() + (size_t)j * 2 * km * sh.N, tmp.data(), tmp.size() * 8);
      o.p = pk_end;
    }
    if (sink) sink(index, key.data());
  }
}

// The same with the per-PublicKey work (copy, seed expansion of the c1 half -- BLAKE2Xb + rejection sampling, 0.1-0.2 ms
// per object at N = 4096 -- and the range check) spread over `pf`: the structure is walked once on the calling thread,
// the objects are independent.  keys[e] = (index, key [k][2][k+1][N]) in object order.
void load_kswitch_keys_parallel(const Shape& sh, const uint8_t* data, size_t len, const ParallelFor& pf,
                                std::vector<std::pair<uint64_t, std::vector<uint64_t>>>& keys) {
  Cursor c{data, data + len};
  const uint8_t* obj_end = c.header();
  Cursor o{c.p, obj_end};
  uint64_t id[4];
  for (int i = 0; i < 4; ++i) id[i] = o.u64();
  if (memcmp(id, sh.key_id, 32) != 0) throw Err{PIRGPU_INVALID_ARGUMENT, "GaloisKeys data is invalid (parms_id mismatch)"};
  uint64_t dim1 = o.u64();
  if (dim1 > sh.N) throw Err{PIRGPU_INVALID_ARGUMENT, "GaloisKeys data is invalid"};
  const uint32_t km = sh.k + 1;
  const size_t pk_words = (size_t)2 * km * sh.N;
  struct Piece {
    size_t entry;
    uint32_t j;
    const uint8_t *p, *end;
  };
  std::vector<Piece> pieces;
  keys.clear();
  for (uint64_t index = 0; index < dim1; ++index) {
    uint64_t dim2 = o.u64();
    if (dim2 == 0) continue;
    if (dim2 != sh.k) throw Err{PIRGPU_INVALID_ARGUMENT, "GaloisKeys data is invalid (decomposition count)"};
    keys.emplace_back(index, std::vector<uint64_t>());
    for (uint64_t j = 0; j < dim2; ++j) {
      const uint8_t* pk_end = o.header();  // PublicKey wrapper
      pieces.push_back(Piece{keys.size() - 1, (uint32_t)j, o.p, pk_end});
      o.p = pk_end;
    }
  }
  for (auto& kv : keys) kv.second.resize((size_t)sh.k * pk_words);
  std::vector<Err> errs(pieces.size(), Err{0, ""});
  auto one = [&](size_t i) {
    try {
      Cursor pk{pieces[i].p, pieces[i].end};
      load_ciphertext_into(pk, sh, true, keys[pieces[i].entry].second.data() + (size_t)pieces[i].j * pk_words);
    } catch (const Err& e) {
      errs[i] = e;
    } catch (const std::exception& e) {
      errs[i] = Err{PIRGPU_INTERNAL, e.what()};
    }
  };
  if (pf) pf(pieces.size(), one);
  else for (size_t i = 0; i < pieces.size(); ++i) one(i);
  for (const Err& e : errs)
    if (e.code) throw e;   // the first failing object in object order, like the sequential loader
}

std::string save_kswitch_keys(const Shape& sh, const std::vector<const uint64_t*>& entries,
                              const std::vector<const uint8_t*>* seeds) {
  const size_t pk_words = (size_t)2 * (sh.k + 1) * sh.N;
  std::string body;
  for (int i = 0; i < 4; ++i) put_u64(body, sh.key_id[i]);
  put_u64(body, entries.size());
  for (size_t e = 0; e < entries.size(); ++e) {
    const uint64_t* key = entries[e];
    put_u64(body, key ? sh.k : 0);
    if (!key) continue;
    const uint8_t* sd = seeds ? (*seeds)[e] : nullptr;  // k seeds of kSeedBytes each, one per decomposition digit
    for (uint32_t j = 0; j < sh.k; ++j)
      body.append(save_public_key(sh, key + j * pk_words, sd ? sd + (size_t)j * kSeedBytes : nullptr));
  }
  std::string out;
  put_header(out, kHeader + body.size());
  out.append(body);
  return out;
}

Shape make_shape(const pirgpu_params& prm) {
  Shape sh;
  sh.N = prm.poly_modulus_degree;
  sh.k = prm.num_data_primes;
  for (uint32_t i = 0; i < sh.k; ++i) sh.q[i] = prm.coeff_modulus[i];
  sh.q[sh.k] = prm.special_prime;
  sh.t = prm.plain_modulus;
  parms_id(sh.N, sh.q, sh.k, sh.t, sh.data_id);
  parms_id(sh.N, sh.q, sh.k + 1, sh.t, sh.key_id);
  return sh;
}


// LoadCiphertexts (serialization.cpp:32-42) of one Ciphertexts message -> residues, count
uint32_t load_query_into(const Shape& sh, const uint8_t* data, size_t len, uint64_t* dst, uint32_t max_cts) {
  Reader qr{data, data + len};
  const size_t ctw = 2ull * sh.k * sh.N;
  std::vector<uint64_t> spill;   // ciphertexts beyond max_cts are still parsed and validated, not kept
  uint32_t nq = 0;
  while (qr.p < qr.end) {
    uint64_t tag;
    if (!qr.varint(tag)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Ciphertexts"};
    const uint8_t* d;
    size_t l;
    if ((tag >> 3) == 1 && (tag & 7) == 2) {
      if (!qr.bytes(d, l)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Ciphertexts.ct"};
      Cursor c{d, d + l};
      if (nq < max_cts) {
        load_ciphertext_into(c, sh, false, dst + (size_t)nq * ctw);
      } else {
        load_ciphertext(c, sh, false, spill);
      }
      ++nq;
    } else if (!qr.skip((uint32_t)(tag & 7))) {
      throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Ciphertexts"};
    }
  }
  return nq;
}

uint32_t load_query(const Shape& sh, const uint8_t* data, size_t len, std::vector<uint64_t>& qbuf) {
  Reader qr{data, data + len};
  std::vector<uint64_t> one;
  qbuf.clear();
  uint32_t nq = 0;
  while (qr.p < qr.end) {
    uint64_t tag;
    if (!qr.varint(tag)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Ciphertexts"};
    const uint8_t* d;
    size_t l;
    if ((tag >> 3) == 1 && (tag & 7) == 2) {
      if (!qr.bytes(d, l)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Ciphertexts.ct"};
      Cursor c{d, d + l};
      load_ciphertext(c, sh, false, one);
      qbuf.insert(qbuf.end(), one.begin(), one.end());
      ++nq;
    } else if (!qr.skip((uint32_t)(tag & 7))) {
      throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Ciphertexts"};
    }
  }
  return nq;
}

}  // namespace wire
}  // namespace pirgpu
