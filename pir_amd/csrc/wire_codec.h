// wire_codec.h -- internal header: pir/proto framing primitives + SEAL 3.5.6 object codec.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pirgpu.h"

namespace pirgpu {
namespace wire {

void blake2b(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen);
// BLAKE2Xb extendable-output function (outlen < 2^32), optionally keyed (keylen <= 64).
void blake2xb(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen, const uint8_t* key, size_t keylen);

// SEAL 3.5.6's seeded PRNG (BlakePRNG) and uniform polynomial sampler: what re-expands the c1 half of a
// seed-compressed Serializable<> object on load (reference serialization.h:104-118 -> Ciphertext::load).
constexpr size_t kSeedBytes = 64;  // random_seed_type = std::array<uint64_t, 8>
struct SealPrng {
  uint8_t seed[kSeedBytes];
  uint64_t counter = 0;
  uint8_t buf[4096];
  size_t head = sizeof(buf);
  explicit SealPrng(const uint8_t seed_bytes[kSeedBytes]);
  void generate(uint8_t* dst, size_t n);
  uint32_t u32() {
    uint32_t v;
    generate(reinterpret_cast<uint8_t*>(&v), 4);
    return v;
  }
};
void sample_poly_uniform(SealPrng& rng, const uint64_t* moduli, uint32_t n_moduli, uint32_t N, uint64_t* out);
void parms_id(uint32_t N, const uint64_t* moduli, size_t n_moduli, uint64_t t, uint64_t out[4]);

// ------------------------------------------------------------------ proto3 primitives

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool varint(uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
      uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return true;
    }
    return false;
  }
  bool bytes(const uint8_t*& data, size_t& len) {
    uint64_t n;
    if (!varint(n) || n > (uint64_t)(end - p)) return false;
    data = p;
    len = (size_t)n;
    p += n;
    return true;
  }
  bool skip(uint32_t wt) {
    uint64_t v;
    const uint8_t* d;
    size_t l;
    switch (wt) {
      case 0: return varint(v);
      case 1: if (end - p < 8) return false; p += 8; return true;
      case 2: return bytes(d, l);
      case 5: if (end - p < 4) return false; p += 4; return true;
      default: return false;
    }
  }
};

void put_varint(std::string& s, uint64_t v);
void put_bytes_field(std::string& s, uint32_t field, const std::string& payload);

// ------------------------------------------------------------------ SEAL 3.5.6 objects

constexpr uint16_t kSealMagic = 0xA15E;
constexpr size_t kHeader = 16;

struct Err {
  int code;
  std::string msg;
};

void put_u64(std::string& s, uint64_t v);
void put_header(std::string& s, uint64_t total_size);

struct Cursor {
  const uint8_t* p;
  const uint8_t* end;
  void need(size_t n) const {
    if ((size_t)(end - p) < n) throw Err{PIRGPU_INVALID_ARGUMENT, "SEAL object truncated"};
  }
  uint64_t u64() {
    need(8);
    uint64_t v;
    memcpy(&v, p, 8);
    p += 8;
    return v;
  }
  uint8_t u8() {
    need(1);
    return *p++;
  }
  // returns the end of the object the header describes
  const uint8_t* header() {
    need(kHeader);
    uint16_t magic = (uint16_t)(p[0] | (p[1] << 8));
    if (magic != kSealMagic || p[2] != 0x10) throw Err{PIRGPU_INVALID_ARGUMENT, "loaded SEALHeader is invalid"};
    if (p[3] != 3) throw Err{PIRGPU_INVALID_ARGUMENT, "incompatible SEAL version"};
    if (p[5] != 0)
      throw Err{PIRGPU_UNIMPLEMENTED, "compressed SEAL objects are not supported (reference builds SEAL without zlib)"};
    uint64_t size;
    memcpy(&size, p + 8, 8);
    if (size < kHeader || size > (uint64_t)(end - p)) throw Err{PIRGPU_INVALID_ARGUMENT, "SEAL object size mismatch"};
    const uint8_t* obj_end = p + size;
    p += kHeader;
    return obj_end;
  }
};

struct Shape {
  uint32_t N, k;
  uint64_t q[PIRGPU_MAX_PRIMES + 1];  // data primes then special
  uint64_t t;
  uint64_t data_id[4], key_id[4];
};

Shape make_shape(const pirgpu_params& prm);

// Ciphertext::load (+ is_valid_for): residues [2][nres][N]; key_level = (k+1)-prime NTT-form object.
void load_ciphertext(Cursor& c, const Shape& sh, bool key_level, std::vector<uint64_t>& out);
// the same into caller-owned memory of 2 * nres * N words (e.g. pinned staging)
void load_ciphertext_into(Cursor& c, const Shape& sh, bool key_level, uint64_t* out);
// Ciphertext::save of a data-level ciphertext appended to `out` (exactly saved_ciphertext_size bytes)
size_t saved_ciphertext_size(const Shape& sh);
void append_ciphertext(std::string& out, const Shape& sh, const uint64_t* ct);
// Everything of that object before the 2 k N raw words (the same bytes for every data-level ciphertext of a context).
void append_ciphertext_prefix(std::string& out, const Shape& sh);
// Ciphertext::save of a size-2 ciphertext: data level (k primes, coefficient form) or, with
// key_level, the (k+1)-prime NTT-form body of a PublicKey.
std::string save_ciphertext(const Shape& sh, const uint64_t* ct, bool key_level = false,
                            const uint8_t* seed = nullptr);
// PublicKey::save = header + Ciphertext::save(key level).
std::string save_public_key(const Shape& sh, const uint64_t* pk, const uint8_t* seed = nullptr);
// KSwitchKeys::load (GaloisKeys / RelinKeys): calls sink(index, key [k][2][k+1][N]) per present entry.
void load_kswitch_keys(const Shape& sh, const uint8_t* data, size_t len,
                       const std::function<void(uint64_t, const uint64_t*)>& sink);
// The same with the objects parsed (and their seeded halves expanded) through `pf` (n, fn): fn(i) for every i < n, on
// whatever threads pf has; nullptr = sequentially.  keys[e] = (index, key) in object order.
using ParallelFor = std::function<void(size_t, const std::function<void(size_t)>&)>;
void load_kswitch_keys_parallel(const Shape& sh, const uint8_t* data, size_t len, const ParallelFor& pf,
                                std::vector<std::pair<uint64_t, std::vector<uint64_t>>>& keys);
// KSwitchKeys::save: entries[i] = key [k][2][k+1][N] or nullptr (absent), i < dim1.
// seeds (optional): per entry k * kSeedBytes bytes -> the seed-compressed Serializable<> form (c1 halves omitted).
std::string save_kswitch_keys(const Shape& sh, const std::vector<const uint64_t*>& entries,
                              const std::vector<const uint8_t*>* seeds = nullptr);

// LoadCiphertexts (serialization.cpp:32-42) of one pir.Ciphertexts message -> residues [n][2][k][N]; returns n.
uint32_t load_query(const Shape& sh, const uint8_t* data, size_t len, std::vector<uint64_t>& qbuf);
// the same into caller-owned memory with room for max_cts ciphertexts; returns the number of ciphertexts in the message
// (all of them parsed and validated; those beyond max_cts are not stored)
uint32_t load_query_into(const Shape& sh, const uint8_t* data, size_t len, uint64_t* dst, uint32_t max_cts);

}  // namespace wire
}  // namespace pirgpu
