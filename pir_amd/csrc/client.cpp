// client.cpp -- libpirclient.so: CPU PIR client (include/pirclient.h), counterpart of the reference's
// PIRClient (pir/cpp/client.cpp) for the GPU server path.  Host-only C++ (g++), no device code.
//
//   createQueryFor / CreateRequest            client.cpp:80-144
//   ProcessResponse / ProcessResponseInteger   client.cpp:146-185
//   ProcessReplyCiphertextDecomp               client.cpp:219-255
//   CiphertextReencoder::Encode / Decode       ct_reencoder.cpp:40-111
//   StringEncoder::decode                      string_encoder.cpp:124-163
//   PIRDatabase::calculate_indices / calculate_item_offset   database.cpp:112-138
//
// The BFV primitives SEAL 3.5.6 provides to the reference (KeyGenerator, Encryptor, Decryptor) are
// implemented here directly: RLWE public-key encryption at key level followed by divide-and-round by
// the special prime, key-switching keys as one RLWE sample per RNS digit carrying p * s', decryption by
// exact scale-and-round of c0 + c1 s.  NTT layout and root choice follow SEAL (minimal primitive 2N-th
// root, bit-reversed output) because Galois keys cross the wire in NTT form.
#include "../../include/pirclient.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/random.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "host_math.h"
#include "wire_codec.h"

namespace pirgpu {
namespace client {

typedef unsigned __int128 u128;
using wire::Err;

// ------------------------------------------------------------------ randomness

// BLAKE2b in counter mode: block_i = BLAKE2b-512(key || i).
struct Rng {
  uint8_t key[64];
  uint64_t ctr = 0;
  uint8_t buf[64];
  size_t pos = 64;
  void seed(const uint8_t* s, size_t n) {
    if (s) {
      wire::blake2b(key, 64, s, n);
    } else {
      size_t got = 0;
      while (got < 64) {
        ssize_t r = getrandom(key + got, 64 - got, 0);
        if (r <= 0) throw Err{PIRGPU_INTERNAL, "getrandom failed"};
        got += (size_t)r;
      }
    }
    ctr = 0;
    pos = 64;
  }
  uint64_t u64() {
    if (pos + 8 > 64) {
      uint8_t in[72];
      memcpy(in, key, 64);
      memcpy(in + 64, &ctr, 8);
      ++ctr;
      wire::blake2b(buf, 64, in, 72);
      pos = 0;
    }
    uint64_t v;
    memcpy(&v, buf + pos, 8);
    pos += 8;
    return v;
  }
  uint64_t below(uint64_t q) {  // uniform in [0, q) by rejection
    const uint64_t lim = UINT64_MAX - (UINT64_MAX % q + 1) % q;
    uint64_t v;
    do v = u64(); while (v > lim);
    return v % q;
  }
  int ternary() { return (int)below(3) - 1; }
  double unit() { return ((double)(u64() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0,1)
  // SEAL's ClippedNormalDistribution(0, 3.2, 19.2), rounded to the nearest integer
  int noise() {
    const double sigma = 3.2, max_dev = 19.2;
    for (;;) {
      double r = sqrt(-2.0 * log(unit())) * cos(2.0 * M_PI * unit()) * sigma;
      if (fabs(r) <= max_dev) return (int)llrint(r);
    }
  }
};

// ------------------------------------------------------------------ one RNS modulus: arithmetic + NTT

struct Modulus {
  uint64_t q = 0;
  uint32_t N = 0, logn = 0;
  std::vector<uint64_t> w, ws, iw, iws;  // psi^bitrev(i) (+Shoup quotients), inverses
  uint64_t ninv = 0, ninv_s = 0;

  uint64_t mul(uint64_t a, uint64_t b) const { return (uint64_t)(((u128)a * b) % q); }
  uint64_t add(uint64_t a, uint64_t b) const { uint64_t s = a + b; return s >= q ? s - q : s; }
  uint64_t sub(uint64_t a, uint64_t b) const { return a >= b ? a - b : a + q - b; }
  uint64_t neg(uint64_t a) const { return a ? q - a : 0; }
  uint64_t from_signed(int64_t v) const { return v < 0 ? q - (uint64_t)(-v) % q : (uint64_t)v % q; }
  uint64_t mul_shoup(uint64_t x, uint64_t wv, uint64_t wq) const {
    uint64_t hi = (uint64_t)(((u128)x * wq) >> 64);
    uint64_t r = x * wv - hi * q;
    return r >= q ? r - q : r;
  }

  void init(uint64_t modulus, uint32_t degree) {
    q = modulus;
    N = degree;
    logn = 0;
    while ((1u << logn) < N) ++logn;
    const uint64_t psi = hm::minimal_primitive_root(2ull * N, q);
    const uint64_t ipsi = hm::invmod_prime(psi, q);
    w.assign(N, 0), ws.assign(N, 0), iw.assign(N, 0), iws.assign(N, 0);
    uint64_t p = 1, ip = 1;
    for (uint32_t i = 0; i < N; ++i) {
      const uint32_t r = hm::bitrev(i, logn);
      w[r] = p, ws[r] = hm::shoup(p, q);
      iw[r] = ip, iws[r] = hm::shoup(ip, q);
      p = mul(p, psi);
      ip = mul(ip, ipsi);
    }
    ninv = hm::invmod_prime(N % q, q);
    ninv_s = hm::shoup(ninv, q);
  }
  // natural order in, bit-reversed order out (SEAL ntt_negacyclic_harvey)
  void ntt(uint64_t* x) const {
    uint32_t t = N;
    for (uint32_t m = 1; m < N; m <<= 1) {
      t >>= 1;
      for (uint32_t i = 0; i < m; ++i) {
        const uint64_t wv = w[m + i], wq = ws[m + i];
        uint64_t* a = x + 2 * i * t;
        for (uint32_t j = 0; j < t; ++j) {
          const uint64_t u = a[j], v = mul_shoup(a[j + t], wv, wq);
          a[j] = add(u, v);
          a[j + t] = sub(u, v);
        }
      }
    }
  }
  void intt(uint64_t* x) const {
    uint32_t t = 1;
    for (uint32_t m = N; m > 1; m >>= 1) {
      const uint32_t h = m >> 1;
      for (uint32_t i = 0; i < h; ++i) {
        const uint64_t wv = iw[h + i], wq = iws[h + i];
        uint64_t* a = x + 2 * i * t;
        for (uint32_t j = 0; j < t; ++j) {
          const uint64_t u = a[j], v = a[j + t];
          a[j] = add(u, v);
          a[j + t] = mul_shoup(sub(u, v), wv, wq);
        }
      }
      t <<= 1;
    }
    for (uint32_t i = 0; i < N; ++i) x[i] = mul_shoup(x[i], ninv, ninv_s);
  }
};

// ------------------------------------------------------------------ little-endian multiword integers

struct Big {
  static constexpr int W = PIRGPU_MAX_PRIMES + 3;
  uint64_t v[W] = {0};
  void muladd(const Big& a, uint64_t b) {  // this += a * b
    u128 carry = 0;
    for (int i = 0; i < W; ++i) {
      u128 cur = (u128)a.v[i] * b + v[i] + carry;
      v[i] = (uint64_t)cur;
      carry = cur >> 64;
    }
  }
  void mul_word(uint64_t b) {
    u128 carry = 0;
    for (int i = 0; i < W; ++i) {
      u128 cur = (u128)v[i] * b + carry;
      v[i] = (uint64_t)cur;
      carry = cur >> 64;
    }
  }
  bool sub(const Big& a) {  // this -= a, returns borrow
    uint64_t borrow = 0;
    for (int i = 0; i < W; ++i) {
      u128 cur = (u128)v[i] - a.v[i] - borrow;
      v[i] = (uint64_t)cur;
      borrow = (uint64_t)(cur >> 64) & 1;
    }
    return borrow != 0;
  }
  void add(const Big& a) {
    uint64_t carry = 0;
    for (int i = 0; i < W; ++i) {
      u128 cur = (u128)v[i] + a.v[i] + carry;
      v[i] = (uint64_t)cur;
      carry = (uint64_t)(cur >> 64);
    }
  }
  int cmp(const Big& a) const {
    for (int i = W - 1; i >= 0; --i)
      if (v[i] != a.v[i]) return v[i] < a.v[i] ? -1 : 1;
    return 0;
  }
  int bits() const {
    for (int i = W - 1; i >= 0; --i)
      if (v[i]) return 64 * i + (64 - __builtin_clzll(v[i]));
    return 0;
  }
  uint64_t mod_word(uint64_t m) const {
    u128 r = 0;
    for (int i = W - 1; i >= 0; --i) r = ((r << 64) | v[i]) % m;
    return (uint64_t)r;
  }
  void shr1() {
    for (int i = 0; i < W; ++i) v[i] = (v[i] >> 1) | (i + 1 < W ? v[i + 1] << 63 : 0);
  }
};

}  // namespace client
}  // namespace pirgpu

using namespace pirgpu;
using namespace pirgpu::client;

struct pirclient {
  pirgpu_params prm;
  wire::Shape sh;
  uint32_t N, k, km;
  uint64_t t;
  uint32_t bits_per_coeff;      // StringEncoder width in use
  uint32_t pt_bits;             // floor(log2 t): reencoder digit width
  uint32_t exp_ratio;           // CiphertextReencoder::ExpansionRatio
  std::vector<uint32_t> local_ratio;
  std::vector<Modulus> mod;     // k data primes, then the special prime
  Rng rng;
  std::string err;

  std::vector<uint64_t> s_ntt;  // [km][N]
  std::vector<int8_t> s_signed; // [N]
  std::vector<uint64_t> pk;     // [2][km][N] NTT form
  std::map<uint32_t, std::vector<uint64_t>> galois;  // elt -> [k][2][km][N]
  std::vector<uint64_t> relin;  // [k][2][km][N]
  std::string galois_blob, relin_blob;  // serialized once (client.cpp:49-54), fully expanded
  std::string galois_blob_seeded, relin_blob_seeded;  // the same keys, seed-compressed (the default on the wire)
  std::map<uint32_t, std::vector<uint8_t>> galois_seeds;
  bool seeded_keys = true;

  // decryption / encryption constants
  std::vector<uint64_t> inv_punct;   // (Q/q_j)^-1 mod q_j
  std::vector<uint64_t> delta_mod;   // floor(Q/t) mod q_j
  std::vector<uint64_t> p_mod, p_inv, half_mod;  // special prime mod q_j, its inverse, (p>>1) mod q_j
  uint64_t q_mod_t;
  Big Q, half_Q;
  std::vector<Big> punct;            // Q/q_j
  int q_bits;

  size_t ct_words() const { return (size_t)2 * k * N; }
  size_t key_words() const { return (size_t)k * 2 * km * N; }

  // ---------------------------------------------------------------- setup
  void init(const pirgpu_params& p, const uint8_t* seed, size_t seed_len) {
    prm = p;
    N = p.poly_modulus_degree;
    k = p.num_data_primes;
    km = k + 1;
    t = p.plain_modulus;
    if (N < 2 || (N & (N - 1)) || k < 1 || k > PIRGPU_MAX_PRIMES || t < 2 || !p.special_prime)
      throw Err{PIRGPU_INVALID_ARGUMENT, "invalid encryption parameters"};
    if (p.num_dimensions < 1 || p.num_dimensions > PIRGPU_MAX_DIMS)
      throw Err{PIRGPU_INVALID_ARGUMENT, "invalid number of dimensions"};
    sh = wire::make_shape(p);
    mod.resize(km);
    for (uint32_t j = 0; j < km; ++j) {
      const uint64_t q = j < k ? p.coeff_modulus[j] : p.special_prime;
      if (!hm::is_prime(q) || (q - 1) % (2ull * N)) throw Err{PIRGPU_INVALID_ARGUMENT, "coeff modulus is not NTT friendly"};
      mod[j].init(q, N);
    }
    pt_bits = hm::bits_per_coeff(t);
    bits_per_coeff = p.bits_per_coeff ? p.bits_per_coeff : pt_bits;  // client.cpp:169-172
    exp_ratio = 0;
    local_ratio.resize(k);
    for (uint32_t j = 0; j < k; ++j) {
      local_ratio[j] = hm::local_expansion_ratio(mod[j].q, pt_bits);
      exp_ratio += local_ratio[j];
    }
    // Q, Q/q_j and friends
    Q = Big();
    Q.v[0] = 1;
    for (uint32_t j = 0; j < k; ++j) Q.mul_word(mod[j].q);
    q_bits = Q.bits();
    half_Q = Q;
    half_Q.shr1();
    punct.assign(k, Big());
    inv_punct.resize(k), delta_mod.resize(k), p_mod.resize(k), p_inv.resize(k), half_mod.resize(k);
    // floor(Q/t) = (Q - Q mod t) / t  -> residues via (Q mod q_j - Q mod t) * t^-1 is not available
    // (Q = 0 mod q_j), so divide the multiword value directly.
    q_mod_t = Q.mod_word(t);
    Big delta = Q;
    {
      u128 r = 0;
      for (int i = Big::W - 1; i >= 0; --i) {
        u128 cur = (r << 64) | delta.v[i];
        delta.v[i] = (uint64_t)(cur / t);
        r = cur % t;
      }
    }
    const uint64_t sp = mod[k].q;
    for (uint32_t j = 0; j < k; ++j) {
      punct[j].v[0] = 1;
      for (uint32_t i = 0; i < k; ++i)
        if (i != j) punct[j].mul_word(mod[i].q);
      inv_punct[j] = hm::invmod_prime(punct[j].mod_word(mod[j].q), mod[j].q);
      delta_mod[j] = delta.mod_word(mod[j].q);
      p_mod[j] = sp % mod[j].q;
      p_inv[j] = hm::invmod_prime(p_mod[j], mod[j].q);
      half_mod[j] = (sp >> 1) % mod[j].q;
    }
    rng.seed(seed, seed_len);
    keygen();
  }

  // ---------------------------------------------------------------- sampling helpers
  // signed small polynomial -> [km][N] residues in NTT form
  void small_to_ntt(const std::vector<int>& e, uint64_t* out) const {
    for (uint32_t j = 0; j < km; ++j) {
      uint64_t* o = out + (size_t)j * N;
      for (uint32_t i = 0; i < N; ++i) o[i] = mod[j].from_signed(e[i]);
      mod[j].ntt(o);
    }
  }
  std::vector<int> sample_noise() {
    std::vector<int> e(N);
    for (auto& v : e) v = rng.noise();
    return e;
  }
  // (-(a s + e), a) at key level, NTT form: out [2][km][N].  `a` is sampled the way SEAL 3.5.6 samples the
  // public half of a symmetric-key sample (encrypt_zero_symmetric): a fresh 64-byte public seed, BlakePRNG,
  // sample_poly_uniform over the key-level moduli -- so that the object can be sent seed-compressed
  // (Serializable<>, reference client.cpp:47-54).  seed_out (optional) receives that seed.
  void rlwe_zero_sym(uint64_t* out, uint8_t* seed_out = nullptr) {
    uint64_t* c0 = out;
    uint64_t* c1 = out + (size_t)km * N;
    uint8_t seed[wire::kSeedBytes];
    for (size_t i = 0; i < wire::kSeedBytes; i += 8) {
      const uint64_t v = rng.u64();
      memcpy(seed + i, &v, 8);
    }
    if (seed_out) memcpy(seed_out, seed, wire::kSeedBytes);
    {
      wire::SealPrng pub(seed);
      uint64_t mods[PIRGPU_MAX_PRIMES + 1];
      for (uint32_t j = 0; j < km; ++j) mods[j] = mod[j].q;
      wire::sample_poly_uniform(pub, mods, km, N, c1);
    }
    small_to_ntt(sample_noise(), c0);
    for (uint32_t j = 0; j < km; ++j) {
      const Modulus& m = mod[j];
      for (uint32_t i = 0; i < N; ++i) {
        const size_t o = (size_t)j * N + i;
        c0[o] = m.neg(m.add(m.mul(c1[o], s_ntt[o]), c0[o]));
      }
    }
  }
  // KSwitchKey for new_key (NTT form [km][N]): out [k][2][km][N]; seeds_out: k seeds of kSeedBytes
  void make_kswitch_key(const uint64_t* new_key, uint64_t* out, uint8_t* seeds_out) {
    for (uint32_t j = 0; j < k; ++j) {
      uint64_t* smp = out + (size_t)j * 2 * km * N;
      rlwe_zero_sym(smp, seeds_out + (size_t)j * wire::kSeedBytes);
      const Modulus& m = mod[j];
      uint64_t* c0j = smp + (size_t)j * N;
      const uint64_t* nk = new_key + (size_t)j * N;
      for (uint32_t i = 0; i < N; ++i) c0j[i] = m.add(c0j[i], m.mul(nk[i], p_mod[j]));
    }
  }
  void keygen() {
    s_signed.resize(N);
    std::vector<int> s(N);
    for (uint32_t i = 0; i < N; ++i) s_signed[i] = (int8_t)(s[i] = rng.ternary());
    s_ntt.resize((size_t)km * N);
    small_to_ntt(s, s_ntt.data());
    pk.resize((size_t)2 * km * N);
    rlwe_zero_sym(pk.data());
    // Galois keys for generate_galois_elts(N) (utils.cpp:7-14): N/2^i + 1, i < log2 N
    std::vector<uint64_t> rotated((size_t)km * N);
    uint32_t logn = mod[0].logn;
    uint64_t max_index = 0;
    for (uint32_t i = 0; i < logn; ++i) {
      const uint32_t g = (N >> i) + 1;
      for (uint32_t j = 0; j < km; ++j) {
        uint64_t* r = rotated.data() + (size_t)j * N;
        for (uint32_t c = 0; c < N; ++c) {
          const uint64_t e = (uint64_t)c * g % (2ull * N);
          const uint64_t v = mod[j].from_signed(s[c]);
          if (e < N) r[e] = v; else r[e - N] = mod[j].neg(v);
        }
        mod[j].ntt(r);
      }
      auto& key = galois[g];
      key.resize(key_words());
      auto& sd = galois_seeds[g];
      sd.resize((size_t)k * wire::kSeedBytes);
      make_kswitch_key(rotated.data(), key.data(), sd.data());
      max_index = std::max<uint64_t>(max_index, (g - 1) >> 1);
    }
    // relinearisation key: s^2
    for (uint32_t j = 0; j < km; ++j)
      for (uint32_t i = 0; i < N; ++i) {
        const size_t o = (size_t)j * N + i;
        rotated[o] = mod[j].mul(s_ntt[o], s_ntt[o]);
      }
    relin.resize(key_words());
    std::vector<uint8_t> relin_seed((size_t)k * wire::kSeedBytes);
    make_kswitch_key(rotated.data(), relin.data(), relin_seed.data());
    explicit_bzero(rotated.data(), rotated.size() * 8);  // held sigma_g(s) / s^2
    explicit_bzero(s.data(), s.size() * sizeof(int));
    // serialize once (client.cpp:49-54): seed-compressed like SEAL's Serializable<GaloisKeys>/<RelinKeys>,
    // and fully expanded (what galois_keys_local + SaveRequest produce, server_test.cpp) on request
    std::vector<const uint64_t*> entries(max_index + 1, nullptr);
    std::vector<const uint8_t*> seeds(max_index + 1, nullptr);
    for (auto& kv : galois) {
      entries[(kv.first - 1) >> 1] = kv.second.data();
      seeds[(kv.first - 1) >> 1] = galois_seeds[kv.first].data();
    }
    std::vector<const uint8_t*> rseeds{relin_seed.data()};
    galois_blob_seeded = wire::save_kswitch_keys(sh, entries, &seeds);
    relin_blob_seeded = wire::save_kswitch_keys(sh, {relin.data()}, &rseeds);
    galois_blob = wire::save_kswitch_keys(sh, entries);
    relin_blob = wire::save_kswitch_keys(sh, {relin.data()});
  }

  // ---------------------------------------------------------------- encrypt / decrypt
  void encrypt(const uint64_t* pt, size_t n_coeffs, uint64_t* ct) {
    if (n_coeffs > N) throw Err{PIRGPU_INVALID_ARGUMENT, "plaintext has more coefficients than the ring degree"};
    for (size_t i = 0; i < n_coeffs; ++i)
      if (pt[i] >= t) throw Err{PIRGPU_INVALID_ARGUMENT, "plaintext coefficient is not reduced modulo plain_modulus"};
    std::vector<uint64_t> u((size_t)km * N), x((size_t)km * N);
    std::vector<int> us(N);
    for (auto& v : us) v = rng.ternary();
    small_to_ntt(us, u.data());
    for (uint32_t comp = 0; comp < 2; ++comp) {
      const uint64_t* pkc = pk.data() + (size_t)comp * km * N;
      const std::vector<int> e = sample_noise();
      for (uint32_t j = 0; j < km; ++j) {
        const Modulus& m = mod[j];
        uint64_t* xj = x.data() + (size_t)j * N;
        for (uint32_t i = 0; i < N; ++i) xj[i] = m.mul(pkc[(size_t)j * N + i], u[(size_t)j * N + i]);
        m.intt(xj);
        for (uint32_t i = 0; i < N; ++i) xj[i] = m.add(xj[i], m.from_signed(e[i]));
      }
      // divide_and_round_q_last: (x + p/2 - [(x + p/2) mod p]) / p
      const Modulus& ms = mod[k];
      uint64_t* last = x.data() + (size_t)k * N;
      const uint64_t half = ms.q >> 1;
      for (uint32_t i = 0; i < N; ++i) last[i] = ms.add(last[i], half);
      for (uint32_t j = 0; j < k; ++j) {
        const Modulus& m = mod[j];
        const uint64_t* xj = x.data() + (size_t)j * N;
        uint64_t* o = ct + ((size_t)comp * k + j) * N;
        for (uint32_t i = 0; i < N; ++i) {
          const uint64_t tmp = m.sub(last[i] % m.q, half_mod[j]);
          o[i] = m.mul(m.sub(xj[i], tmp), p_inv[j]);
        }
      }
    }
    // multiply_add_plain_with_scaling_variant: c0 += floor(Q/t) m + floor((m (Q mod t) + (t+1)/2) / t)
    const uint64_t half_t = (t + 1) >> 1;
    for (size_t i = 0; i < n_coeffs; ++i) {
      if (!pt[i]) continue;
      const uint64_t fix = (uint64_t)(((u128)pt[i] * q_mod_t + half_t) / t);
      for (uint32_t j = 0; j < k; ++j) {
        const Modulus& m = mod[j];
        uint64_t* o = ct + (size_t)j * N;
        o[i] = m.add(o[i], m.add(m.mul(pt[i] % m.q, delta_mod[j]), fix % m.q));
      }
    }
  }

  void check_ct(const uint64_t* ct) const {
    for (uint32_t c = 0; c < 2; ++c)
      for (uint32_t j = 0; j < k; ++j) {
        const uint64_t* v = ct + ((size_t)c * k + j) * N;
        for (uint32_t i = 0; i < N; ++i)
          if (v[i] >= mod[j].q) throw Err{PIRGPU_INVALID_ARGUMENT, "ciphertext data is invalid (coefficient out of range)"};
      }
  }
  // [c0 + c1 s]_{q_j}, coefficient form: out [k][N]
  void phase(const uint64_t* ct, uint64_t* out) const {
    for (uint32_t j = 0; j < k; ++j) {
      const Modulus& m = mod[j];
      uint64_t* o = out + (size_t)j * N;
      memcpy(o, ct + ((size_t)k + j) * N, (size_t)N * 8);
      m.ntt(o);
      for (uint32_t i = 0; i < N; ++i) o[i] = m.mul(o[i], s_ntt[(size_t)j * N + i]);
      m.intt(o);
      const uint64_t* c0 = ct + (size_t)j * N;
      for (uint32_t i = 0; i < N; ++i) o[i] = m.add(o[i], c0[i]);
    }
  }
  // round(t x / Q) mod t with x = CRT(x_j): since x = sum_j y_j Q/q_j - v Q (y_j = x_j (Q/q_j)^-1 mod q_j),
  // t x / Q = sum_j t y_j / q_j (mod t); integer and 64-bit fixed-point fractional parts are summed apart.
  void decrypt(const uint64_t* ct, uint64_t* pt) const {
    check_ct(ct);
    std::vector<uint64_t> ph((size_t)k * N);
    phase(ct, ph.data());
    for (uint32_t i = 0; i < N; ++i) {
      u128 ipart = 0, frac = 0;
      for (uint32_t j = 0; j < k; ++j) {
        const Modulus& m = mod[j];
        const uint64_t y = m.mul(ph[(size_t)j * N + i], inv_punct[j]);
        const u128 ty = (u128)t * y;
        ipart += (uint64_t)(ty / m.q);
        frac += (((u128)(uint64_t)(ty % m.q)) << 64) / m.q;
      }
      frac += (u128)1 << 63;
      ipart += frac >> 64;
      pt[i] = (uint64_t)(ipart % t);
    }
  }
  // Decryptor::invariant_noise_budget: bits(Q) - bits(|[t x]_Q|_inf) - 1, floored at 0
  int noise_budget(const uint64_t* ct) const {
    check_ct(ct);
    std::vector<uint64_t> ph((size_t)k * N);
    phase(ct, ph.data());
    int worst = 0;
    for (uint32_t i = 0; i < N; ++i) {
      Big acc;
      u128 frac = 0;
      uint64_t ipart = 0;
      for (uint32_t j = 0; j < k; ++j) {
        const Modulus& m = mod[j];
        const uint64_t tx = m.mul(ph[(size_t)j * N + i], t % m.q);
        const uint64_t y = m.mul(tx, inv_punct[j]);
        acc.muladd(punct[j], y);
        frac += (((u128)y) << 64) / m.q;
        ipart += (uint64_t)(frac >> 64);
        frac = (uint64_t)frac;
      }
      // acc = [t x]_Q + v Q with v = floor(sum y_j / q_j) up to fixed-point error: correct by at most one Q
      Big vq = Q;
      vq.mul_word(ipart);
      if (acc.sub(vq)) acc.add(Q);
      while (acc.cmp(Q) >= 0) acc.sub(Q);
      if (acc.cmp(half_Q) > 0) {
        Big r = Q;
        r.sub(acc);
        acc = r;
      }
      worst = std::max(worst, acc.bits());
    }
    return std::max(0, q_bits - worst - 1);
  }

  // ---------------------------------------------------------------- reencoder
  void reencode(const uint64_t* ct, uint64_t* pts) const {  // ct_reencoder.cpp:40-73
    const uint64_t mask = (1ull << pt_bits) - 1;
    uint64_t* o = pts;
    for (uint32_t poly = 0; poly < 2; ++poly)
      for (uint32_t j = 0; j < k; ++j) {
        const uint64_t* src = ct + ((size_t)poly * k + j) * N;
        for (uint32_t d = 0, shift = 0; d < local_ratio[j]; ++d, shift += pt_bits, o += N)
          for (uint32_t c = 0; c < N; ++c) o[c] = (src[c] >> shift) & mask;
      }
  }
  void redecode(const uint64_t* pts, uint64_t* ct) const {  // ct_reencoder.cpp:79-111
    const uint64_t* in = pts;
    for (uint32_t poly = 0; poly < 2; ++poly)
      for (uint32_t j = 0; j < k; ++j) {
        uint64_t* dst = ct + ((size_t)poly * k + j) * N;
        for (uint32_t d = 0, shift = 0; d < local_ratio[j]; ++d, shift += pt_bits, in += N)
          for (uint32_t c = 0; c < N; ++c) dst[c] = shift ? dst[c] + (in[c] << shift) : in[c];
      }
  }

  // ---------------------------------------------------------------- PIR client logic
  uint64_t dim_sum() const {
    uint64_t s = 0;
    for (uint32_t i = 0; i < prm.num_dimensions; ++i) s += prm.dimensions[i];
    return s;
  }
  uint32_t query_ct_count() const { return (uint32_t)(dim_sum() / N + 1); }
  uint64_t reply_ct_count() const {
    uint64_t n = 1;
    for (uint32_t d = 1; d < prm.num_dimensions; ++d) n *= 2ull * exp_ratio;
    return n;
  }
  // PIRDatabase::calculate_indices (database.cpp:112-131)
  std::vector<uint64_t> calculate_indices(uint64_t index) const {
    uint64_t pt_index = index / prm.items_per_plaintext;
    std::vector<uint64_t> res(prm.num_dimensions, 0);
    for (int i = (int)prm.num_dimensions - 1; i >= 0; --i) {
      res[i] = pt_index % prm.dimensions[i];
      pt_index /= prm.dimensions[i];
    }
    return res;
  }
  // PIRDatabase::calculate_item_offset (database.cpp:133-138)
  uint64_t calculate_item_offset(uint64_t index) const {
    return (index % prm.items_per_plaintext) * prm.bytes_per_item;
  }

  // PIRClient::createQueryFor (client.cpp:92-144).  The query is the concatenation of one one-hot selection
  // vector per dimension, cut into ring-degree sized ciphertexts: dimension l's hot slot sits at global position
  // (d_0 + ... + d_{l-1}) + index_l, i.e. in ciphertext pos / N at coefficient pos % N, and carries the inverse of
  // the factor the expansion of that ciphertext multiplies in (N for full ciphertexts, next_power_two(dim_sum % N)
  // for the last one, client.cpp:122-125).
  void create_query(uint64_t desired_index, uint64_t* out) {
    if (desired_index >= prm.num_items)
      throw Err{PIRGPU_INVALID_ARGUMENT, "invalid index " + std::to_string(desired_index)};
    const std::vector<uint64_t> indices = calculate_indices(desired_index);
    const uint64_t ds = dim_sum();
    const uint32_t n_cts = query_ct_count();
    std::vector<std::pair<uint64_t, uint64_t>> hot;  // (global position, scale inverse mod t)
    uint64_t first_slot = 0;
    for (uint32_t l = 0; l < prm.num_dimensions; ++l) {
      const uint64_t pos = first_slot + indices[l];
      const bool last_ct = pos / N + 1 == n_cts;
      hot.emplace_back(pos, invert_mod_t(last_ct ? hm::next_power_two(ds % N) : N));
      first_slot += prm.dimensions[l];
    }
    std::vector<uint64_t> pt(N);
    for (uint32_t c = 0; c < n_cts; ++c) {
      std::fill(pt.begin(), pt.end(), 0);
      for (const auto& h : hot)
        if (h.first / N == c) pt[h.first % N] = h.second;
      encrypt(pt.data(), N, out + (size_t)c * ct_words());
    }
  }
  uint64_t invert_mod_t(uint64_t m) const {  // client.cpp:69-78 (try_invert_uint_mod: extended Euclid)
    int64_t a = (int64_t)(m % t), b = (int64_t)t, x0 = 1, x1 = 0;
    while (b) {
      int64_t qq = a / b, r = a % b;
      a = b, b = r;
      int64_t nx = x0 - qq * x1;
      x0 = x1, x1 = nx;
    }
    if (a != 1) throw Err{PIRGPU_INTERNAL, "Could not invert value"};
    return (uint64_t)(x0 < 0 ? x0 + (int64_t)t : x0);
  }

  void process_reply(const uint64_t* reply, size_t n_cts, uint64_t* pt_out) const {  // client.cpp:187-255
    if (prm.use_ciphertext_multiplication) {
      if (n_cts != 1)
        throw Err{PIRGPU_INVALID_ARGUMENT, "Number of ciphertexts in reply must be 1 when using CT multiplication"};
      decrypt(reply, pt_out);
      return;
    }
    const size_t ratio = (size_t)exp_ratio * 2;
    if (n_cts != reply_ct_count())
      throw Err{PIRGPU_INVALID_ARGUMENT, "Number of ciphertexts in reply does not match expected"};
    std::vector<uint64_t> cts(reply, reply + n_cts * ct_words()), pts;
    size_t n = n_cts;
    for (uint32_t d = 0; d < prm.num_dimensions; ++d) {
      pts.resize(n * N);
      for (size_t i = 0; i < n; ++i) decrypt(cts.data() + i * ct_words(), pts.data() + i * N);
      if (n <= 1) break;
      n /= ratio;
      for (size_t i = 0; i < n; ++i) redecode(pts.data() + i * ratio * N, cts.data() + i * ct_words());
    }
    memcpy(pt_out, pts.data(), (size_t)N * 8);
  }

  // StringEncoder::decode (string_encoder.cpp:124-163), same shift/or sequence on 8-bit chars
  void string_decode(const uint64_t* pt, size_t length, size_t byte_offset, uint8_t* out) const {
    const size_t bpc = bits_per_coeff;
    // pt.coeff_count() of a decrypted SEAL plaintext: Decryptor::decrypt trims the result to its significant
    // coefficients (at least one), and the reference's bound (string_encoder.cpp:126) is taken on that count
    size_t coeff_count = N;
    while (coeff_count > 1 && !pt[coeff_count - 1]) --coeff_count;
    if (byte_offset + length > coeff_count * bpc / 8)
      throw Err{PIRGPU_INVALID_ARGUMENT, "Requested decode beyond end of data in polynomial"};
    if (length == 0) return;
    const size_t start = byte_offset * 8 / bpc;
    size_t coeff_bits = (start + 1) * bpc - byte_offset * 8;
    memset(out, 0, length);
    size_t idx = 0, remain = 8;
    for (size_t i = start; i < coeff_count; ++i) {
      while (coeff_bits > 0) {
        const size_t n = std::min(coeff_bits, remain);
        out[idx] = (uint8_t)((out[idx] << n) | (uint8_t)(pt[i] >> (coeff_bits - n)));
        coeff_bits -= n;
        remain -= n;
        if (remain == 0) {
          if (++idx >= length) return;
          remain = 8;
        }
      }
      coeff_bits = bpc;
    }
  }

  // IntegerEncoder::decode_int64 (SEAL 3.5.6, base 2): Horner evaluation at x = 2 of the centred coefficients
  int64_t decode_int64(const uint64_t* pt) const {
    const uint64_t neg_threshold = (t + 1) >> 1;
    __int128 acc = 0;
    int top = (int)N - 1;
    while (top >= 0 && !pt[top]) --top;
    for (int i = top; i >= 0; --i) {
      acc *= 2;
      if (pt[i] >= neg_threshold) acc -= (__int128)(t - pt[i]); else acc += (__int128)pt[i];
      if (acc > (__int128)INT64_MAX || acc < (__int128)INT64_MIN)
        throw Err{PIRGPU_INTERNAL, "output out of range"};
    }
    return (int64_t)acc;
  }

  // ---------------------------------------------------------------- wire level
  // SaveRequest (serialization.cpp:44-73): one Ciphertexts per query, then this client's GaloisKeys and RelinKeys
  std::string save_request(const uint64_t* queries, size_t n) const {
    std::string out;
    const uint32_t nq = query_ct_count();
    for (size_t i = 0; i < n; ++i) {
      const uint64_t* q = queries + i * nq * ct_words();
      std::string cts;
      for (uint32_t c = 0; c < nq; ++c) wire::put_bytes_field(cts, 1, wire::save_ciphertext(sh, q + c * ct_words()));
      wire::put_bytes_field(out, 1, cts);
    }
    wire::put_bytes_field(out, 2, seeded_keys ? galois_blob_seeded : galois_blob);
    wire::put_bytes_field(out, 3, seeded_keys ? relin_blob_seeded : relin_blob);
    return out;
  }
  std::string create_request(const uint64_t* indexes, size_t n) {  // client.cpp:80-90
    const uint32_t nq = query_ct_count();
    std::vector<uint64_t> q(n * (size_t)nq * ct_words());
    for (size_t i = 0; i < n; ++i) create_query(indexes[i], q.data() + i * nq * ct_words());
    return save_request(q.data(), n);
  }
  // pir.Response (payload.proto:39-42): the Ciphertexts sub-messages
  static std::vector<std::pair<const uint8_t*, size_t>> parse_response(const uint8_t* data, size_t len) {
    std::vector<std::pair<const uint8_t*, size_t>> replies;
    wire::Reader r{data, data + len};
    while (r.p < r.end) {
      uint64_t tag;
      if (!r.varint(tag)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Response"};
      const uint8_t* d;
      size_t l;
      if ((tag >> 3) == 1 && (tag & 7) == 2) {
        if (!r.bytes(d, l)) throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Response.reply"};
        replies.emplace_back(d, l);
      } else if (!r.skip((uint32_t)(tag & 7))) {
        throw Err{PIRGPU_INVALID_ARGUMENT, "malformed Response"};
      }
    }
    return replies;
  }
  void reply_plaintext(const std::pair<const uint8_t*, size_t>& msg, uint64_t* pt) const {
    std::vector<uint64_t> cts;
    const uint32_t n = wire::load_query(sh, msg.first, msg.second, cts);
    process_reply(cts.data(), n, pt);
  }
};

// ==================================================================== C ABI

static thread_local std::string g_create_error;

template <class F>
static int guarded(pirclient* c, F&& f) {
  try {
    f();
    return PIRGPU_OK;
  } catch (const Err& e) {
    if (c) c->err = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    if (c) c->err = e.what();
    return PIRGPU_INTERNAL;
  }
}

extern "C" {

int pirclient_create(const pirgpu_params* params, const uint8_t* seed, size_t seed_len, pirclient** out) {
  if (!params || !out) {
    g_create_error = "null argument";
    return PIRGPU_INVALID_ARGUMENT;
  }
  *out = nullptr;
  std::unique_ptr<pirclient> c(new pirclient());
  try {
    c->init(*params, seed, seed_len);
  } catch (const Err& e) {
    g_create_error = e.msg;
    return e.code;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return PIRGPU_INTERNAL;
  }
  *out = c.release();
  return PIRGPU_OK;
}

void pirclient_destroy(pirclient* c) {
  if (!c) return;
  // wipe the secret key and the generator state before the memory is returned
  if (!c->s_ntt.empty()) explicit_bzero(c->s_ntt.data(), c->s_ntt.size() * sizeof(uint64_t));
  if (!c->s_signed.empty()) explicit_bzero(c->s_signed.data(), c->s_signed.size());
  explicit_bzero(&c->rng, sizeof(c->rng));
  delete c;
}
const char* pirclient_last_error(const pirclient* c) { return c ? c->err.c_str() : "null client"; }
const char* pirclient_create_error(void) { return g_create_error.c_str(); }
int pirclient_set_seeded_keys(pirclient* c, int enabled) {
  if (!c) return PIRGPU_INVALID_ARGUMENT;
  c->seeded_keys = enabled != 0;
  return PIRGPU_OK;
}

void pirclient_free(void* p) { free(p); }

int pirclient_create_request(pirclient* c, const uint64_t* indexes, size_t n_indexes, uint8_t** request,
                             size_t* request_len) {
  if (!c || (!indexes && n_indexes) || !request || !request_len) return PIRGPU_INVALID_ARGUMENT;
  *request = nullptr;
  *request_len = 0;
  return guarded(c, [&] {
    const std::string s = c->create_request(indexes, n_indexes);
    uint8_t* buf = (uint8_t*)malloc(s.size() ? s.size() : 1);
    if (!buf) throw Err{PIRGPU_INTERNAL, "out of memory"};
    memcpy(buf, s.data(), s.size());
    *request = buf;
    *request_len = s.size();
  });
}

int pirclient_save_request(pirclient* c, const uint64_t* queries, size_t n_queries, uint8_t** request, size_t* request_len) {
  if (!c || (!queries && n_queries) || !request || !request_len) return PIRGPU_INVALID_ARGUMENT;
  *request = nullptr;
  *request_len = 0;
  return guarded(c, [&] {
    const std::string s = c->save_request(queries, n_queries);
    uint8_t* buf = (uint8_t*)malloc(s.size() ? s.size() : 1);
    if (!buf) throw Err{PIRGPU_INTERNAL, "out of memory"};
    memcpy(buf, s.data(), s.size());
    *request = buf;
    *request_len = s.size();
  });
}

int pirclient_load_response(pirclient* c, const uint8_t* response, size_t response_len, uint64_t* replies_out,
                            size_t cap_replies, size_t* n_replies) {
  if (!c || (!response && response_len) || !n_replies) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] {
    const auto replies = pirclient::parse_response(response, response_len);
    if (cap_replies < replies.size() || (!replies_out && !replies.empty())) throw Err{PIRGPU_INVALID_ARGUMENT, "replies_out too small"};
    const size_t per = (size_t)c->reply_ct_count() * c->ct_words();
    std::vector<uint64_t> cts;
    for (size_t i = 0; i < replies.size(); ++i) {
      const uint32_t n = wire::load_query(c->sh, replies[i].first, replies[i].second, cts);
      if (n != c->reply_ct_count()) throw Err{PIRGPU_INVALID_ARGUMENT, "Number of ciphertexts in reply does not match expected"};
      memcpy(replies_out + i * per, cts.data(), per * 8);
    }
    *n_replies = replies.size();
  });
}

int pirclient_process_response(pirclient* c, const uint64_t* indexes, size_t n_indexes, const uint8_t* response,
                               size_t response_len, uint8_t* items_out, size_t items_cap) {
  if (!c || (!indexes && n_indexes) || (!response && response_len)) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] {
    const auto replies = pirclient::parse_response(response, response_len);
    if (replies.size() != n_indexes)
      throw Err{PIRGPU_INVALID_ARGUMENT, "Number of indexes must match number of replies"};
    const size_t item = c->prm.bytes_per_item;
    if (items_cap < n_indexes * item || (!items_out && n_indexes * item != 0))
      throw Err{PIRGPU_INVALID_ARGUMENT, "items_out too small"};
    std::vector<uint64_t> pt(c->N);
    for (size_t i = 0; i < n_indexes; ++i) {
      c->reply_plaintext(replies[i], pt.data());
      c->string_decode(pt.data(), item, c->calculate_item_offset(indexes[i]), items_out + i * item);
    }
  });
}

int pirclient_process_response_integer(pirclient* c, const uint8_t* response, size_t response_len, int64_t* out,
                                       size_t out_cap, size_t* n_out) {
  if (!c || (!response && response_len) || !n_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] {
    const auto replies = pirclient::parse_response(response, response_len);
    if (out_cap < replies.size() || (!out && !replies.empty())) throw Err{PIRGPU_INVALID_ARGUMENT, "out too small"};
    std::vector<uint64_t> pt(c->N);
    for (size_t i = 0; i < replies.size(); ++i) {
      c->reply_plaintext(replies[i], pt.data());
      out[i] = c->decode_int64(pt.data());
    }
    *n_out = replies.size();
  });
}

uint32_t pirclient_query_ct_count(const pirclient* c) { return c ? c->query_ct_count() : 0; }
uint64_t pirclient_reply_ct_count(const pirclient* c) { return c ? c->reply_ct_count() : 0; }

int pirclient_create_query(pirclient* c, uint64_t index, uint64_t* query_out, size_t cap_cts, uint32_t* n_cts) {
  if (!c || !query_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] {
    if (cap_cts < c->query_ct_count()) throw Err{PIRGPU_INVALID_ARGUMENT, "query_out too small"};
    c->create_query(index, query_out);
    if (n_cts) *n_cts = c->query_ct_count();
  });
}

int pirclient_galois_key(const pirclient* cc, uint32_t elt, uint64_t* key_out) {
  pirclient* c = const_cast<pirclient*>(cc);
  if (!c || !key_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] {
    auto it = c->galois.find(elt);
    if (it == c->galois.end()) throw Err{PIRGPU_INVALID_ARGUMENT, "no Galois key for element " + std::to_string(elt)};
    memcpy(key_out, it->second.data(), it->second.size() * 8);
  });
}

int pirclient_process_reply(pirclient* c, const uint64_t* reply, size_t n_cts, uint64_t* plaintext_out) {
  if (!c || !reply || !plaintext_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] { c->process_reply(reply, n_cts, plaintext_out); });
}

int pirclient_encrypt(pirclient* c, const uint64_t* plaintext, size_t n_coeffs, uint64_t* ct_out) {
  if (!c || (!plaintext && n_coeffs) || !ct_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] { c->encrypt(plaintext, n_coeffs, ct_out); });
}

int pirclient_decrypt(pirclient* c, const uint64_t* ct, uint64_t* plaintext_out) {
  if (!c || !ct || !plaintext_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] { c->decrypt(ct, plaintext_out); });
}

int pirclient_noise_budget(pirclient* c, const uint64_t* ct, int* bits) {
  if (!c || !ct || !bits) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] { *bits = c->noise_budget(ct); });
}

int pirclient_reencode(const pirclient* cc, const uint64_t* ct, uint64_t* plaintexts_out, size_t cap_pts,
                       uint32_t* n_pts) {
  pirclient* c = const_cast<pirclient*>(cc);
  if (!c || !ct || !plaintexts_out) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] {
    if (cap_pts < 2ull * c->exp_ratio) throw Err{PIRGPU_INVALID_ARGUMENT, "plaintexts_out too small"};
    c->reencode(ct, plaintexts_out);
    if (n_pts) *n_pts = 2 * c->exp_ratio;
  });
}

int pirclient_string_decode(const pirclient* cc, const uint64_t* plaintext, size_t length, size_t byte_offset,
                            uint8_t* out) {
  pirclient* c = const_cast<pirclient*>(cc);
  if (!c || !plaintext || (!out && length)) return PIRGPU_INVALID_ARGUMENT;
  return guarded(c, [&] { c->string_decode(plaintext, length, byte_offset, out); });
}

}  // extern "C"
