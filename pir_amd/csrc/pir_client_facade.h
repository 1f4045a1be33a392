// pir_client_facade.h -- header-only C++ mirror of the reference's PIRClient (pir/cpp/client.h:34-97)
// over the C ABI of libpirclient.so (include/pirclient.h).  CPU only; pairs with pir_facade.h
// (PIRDatabase / PIRServer over libpirgpu.so) for a complete C++ round trip:
//   auto client = pir::PIRClient::Create(params);                    // client.cpp:61-67
//   auto request = (*client)->CreateRequest({index});                // client.cpp:80-90   serialized pir.Request
//   auto response = (*server)->ProcessRequest(*request);             // server.cpp:44-65
//   auto items = (*client)->ProcessResponse({index}, *response);     // client.cpp:160-185
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/pirclient.h"
#include "pir_facade.h"

namespace pir {

class PIRClient {
 public:
  ~PIRClient() { pirclient_destroy(c_); }
  PIRClient(const PIRClient&) = delete;

  // client.cpp:61-67 (+ initialize(): key generation, serialized Galois / relinearisation keys).
  // A non-empty seed makes key generation and encryption deterministic (tests).
  static StatusOr<std::unique_ptr<PIRClient>> Create(std::shared_ptr<PIRParameters> params,
                                                     const std::string& seed = std::string()) {
    if (params->coeff_modulus.size() < 2 || params->coeff_modulus.size() > PIRGPU_MAX_PRIMES + 1 ||
        params->dimensions.empty() || params->dimensions.size() > PIRGPU_MAX_DIMS)
      return InvalidArgumentError("invalid parameters");
    pirgpu_params p{};
    p.poly_modulus_degree = params->poly_modulus_degree;
    p.num_data_primes = static_cast<uint32_t>(params->coeff_modulus.size() - 1);
    for (uint32_t i = 0; i < p.num_data_primes; ++i) p.coeff_modulus[i] = params->coeff_modulus[i];
    p.special_prime = params->coeff_modulus.back();
    p.plain_modulus = params->plain_modulus;
    p.num_dimensions = static_cast<uint32_t>(params->dimensions.size());
    for (size_t i = 0; i < params->dimensions.size(); ++i) p.dimensions[i] = params->dimensions[i];
    p.num_pt = params->num_pt;
    p.num_items = params->num_items;
    p.bytes_per_item = params->bytes_per_item;
    p.items_per_plaintext = params->items_per_plaintext;
    p.bits_per_coeff = params->bits_per_coeff;
    p.use_ciphertext_multiplication = params->use_ciphertext_multiplication ? 1 : 0;
    pirclient* c = nullptr;
    int rc = pirclient_create(&p, seed.empty() ? nullptr : reinterpret_cast<const uint8_t*>(seed.data()), seed.size(), &c);
    if (rc) return Status(static_cast<StatusCode>(rc), pirclient_create_error());
    return std::unique_ptr<PIRClient>(new PIRClient(c, std::move(params)));
  }

  // client.cpp:80-90 -> serialized pir.Request
  StatusOr<std::string> CreateRequest(const std::vector<std::size_t>& indexes) const {
    std::vector<uint64_t> idx(indexes.begin(), indexes.end());
    uint8_t* buf = nullptr;
    size_t len = 0;
    int rc = pirclient_create_request(c_, idx.data(), idx.size(), &buf, &len);
    if (rc) return Err(rc);
    std::string out(reinterpret_cast<const char*>(buf), len);
    pirclient_free(buf);
    return out;
  }

  // client.cpp:160-185: serialized pir.Response -> one item per index
  StatusOr<std::vector<std::string>> ProcessResponse(const std::vector<std::size_t>& indexes,
                                                     const std::string& response) const {
    std::vector<uint64_t> idx(indexes.begin(), indexes.end());
    const size_t item = params_->bytes_per_item;
    std::string flat(idx.size() * item, '\0');
    int rc = pirclient_process_response(c_, idx.data(), idx.size(), reinterpret_cast<const uint8_t*>(response.data()),
                                        response.size(), reinterpret_cast<uint8_t*>(&flat[0]), flat.size());
    if (rc) return Err(rc);
    std::vector<std::string> out(idx.size());
    for (size_t i = 0; i < idx.size(); ++i) out[i] = flat.substr(i * item, item);
    return out;
  }

  // client.cpp:146-158
  StatusOr<std::vector<int64_t>> ProcessResponseInteger(const std::string& response) const {
    std::vector<int64_t> out(4096);
    size_t n = 0;
    int rc = pirclient_process_response_integer(c_, reinterpret_cast<const uint8_t*>(response.data()), response.size(),
                                                out.data(), out.size(), &n);
    if (rc) return Err(rc);
    out.resize(n);
    return out;
  }

  // The SEAL objects the reference's tests reach through friend access (client_test.cpp:55-58), residue level.
  StatusOr<Ciphertext> Encrypt(const std::vector<uint64_t>& plaintext) const {
    Ciphertext ct(CtWords());
    int rc = pirclient_encrypt(c_, plaintext.data(), plaintext.size(), ct.data());
    if (rc) return Err(rc);
    return ct;
  }
  StatusOr<std::vector<uint64_t>> Decrypt(const Ciphertext& ct) const {
    std::vector<uint64_t> pt(params_->poly_modulus_degree);
    int rc = pirclient_decrypt(c_, ct.data(), pt.data());
    if (rc) return Err(rc);
    return pt;
  }
  // createQueryFor (client.cpp:92-144) -> the query ciphertexts
  StatusOr<std::vector<Ciphertext>> CreateQueryFor(size_t index) const {
    const uint32_t n = pirclient_query_ct_count(c_);
    std::vector<uint64_t> flat(static_cast<size_t>(n) * CtWords());
    uint32_t got = 0;
    int rc = pirclient_create_query(c_, index, flat.data(), n, &got);
    if (rc) return Err(rc);
    std::vector<Ciphertext> out(got);
    for (uint32_t i = 0; i < got; ++i) out[i].assign(flat.begin() + i * CtWords(), flat.begin() + (i + 1) * CtWords());
    return out;
  }
  // Galois keys of initialize() for PIRServer::SetGaloisKeys (residue-level path)
  StatusOr<GaloisKeys> GaloisKeysForServer() const {
    GaloisKeys keys;
    const uint32_t N = params_->poly_modulus_degree;
    const size_t k = params_->coeff_modulus.size() - 1;
    for (uint32_t n = N; n > 1; n >>= 1) {   // generate_galois_elts (utils.cpp:7-14): N/2^i + 1
      const uint32_t elt = n + 1;
      std::vector<uint64_t> key(k * 2 * (k + 1) * N);
      int rc = pirclient_galois_key(c_, elt, key.data());
      if (rc) return Err(rc);
      keys.emplace(elt, std::move(key));
    }
    return keys;
  }
  size_t CtWords() const { return 2 * (params_->coeff_modulus.size() - 1) * params_->poly_modulus_degree; }

 private:
  PIRClient(pirclient* c, std::shared_ptr<PIRParameters> params) : c_(c), params_(std::move(params)) {}
  Status Err(int rc) const { return Status(static_cast<StatusCode>(rc), pirclient_last_error(c_)); }
  pirclient* c_;
  std::shared_ptr<PIRParameters> params_;
};

}  // namespace pir
