// pir_facade.h -- header-only C++17 mirror of the reference's server-side classes over the C ABI.
//
// Same class and method names, argument meaning and error behaviour as
//   pir::PIRDatabase  (reference pir/cpp/database.h:37-133)
//   pir::PIRServer    (reference pir/cpp/server.h:33-145)
// with the reference's abseil / SEAL / protobuf types replaced by minimal stand-ins that
// live in this header (none of those libraries exist in this image):
//   pir::Status / pir::StatusOr<T>   absl::Status(Or) with the numeric absl::StatusCode values
//   pir::Ciphertext                  seal::Ciphertext: uint64_t[2][k][N] in SEAL's own layout
//   pir::GaloisKeys                  map galois_elt -> uint64_t[k][2][k+1][N] (NTT form)
//   std::string request / response   serialized pir.Request / pir.Response (payload.proto)
// Every method is a thin call into libpirgpu (include/pirgpu.h); nothing is computed here.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pirgpu.h"

namespace pir {

enum class StatusCode : int { kOk = 0, kInvalidArgument = 3, kFailedPrecondition = 9, kUnimplemented = 12, kInternal = 13 };

class Status {
 public:
  Status() = default;
  Status(StatusCode code, std::string msg) : code_(code), msg_(std::move(msg)) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const std::string& message() const { return msg_; }

 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
inline Status InvalidArgumentError(std::string m) { return Status(StatusCode::kInvalidArgument, std::move(m)); }

template <typename T>
class StatusOr {
 public:
  StatusOr(Status s) : status_(std::move(s)) {}           // NOLINT
  StatusOr(T v) : value_(std::move(v)) {}                 // NOLINT
  bool ok() const { return status_.ok(); }
  const Status& status() const { return status_; }
  T& value() { return value_; }
  T& operator*() { return value_; }
  T* operator->() { return &value_; }

 private:
  Status status_;
  T value_{};
};

// PIRParameters (payload.proto:45-69) with the SEAL EncryptionParameters flattened.
struct PIRParameters {
  uint32_t poly_modulus_degree = 4096;
  std::vector<uint64_t> coeff_modulus;   // data primes followed by the special prime (SEAL order)
  uint64_t plain_modulus = 0;
  uint64_t num_items = 0, num_pt = 0;
  std::vector<uint32_t> dimensions;
  uint32_t bytes_per_item = 0, items_per_plaintext = 0, bits_per_coeff = 0;
  bool use_ciphertext_multiplication = false;
  size_t DimensionsSum() const {          // PIRContext::DimensionsSum, context.h:59-62
    size_t s = 0;
    for (auto d : dimensions) s += d;
    return s;
  }
};

using Ciphertext = std::vector<uint64_t>;                        // [2][k][N]
using GaloisKeys = std::map<uint32_t, std::vector<uint64_t>>;    // elt -> [k][2][k+1][N]

namespace detail {
inline Status FromRc(pirgpu_ctx* c, int rc) {
  return rc == 0 ? OkStatus() : Status(static_cast<StatusCode>(rc), c ? pirgpu_last_error(c) : pirgpu_create_error());
}
}  // namespace detail

class PIRDatabase {
 public:
  ~PIRDatabase() { pirgpu_destroy(ctx_); }
  PIRDatabase(const PIRDatabase&) = delete;

  // database.cpp:40-44
  static StatusOr<std::shared_ptr<PIRDatabase>> Create(std::shared_ptr<PIRParameters> params, int device = 0) {
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
    p.device = device;
    pirgpu_ctx* ctx = nullptr;
    int rc = pirgpu_create(&p, &ctx);
    if (rc) return detail::FromRc(nullptr, rc);
    return std::shared_ptr<PIRDatabase>(new PIRDatabase(ctx, std::move(params)));
  }

  // database.cpp:52-58
  static StatusOr<std::shared_ptr<PIRDatabase>> Create(const std::vector<std::string>& rawdb,
                                                       std::shared_ptr<PIRParameters> params, int device = 0) {
    auto db = Create(std::move(params), device);
    if (!db.ok()) return db.status();
    Status s = (*db)->populate(rawdb);
    if (!s.ok()) return s;
    return db;
  }

  // database.cpp:84-110
  Status populate(const std::vector<std::string>& rawdb) {
    if (rawdb.size() != params_->num_items)
      return InvalidArgumentError("Database size " + std::to_string(rawdb.size()) + " does not match params value " +
                                  std::to_string(params_->num_items));
    std::string flat;
    flat.reserve(rawdb.size() * params_->bytes_per_item);
    for (const auto& s : rawdb) {
      if (s.size() != params_->bytes_per_item) return InvalidArgumentError("item size does not match parameters");
      flat += s;
    }
    return detail::FromRc(ctx_, pirgpu_db_load_items(ctx_, reinterpret_cast<const uint8_t*>(flat.data()),
                                                     rawdb.size(), params_->bytes_per_item));
  }

  // database.cpp:290-316 (selection vector in coefficient form; it is not mutated here)
  StatusOr<std::vector<Ciphertext>> multiply(const std::vector<Ciphertext>& selection_vector) const {
    const size_t words = CtWords();
    std::vector<uint64_t> in(selection_vector.size() * words);
    for (size_t i = 0; i < selection_vector.size(); ++i) {
      if (selection_vector[i].size() != words) return InvalidArgumentError("ciphertext has the wrong shape");
      std::copy(selection_vector[i].begin(), selection_vector[i].end(), in.begin() + i * words);
    }
    const uint64_t n = pirgpu_reply_ct_count(ctx_);
    std::vector<uint64_t> out(n * words);
    uint64_t got = 0;
    int rc = pirgpu_multiply(ctx_, in.data(), selection_vector.size(), out.data(), n, &got);
    if (rc) return detail::FromRc(ctx_, rc);
    std::vector<Ciphertext> result(got);
    for (uint64_t i = 0; i < got; ++i) result[i].assign(out.begin() + i * words, out.begin() + (i + 1) * words);
    return result;
  }

  std::size_t size() const { return pirgpu_db_size(ctx_); }   // database.h:97

  // database.cpp:318-326
  std::vector<uint32_t> calculate_indices(uint32_t index) const {
    uint32_t pt_index = index / params_->items_per_plaintext;
    std::vector<uint32_t> results(params_->dimensions.size(), 0);
    for (int i = static_cast<int>(results.size()) - 1; i >= 0; --i) {
      results[i] = pt_index % params_->dimensions[i];
      pt_index = pt_index / params_->dimensions[i];
    }
    return results;
  }
  // database.cpp:328-332
  size_t calculate_item_offset(uint32_t index) const {
    uint32_t pt_index = index / params_->items_per_plaintext;
    return (index - pt_index * params_->items_per_plaintext) * params_->bytes_per_item;
  }

  size_t CtWords() const { return 2 * (params_->coeff_modulus.size() - 1) * params_->poly_modulus_degree; }
  pirgpu_ctx* handle() const { return ctx_; }
  const std::shared_ptr<PIRParameters>& Params() const { return params_; }

 private:
  PIRDatabase(pirgpu_ctx* ctx, std::shared_ptr<PIRParameters> params) : ctx_(ctx), params_(std::move(params)) {}
  pirgpu_ctx* ctx_;
  std::shared_ptr<PIRParameters> params_;
};

class PIRServer {
 public:
  // server.cpp:35-42
  static StatusOr<std::unique_ptr<PIRServer>> Create(std::shared_ptr<PIRDatabase> db,
                                                     std::shared_ptr<PIRParameters> params) {
    if (params->num_pt != db->size()) return InvalidArgumentError("database size mismatch");
    return std::unique_ptr<PIRServer>(new PIRServer(std::move(db), std::move(params)));
  }

  // server.cpp:44-65 on serialized pir.Request / pir.Response
  StatusOr<std::string> ProcessRequest(const std::string& request) const {
    uint8_t* resp = nullptr;
    size_t len = 0;
    int rc = pirgpu_process_request(db_->handle(), reinterpret_cast<const uint8_t*>(request.data()), request.size(),
                                    &resp, &len);
    if (rc) return detail::FromRc(db_->handle(), rc);
    std::string out(reinterpret_cast<const char*>(resp), len);
    pirgpu_free(resp);
    return out;
  }

  // The same for several independent requests (different clients) served TOGETHER -- no reference counterpart (the
  // reference is single-threaded); result[i] is what ProcessRequest(requests[i]) would have returned.
  std::vector<StatusOr<std::string>> ProcessRequests(const std::vector<std::string>& requests) const {
    const uint32_t n = static_cast<uint32_t>(requests.size());
    std::vector<const uint8_t*> ptrs(n);
    std::vector<size_t> lens(n), rlens(n, 0);
    std::vector<uint8_t*> resps(n, nullptr);
    std::vector<int> status(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
      ptrs[i] = reinterpret_cast<const uint8_t*>(requests[i].data());
      lens[i] = requests[i].size();
    }
    pirgpu_process_requests(db_->handle(), n, ptrs.data(), lens.data(), resps.data(), rlens.data(), status.data());
    std::vector<StatusOr<std::string>> out;
    out.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
      if (status[i]) {
        out.emplace_back(Status(static_cast<StatusCode>(status[i]), pirgpu_request_error(i)));
      } else {
        out.emplace_back(std::string(reinterpret_cast<const char*>(resps[i]), rlens[i]));
        pirgpu_free(resps[i]);
      }
    }
    return out;
  }

  // ProcessRequests in two halves (pirgpu_process_requests_begin / _end): ONE calling thread keeps two calls in flight --
  // Begin(next) before End(previous) -- so the next call's parsing / staging / queueing run under the previous call's
  // tail.  The request strings must stay alive and untouched until End; every Begin is matched by exactly one End.
  // Dropping a PendingRequests without ProcessRequestsEnd (an early return or an exception in the caller) ends the call
  // in the destructor: the library's serving thread writes into the arrays below until then.
  class PendingRequests {
   public:
    PendingRequests() = default;
    PendingRequests(const PendingRequests&) = delete;
    PendingRequests& operator=(const PendingRequests&) = delete;
    PendingRequests(PendingRequests&& o) noexcept { take(o); }
    PendingRequests& operator=(PendingRequests&& o) noexcept {
      if (this != &o) {
        finish();
        take(o);
      }
      return *this;
    }
    ~PendingRequests() { finish(); }
    bool valid() const { return call_ != nullptr; }

   private:
    void take(PendingRequests& o) {   // the vectors' heap blocks (whose addresses the library holds) move with them
      call_ = o.call_;
      o.call_ = nullptr;
      ptrs_ = std::move(o.ptrs_);
      lens_ = std::move(o.lens_);
      rlens_ = std::move(o.rlens_);
      resps_ = std::move(o.resps_);
      status_ = std::move(o.status_);
    }
    void finish() {                   // an unclaimed call: wait for it and free what it produced
      if (!call_) return;
      pirgpu_process_requests_end(call_);
      call_ = nullptr;
      for (size_t i = 0; i < resps_.size(); ++i)
        if (!status_[i] && resps_[i]) pirgpu_free(resps_[i]);
    }
    friend class PIRServer;
    void* call_ = nullptr;
    std::vector<const uint8_t*> ptrs_;
    std::vector<size_t> lens_, rlens_;
    std::vector<uint8_t*> resps_;
    std::vector<int> status_;
  };
  StatusOr<std::unique_ptr<PendingRequests>> ProcessRequestsBegin(const std::vector<std::string>& requests) const {
    auto p = std::unique_ptr<PendingRequests>(new PendingRequests());   // heap: the arrays' addresses are handed to the library
    const uint32_t n = static_cast<uint32_t>(requests.size());
    p->ptrs_.resize(n);
    p->lens_.resize(n);
    p->rlens_.assign(n, 0);
    p->resps_.assign(n, nullptr);
    p->status_.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
      p->ptrs_[i] = reinterpret_cast<const uint8_t*>(requests[i].data());
      p->lens_[i] = requests[i].size();
    }
    const int rc = pirgpu_process_requests_begin(db_->handle(), n, p->ptrs_.data(), p->lens_.data(), p->resps_.data(),
                                                 p->rlens_.data(), p->status_.data(), &p->call_);
    if (rc) return Status(static_cast<StatusCode>(rc), "pirgpu_process_requests_begin failed");
    return p;
  }
  std::vector<StatusOr<std::string>> ProcessRequestsEnd(std::unique_ptr<PendingRequests> p) const {
    std::vector<StatusOr<std::string>> out;
    if (!p || !p->call_) return out;
    pirgpu_process_requests_end(p->call_);
    p->call_ = nullptr;
    const uint32_t n = static_cast<uint32_t>(p->status_.size());
    out.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
      if (p->status_[i]) {
        out.emplace_back(Status(static_cast<StatusCode>(p->status_[i]), pirgpu_request_error(i)));
      } else {
        out.emplace_back(std::string(reinterpret_cast<const char*>(p->resps_[i]), p->rlens_[i]));
        pirgpu_free(p->resps_[i]);
      }
    }
    return out;
  }

  // what SEALDeserialize<GaloisKeys> yields (server.cpp:46-48)
  Status SetGaloisKeys(const GaloisKeys& keys) const {
    const size_t k = params_->coeff_modulus.size() - 1;
    const size_t key_words = k * 2 * (k + 1) * params_->poly_modulus_degree;
    for (auto it = keys.begin(); it != keys.end(); ++it)
      if (it->second.size() != key_words) return InvalidArgumentError("Galois key has the wrong size");
    int rc = pirgpu_clear_galois_keys(db_->handle());
    for (auto it = keys.begin(); rc == 0 && it != keys.end(); ++it)
      rc = pirgpu_set_galois_key(db_->handle(), it->first, it->second.data());
    return detail::FromRc(db_->handle(), rc);
  }

  // server.cpp:67-76
  Status substitute_power_x_inplace(Ciphertext& ct, uint32_t power) const {
    if (ct.size() != db_->CtWords()) return InvalidArgumentError("ciphertext has the wrong size");
    return detail::FromRc(db_->handle(), pirgpu_substitute_power_x(db_->handle(), ct.data(), power));
  }
  // server.cpp:78-103
  void multiply_inverse_power_of_x(const Ciphertext& encrypted, uint32_t k, Ciphertext& destination) const {
    if (encrypted.size() != db_->CtWords()) {  // the reference's signature has no status to report through
      destination.clear();
      return;
    }
    destination.resize(encrypted.size());
    pirgpu_multiply_inverse_power_of_x(db_->handle(), encrypted.data(), k, destination.data());
  }
  // server.cpp:105-146
  StatusOr<std::vector<Ciphertext>> oblivious_expansion(const Ciphertext& ct, size_t num_items) const {
    const size_t words = db_->CtWords();
    if (ct.size() != words) return InvalidArgumentError("ciphertext has the wrong size");
    std::vector<uint64_t> out(std::max<size_t>(num_items, 1) * words);
    int rc = pirgpu_expand(db_->handle(), ct.data(), static_cast<uint32_t>(num_items), out.data());
    if (rc) return detail::FromRc(db_->handle(), rc);
    std::vector<Ciphertext> result(num_items);
    for (size_t i = 0; i < num_items; ++i) result[i].assign(out.begin() + i * words, out.begin() + (i + 1) * words);
    return result;
  }
  // server.cpp:148-171
  StatusOr<std::vector<Ciphertext>> oblivious_expansion(const std::vector<Ciphertext>& cts, size_t total_items) const {
    const size_t words = db_->CtWords();
    std::vector<uint64_t> in(cts.size() * words), out(std::max<size_t>(total_items, 1) * words);
    for (size_t i = 0; i < cts.size(); ++i) {
      if (cts[i].size() != words) return InvalidArgumentError("ciphertext has the wrong size");
      std::copy(cts[i].begin(), cts[i].end(), in.begin() + i * words);
    }
    int rc = pirgpu_expand_multi(db_->handle(), in.data(), static_cast<uint32_t>(cts.size()), total_items, out.data());
    if (rc) return detail::FromRc(db_->handle(), rc);
    std::vector<Ciphertext> result(total_items);
    for (size_t i = 0; i < total_items; ++i) result[i].assign(out.begin() + i * words, out.begin() + (i + 1) * words);
    return result;
  }

  const std::shared_ptr<PIRDatabase>& Database() const { return db_; }

 private:
  PIRServer(std::shared_ptr<PIRDatabase> db, std::shared_ptr<PIRParameters> params)
      : db_(std::move(db)), params_(std::move(params)) {}
  std::shared_ptr<PIRDatabase> db_;
  std::shared_ptr<PIRParameters> params_;
};

}  // namespace pir
