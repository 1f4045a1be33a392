// facade_test.cpp -- drives the C++ mirror of PIRDatabase / PIRServer (pir_amd/csrc/pir_facade.h)
// the way the reference's server_test.cpp drives the real classes.  Inputs (parameters, raw
// database, serialized Request) are written by tests/test_cpp_facade.py; the serialized
// Response is written back for comparison with the CPU oracle.
//   usage: facade_test <params.txt> <db.bin> <request.bin> <response.bin>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>

#include "../../pir_amd/csrc/pir_facade.h"

static std::string slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

#define EXPECT(cond)                                                      \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 5) return 2;
  auto params = std::make_shared<pir::PIRParameters>();
  {
    std::ifstream f(argv[1]);
    size_t nmod, ndim;
    f >> params->poly_modulus_degree >> params->plain_modulus >> params->num_items >> params->num_pt >>
        params->bytes_per_item >> params->items_per_plaintext >> params->bits_per_coeff >> nmod >> ndim;
    params->coeff_modulus.resize(nmod);
    for (auto& q : params->coeff_modulus) f >> q;
    params->dimensions.resize(ndim);
    for (auto& d : params->dimensions) f >> d;
  }
  const std::string raw = slurp(argv[2]);
  std::vector<std::string> rawdb(params->num_items);
  for (size_t i = 0; i < rawdb.size(); ++i) rawdb[i] = raw.substr(i * params->bytes_per_item, params->bytes_per_item);

  // PIRServer::Create with an unpopulated database -> InvalidArgument (server.cpp:37-39)
  {
    auto empty = pir::PIRDatabase::Create(params);
    EXPECT(empty.ok());
    auto srv = pir::PIRServer::Create(*empty, params);
    EXPECT(!srv.ok() && srv.status().code() == pir::StatusCode::kInvalidArgument);
    // populate with the wrong number of items -> InvalidArgument (database.cpp:85-90)
    std::vector<std::string> fewer(rawdb.begin(), rawdb.end() - 1);
    EXPECT((*empty)->populate(fewer).code() == pir::StatusCode::kInvalidArgument);
  }
  // CT multiplication mode is Unimplemented on this path
  {
    auto p2 = std::make_shared<pir::PIRParameters>(*params);
    p2->use_ciphertext_multiplication = true;
    auto db = pir::PIRDatabase::Create(p2);
    EXPECT(!db.ok() && db.status().code() == pir::StatusCode::kUnimplemented);
  }

  auto db = pir::PIRDatabase::Create(rawdb, params);
  if (!db.ok()) {
    std::fprintf(stderr, "Create failed: %s\n", db.status().message().c_str());
    return 1;
  }
  EXPECT((*db)->size() == params->num_pt);
  auto server = pir::PIRServer::Create(*db, params);
  EXPECT(server.ok());

  // malformed request -> InvalidArgument, like SEALDeserialize failing (serialization.h:113-115)
  auto bad = (*server)->ProcessRequest(std::string("\x12\x03zzz", 5));
  EXPECT(!bad.ok() && bad.status().code() == pir::StatusCode::kInvalidArgument);

  auto response = (*server)->ProcessRequest(slurp(argv[3]));
  if (!response.ok()) {
    std::fprintf(stderr, "ProcessRequest failed: %s\n", response.status().message().c_str());
    return 1;
  }
  std::ofstream(argv[4], std::ios::binary) << *response;

  // the same request through ProcessRequests and through its two-halves form (one caller, two calls in flight):
  // identical bytes back, a malformed request fails alone
  {
    const std::vector<std::string> reqs = {slurp(argv[3]), std::string("\x12\x03zzz", 5), slurp(argv[3])};
    auto together = (*server)->ProcessRequests(reqs);
    EXPECT(together.size() == 3 && together[0].ok() && !together[1].ok() && together[2].ok());
    EXPECT(*together[0] == *response && *together[2] == *response);
    auto a = (*server)->ProcessRequestsBegin(reqs);
    auto b = (*server)->ProcessRequestsBegin(reqs);      // second call handed over before the first is waited for
    EXPECT(a.ok() && b.ok() && (*a)->valid() && (*b)->valid());
    auto ra = (*server)->ProcessRequestsEnd(std::move(*a));
    auto rb = (*server)->ProcessRequestsEnd(std::move(*b));
    EXPECT(ra.size() == 3 && rb.size() == 3);
    EXPECT(ra[0].ok() && *ra[0] == *response && !ra[1].ok() && ra[1].status().code() == pir::StatusCode::kInvalidArgument);
    EXPECT(rb[2].ok() && *rb[2] == *response);
    // a pending call that is dropped without End (early return in the caller) is ended by the destructor, and a
    // moved-from handle no longer owns it
    {
      auto c = (*server)->ProcessRequestsBegin(reqs);
      EXPECT(c.ok() && (*c)->valid());
      pir::PIRServer::PendingRequests moved(std::move(**c));
      EXPECT(moved.valid() && !(*c)->valid());
    }
    auto again = (*server)->ProcessRequests(reqs);
    EXPECT(again.size() == 3 && again[0].ok() && *again[0] == *response);
  }

  // selection vector of the wrong size -> InvalidArgument (database.cpp:297-300)
  std::vector<pir::Ciphertext> sv(params->DimensionsSum() + 1, pir::Ciphertext((*db)->CtWords(), 0));
  auto mul = (*db)->multiply(sv);
  EXPECT(!mul.ok() && mul.status().code() == pir::StatusCode::kInvalidArgument);
  // expansion of more items than the ring degree -> InvalidArgument (server.cpp:111-114)
  auto exp = (*server)->oblivious_expansion(pir::Ciphertext((*db)->CtWords(), 1), params->poly_modulus_degree + 1);
  EXPECT(!exp.ok() && exp.status().code() == pir::StatusCode::kInvalidArgument);
  std::puts("facade_test OK");
  return 0;
}
