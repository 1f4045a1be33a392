// client_facade_test.cpp -- drives the C++ mirror of PIRClient (pir_amd/csrc/pir_client_facade.h) the way the
// reference's client_test.cpp drives the real class.  CPU only.
//   usage: client_facade_test <params.txt>          (parameters written by tests/test_cpp_facade.py)
//   usage: client_facade_test <params.txt> <db.bin> (GPU): C++ round trip client -> PIRServer -> client
#include <cstdio>
#include <fstream>
#include <iterator>

#include "../../pir_amd/csrc/pir_client_facade.h"

#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                              \
    }                                                                        \
  } while (0)

static std::shared_ptr<pir::PIRParameters> read_params(const char* path) {
  auto params = std::make_shared<pir::PIRParameters>();
  std::ifstream f(path);
  size_t nmod, ndim;
  f >> params->poly_modulus_degree >> params->plain_modulus >> params->num_items >> params->num_pt >>
      params->bytes_per_item >> params->items_per_plaintext >> params->bits_per_coeff >> nmod >> ndim;
  params->coeff_modulus.resize(nmod);
  for (auto& q : params->coeff_modulus) f >> q;
  params->dimensions.resize(ndim);
  for (auto& d : params->dimensions) f >> d;
  return params;
}

static uint64_t next_power_two(uint64_t n) {  // utils.h:29-37
  uint64_t m = 1;
  while (m < n) m <<= 1;
  return m;
}

int main(int argc, char** argv) {
  if (argc != 2 && argc != 3) return 2;
  auto params = read_params(argv[1]);
  const uint64_t t = params->plain_modulus;
  auto client = pir::PIRClient::Create(params, "cpp-client-test");
  if (!client.ok()) {
    std::fprintf(stderr, "Create failed: %s\n", client.status().message().c_str());
    return 1;
  }

  if (argc == 2) {
    // ---- client_test.cpp:95-127 (TestCreateRequestD2): 82 items, dims [10, 9], index 42 -> row 4, col 6
    EXPECT(params->dimensions.size() == 2 && params->dimensions[0] == 10 && params->dimensions[1] == 9);
    auto query = (*client)->CreateQueryFor(42);
    EXPECT(query.ok() && query->size() == 1);
    auto pt = (*client)->Decrypt((*query)[0]);
    EXPECT(pt.ok());
    const uint64_t m = next_power_two(10 + 9);
    for (size_t i = 0; i < pt->size(); ++i) {
      if (i == 4 || i == 10 + 6)
        EXPECT(((*pt)[i] * m) % t == 1);
      else
        EXPECT((*pt)[i] == 0);
    }
    // client_test.cpp:269-272 (TestCreateRequest_InvalidIndex)
    auto bad = (*client)->CreateRequest({params->num_items + 1});
    EXPECT(!bad.ok() && bad.status().code() == pir::StatusCode::kInvalidArgument);
    // Encryptor / Decryptor round trip
    std::vector<uint64_t> msg = {1, 2, t - 1, 0, 77};
    auto ct = (*client)->Encrypt(msg);
    EXPECT(ct.ok());
    auto back = (*client)->Decrypt(*ct);
    EXPECT(back.ok());
    for (size_t i = 0; i < msg.size(); ++i) EXPECT((*back)[i] == msg[i]);
    // request carries one query plus keys; a reply with the wrong ciphertext count is rejected (client.cpp:229-232)
    auto req = (*client)->CreateRequest({5});
    EXPECT(req.ok() && req->size() > 1000000);
    auto resp = (*client)->ProcessResponse({5}, std::string());
    EXPECT(!resp.ok() && resp.status().code() == pir::StatusCode::kInvalidArgument);   // 1 index, 0 replies
    std::puts("client_facade_test OK");
    return 0;
  }

  // ---- correctness_test.cpp:95-113 in C++ only: PIRClient -> PIRServer (GPU) -> PIRClient
  std::ifstream f(argv[2], std::ios::binary);
  const std::string raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::vector<std::string> rawdb(params->num_items);
  for (size_t i = 0; i < rawdb.size(); ++i) rawdb[i] = raw.substr(i * params->bytes_per_item, params->bytes_per_item);
  auto db = pir::PIRDatabase::Create(rawdb, params);
  EXPECT(db.ok());
  auto server = pir::PIRServer::Create(*db, params);
  EXPECT(server.ok());
  const std::vector<std::size_t> indexes = {0, params->num_items / 2, params->num_items - 1};
  auto request = (*client)->CreateRequest(indexes);
  EXPECT(request.ok());
  auto response = (*server)->ProcessRequest(*request);
  if (!response.ok()) {
    std::fprintf(stderr, "ProcessRequest failed: %s\n", response.status().message().c_str());
    return 1;
  }
  auto items = (*client)->ProcessResponse(indexes, *response);
  EXPECT(items.ok() && items->size() == indexes.size());
  for (size_t i = 0; i < indexes.size(); ++i) EXPECT((*items)[i] == rawdb[indexes[i]]);
  // two clients, served together (pirgpu_process_requests): each decodes its own items; a malformed request in
  // between fails alone
  auto client2 = pir::PIRClient::Create(params);
  EXPECT(client2.ok());
  const std::vector<std::size_t> idx2 = {params->num_items / 3};
  auto request2 = (*client2)->CreateRequest(idx2);
  EXPECT(request2.ok());
  auto both = (*server)->ProcessRequests({*request, std::string("\x12\x03zzz", 5), *request2});
  EXPECT(both.size() == 3 && both[0].ok() && !both[1].ok() && both[2].ok());
  EXPECT(both[1].status().code() == pir::StatusCode::kInvalidArgument);
  EXPECT(*both[0] == *response);                       // the server is deterministic: together == alone
  auto items2 = (*client2)->ProcessResponse(idx2, *both[2]);
  EXPECT(items2.ok() && items2->size() == 1 && (*items2)[0] == rawdb[idx2[0]]);
  std::puts("client_facade_test round trip OK");
  return 0;
}
