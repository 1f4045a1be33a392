/* pirclient.h -- C ABI of libpirclient.so: the CPU-side PIR client.
 *
 * The client stays on the CPU in the reference (north_star: "PIRClient ... stay on CPU"); this library
 * is its counterpart for libpirgpu's server path so that the repository is a full round trip:
 * key generation, query creation, reply decryption and recursive decode.  Pure host code (g++, no HIP,
 * no device): it links nothing from libpirgpu.so and runs on machines without a GPU.
 *
 * Each entry point names the reference interface it replaces (file:line under pir/cpp/).
 * Residue arrays use SEAL's in-memory layout, identical to pirgpu.h:
 *   ciphertext  [2][k][N] uint64 (coefficient form, data level)
 *   galois key  [k][2][k+1][N] uint64 (NTT form, key level; one RLWE sample per RNS digit)
 *   plaintext   [N] uint64 coefficients < plain_modulus
 * Status codes are the pirgpu.h ones (numeric absl::StatusCode).
 *
 * Randomness: a BLAKE2b counter-mode generator keyed from getrandom(2), or from a caller-supplied seed
 * (deterministic, for tests).  SEAL's own Blake2xb stream is NOT reproduced, so keys are always sent
 * fully expanded (never seed-compressed): a request built here is what server_test.cpp builds with
 * galois_keys_local + SaveRequest.
 */
#ifndef PIRCLIENT_H_
#define PIRCLIENT_H_

#include <stddef.h>
#include <stdint.h>

#include "pirgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pirclient pirclient;

/* PIRClient::Create(params) + initialize() -- client.cpp:38-67: key generation (secret, public, Galois keys
 * for generate_galois_elts(N), relinearisation key) and the serialized key fields of the request.
 * seed == NULL: keyed from the OS; otherwise deterministic in (seed, seed_len).
 * shard_begin/shard_end and device in `params` are ignored. */
int pirclient_create(const pirgpu_params* params, const uint8_t* seed, size_t seed_len, pirclient** out);
void pirclient_destroy(pirclient* c);
const char* pirclient_last_error(const pirclient* c);
/* message of a failed pirclient_create (thread-local) */
const char* pirclient_create_error(void);

/* PIRClient::CreateRequest(indexes) -- client.cpp:80-90: serialized pir.Request (payload.proto:27-36) holding
 * one Ciphertexts per index, the GaloisKeys and the RelinKeys.  *request is malloc'd; release with pirclient_free. */
int pirclient_create_request(pirclient* c, const uint64_t* indexes, size_t n_indexes, uint8_t** request,
                             size_t* request_len);
/* PIRClient::ProcessResponse(indexes, response) -- client.cpp:160-185: serialized pir.Response ->
 * n_indexes items of bytes_per_item bytes each, written back to back into items_out. */
int pirclient_process_response(pirclient* c, const uint64_t* indexes, size_t n_indexes, const uint8_t* response,
                               size_t response_len, uint8_t* items_out, size_t items_cap);
/* PIRClient::ProcessResponseInteger(response) -- client.cpp:146-158 (IntegerEncoder::decode_int64 per reply). */
int pirclient_process_response_integer(pirclient* c, const uint8_t* response, size_t response_len, int64_t* out,
                                       size_t out_cap, size_t* n_out);
void pirclient_free(void* p);
/* SaveRequest(queries, galois_keys, relin_keys, request) -- serialization.cpp:44-73 -- with this client's keys: a
 * serialized pir.Request around n_queries caller-supplied query ciphertext sets (residues [n_queries][query_ct_count]
 * [2][k][N], e.g. from pirclient_create_query), so that the SAME ciphertexts can go through the residue-level and the
 * wire-level server entry points.  *request is malloc'd; release with pirclient_free. */
int pirclient_save_request(pirclient* c, const uint64_t* queries, size_t n_queries, uint8_t** request, size_t* request_len);
/* LoadCiphertexts over every Response.reply -- serialization.cpp:32-42: serialized pir.Response -> residues
 * [n_replies][reply_ct_count][2][k][N] (cap_replies = room in replies_out, in replies). */
int pirclient_load_response(pirclient* c, const uint8_t* response, size_t response_len, uint64_t* replies_out,
                            size_t cap_replies, size_t* n_replies);
/* Key fields of the requests: seed-compressed (default; what SEAL 3.5.6's Serializable<GaloisKeys> /
 * Serializable<RelinKeys> of PIRClient::initialize produce, client.cpp:47-54: every key sample carries c0 and the
 * 64-byte seed its uniform half is re-sampled from) or, with enabled == 0, fully expanded objects. */
int pirclient_set_seeded_keys(pirclient* c, int enabled);

/* ---- residue-level halves of the same calls (what sits between SEAL objects in the reference) ---- */

/* number of query ciphertexts: dim_sum / N + 1 (client.cpp:110) */
uint32_t pirclient_query_ct_count(const pirclient* c);
/* PIRClient::createQueryFor -- client.cpp:92-144.  query_out: [query_ct_count][2][k][N]. */
int pirclient_create_query(pirclient* c, uint64_t index, uint64_t* query_out, size_t cap_cts, uint32_t* n_cts);
/* The serialized-once keys of initialize() as residues: KSwitchKey for Galois element `elt`
 * (must be one of generate_galois_elts(N)); key_out [k][2][k+1][N]. */
int pirclient_galois_key(const pirclient* c, uint32_t elt, uint64_t* key_out);
/* PIRClient::ProcessReply -- client.cpp:187-255 (ProcessReplyCiphertextDecomp: decrypt, CiphertextReencoder::Decode,
 * repeat once per dimension).  reply [n_cts][2][k][N] -> plaintext_out [N]. */
int pirclient_process_reply(pirclient* c, const uint64_t* reply, size_t n_cts, uint64_t* plaintext_out);
/* expected reply size: (2 * ExpansionRatio)^(d-1) (client.cpp:224-226) */
uint64_t pirclient_reply_ct_count(const pirclient* c);

/* ---- the SEAL objects PIRClientTest reaches through friend access (client_test.cpp:55-58) ---- */

/* seal::Encryptor::encrypt (public key).  plaintext [n_coeffs <= N] -> ct_out [2][k][N]. */
int pirclient_encrypt(pirclient* c, const uint64_t* plaintext, size_t n_coeffs, uint64_t* ct_out);
/* seal::Decryptor::decrypt.  ct [2][k][N] -> plaintext_out [N]. */
int pirclient_decrypt(pirclient* c, const uint64_t* ct, uint64_t* plaintext_out);
/* seal::Decryptor::invariant_noise_budget in bits (0 = decryption no longer reliable). */
int pirclient_noise_budget(pirclient* c, const uint64_t* ct, int* bits);
/* CiphertextReencoder::Encode -- ct_reencoder.cpp:40-73: ct -> [2*ExpansionRatio][N] plaintexts. */
int pirclient_reencode(const pirclient* c, const uint64_t* ct, uint64_t* plaintexts_out, size_t cap_pts,
                       uint32_t* n_pts);
/* StringEncoder::decode -- string_encoder.cpp:124-163 with the context's bits_per_coeff. */
int pirclient_string_decode(const pirclient* c, const uint64_t* plaintext, size_t length, size_t byte_offset,
                            uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PIRCLIENT_H_ */
