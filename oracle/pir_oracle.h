/*
 * pir_oracle.h -- CPU restatement of the OpenMined/PIR server query path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the CPU baseline.  The product path (pir_amd/) never
 * links, imports or falls back to this code.
 *
 * PARITY STATUS: "parity unpinned" at ciphertext-bit level.  The arithmetic of
 * the reference lives in Microsoft SEAL 3.5.6 (reference pir/deps.bzl:64-71),
 * which is absent from /root/reference and from this image, and the reference's
 * tests hold no ciphertext golden vectors.  The restatement is pinned instead
 * on every plaintext-level known answer of the reference's own tests
 * (server_test.cpp:291-305, :333-339, :376-383, :423-428, ...; see
 * tests/test_oracle_known_answers.py) driven through the CPU client in
 * oracle/client.py.
 *
 * Layout conventions (SEAL's in-memory layout, cf. reference server.cpp:98,
 * ct_reencoder.cpp:61):
 *   ciphertext  : uint64_t[2][k][N]      (poly, residue, coefficient)
 *   NTT plaintext: uint64_t[k][N]
 *   Galois key  : uint64_t[k][2][k+1][N] (digit, component, key-level residue; NTT form)
 * Moduli: q[0..k-1] are the data primes, q[k] is the key-switching special prime.
 */
#ifndef PIR_ORACLE_H_
#define PIR_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXK 8 /* max number of data primes */
#define ORC_MAXDIMS 8

typedef struct orc_ctx orc_ctx;

/* status codes = absl::StatusCode values used by the reference */
#define ORC_OK 0
#define ORC_INVALID_ARGUMENT 3
#define ORC_INTERNAL 13

/* moduli: k data primes followed by the special prime (0 = no special prime,
 * key switching unavailable).  All primes must be == 1 mod 2N and < 2^61. */
orc_ctx* orc_create(uint32_t N, uint32_t k, const uint64_t* moduli, uint64_t t);
void orc_destroy(orc_ctx* c);

uint32_t orc_N(const orc_ctx* c);
uint32_t orc_k(const orc_ctx* c);
uint64_t orc_modulus(const orc_ctx* c, uint32_t i); /* i == k -> special */
uint64_t orc_plain_modulus(const orc_ctx* c);
uint64_t orc_psi(const orc_ctx* c, uint32_t i); /* minimal primitive 2N-th root */

/* number theory helpers */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q); /* q prime */
int orc_is_prime(uint64_t n);
uint64_t orc_minimal_primitive_root(uint64_t two_n, uint64_t q);

/* negacyclic NTT, natural order in, bit-reversed order out. */
void orc_ntt_fwd(const orc_ctx* c, uint32_t mod_idx, uint64_t* poly);
void orc_ntt_inv(const orc_ctx* c, uint32_t mod_idx, uint64_t* poly);
void orc_ct_ntt_fwd(const orc_ctx* c, uint64_t* ct); /* Evaluator::transform_to_ntt_inplace(ct) */
void orc_ct_ntt_inv(const orc_ctx* c, uint64_t* ct); /* Evaluator::transform_from_ntt_inplace */

/* dyadic product / sum of two length-N residue vectors mod q[mod_idx] */
void orc_dyadic_mul(const orc_ctx* c, uint32_t mod_idx, const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_poly_add(const orc_ctx* c, uint32_t mod_idx, const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_poly_sub(const orc_ctx* c, uint32_t mod_idx, const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_poly_neg(const orc_ctx* c, uint32_t mod_idx, const uint64_t* a, uint64_t* out);

/* SEAL GaloisTool::apply_galois on one residue polynomial (coefficient form). */
void orc_apply_galois_poly(const orc_ctx* c, uint32_t mod_idx, const uint64_t* in, uint32_t galois_elt, uint64_t* out);
/* SEAL util::negacyclic_shift_poly_coeffmod */
void orc_negacyclic_shift_poly(const orc_ctx* c, uint32_t mod_idx, const uint64_t* in, uint32_t shift, uint64_t* out);

/* SEAL RNSTool::divide_and_round_q_last (used by key switching and by the
 * client's pk-encryption): in = [k+1][N] coefficient form over q[0..k];
 * out = [k][N] = round(in / q[k]) mod q[j]. */
void orc_divide_round_special(const orc_ctx* c, const uint64_t* in, uint64_t* out);

/* reference server.cpp:67-76 (Evaluator::apply_galois_inplace). ct coefficient form. */
int orc_apply_galois_ct(const orc_ctx* c, uint64_t* ct, uint32_t galois_elt, const uint64_t* galois_key);
/* reference server.cpp:78-103 */
void orc_multiply_inverse_power_of_x(const orc_ctx* c, const uint64_t* ct, uint32_t k, uint64_t* out);
/* Evaluator::add_inplace */
void orc_ct_add_inplace(const orc_ctx* c, uint64_t* a, const uint64_t* b);

/* galois_keys[j] = key for element (N >> j) + 1, j < log2(N); NULL where absent. */
/* reference server.cpp:105-146; out holds num_items ciphertexts. */
int orc_oblivious_expansion(const orc_ctx* c, const uint64_t* ct, uint32_t num_items,
                            const uint64_t* const* galois_keys, uint64_t* out);
/* reference server.cpp:148-171 */
int orc_oblivious_expansion_multi(const orc_ctx* c, const uint64_t* cts, uint32_t num_cts, uint64_t total_items,
                                  const uint64_t* const* galois_keys, uint64_t* out);

/* Evaluator::transform_to_ntt_inplace(Plaintext, first_parms_id): coeffs[ncoeff] (< t) -> out[k][N] */
void orc_plain_lift_ntt(const orc_ctx* c, const uint64_t* coeffs, uint32_t ncoeff, uint64_t* out);
/* Evaluator::multiply_plain on NTT operands: out = ct (.) pt */
void orc_multiply_plain_ntt(const orc_ctx* c, const uint64_t* ct_ntt, const uint64_t* pt_ntt, uint64_t* out);

/* reference ct_reencoder.cpp */
uint32_t orc_bits_per_coeff(uint64_t t); /* (uint32) log2(t) */
uint32_t orc_expansion_ratio(const orc_ctx* c);
void orc_reencode(const orc_ctx* c, const uint64_t* ct, uint64_t* pts /* [2*ER][N] */);
void orc_redecode(const orc_ctx* c, const uint64_t* pts, const uint32_t* pt_ncoeff /* may be NULL = N */,
                  uint64_t* ct);

/* reference database.cpp:118-316.  db_ntt: P plaintexts [k][N] in NTT form.
 * sv: dim_sum ciphertexts, mutated in place exactly as the reference mutates its
 * selection vector (database.cpp:190,222); sv_is_ntt: dim_sum flags (in/out).
 * out: (2*ER)^(nd-1) ciphertexts, coefficient form. */
int orc_db_multiply(const orc_ctx* c, const uint64_t* db_ntt, uint64_t P, const uint32_t* dims, uint32_t nd,
                    uint64_t* sv, uint8_t* sv_is_ntt, uint64_t sv_count, uint64_t* out, uint64_t* out_count);
uint64_t orc_reply_ct_count(const orc_ctx* c, uint32_t nd);

/* reference server.cpp:173-195 minus (de)serialisation: expand + multiply. */
int orc_process_query(const orc_ctx* c, const uint64_t* db_ntt, uint64_t P, const uint32_t* dims, uint32_t nd,
                      const uint64_t* query_cts, uint32_t num_query_cts, const uint64_t* const* galois_keys,
                      uint64_t* out, uint64_t* out_count);

/* reference string_encoder.cpp */
uint64_t orc_items_per_plaintext(uint32_t N, uint32_t bits_per_coeff, uint64_t item_size);
uint64_t orc_max_bytes_per_plaintext(uint32_t N, uint32_t bits_per_coeff);
int orc_string_encode(const uint8_t* bytes, uint64_t nbytes, uint32_t bits_per_coeff, uint32_t N, uint64_t* coeffs,
                      uint32_t* num_coeff);
int orc_string_decode(const uint64_t* coeffs, uint32_t coeff_count, uint32_t bits_per_coeff, uint64_t length,
                      uint64_t byte_offset, uint8_t* out);

/* reference database.cpp:84-110: raw items -> NTT plaintexts. */
int orc_db_encode(const orc_ctx* c, const uint8_t* items, uint64_t num_items, uint64_t bytes_per_item,
                  uint64_t items_per_pt, uint32_t bits_per_coeff, uint64_t* db_ntt /* num_pt x [k][N] */,
                  uint64_t num_pt);

/* reference database.cpp:318-342 */
void orc_calculate_dimensions(uint32_t db_size, uint32_t num_dimensions, uint32_t* out);
void orc_calculate_indices(uint32_t index, uint32_t items_per_pt, const uint32_t* dims, uint32_t nd, uint32_t* out);
uint64_t orc_calculate_item_offset(uint32_t index, uint32_t items_per_pt, uint32_t bytes_per_item);

/* reference utils.cpp / utils.h */
uint32_t orc_ceil_log2(uint32_t v);
uint32_t orc_log2(uint32_t v);
uint64_t orc_next_power_two(uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* PIR_ORACLE_H_ */
