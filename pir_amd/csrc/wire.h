// wire.h -- pir/proto wire format + SEAL 3.5.6 object codec (host side).
// Codec test hooks (not part of the product ABI in include/pirgpu.h).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
void pirgpu_wire_parms_id(uint32_t N, const uint64_t* moduli, size_t n_moduli, uint64_t t, uint64_t out[4]);
void pirgpu_wire_blake2b(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen);
void pirgpu_wire_blake2xb(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen, const uint8_t* key,
                          size_t keylen);
void pirgpu_wire_sample_poly_uniform(const uint8_t* seed, const uint64_t* moduli, uint32_t n_moduli, uint32_t N,
                                     uint64_t* out);
struct pirgpu_params;
int pirgpu_wire_load_kswitch_key(const struct pirgpu_params* params, const uint8_t* blob, size_t len, uint64_t index,
                                 uint64_t* out);
// Device-free validation of a serialized pir.Request (same checks as pirgpu_process_request).
struct pirgpu_params;
int pirgpu_wire_validate_request(const struct pirgpu_params* params, const uint8_t* request, size_t request_len,
                                 uint32_t* n_queries);
// A pinned key set is never evicted or released (one pin per request in flight that uses it; slot handles as handed out
// by pirgpu_keyset_lookup / _claim).
struct pirgpu_ctx;
int pirgpu_keyset_pin(struct pirgpu_ctx* ctx, uint32_t slot);
int pirgpu_keyset_unpin(struct pirgpu_ctx* ctx, uint32_t slot);
// The host copy of the key object resident key set `slot` was installed from (0 if the slot is empty).  The pointer stays
// valid while the slot is pinned (no eviction, no reinstall): the wire layer
// compares it with a request's bytes on worker threads without taking the context's lock.
size_t pirgpu_keyset_blob(struct pirgpu_ctx* ctx, uint32_t slot, const uint8_t** blob);
// Slot pirgpu_query_use_keyset last selected, as a handle carrying the generation it was selected with.
uint32_t pirgpu_current_keyset(struct pirgpu_ctx* ctx);
// The same selection as the raw (slot index, generation) pair, saved and put back UNCHECKED: the wire layer serves a lone
// request with its client's set selected and restores the direct API's selection exactly as it was -- one that had gone
// stale meanwhile stays stale (its next use fails with FailedPrecondition) instead of being re-validated against the
// slot's next tenant.
void pirgpu_keyset_selection_get(struct pirgpu_ctx* ctx, uint32_t sel[2]);
void pirgpu_keyset_selection_set(struct pirgpu_ctx* ctx, const uint32_t sel[2]);
// Drops the wire layer's per-context state (called by pirgpu_destroy).
void pirgpu_wire_forget(struct pirgpu_ctx* ctx);
// Request-level critical section (recursive with the per-call lock of the ABI entry points): the wire layer holds it
// while it stages + queues a window, installs a client's keys, or serves a lone query; NOT while a queued window runs.
void pirgpu_request_lock(struct pirgpu_ctx* ctx);
void pirgpu_request_unlock(struct pirgpu_ctx* ctx);
// pirgpu_query_fetch in two halves (both downloads queued at once): *first_part ciphertexts land first (fetch_wait(0)),
// the rest with fetch_wait(1); `reply` is pinned host memory.  Lets the caller serialise one half under the other's transfer.
int pirgpu_query_fetch_begin(struct pirgpu_ctx* ctx, uint64_t* reply, uint64_t cap_cts, uint64_t* count, uint64_t* first_part);
int pirgpu_query_fetch_wait(struct pirgpu_ctx* ctx, int part);
// Queries in flight as last set with pirgpu_set_concurrency (1 by default).
uint32_t pirgpu_get_concurrency(struct pirgpu_ctx* ctx);
#ifdef __cplusplus
}
#endif
