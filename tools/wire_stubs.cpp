// link stubs for the device-side entry points wire.cpp references (never called by the validation path)
#include "../include/pirgpu.h"
#include "../pir_amd/csrc/wire.h"
extern "C" {
int pirgpu_get_params(const pirgpu_ctx*, pirgpu_params*) { return 13; }
uint64_t pirgpu_reply_ct_count(const pirgpu_ctx*) { return 0; }
int pirgpu_clear_galois_keys(pirgpu_ctx*) { return 13; }
int pirgpu_set_galois_key(pirgpu_ctx*, uint32_t, const uint64_t*) { return 13; }
const char* pirgpu_last_error(const pirgpu_ctx*) { return ""; }
void pirgpu_set_error(pirgpu_ctx*, const char*) {}
int pirgpu_process_query(pirgpu_ctx*, const uint64_t*, uint32_t, uint64_t*, uint64_t, uint64_t*) { return 13; }
int pirgpu_set_concurrency(pirgpu_ctx*, uint32_t) { return 13; }
int pirgpu_batch_stage(pirgpu_ctx*, const uint64_t*, uint32_t, uint32_t) { return 13; }
int pirgpu_batch_run(pirgpu_ctx*) { return 13; }
int pirgpu_batch_fetch(pirgpu_ctx*, uint64_t*, uint64_t, uint64_t*) { return 13; }
int pirgpu_batch_set_host_replies(pirgpu_ctx*, uint64_t*, uint64_t) { return 13; }
int pirgpu_batch_next_host_replies(pirgpu_ctx*, uint32_t*) { return 13; }
int pirgpu_keyset_lookup(pirgpu_ctx*, const uint8_t*, size_t, int, uint32_t*) { return 13; }
int pirgpu_keyset_verify(pirgpu_ctx*, uint32_t, const uint8_t*, size_t) { return 0; }
size_t pirgpu_keyset_blob(pirgpu_ctx*, uint32_t, const uint8_t** b) { if (b) *b = nullptr; return 0; }
int pirgpu_keyset_claim(pirgpu_ctx*, const uint8_t*, size_t, uint32_t*) { return 13; }
int pirgpu_keyset_release(pirgpu_ctx*, uint32_t) { return 13; }
int pirgpu_keyset_set_key(pirgpu_ctx*, uint32_t, uint32_t, const uint64_t*) { return 13; }
int pirgpu_keyset_set_keys(pirgpu_ctx*, uint32_t, uint32_t, const uint32_t*, const uint64_t* const*) { return 13; }
int pirgpu_keyset_stats(pirgpu_ctx*, uint64_t*) { return 13; }
int pirgpu_query_use_keyset(pirgpu_ctx*, uint32_t) { return 13; }
uint32_t pirgpu_current_keyset(pirgpu_ctx*) { return 0; }
void pirgpu_keyset_selection_get(pirgpu_ctx*, uint32_t sel[2]) { sel[0] = sel[1] = 0; }
void pirgpu_keyset_selection_set(pirgpu_ctx*, const uint32_t*) {}
int pirgpu_batch_set_keysets(pirgpu_ctx*, const uint32_t*, uint32_t) { return 13; }
int pirgpu_query_stage(pirgpu_ctx*, const uint64_t*, uint32_t) { return 13; }
int pirgpu_query_stage_async(pirgpu_ctx*, const uint64_t*, uint32_t) { return 13; }
int pirgpu_query_run(pirgpu_ctx*) { return 13; }
int pirgpu_sync(pirgpu_ctx*) { return 13; }
int pirgpu_query_fetch(pirgpu_ctx*, uint64_t*, uint64_t, uint64_t*) { return 13; }
int pirgpu_query_fetch_begin(pirgpu_ctx*, uint64_t*, uint64_t, uint64_t*, uint64_t*) { return 13; }
int pirgpu_query_fetch_wait(pirgpu_ctx*, int) { return 13; }
uint64_t* pirgpu_host_query_buffer(pirgpu_ctx*, uint32_t) { return nullptr; }
uint64_t* pirgpu_host_reply_buffer(pirgpu_ctx*, uint32_t) { return nullptr; }
int pirgpu_keyset_pin(pirgpu_ctx*, uint32_t) { return 13; }
int pirgpu_keyset_unpin(pirgpu_ctx*, uint32_t) { return 13; }
int pirgpu_batch_stage_async(pirgpu_ctx*, const uint64_t*, uint32_t, uint32_t) { return 13; }
int pirgpu_batch_unstage(pirgpu_ctx*) { return 13; }
int pirgpu_batch_select(pirgpu_ctx*, uint32_t) { return 13; }
void pirgpu_request_lock(pirgpu_ctx*) {}
void pirgpu_request_unlock(pirgpu_ctx*) {}
uint32_t pirgpu_get_concurrency(pirgpu_ctx*) { return 1; }
}
