// link stubs for the device-side entry points wire.cpp references (never called by the validation path)
#include "../include/pirgpu.h"
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
int pirgpu_keys_blob_matches(pirgpu_ctx*, const uint8_t*, size_t) { return 0; }
void pirgpu_keys_blob_set(pirgpu_ctx*, const uint8_t*, size_t) {}
uint32_t pirgpu_get_concurrency(pirgpu_ctx*) { return 1; }
}
