/*
 * pirgpu.h -- C ABI of the MI355X-native PIR server query path (libpirgpu.so).
 *
 * The reference (OpenMined/PIR) has no FFI seam: its boundary is the C++ class
 * API of PIRServer / PIRDatabase over Microsoft SEAL objects.  Every entry
 * point below replaces one reference interface, cited as file:line relative to
 * /root/reference.  The host-side C++ mirror of the reference classes
 * (pir_amd/csrc/pir_facade.h) and the ctypes binding (pir_amd/capi.py) are thin
 * wrappers over exactly these symbols; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer, the
 *     library owns all device memory;
 *   - residue arrays use SEAL's in-memory layout (reference server.cpp:98,
 *     ct_reencoder.cpp:61):
 *        ciphertext   uint64_t[2][k][N]       (poly, residue, coefficient)
 *        Galois key   uint64_t[k][2][k+1][N]  (digit, component, key-level residue), NTT form
 *   - every function returns 0 on success or the numeric absl::StatusCode the
 *     reference would have returned (3 InvalidArgument, 9 FailedPrecondition,
 *     12 Unimplemented, 13 Internal); pirgpu_last_error() gives the message;
 *   - one context drives one GPU; every entry point takes the context's lock
 *     for its own duration, and pirgpu_process_request(s) holds it while a window of
 *     requests is served (key lookup / installation + every query); requests that
 *     arrive meanwhile are queued and served together afterwards.  Sequences the
 *     CALLER composes out of several calls (set keys + query, stage / run / fetch)
 *     are only atomic if the caller serialises them itself;
 *   - every result is a canonical residue in [0, q_j) and is bit-identical to
 *     the reference's SEAL CPU path on the same inputs.
 */
#ifndef PIRGPU_H_
#define PIRGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PIRGPU_MAX_PRIMES 8
#define PIRGPU_MAX_DIMS 8

#define PIRGPU_OK 0
#define PIRGPU_INVALID_ARGUMENT 3
#define PIRGPU_FAILED_PRECONDITION 9
#define PIRGPU_UNIMPLEMENTED 12
#define PIRGPU_INTERNAL 13

typedef struct pirgpu_ctx pirgpu_ctx;

/* What PIRContext + PIRParameters carry (reference context.h:36-84,
 * pir/proto/payload.proto:45-69), flattened. */
typedef struct pirgpu_params {
  uint32_t poly_modulus_degree;            /* N: 2048, 4096, 8192 or 16384 */
  uint32_t num_data_primes;                /* k: ciphertext level (first_context_data) */
  uint64_t coeff_modulus[PIRGPU_MAX_PRIMES]; /* q_0..q_{k-1} */
  uint64_t special_prime;                  /* key-switching prime p (last of SEAL's coeff_modulus) */
  uint64_t plain_modulus;                  /* t */
  uint32_t num_dimensions;                 /* d */
  uint32_t dimensions[PIRGPU_MAX_DIMS];    /* PIRParameters.dimensions */
  uint64_t num_pt;                         /* PIRParameters.num_pt */
  uint64_t num_items;                      /* PIRParameters.num_items (0 if coefficient-loaded) */
  uint32_t bytes_per_item;                 /* PIRParameters.bytes_per_item */
  uint32_t items_per_plaintext;            /* PIRParameters.items_per_plaintext */
  uint32_t bits_per_coeff;                 /* PIRParameters.bits_per_coeff (0 = floor(log2 t)) */
  uint32_t use_ciphertext_multiplication;  /* must be 0: CT x CT mode is Unimplemented */
  int32_t device;                          /* HIP device ordinal */
  /* Row sharding for multi-GPU (not in the reference): this context holds the
   * top-level indices [shard_begin, shard_end) of dimension 0 and produces the
   * partial reply for them; 0,0 = the whole database (an empty shard is any begin == end != 0,
   * e.g. dimensions[0],dimensions[0]; its partial reply is all zero). */
  uint32_t shard_begin;
  uint32_t shard_end;
  /* Slot sharding for multi-GPU (not in the reference; d = 2): this context holds the NTT slots
   * [slot_begin, slot_end) -- multiples of 16 out of the ring's k * N, device order -- of EVERY plaintext and serves the
   * pirgpu_slots_* step only; 0,0 = all slots.  (The base case of PIRDatabase::multiply, reference database.cpp:185-194,
   * is a dyadic product in NTT form: independent per slot.)
   * Memory: AFTER pirgpu_db_finalize(ctx, 1) such a context holds (slot_end - slot_begin) / (k N) of the packed database.
   * WHILE it is being loaded it holds the full u64 staging copy of every plaintext as well (every rank encodes and
   * transforms the whole database, then packs its own slots out of it): the load-time peak per rank is the whole
   * database at 8 bytes per residue + its share of the operand layout, so slot sharding spreads the scan's bytes, not yet a
   * database larger than one GPU's memory (that needs the encode + pack in row blocks). */
  uint32_t slot_begin;
  uint32_t slot_end;
} pirgpu_params;

/* PIRContext::Create + PIRDatabase::Create(params) (reference context.cpp:37-50,
 * database.cpp:40-44): validates the parameters, builds NTT tables on the device. */
int pirgpu_create(const pirgpu_params* params, pirgpu_ctx** out);
void pirgpu_destroy(pirgpu_ctx* ctx);
/* Message of the calling thread's last failed call on this context (falls back to the context's last
 * failure when this thread has none). */
const char* pirgpu_last_error(const pirgpu_ctx* ctx);
/* Records an error message on the context (used by the wire-level layer). */
void pirgpu_set_error(pirgpu_ctx* ctx, const char* message);
/* The parameters the context was created with (shard range resolved). */
int pirgpu_get_params(const pirgpu_ctx* ctx, pirgpu_params* out);
/* last error of a failed pirgpu_create (no context to ask) */
const char* pirgpu_create_error(void);

/* PIRDatabase::populate(vector<string>) (reference database.cpp:84-110,
 * string_encoder.cpp:58-122).  items: num_items x bytes_per_item raw bytes,
 * item-major; must equal params.num_items.  Bit packing, plain lift and forward
 * NTT all run on the device; the encoded database stays resident in HBM. */
int pirgpu_db_load_items(pirgpu_ctx* ctx, const uint8_t* items, uint64_t num_items, uint32_t bytes_per_item);
/* PIRDatabase::populate from already-encoded plaintexts (the IntegerEncoder path,
 * reference database.cpp:60-82): coeffs = n_pt x N coefficients, each < t,
 * zero padded; plaintext indices [first_pt, first_pt + n_pt). */
int pirgpu_db_load_coeffs(pirgpu_ctx* ctx, uint64_t first_pt, uint64_t n_pt, const uint64_t* coeffs);
/* PIRDatabase::size() (reference database.h:97) -- plaintexts loaded so far. */
uint64_t pirgpu_db_size(const pirgpu_ctx* ctx);
/* Optional: bring the scan's operand-layout copy of the database up to date now (otherwise done lazily
 * by the first query after a load).  With release_staging != 0 (d >= 2 only) the u64 staging copy that
 * loads write into is freed afterwards: the database then occupies L/8 of its u64 size (plus tile
 * padding) and cannot be reloaded (FailedPrecondition); pirgpu_db_read_plaintext keeps working.
 * No reference counterpart (the reference keeps one vector<Plaintext>, database.h:126-133). */
int pirgpu_db_finalize(pirgpu_ctx* ctx, int release_staging);
/* SEAL's default build throws logic_error("result ciphertext is transparent") from multiply_plain when a
 * database plaintext is identically zero, which the reference surfaces as InternalError for every query
 * (database.cpp:308-315).  Default (allow == 0): same status.  allow != 0: return the mathematically defined
 * reply instead (what a SEAL built without SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT computes). */
int pirgpu_set_transparent_policy(pirgpu_ctx* ctx, int allow);
/* Row-sharded servers (not in the reference): the reference's transparent-ciphertext failure depends on the WHOLE
 * database, so the ranks exchange their shard's count once after loading (pir_amd.distributed.sync_zero_plaintexts:
 * one all-reduce) and tell their context how many identically-zero plaintexts the other shards hold; every rank then
 * takes the same decision.  pirgpu_check_ready returns what the next query would fail with before any kernel or
 * collective has run: FailedPrecondition (database not fully loaded), Internal (transparent), else 0. */
uint64_t pirgpu_zero_plaintexts(const pirgpu_ctx* ctx);
int pirgpu_set_remote_zero_plaintexts(pirgpu_ctx* ctx, uint64_t count);
int pirgpu_check_ready(pirgpu_ctx* ctx);
/* Test hook: read back one encoded plaintext [k][N] (NTT form) from HBM. */
int pirgpu_db_read_plaintext(pirgpu_ctx* ctx, uint64_t pt_index, uint64_t* out);

/* What SEALDeserialize<GaloisKeys> yields per request (reference server.cpp:46-48):
 * install the key for one Galois element.  Keys stay on the device until cleared.
 * (These two operate on key set slot 0, the default set of every query entry point.) */
int pirgpu_set_galois_key(pirgpu_ctx* ctx, uint32_t galois_elt, const uint64_t* key);
int pirgpu_clear_galois_keys(pirgpu_ctx* ctx);

/* Per-client key sets.  In the reference the Galois keys are locals of one ProcessRequest call (server.cpp:46-48),
 * so requests of different clients are independent; here up to `capacity` clients' keys stay resident in HBM
 * (slots 1..capacity, least recently used evicted; slot 0 is the set above), every query names the slot it is
 * switched with, and the queries of ONE batch group may use different slots (the key-switch kernels index the key
 * of each query's client).
 *   keyset_lookup   finds the resident set that was claimed with exactly these `id` bytes (the wire layer uses the
 *                   serialized GaloisKeys object, any client identifier works); *slot = 0 when not resident.
 *                   verify = 0 accepts a match of length + a sampled fingerprint only; the caller must then confirm
 *                   with keyset_verify (byte-for-byte) before it releases a result computed with that slot.
 *   keyset_claim    empties a free slot -- or the least recently used one, after waiting for the work in flight --
 *                   and tags it with `id`; install the keys with keyset_set_key.  FailedPrecondition when every slot
 *                   belongs to the requests currently being processed together.
 *   keyset_release  empties a slot.
 *   query_use_keyset  slot for pirgpu_process_query / query_run / expand / substitute (default 0).
 *   batch_set_keysets one slot per staged query of the batch (after pirgpu_batch_stage, which resets them to 0).
 *   keyset_stats    [0] resident sets, [1] keys uploaded so far, [2] evictions, [3] capacity.
 * A `slot` is a HANDLE: slot index + the generation of the set living there (0 = the default set).  Once a set has been
 * evicted or released, every entry point that is given its old handle fails with FailedPrecondition ("stale key set
 * handle") instead of switching a query with whichever client's keys moved in -- claim it again.  The same holds for
 * handles the context REMEMBERS: a staged batch (batch_set_keysets) or the query_use_keyset selection whose set has been
 * evicted since fails its next run / expansion with FailedPrecondition.  keyset_claim never evicts a set that requests
 * in flight are pinned to (the wire layer's windows). */
int pirgpu_set_keyset_capacity(pirgpu_ctx* ctx, uint32_t capacity);
int pirgpu_keyset_lookup(pirgpu_ctx* ctx, const uint8_t* id, size_t id_len, int verify, uint32_t* slot);
int pirgpu_keyset_verify(pirgpu_ctx* ctx, uint32_t slot, const uint8_t* id, size_t id_len);
int pirgpu_keyset_claim(pirgpu_ctx* ctx, const uint8_t* id, size_t id_len, uint32_t* slot);
int pirgpu_keyset_release(pirgpu_ctx* ctx, uint32_t slot);
int pirgpu_keyset_set_key(pirgpu_ctx* ctx, uint32_t slot, uint32_t galois_elt, const uint64_t* key);
/* n keys of one set in one call (one wait for all uploads instead of one per key). */
int pirgpu_keyset_set_keys(pirgpu_ctx* ctx, uint32_t slot, uint32_t n, const uint32_t* galois_elts, const uint64_t* const* keys);
int pirgpu_query_use_keyset(pirgpu_ctx* ctx, uint32_t slot);
int pirgpu_batch_set_keysets(pirgpu_ctx* ctx, const uint32_t* slots, uint32_t count);
int pirgpu_keyset_stats(pirgpu_ctx* ctx, uint64_t stats[4]);

/* PIRServer::processQuery minus (de)serialisation (reference server.cpp:173-195):
 * query = nq ciphertexts (coefficient form), reply = reply_count ciphertexts
 * (coefficient form).  reply_capacity is in ciphertexts. */
int pirgpu_process_query(pirgpu_ctx* ctx, const uint64_t* query, uint32_t nq, uint64_t* reply,
                         uint64_t reply_capacity, uint64_t* reply_count);
/* (2 * ExpansionRatio)^(d-1) (reference client.cpp:224-226, ct_reencoder.cpp:29-38) */
uint64_t pirgpu_reply_ct_count(const pirgpu_ctx* ctx);
/* CiphertextReencoder::ExpansionRatio (reference ct_reencoder.cpp:29-38) */
uint32_t pirgpu_expansion_ratio(const pirgpu_ctx* ctx);

/* Device-resident split of pirgpu_process_query for pipelining and measurement:
 * stage = H2D of the query, run = every kernel of the path (asynchronous on the
 * context's stream), fetch = D2H of the reply, sync = wait for the stream. */
int pirgpu_query_stage(pirgpu_ctx* ctx, const uint64_t* query, uint32_t nq);
/* query_stage without the wait: the upload is queued on the context's stream in front of query_run's kernels;
 * `pinned_query` (pirgpu_host_query_buffer or other pinned memory) must stay untouched until query_fetch returns. */
int pirgpu_query_stage_async(pirgpu_ctx* ctx, const uint64_t* pinned_query, uint32_t nq);
int pirgpu_query_run(pirgpu_ctx* ctx);
int pirgpu_query_fetch(pirgpu_ctx* ctx, uint64_t* reply, uint64_t reply_capacity, uint64_t* reply_count);
int pirgpu_sync(pirgpu_ctx* ctx);
/* hipDeviceSynchronize() on the context's device, through the HIP runtime the library itself is linked against: every
 * stream of the device, the library's own included.  (A measurement harness brackets its timed region with this when it
 * has no other handle on the runtime -- bench.py at one GPU, which does not import torch; not in the reference.) */
int pirgpu_device_synchronize(pirgpu_ctx* ctx);

/* Batch mode -- the `for (const auto& query : request.query())` loop of
 * PIRServer::ProcessRequest (reference server.cpp:60-63) with several queries in flight:
 * set_concurrency gives the context n_workers independent working sets (stream +
 * intermediates, ~0.4 GB each at N=4096 / dim_sum=324); batch_stage uploads `count`
 * queries of nq ciphertexts each; batch_run enqueues all of them (asynchronous) in rounds of
 * n_workers queries: groups of up to 8 queries are expanded together (the expansion kernels run
 * over nodes x queries) and, with the int8-MFMA scan (d >= 2), share one pass over the database;
 * consecutive groups alternate between two streams so that scan and expansion overlap
 * (n_workers = 16 keeps both busy); batch_fetch waits and downloads count x reply_ct_count
 * ciphertexts, reply i answering query i. */
int pirgpu_set_concurrency(pirgpu_ctx* ctx, uint32_t n_workers);
int pirgpu_batch_stage(pirgpu_ctx* ctx, const uint64_t* queries, uint32_t nq, uint32_t count);
int pirgpu_batch_run(pirgpu_ctx* ctx);
/* batch_stage without the wait: `pinned_queries` (pirgpu_host_query_buffer, or any pinned memory that stays untouched
 * until the batch has run) goes up in pieces of 8 queries, and each group of a later pirgpu_batch_run waits -- on the
 * device -- only for the pieces that hold its queries.  batch_unstage forgets the staged queries (and with them the
 * references to key sets that pirgpu_batch_set_keysets left).
 * A context has TWO independent sets of batch state (staged queries, replies, their key sets, the host-reply target and
 * its group events, pinned staging): pirgpu_batch_select picks the set the CALLING THREAD's later pirgpu_batch_* /
 * pirgpu_host_*_buffer calls work on (default 0).  Lanes and streams are shared, so batches run from the two sets
 * execute back to back in the order they were queued: one can be parsed, staged and queued while the other is still on
 * the GPU or on its way back to the host (what pirgpu_process_request(s) do with two request windows in flight). */
int pirgpu_batch_stage_async(pirgpu_ctx* ctx, const uint64_t* pinned_queries, uint32_t nq, uint32_t count);
int pirgpu_batch_unstage(pirgpu_ctx* ctx);
int pirgpu_batch_select(pirgpu_ctx* ctx, uint32_t which);
int pirgpu_batch_fetch(pirgpu_ctx* ctx, uint64_t* replies, uint64_t reply_capacity, uint64_t* reply_count);
/* Replies on their way to the host while the batch is still running: with a PINNED host buffer set here (capacity in
 * ciphertexts; NULL switches it off), every group of a batch that fits it downloads its replies on its own stream as soon
 * as they exist; pirgpu_batch_fetch into that same buffer then only waits.  (What the wire layer does with
 * pirgpu_host_reply_buffer: 64 MB of replies per 64 queries no longer cross PCIe after the last kernel.) */
int pirgpu_batch_set_host_replies(pirgpu_ctx* ctx, uint64_t* pinned_host, uint64_t capacity);
/* Blocks until the NEXT group of the batch just run has its replies in that host buffer and reports how many leading
 * queries are complete (*ready; the batch's size once every group has been reported): the caller can serialise replies
 * [previous ready, *ready) while the later groups are still being computed. */
int pirgpu_batch_next_host_replies(pirgpu_ctx* ctx, uint32_t* ready);
/* Multi-GPU, query-parallel expansion (not in the reference).  batch_expand runs
 * only oblivious_expansion + the selector NTT for the staged queries [first, first+count) and writes
 * their selection vectors (count x dim_sum ciphertexts, NTT form, device order) to caller-owned DEVICE
 * memory, e.g. this rank's slice of an RCCL all-gather buffer.  batch_run_selectors then runs
 * PIRDatabase::multiply for `count` queries whose selection vectors are already in DEVICE memory
 * (query i at device_sv + i * dim_sum ciphertexts) on this context's database shard; replies go to
 * the batch reply buffer like pirgpu_batch_run's. */
int pirgpu_batch_expand(pirgpu_ctx* ctx, uint32_t first, uint32_t count, uint64_t* device_dst);
int pirgpu_batch_run_selectors(pirgpu_ctx* ctx, const uint64_t* device_sv, uint32_t count);
/* Row-sharded multi-GPU runs with the PACKED selector exchange (d = 2 and the int8-MFMA scan on every rank;
 * DESIGN.md section 7).  What every rank needs from a query is (a) all its column selectors, in the scan's
 * B-operand layout (digit-packed signed bytes, one buffer per group of <= 8 queries), and (b) only its own rows'
 * dimension-0 selectors (NTT form, u64).  packed_selector_bytes: size of one group buffer (identical on every rank;
 * 0 if the packed exchange does not apply to this context).  batch_expand_packed: oblivious expansion + selector NTT
 * of the staged queries [first, first+count) in groups of 8 (group g = queries first+8g ..); writes group g's packed
 * column selectors to device_packed + g * packed_selector_bytes and, for every destination rank s holding rows
 * [row_cuts[s], row_cuts[s+1]), the block [count][rows of s][2][k][N] of row selectors, blocks back to back in
 * device_rows (= the send buffer of an all-to-all with split sizes count * rows_s * 2kN words).
 * batch_run_packed: PIRDatabase::multiply on this shard for n_ranks * per_rank queries whose packed groups arrive as
 * [source rank][group] (the all-gather of every rank's device_packed) and whose row selectors for THIS shard's rows
 * arrive as [query][my rows][2][k][N] (the all-to-all's receive buffer); replies go to the batch reply buffer,
 * query i = source rank i / per_rank, its query i % per_rank.  Needs pirgpu_set_concurrency(>= 8). */
uint64_t pirgpu_packed_selector_bytes(pirgpu_ctx* ctx);
int pirgpu_batch_expand_packed(pirgpu_ctx* ctx, uint32_t first, uint32_t count, uint8_t* device_packed,
                               uint64_t* device_rows, const uint32_t* row_cuts, uint32_t n_ranks);
int pirgpu_batch_run_packed(pirgpu_ctx* ctx, const uint8_t* device_packed, uint32_t n_ranks, uint32_t per_rank,
                            const uint64_t* device_rows);
/* Waits for the batch and copies its replies into caller-owned DEVICE memory (multi-GPU reduce). */
int pirgpu_batch_reply_copy_to_device(pirgpu_ctx* ctx, uint64_t* device_dst, uint64_t capacity);

/* Pipelined multi-GPU step: the same entry points without any host synchronisation, plus the
 * device-side ordering a caller needs to hang its collectives between them.  The library queues work on its own HIP
 * streams (a main stream, the batch lanes, the workers):
 *   stream_handle   the main stream (a hipStream_t), e.g. for torch.cuda.ExternalStream -- record / wait events on it;
 *   join            the main stream waits, on the device, for everything queued on lanes and workers so far;
 *   join_stream     the same, but it is `stream` (a hipStream_t of the caller) that waits -- the main stream does not,
 *                   so a later fork does not make every lane wait for every other lane's earlier work;
 *   fork            lanes and workers wait for everything the main stream has been made to wait for;
 *   batch_set_reply_buffer  later batches write their replies into the caller's device buffer (capacity in
 *                   ciphertexts; NULL restores the context's own buffer): a collective can read them where they are;
 *   batch_expand_packed_async      = batch_expand_packed, returns once the work is queued (join to get a completion point);
 *   batch_reply_copy_to_device_async  join + copy on the main stream, no wait;
 *   reduce_fixup_device_async      x mod q_j queued on `stream` (a hipStream_t of the caller; NULL = the main stream).
 * pirgpu_batch_run_packed / _run_selectors / _batch_run never wait for the device in either form. */
void* pirgpu_stream_handle(pirgpu_ctx* ctx);
int pirgpu_join(pirgpu_ctx* ctx);
int pirgpu_join_stream(pirgpu_ctx* ctx, void* stream);
int pirgpu_batch_set_reply_buffer(pirgpu_ctx* ctx, uint64_t* device_buf, uint64_t capacity);
int pirgpu_fork(pirgpu_ctx* ctx);
int pirgpu_batch_expand_packed_async(pirgpu_ctx* ctx, uint32_t first, uint32_t count, uint8_t* device_packed,
                                     uint64_t* device_rows, const uint32_t* row_cuts, uint32_t n_ranks);
int pirgpu_batch_reply_copy_to_device_async(pirgpu_ctx* ctx, uint64_t* device_dst, uint64_t capacity);
int pirgpu_reduce_fixup_device_async(pirgpu_ctx* ctx, uint64_t* device_ptr, uint64_t count, void* stream);
/* Row selectors of the packed exchange in 5 bytes per residue instead of 8 (moduli below 2^40: pack40_supported = 1):
 * `words` u64 residues (a multiple of 4) <-> words * 5 / 4 dwords in DEVICE memory, queued on `stream` (NULL: the main
 * stream).  Every 4 words become 5 dwords, so the packed form of a buffer can be cut wherever the word form is cut at a
 * multiple of 4 words -- e.g. into the per-rank pieces of an all-to-all. */
int pirgpu_pack40_supported(pirgpu_ctx* ctx);
int pirgpu_pack40_device_async(pirgpu_ctx* ctx, const uint64_t* words_in, uint32_t* packed, uint64_t words, void* stream);
int pirgpu_unpack40_device_async(pirgpu_ctx* ctx, const uint32_t* packed, uint64_t* words_out, uint64_t words, void* stream);

/* Slot-sharded multi-GPU step (not in the reference; d = 2, int8-MFMA scan).  G ranks hold the
 * slots [slot_cuts[g], slot_cuts[g+1]) of every plaintext (contexts created with slot_begin / slot_end; a context that
 * holds all slots serves the step too, as the G = 1 case).  One step of B queries, B / G per rank:
 *   slots_expand   oblivious expansion + selector NTT (reference server.cpp:105-171, database.cpp:190,222) of the staged
 *                  queries [first, first + count) in groups of 8; the whole NTT-form selection vectors stay in device_sv
 *                  (count x dim_sum ciphertexts, the context's own element type -- only slots_finish reads them) and the
 *                  column selectors are packed into the scan's B-operand layout, cut by destination rank: piece
 *                  [rank r][group g] of device_packed = the slots of rank r of group g, pirgpu_slots_packed_bytes(ctx,
 *                  slots of r) bytes each -- the send buffer of an all-to-all whose split sizes are groups x that;
 *   slots_scan     the base case of PIRDatabase::multiply (database.cpp:185-194,238-247) on this context's slots for
 *                  n_ranks x per_rank queries in ONE launch: device_packed = the all-to-all's receive buffer,
 *                  [source rank][group][my slots of that group]; row sums go to device_rowsums as
 *                  [source rank][query][row][component][my slots] u64 -- the send buffer of the all-to-all back;
 *   slots_finish   for the `count` queries this rank expanded: device_rowsums = that all-to-all's receive buffer,
 *                  [rank h][query][row][component][slots of h]; puts the pieces together and runs the rest of
 *                  PIRDatabase::multiply (inverse NTT, CiphertextReencoder::Encode, upper level, database.cpp:196-254)
 *                  with the row selectors in device_sv; reply i (reply_ct_count ciphertexts) at device_replies + i.
 * All three only queue work on the context's lanes.  after / then (hipStream_t of the caller, NULL = none): the lanes a
 * call uses first wait -- on the device -- for everything queued on `after` so far, and `then` waits for everything the
 * call queued: the caller's collectives are ordered against the step without a host wait and without a barrier over
 * all lanes.  Needs pirgpu_set_concurrency(>= 8).  No rank exchanges row selectors and there is no reduce: every reply
 * is computed whole by the rank that owns its query. */
uint64_t pirgpu_slots_packed_bytes(pirgpu_ctx* ctx, uint32_t slots);
int pirgpu_slots_expand_async(pirgpu_ctx* ctx, uint32_t first, uint32_t count, uint8_t* device_packed, uint64_t* device_sv,
                              const uint32_t* slot_cuts, uint32_t n_ranks, void* after, void* then);
int pirgpu_slots_scan_async(pirgpu_ctx* ctx, const uint8_t* device_packed, uint32_t n_ranks, uint32_t per_rank,
                            uint64_t* device_rowsums, void* after, void* then);
int pirgpu_slots_finish_async(pirgpu_ctx* ctx, const uint64_t* device_rowsums, uint32_t count, const uint64_t* device_sv,
                              const uint32_t* slot_cuts, uint32_t n_ranks, uint64_t* device_replies, void* after,
                              void* then);

/* PIRServer::oblivious_expansion (reference server.cpp:105-146): one ciphertext
 * -> num_items ciphertexts, coefficient form. */
int pirgpu_expand(pirgpu_ctx* ctx, const uint64_t* ct, uint32_t num_items, uint64_t* out);
/* PIRServer::oblivious_expansion, multi-ciphertext overload (reference server.cpp:148-171). */
int pirgpu_expand_multi(pirgpu_ctx* ctx, const uint64_t* cts, uint32_t num_cts, uint64_t total_items,
                        uint64_t* out);
/* PIRServer::substitute_power_x_inplace (reference server.cpp:67-76). */
int pirgpu_substitute_power_x(pirgpu_ctx* ctx, uint64_t* ct, uint32_t power);
/* PIRServer::multiply_inverse_power_of_x (reference server.cpp:78-103). */
int pirgpu_multiply_inverse_power_of_x(pirgpu_ctx* ctx, const uint64_t* ct, uint32_t k, uint64_t* out);
/* PIRDatabase::multiply (reference database.cpp:290-316): selection vector of
 * sv_count ciphertexts in coefficient form -> reply ciphertexts. */
int pirgpu_multiply(pirgpu_ctx* ctx, const uint64_t* selection_vector, uint64_t sv_count, uint64_t* reply,
                    uint64_t reply_capacity, uint64_t* reply_count);

/* Evaluator::transform_to_ntt_inplace / transform_from_ntt_inplace on count
 * ciphertexts (reference database.cpp:190,252) -- test hooks for the NTT kernels.
 * key_level = 0: ciphertext layout [2][k][N] over q_0..q_{k-1};
 * key_level = 1: polys is count x [k+1][N] over q_0..q_{k-1},p. */
int pirgpu_ntt_forward(pirgpu_ctx* ctx, uint64_t* polys, uint64_t count, int key_level);
int pirgpu_ntt_inverse(pirgpu_ctx* ctx, uint64_t* polys, uint64_t count, int key_level);

/* Multi-GPU reduce step (not in the reference): in place, x[i] <- x[i] mod q_j
 * over count ciphertexts whose words hold integer sums of per-shard partial
 * replies (each partial < q_j, so an 8-way sum fits 64 bits). device_ptr is a
 * device pointer (e.g. a torch tensor's data_ptr after the RCCL all-reduce). */
int pirgpu_reduce_fixup_device(pirgpu_ctx* ctx, uint64_t* device_ptr, uint64_t count);
/* Device pointer of the staged reply (valid until the next run) for collectives. */
uint64_t* pirgpu_reply_device_ptr(pirgpu_ctx* ctx);
/* Copies the reply of the last run into caller-owned DEVICE memory (e.g. a torch
 * tensor handed to RCCL); waits for the copy. capacity is in ciphertexts. */
int pirgpu_reply_copy_to_device(pirgpu_ctx* ctx, uint64_t* device_dst, uint64_t capacity);

/* Wire-level entry: PIRServer::ProcessRequest (reference server.cpp:44-65).
 * request = serialized pir.Request (pir/proto/payload.proto:27-36); on success
 * *response points to a malloc'd serialized pir.Response (payload.proto:39-42)
 * that the caller releases with pirgpu_free.  The SEAL 3.5.6 objects inside the bytes fields may be fully
 * expanded or seed-compressed (Serializable<>, what the reference client sends for its keys); relin_keys, when
 * present, are parsed and validated like server.cpp:53-58.  Atomic on the context (see Conventions). */
int pirgpu_process_request(pirgpu_ctx* ctx, const uint8_t* request, size_t request_len, uint8_t** response,
                           size_t* response_len);
/* The same for n independent requests (different clients) served TOGETHER: every client's Galois keys are looked up
 * among / installed into the resident key sets, all queries go through the batch pipeline as one batch (groups of 8
 * expanded together, each query with its own client's keys, one database pass per group), responses[i] answers
 * requests[i].  status[i] = what pirgpu_process_request would have returned for request i (a failing request does not
 * affect the others); the return value is the first non-zero status.  The requests are served in WINDOWS of at most 64
 * queries and capacity / 2 clients (pirgpu_set_keyset_capacity), two windows in flight: the next one is parsed, staged
 * and queued while the previous one's groups are still on the GPU and its replies are being downloaded and serialised.
 * Threads that call pirgpu_process_request(s) concurrently on one context share those two slots: requests that arrive
 * while both are taken are queued and served together by the next thread that gets one.  Host staging buffers the wire
 * layer uses (pinned, one pair per batch set). */
int pirgpu_process_requests(pirgpu_ctx* ctx, uint32_t n, const uint8_t* const* requests, const size_t* request_lens,
                            uint8_t** responses, size_t* response_lens, int* status);
/* Message of request i of the calling thread's last pirgpu_process_requests / _end call ("" if it succeeded). */
const char* pirgpu_request_error(uint32_t i);
/* The same call in two halves, for ONE calling thread that wants two calls in flight (the reference's harness calls
 * ProcessRequest synchronously, benchmark.cpp:71-79; a server loop need not): begin hands the call to a serving thread of
 * the library and returns at once, end waits for it and returns what pirgpu_process_requests would have.  Every array
 * passed to begin (requests, lengths, responses, status) must stay alive and untouched until end returns; responses and
 * status are valid after end.  begin(i + 1) before end(i) lets call i + 1's parsing, staging and queueing run under
 * call i's tail -- what two calling threads get (two request windows in flight), without the caller owning a second
 * thread.  Every begin must be matched by exactly one end, and the context must outlive every call still pending. */
int pirgpu_process_requests_begin(pirgpu_ctx* ctx, uint32_t n, const uint8_t* const* requests, const size_t* request_lens,
                                  uint8_t** responses, size_t* response_lens, int* status, void** call);
int pirgpu_process_requests_end(void* call);
uint64_t* pirgpu_host_query_buffer(pirgpu_ctx* ctx, uint32_t queries);
uint64_t* pirgpu_host_reply_buffer(pirgpu_ctx* ctx, uint32_t queries);
/* Releases a response buffer.  The library keeps up to 256 MB of released buffers for later responses (a fresh megabyte
 * per reply would cost an mmap, its page faults and a munmap every time); pass only pointers the library returned. */
void pirgpu_free(void* p);

/* Measurement: mean duration of the phases of the pirgpu_query_run calls made since
 * profiling was enabled (or since the previous call of this function), taken with
 * HIP events on the context's own stream.  phase_ms[0..5] = expansion +
 * selection-vector NTT, (reserved, 0), database scan (the streaming kernel alone),
 * upper levels (inverse NTT of the scan rows, re-encode, multiply-accumulate),
 * final inverse NTT, total.  *runs receives the number of runs averaged. */
int pirgpu_last_timings(pirgpu_ctx* ctx, float phase_ms[6], uint32_t* runs);
/* Enable/disable the phase events above (default off: zero overhead). */
int pirgpu_set_profiling(pirgpu_ctx* ctx, int enabled);
/* Measurement, batch pipeline: with profiling enabled, the database-pass launches of the batches run since (up to 64)
 * are bracketed with HIP events on their lane's stream.  Mean / minimum duration of those launches -- the scan as it
 * actually runs inside the headline step: `workgroups` persistent workgroups (0 = one per CU) beside the other lane's
 * kernels, `queries` queries per pass -- and their number; waits for the batch and resets the collection. */
int pirgpu_batch_scan_timings(pirgpu_ctx* ctx, float* mean_ms, float* min_ms, uint32_t* launches, uint32_t* workgroups,
                              uint32_t* queries);
/* Bytes a single-query pass over the database reads: the digit-packed operand-layout copy when that pass
 * is the int8-MFMA scan (info[7] of pirgpu_scan_info), else num_pt(shard) * k * N * 8. */
uint64_t pirgpu_scan_bytes(const pirgpu_ctx* ctx);
/* How this context scans its database.  info[0] = 1 if the digit-sliced int8-MFMA scan is active
 * (0: 64-bit multiply-accumulate kernels), info[1] = digits per residue, info[2] = column chunks per
 * pass, info[3] = k-steps (64 columns) per chunk, info[4] = queries per database pass in batch mode,
 * info[5] = rows, info[6] = columns of the scanned matrix, info[7] = flags: bit 0 = single queries use the MFMA
 * scan as well (matrices wider than one column chunk may scan single queries with the 64-bit kernels), bit 1 = the top
 * digit of database and selectors is stored as a nibble (L - 1/2 bytes per residue instead of L). */
int pirgpu_scan_info(pirgpu_ctx* ctx, uint32_t info[8]);
/* Options by name (case-insensitive), e.g. "scan_mfma" (0 keeps the 64-bit multiply-accumulate scan for d >= 2),
 * "scan_mfma_wide" (0 / 1 forces the 8-wave / 4-wave scan kernel), "upper_blocks", "fuse_last", "last_ntt",
 * "tree40", "sel_f64", "split_upper", "loop_transforms" (0: one transform per workgroup everywhere),
 * "scan_mfma_wgs_batch" (workgroups of a database pass that shares the chip with another group) -- DESIGN.md section 6
 * lists them.  A name that was not set falls back to the environment variable
 * PIRGPU_<NAME> (the A/B scripts under tools/ use that), then to the built-in default; get_option returns -1 for
 * "built-in default".  Options that shape the workspace must be set before the context is first used
 * (FailedPrecondition afterwards).  The arithmetic flavour (PIRGPU_NTT_MODE) is fixed at pirgpu_create.
 * Environment variables are honoured only when PIRGPU_ALLOW_ENV=1 is set too (tests, A/B scripts): by default a
 * context's behaviour depends on its parameters and pirgpu_set_option alone. */
int pirgpu_set_option(pirgpu_ctx* ctx, const char* name, int64_t value);
int pirgpu_get_option(pirgpu_ctx* ctx, const char* name, int64_t* value);
/* Arithmetic flavour of the transform kernels of this context: 0 = 64-bit integer Shoup/Harvey butterflies (any
 * modulus < 2^61), 1 = exact fp64 (all moduli < 2^46), 2 = exact fp64 with per-stage renormalisation (< 2^49).
 * Chosen from the largest modulus; PIRGPU_NTT_MODE=0|2 in the environment at pirgpu_create forces a more general
 * flavour (all three produce identical residues -- tests/test_gpu_ntt_modes.py). */
int pirgpu_ntt_mode(const pirgpu_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PIRGPU_H_ */
