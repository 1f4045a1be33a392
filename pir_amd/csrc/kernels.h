// kernels.h -- launch wrappers of the gfx950 kernels (kernels.hip, ntt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_params.h"

namespace pirgpu {

constexpr int kMaxScanChunks = 16;    // column chunks of one MFMA scan pass (wider matrices use the 64-bit kernels)
constexpr int kMaxMfmaQueries = 8;   // (query, comp) pairs fill the 16 columns of one MFMA tile
constexpr int kMaxScanGroups = 16;   // groups of <= 8 queries one scan launch can serve (slot-sharded multi-GPU step)
constexpr int kMaxSlices = 16;       // slot ranges (= ranks) a packed selector buffer / row-sum exchange can be cut into

constexpr uint32_t kKsWideLevel = 256;   // nodes per launch from which the key-switch kernels use the XCD-aware 1-D grid
inline bool ks_digit_takes_c0(uint32_t nodes) { return nodes >= kKsWideLevel && nodes % 8 == 0; }

struct MfmaPtrs {                    // one pointer per query of a group (by-value kernel argument)
  const void* p[kMaxMfmaQueries];
};

// Groups of one scan launch (by-value kernel argument): group g reads its packed selectors at sel[g] and writes query q
// to out[g] + q * out_qstride.
struct ScanGroups {
  uint32_t n;
  const uint8_t* sel[kMaxScanGroups];
  uint64_t* out[kMaxScanGroups];
  uint8_t nq[kMaxScanGroups];
};

// Slot ranges [cut[r], cut[r+1]) (multiples of 16, cut[0] = 0, cut[n] = k N) and, for sel_pack, the byte offset of each
// range's piece in the output buffer.
struct SliceMap {
  uint32_t n;
  uint32_t cut[kMaxSlices + 1];
  uint64_t off[kMaxSlices];
};

// Galois keys for ONE Galois element of the queries expanded together: tree ciphertext n (= node * B + query) is
// switched with p[n % B].  Keys are per request in the reference (PIRServer::ProcessRequest deserialises them into a
// local, server.cpp:46-48), so the queries of one group may come from different clients; B = 1: one key for all.
struct KeyPtrs {
  const uint64_t* p[kMaxMfmaQueries];
  uint32_t B;
};

// Kernels that contain an NTT, for one ring degree (ntt_kernels.hip is compiled once
// per degree).  `mode` is an NttMode.
struct NttOps {
  hipError_t (*configure)(int mode);
  hipError_t (*ntt_batch)(hipStream_t st, int mode, const DevParams* P, uint64_t* data, uint64_t n_polys,
                          uint32_t mod_period, uint32_t mod_base, bool inverse);
  // src_is_tree: the source holds the expansion tree's element type (doubles in the fp64 flavours)
  hipError_t (*ct_ntt_fwd_oop)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* src,
                               uint64_t* dst, uint64_t n_cts, bool src_is_tree);
  // same transform for B interleaved queries (source ciphertext i*B + q -> dst.p[q] + i): batched expansion
  hipError_t (*ct_ntt_fwd_split)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* src,
                                 const MfmaPtrs& dst, uint32_t B, uint64_t n_cts_total);
  hipError_t (*db_encode)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* coeffs,
                          const uint8_t* bytes, uint64_t bytes_per_pt, uint64_t total_bytes, uint32_t bits,
                          uint64_t n_pt, uint64_t* db);
  // c0_out != nullptr (fp64 flavours, ks_digit_takes_c0(nodes)): the launch also writes NTT(c0) of every node into the
  // product buffer c0_out (NTT-domain last level), sparing ks_last_ntt its own launch for that (c0_done)
  // tree40: res_in holds the tree in the 5-byte form (wide levels of the fused expansion, ks_mac_combine's tout40)
  hipError_t (*ks_digit)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* res_in,
                         uint32_t galois_elt, uint32_t nodes, uint64_t* dig, bool pack40, uint64_t* c0_out, bool tree40,
                         bool loop_targets);   // loop_targets (fp64 flavours, wide levels): one workgroup per source, k + 1 transforms each
  // key-level moduli I_base .. I_base + I_count - 1 (all: 0, k + 1; the special prime alone: k, 1)
  hipError_t (*ks_mac_intt)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* dig,
                            const KeyPtrs& key, uint32_t nodes, uint64_t* prod, bool pack40, uint32_t I_base,
                            uint32_t I_count);
  // n_queries queries in one launch: query q reads src + q * src_qstride, selectors svq.p[q], writes part + q * part_qstride;
  // sel_f64 (fp64 flavours): the selectors are canonical residues stored as exact doubles (lane-internal form)
  hipError_t (*upper_fused)(hipStream_t st, int mode, const DevParams* P, uint32_t k, uint32_t enc_count,
                            const uint64_t* src, const MfmaPtrs& svq, uint64_t* part, uint32_t n_rows,
                            uint32_t n_dim, uint32_t n_children_total, uint32_t sv_first, uint32_t C,
                            uint32_t chunk_len, uint32_t n_chunks, uint32_t n_queries, uint64_t src_qstride,
                            uint64_t part_qstride, bool sel_f64);
  // last expansion level fused with the selectors' forward NTT (fp64 flavours): tree_cts tree ciphertexts
  // (index = slot * B + query) -> selectors slot and slot + shift_pow of query q at dst.p[q], if < n_items
  hipError_t (*ks_last_level)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* tree,
                              const uint64_t* prod, uint32_t galois_elt, uint32_t shift_pow, uint32_t n_items,
                              uint32_t B, const MfmaPtrs& dst, uint32_t tree_cts, bool pack40);
  // split upper level, part 1 (fp64 flavours, large rings): re-encode + lift + forward NTT of the children
  // [b0, b0 + blk) of every row into scratch [query][row][cc][child in block][chunk][target modulus][N] doubles
  hipError_t (*upper_ntt)(hipStream_t st, int mode, const DevParams* P, uint32_t k, uint32_t enc_count,
                          const uint64_t* src, uint64_t* scratch, uint32_t n_rows, uint32_t n_dim,
                          uint32_t n_children_total, uint32_t C, uint32_t b0, uint32_t blk, uint32_t n_queries,
                          uint64_t src_qstride, bool loop_source);   // loop_source: one workgroup per (child, source polynomial)
  // data residues of one level below the last: MAC + inverse transform + combine with the special-prime product
  // (already in `prod`) + tree butterfly, tree_in -> tree_out (fp64 flavours); tin40 / tout40: that tree buffer holds
  // 5-byte polynomials (5 N bytes each, offset form) instead of doubles -- tout40 needs shift_pow < N / 16
  hipError_t (*ks_mac_combine)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* dig,
                               const KeyPtrs& key, const uint64_t* prod, const uint64_t* tree_in, uint32_t galois_elt,
                               uint32_t nodes, uint32_t shift_pow, uint64_t* tree_out, bool pack40, bool tin40,
                               bool tout40, const uint64_t* xpow_c0ntt,    // xpow_c0ntt != null: the tree's c0 is in NTT form
                               const uint16_t* perm,    // ... and sigma_g's table for it (ctx.hip galois_perm_table)
                               bool c0_split);          // ... with component 0 as a launch of its own
  // last expansion level in the NTT domain (fp64 flavours): `prod` holds the special-prime products (ks_mac_intt with
  // I_base = k) and receives NTT(a_0); xpow = NTT_j(x^(-shift_pow)), [k][N] doubles; galois_inv = galois_elt^-1 mod 2N
  hipError_t (*ks_last_ntt)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* tree,
                            const uint64_t* dig, const KeyPtrs& key, uint64_t* prod, const uint64_t* xpow,
                            uint32_t galois_elt, uint32_t galois_inv, uint32_t shift_pow, uint32_t n_items, uint32_t B,
                            const MfmaPtrs& dst, uint32_t nodes, bool pack40, bool out_f64, bool c0_done, int c0_tree,
                            const uint16_t* perm);   // [2][N]: sigma_g, sigma_(g^-1) on NTT positions as padded LDS indices
  // slot-sharded step: inverse NTT of the row sums of queries q0 .. q0 + nq - 1, gathered from the per-rank slot pieces
  // of the all-to-all receive buffer (nq_total queries per piece, RC = 2 rows polynomials per modulus and query) into
  // dst[(q - q0)][row, comp][k][N] -- launch_slots_assemble + ntt_batch(inverse) in one pass
  hipError_t (*ntt_inv_gather)(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* src,
                               uint64_t* dst, const SliceMap& map, uint32_t RC, uint32_t nq, uint32_t nq_total,
                               uint32_t q0);
  // c0 of `cts` tree ciphertexts (doubles) into NTT form, in place: the tree keeps c0 in NTT form from the first fused
  // level on (ks_combine_c0_ntt)
  hipError_t (*tree_c0_fwd)(hipStream_t st, int mode, const DevParams* P, uint32_t k, uint64_t* tree, uint32_t cts);
};

// pack_bytes: width of the packed key-switch intermediates the kernels are built for (5, 6 or 7; ntt_kernels.hip)
const NttOps* ntt_ops_for(uint32_t N, int pack_bytes = 5);  // nullptr for unsupported degrees

hipError_t launch_ntt_reorder(hipStream_t st, uint32_t N, const uint64_t* in, uint64_t* out, uint64_t n_polys,
                              bool to_device, bool as_f64);
// u64 residues < 2^40 <-> 5 bytes each (4 words <-> 5 dwords); words must be a multiple of 4
hipError_t launch_pack40x4(hipStream_t st, const uint64_t* in, uint32_t* out, uint64_t words);
hipError_t launch_unpack40x4(hipStream_t st, const uint32_t* in, uint64_t* out, uint64_t words);
hipError_t launch_ks_combine(hipStream_t st, const DevParams* P, int mode, uint32_t N, uint32_t k,
                             const uint64_t* res_in, const uint64_t* prod, uint32_t galois_inv, uint32_t nodes,
                             uint32_t shift_pow, bool expand_step, uint32_t hi_limit, bool pack40, uint64_t* res_out,
                             int pack_bytes = 5);
// to_tree with chunk_words != 0: the input consists of chunks of chunk_words words that lie in_stride words apart
hipError_t launch_tree_convert(hipStream_t st, const DevParams* P, int mode, const uint64_t* in, uint64_t* out,
                               uint64_t words, bool to_tree, uint64_t chunk_words = 0, uint64_t in_stride = 0);
hipError_t launch_monomial_shift(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* in,
                                 uint32_t shift, uint64_t count, uint64_t* out);
hipError_t launch_scan(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* db,
                       const uint64_t* sv, uint64_t* out, uint32_t rows, uint32_t cols, uint64_t num_pt,
                       uint32_t nsplit, uint32_t cols_per_split, uint32_t rows_per_thread, uint32_t block,
                       bool limb);
hipError_t launch_scan_mq(hipStream_t st, const DevParams* P, uint32_t N, uint32_t k, const uint64_t* db,
                          const uint64_t* const* sv, uint64_t* const* out, uint32_t nq, uint32_t rows,
                          uint32_t cols, uint32_t rows_per_wave, bool limb);
// n_queries > 1: query q folds part + q * part_qstride into out + q * out_qstride (one launch for a group)
hipError_t launch_reduce_splits(hipStream_t st, const DevParams* P, const uint64_t* part, uint32_t nsplit,
                                uint64_t words, uint64_t* out, uint32_t n_queries = 1, uint64_t part_qstride = 0,
                                uint64_t out_qstride = 0);


// split upper level, part 2: acc[query][slot][comp][jt][i] (+)= sum over the block's children of
// scratch (.) selector, elementwise; `first` starts the sums, `last` writes canonical u64 residues to `out`
// (reduce_splits' output layout) instead of keeping signed doubles in `acc`
hipError_t launch_upper_mac(hipStream_t st, const DevParams* P, const uint64_t* scratch, const MfmaPtrs& svq,
                            uint64_t* acc, uint64_t* out, uint32_t n_queries, uint32_t n_rows, uint32_t C,
                            uint32_t enc_count, uint32_t k, uint32_t N, uint32_t sv_first, uint32_t b0, uint32_t blk,
                            uint32_t n_dim, bool first, bool last, uint64_t acc_qstride, uint64_t out_qstride);

// ---- digit-sliced int8-MFMA scan (scan_mfma.hip) ----
struct MfmaGeom {
  uint32_t L;        // balanced base-256 digits per residue (5, 6, 7); 0 = not applicable
  uint32_t RT;       // row tiles (16 rows)
  uint32_t KG;       // column groups (16 columns)
  uint32_t KS;       // k-steps (64 columns) per chunk, selectors of one chunk live in registers
  uint32_t GC;       // column groups per chunk = ceil(KG / nchunks) <= 4 KS
  uint32_t nchunks;  // column chunks (grid.y); > 1 leaves partial sums to reduce_splits_kernel
  uint32_t NW;       // waves (= slots) per workgroup: 8 (two waves per SIMD, KS <= 3) or 4 (one per SIMD, KS <= 7)
  uint32_t top4;     // 1: the top digit of both operands is stored as nibbles (residues < 2^(8 (L - 1) + 4))
  uint32_t tile_bytes;  // bytes of the L digit tiles of one (row tile, column group): L * 256, or (L - 1) * 256 + 128
  size_t db_bytes, sel_bytes;
};

// wide_override: -1 = choose by width, 0 / 1 = force the 8-wave / 4-wave kernel
MfmaGeom mfma_geometry(const DevParams& hp, uint32_t rows, uint32_t cols, int wide_override = -1, bool allow_top4 = true);
// slot0 / nslots: the slots of the ring the packed copy holds (nslots = 0: all from slot0 on), local index j - slot0
hipError_t launch_db_pack(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint64_t* db, uint8_t* dbp,
                          uint32_t rows, uint32_t cols, uint32_t kN, uint32_t slot0 = 0, uint32_t nslots = 0);
hipError_t launch_db_unpack(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp, uint64_t* out,
                            uint32_t row, uint32_t col, uint32_t kN);
// map: the output cut into per-rank pieces by slot ranges (nullptr: one piece)
hipError_t launch_sel_pack(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const MfmaPtrs& sv, uint32_t nq,
                           uint8_t* selp, uint32_t cols, uint32_t kN, bool sel_f64 = false, const SliceMap* map = nullptr);
// wgs: persistent workgroups of the launch (0 = one per CU; the batch pipeline asks for fewer, see scan_mfma.hip)
// f64_fold: fold the digit diagonals in exact fp64 arithmetic (every data modulus below 2^50)
// one group of nq <= 8 queries over the whole ring; query q writes [rows][2][kN] at out + q * out_qstride
hipError_t launch_scan_mfma(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp,
                            const uint8_t* selp, uint64_t* out, uint64_t out_qstride, uint32_t nq, uint32_t rows,
                            uint32_t kN, uint64_t chunk_stride, uint32_t wgs = 0, bool f64_fold = false);
// several groups over the slots [slot0, slot0 + nslots) in one launch (dbp / selectors / output indexed by local slot,
// an output row is out_rstride slots long)
hipError_t launch_scan_mfma_groups(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp,
                                   const ScanGroups& grp, uint32_t rows, uint64_t chunk_stride, uint32_t wgs, bool f64_fold,
                                   uint32_t slot0, uint32_t nslots, uint64_t out_qstride, uint32_t out_rstride,
                                   bool blk_major = false);   // blk_major: units ordered (slot block, group) instead of (group, slot block)
// row sums of queries q0 .. q0 + nq - 1 (of the nq_total the receive buffer holds per rank) back from the per-rank slot
// pieces of the all-to-all receive buffer to [query][row, comp][kN] at dst + (q - q0) * dst_qstride
hipError_t launch_slots_assemble(hipStream_t st, const uint64_t* src, uint64_t* dst, const SliceMap& map, uint32_t RC,
                                 uint32_t kN, uint32_t nq, uint64_t dst_qstride, uint32_t nq_total, uint32_t q0);

}  // namespace pirgpu
