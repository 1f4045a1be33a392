// scan_mfma.hip -- digit-sliced database scan on the int8 matrix cores (gfx950).
//
// Base case of PIRDatabase::multiply (reference database.cpp:185-194,238-247) for d >= 2: for every
// residue slot j (kN of them) and every database row r
//     out[query][r][comp][j] = sum_c  DB[r][c][j] * S[query][c][comp][j]   (mod q_j)
// i.e. per slot j an exact integer matrix product [rows x cols] * [cols x (2 * queries)].  The plain
// 64-bit multiply-accumulate kernels (kernels.hip) are HBM-bound for one query but VALU-bound as soon
// as several queries share a database pass (each 36x36-bit product costs four v_mad_u64_u32).  Here
// both operands are stored as balanced base-256 digits (signed bytes, L per residue) and the L*L
// digit products run on v_mfma_i32_16x16x64_i8:
//     A tile = 16 rows x 64 columns of digit a of the database      (one 16-byte load per lane)
//     B tile = 64 columns x 16 (query, comp) pairs of digit b of the selectors
//     T[a+b] += A_a * B_b      (the K accumulation and the digit diagonal share one int32 accumulator)
// then value = sum_s T[s] 2^(8 s) is folded and reduced modulo q_j exactly.  All arithmetic is integer
// and exact (|T| < L * 2^14 * 64 * KS < 2^25), so replies stay bit-identical to the reference.  The
// database is stored ONCE in the operand layout of the instruction, which also shrinks it from
// 8 to L bytes per residue (5 for the 36-bit moduli of N = 4096): the pass reads 0.74x the bytes of
// the u64 layout (incl. padding to 16 x 16 tiles) and serves up to 8 queries.
//
// Layouts (bytes):
//   database   [j][chunk][rt = r / 16][kg within the chunk][a][r % 16][c % 16]   (kg = c / 16; a chunk = the 4 * KS
//              column groups one launch row keeps selectors for; one chunk -> [j][rt][kg][a][..] as before)
//   selectors  [j][kg][b][x = 2 * query + comp][c % 16]
//   TOP4 (moduli below 2^(8 (L-1) + 4): 36 bits at L = 5, 44 at L = 6): the top digit's 16 x 16 tile is 128 bytes of
//   nibbles instead of 256 bytes (pack_top4 / expand_top4 below) -- L - 1/2 bytes per residue in both operands.
// A workgroup is 8 waves = 8 consecutive slots j (one per wave); results are staged through LDS so
// that each (row, x) is written as one 64-byte run of 8 slots.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>

#include "arith.h"
#include "kernels.h"

namespace pirgpu {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned __int128 u128;

#ifndef PIRGPU_SCAN_DIRECT
#define PIRGPU_SCAN_DIRECT 0
#endif
#ifndef PIRGPU_SCAN_PAD
#define PIRGPU_SCAN_PAD 1     // pad words per (row, x) run of the result staging: 1 = two 8-byte LDS reads per thread, 2 = one 16-byte read
#endif

// Byte offset of tile (slot j, row tile rt, column group kg, digit a) in the packed database.  Column chunks are
// the outer dimension inside a slot, so the part of a slot's slab one workgroup row streams is contiguous
// (a matrix wider than one chunk used to be read as 14-18 KB pieces with gaps: 3.8-4.2 TB/s instead of ~5.8).
// TB = bytes of the L digit tiles of one (row tile, column group): L * 256, or (L - 1) * 256 + 128 when the top digit is
// stored as nibbles (tile_bytes below); digit a starts a * 256 bytes into the block.
__host__ __device__ __forceinline__ size_t db_tile_offset(uint32_t j, uint32_t rt, uint32_t kg, uint32_t a, uint32_t TB,
                                                          uint32_t RT, uint32_t KG, uint32_t GC) {
  const uint32_t ch = kg / GC, kg0 = ch * GC;
  const uint32_t gc = KG - kg0 < GC ? KG - kg0 : GC;   // groups of this chunk
  return ((size_t)j * RT * KG + (size_t)RT * kg0 + (size_t)rt * gc + (kg - kg0)) * TB + (size_t)a * 256;
}

// Top digit as a NIBBLE (TOP4).  A residue below 2^(8 (L-1) + 4) -- 36 bits at L = 5, 44 at L = 6, 52 at L = 7: every
// BASELINE chain -- needs only 4 bits of its top digit if it is centred asymmetrically: v = x for x <= vmax, else x - q,
// with vmax = 7 * 256^(L-1) + 127 (256^(L-1) - 1) / 255 the largest value whose balanced low digits leave a top digit of
// 7; x - q then never goes below the value whose top digit is -8 (the two ranges together span exactly 2^(8 (L-1) + 4)).
// The top digit tile of 16 x 16 entries is stored in 128 bytes, 8 per row: byte i of the first word holds columns i
// (low nibble) and i + 4, byte i of the second word columns 8 + i and 12 + i, so that a row unpacks to its 16 signed
// bytes with shifts and masks only (expand_top4).  Database and packed selectors shrink from L to L - 1/2 bytes per
// residue: 10 % fewer bytes to stream per pass -- and to send to every GPU of a row-sharded job -- at L = 5.
template <int L>
__host__ __device__ __forceinline__ constexpr int64_t top4_vmax() {
  int64_t p = 1;
  for (int a = 0; a < L - 1; ++a) p *= 256;
  return 7 * p + 127 * ((p - 1) / 255);
}
__host__ __device__ __forceinline__ constexpr uint32_t tile_bytes(uint32_t L, bool top4) {
  return top4 ? (L - 1) * 256 + 128 : L * 256;
}
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sx4(uint32_t v) {   // four nibbles (one per byte, 0..15) -> four signed bytes
  return v | (((v >> 3) & 0x01010101u) * 0xF0u);
}
__device__ __forceinline__ v4i expand_top4(v2i w) {
  const uint32_t w0 = (uint32_t)w[0], w1 = (uint32_t)w[1];
  return v4i{(int)sx4(w0 & 0x0F0F0F0Fu), (int)sx4((w0 >> 4) & 0x0F0F0F0Fu), (int)sx4(w1 & 0x0F0F0F0Fu),
              (int)sx4((w1 >> 4) & 0x0F0F0F0Fu)};
}
// 16 signed bytes of a row (each in [-8, 7]) -> the 8 bytes above
__device__ __forceinline__ v2i pack_top4(const uint8_t (&o)[16]) {
  uint32_t w[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[h] |= (uint32_t)((o[8 * h + i] & 0xF) | ((o[8 * h + 4 + i] & 0xF) << 4)) << (8 * i);
  return v2i{(int)w[0], (int)w[1]};
}

// centred residue -> L balanced base-256 digits (TOP4: centred so that the top digit lies in [-8, 7])
template <int L, bool TOP4 = false>
__device__ __forceinline__ void to_digits(uint64_t x, uint64_t q, int8_t (&d)[L]) {
  int64_t v = TOP4 ? ((int64_t)x > top4_vmax<L>() ? (int64_t)x - (int64_t)q : (int64_t)x)
                   : (x > (q >> 1) ? (int64_t)x - (int64_t)q : (int64_t)x);
#pragma unroll
  for (int a = 0; a < L; ++a) {
    d[a] = (int8_t)(v & 0xFF);
    v = (v - d[a]) >> 8;
  }
}

// database u64 [rows][cols][kN] (zero-padded rows) -> packed.  block = 16 rows x 16 slots; grid = (slots/16, RT, KG).
// slot0: first slot of the packed copy (a slot-sharded context packs only its own slots, local index j - slot0)
template <int L, bool TOP4>
__global__ void __launch_bounds__(256)
db_pack_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ db, uint8_t* __restrict__ dbp,
               uint32_t rows, uint32_t cols, uint32_t kN, uint32_t RT, uint32_t KG, uint32_t GC, uint32_t slot0) {
  constexpr uint32_t TB = tile_bytes(L, TOP4);
  const uint32_t j = slot0 + blockIdx.x * 16 + (threadIdx.x & 15);
  const uint32_t r16 = threadIdx.x >> 4;
  const uint32_t rt = blockIdx.y, kg = blockIdx.z;
  const uint32_t r = rt * 16 + r16;
  const uint64_t q = P->mod[j >> P->logN].q;
  uint8_t o[L][16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const uint32_t c = kg * 16 + t;
    uint64_t x = 0;
    if (r < rows && c < cols) x = db[((size_t)r * cols + c) * kN + j];
    int8_t d[L];
    to_digits<L, TOP4>(x, q, d);
#pragma unroll
    for (int a = 0; a < L; ++a) o[a][t] = (uint8_t)d[a];
  }
#pragma unroll
  for (int a = 0; a < L; ++a) {
    if (TOP4 && a == L - 1) {
      *reinterpret_cast<v2i*>(dbp + db_tile_offset(j - slot0, rt, kg, a, TB, RT, KG, GC) + r16 * 8) = pack_top4(o[a]);
    } else {
      v4i v;
      __builtin_memcpy(&v, o[a], 16);
      *reinterpret_cast<v4i*>(dbp + db_tile_offset(j - slot0, rt, kg, a, TB, RT, KG, GC) + r16 * 16) = v;
    }
  }
}

// selectors (per query u64 [cols][2][kN], NTT form) -> packed.  block = 16 x * 16 slots; grid = (kN/16, KG).
// map (slot-sharded multi-GPU step): the packed buffer is cut by slot ranges -- slots [cut[r], cut[r+1]) (multiples of
// 16) start at byte off[r], local slot index inside -- so that every destination rank's slice is one contiguous piece
// of an all-to-all send buffer; n = 0: one piece, the plain layout.
template <int L, bool TOP4>
__global__ void __launch_bounds__(256)
sel_pack_kernel(const DevParams* __restrict__ P, MfmaPtrs sv, uint32_t nq, uint8_t* __restrict__ selp, uint32_t cols,
                uint32_t kN, uint32_t KG, int sel_f64, SliceMap map) {
  constexpr uint32_t TB = tile_bytes(L, TOP4);
  const uint32_t j = blockIdx.x * 16 + (threadIdx.x & 15);
  const uint32_t x = threadIdx.x >> 4;
  const uint32_t kg = blockIdx.y;
  const uint64_t q = P->mod[j >> P->logN].q;
  const uint32_t qi = x >> 1, comp = x & 1;
  if (qi >= nq) return;   // scan_mfma_kernel does not read the columns of absent queries
  const uint64_t* src = (const uint64_t*)sv.p[qi];
  uint8_t o[L][16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const uint32_t c = kg * 16 + t;
    uint64_t v = 0;
    if (c < cols) v = src[((size_t)c * 2 + comp) * kN + j];
    if (sel_f64) v = f64_to_u64(__longlong_as_double((long long)v));   // lane-internal selectors: exact doubles
    int8_t d[L];
    to_digits<L, TOP4>(v, q, d);
#pragma unroll
    for (int b = 0; b < L; ++b) o[b][t] = (uint8_t)d[b];
  }
  uint32_t jl = j;
  size_t piece = 0;
  if (map.n) {
    uint32_t r = 0;
    while (r + 1 < map.n && j >= map.cut[r + 1]) ++r;
    jl = j - map.cut[r];
    piece = map.off[r];
  }
  uint8_t* blk = selp + piece + ((size_t)jl * KG + kg) * TB;
#pragma unroll
  for (int b = 0; b < L; ++b) {
    if (TOP4 && b == L - 1) {
      *reinterpret_cast<v2i*>(blk + b * 256 + x * 8) = pack_top4(o[b]);
    } else {
      v4i v;
      __builtin_memcpy(&v, o[b], 16);
      *reinterpret_cast<v4i*>(blk + b * 256 + x * 16) = v;
    }
  }
}

// one plaintext (row r, column c) back from the operand layout: out[j] = residue in [0, q_j), device slot order
template <int L, bool TOP4>
__global__ void __launch_bounds__(256)
db_unpack_kernel(const DevParams* __restrict__ P, const uint8_t* __restrict__ dbp, uint64_t* __restrict__ out,
                 uint32_t r, uint32_t c, uint32_t kN, uint32_t RT, uint32_t KG, uint32_t GC) {
  constexpr uint32_t TB = tile_bytes(L, TOP4);
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= kN) return;
  const uint64_t q = P->mod[j >> P->logN].q;
  const uint8_t* blk = dbp + db_tile_offset(j, r >> 4, c >> 4, 0, TB, RT, KG, GC);
  const uint8_t* p = blk + (r & 15) * 16 + (c & 15);
  int64_t v = 0;
#pragma unroll
  for (int a = L - 1; a >= 0; --a) {
    if (TOP4 && a == L - 1) {
      const uint32_t cc = c & 15;
      const uint8_t byte = blk[(size_t)a * 256 + (r & 15) * 8 + (cc >> 3) * 4 + (cc & 3)];
      const int nib = (cc & 4) ? (byte >> 4) : (byte & 0xF);
      v = nib >= 8 ? nib - 16 : nib;
    } else {
      v = v * 256 + (int8_t)p[(size_t)a * 256];
    }
  }
  out[j] = v < 0 ? (uint64_t)(v + (int64_t)q) : (uint64_t)v;
}

__device__ __forceinline__ v4i load_tile(const uint8_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
}
__device__ __forceinline__ v2i load_tile8(const uint8_t* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const v2i*>(p));
}

// Workgroups per column chunk: first[c] .. first[c+1] run chunk c (equal shares of equal chunks).
struct ChunkPlan {
  uint32_t first[kMaxScanChunks + 1];
  uint32_t nchunks;
};

// grid = sum of the plan's workgroups; block = NW waves (wave w <-> slot j0 + w).  NW = 8: two waves per SIMD, 256
// registers each -- selectors of up to 3 k-steps; NW = 4 ("wide"): one wave per SIMD with the whole 512-entry register
// file (VGPRs + AGPRs: MFMA operands may live in either), selectors of up to 7 k-steps = 28 column groups in registers,
// so the matrices of cfg 4 / 5 (27 / 21 column groups) are scanned in ONE chunk with all but the last k-step full.  A workgroup is persistent: it walks
// the units u = i, i + n_c, ... (i = its index among the n_c workgroups of its chunk; a unit = one slot block of one
// group of queries, u = group * nblocks + block) and, while it folds and stores the last row tile of one unit, the
// selector tiles and the first database tiles of its next unit are already in flight (one workgroup per CU at this
// register count, so nothing else would hide that latency).
// Chunk ch covers column groups [ch * GC, (ch + 1) * GC) with GC = ceil(KG / nchunks) <= 4 KS -- equal chunks, so
// that equal shares of the chip finish together (7 k-steps split 3/3/1 left a third of the CUs idle for two thirds
// of the pass) -- and writes to out + ch * chunk_stride (partial sums when nchunks > 1).
// Groups (ScanGroups): one launch serves up to kMaxScanGroups groups of <= 8 queries over the SAME database slots --
// group g reads its packed selectors at sel[g] and writes query q of the group to out[g] + q * out_qstride.  One group
// is the single-GPU pass; several are the slot-sharded multi-GPU step, where a rank scans its slots for every query of
// the step in one launch.
// Slots: the launch covers the slots [slot0, slot0 + nslots) of the ring's k N; database, packed selectors and output
// are indexed by the LOCAL slot j - slot0 (a slot-sharded context holds only its own slots of every plaintext), a row
// of the output is out_rstride slots long.  slot0 = 0, nslots = out_rstride = k N: the whole database.
// TOP4: the top digit of both operands is stored as nibbles (above): its database tiles stay packed in two registers per
// lane until their k-step is due; the selectors' are unpacked once per unit.
// F64F: the digit diagonals are folded in exact fp64 arithmetic (all data moduli < 2^50): four diagonals combine exactly
// in one double (|C| < 2^49.01), chunk c times 2^(32 c) mod q with the 6-operation exact product of arith.h --
// ~45 full-rate operations per value against ~170 mostly quarter-rate 64-bit integer ones.  Same canonical residues.
template <int L, int KS, int NW, bool TOP4, bool F64F = false>
__global__ void __launch_bounds__(NW * 64)
scan_mfma_kernel(const DevParams* __restrict__ P, const uint8_t* __restrict__ dbp, ScanGroups grp, uint32_t rows,
                 uint32_t RT, uint32_t KG, uint64_t chunk_stride, ChunkPlan plan, uint32_t GC, uint32_t slot0,
                 uint32_t nslots, uint64_t out_qstride, uint32_t out_rstride, uint32_t blk_major) {
  constexpr uint32_t TB = tile_bytes(L, TOP4);
  constexpr int LF = TOP4 ? L - 1 : L;   // digits stored as full bytes
  constexpr int NS = 2 * L - 1;        // digit diagonals
  constexpr int NG = (NS + 4) / 5;     // groups of five diagonals (40 bits)
  // results of one row tile, [row][x][slot]: a (row, x) run is padded to 9 words so that the 16 lanes of a row (x = 0..15,
  // 72 bytes apart) hit 16 different 8-byte bank pairs when a wave stores its slot (64 bytes apart they hit two)
  // kDirect (-DPIRGPU_SCAN_DIRECT=1): no staging and no workgroup barrier -- every lane stores its four values itself
  // (8 bytes each; the eight waves' stores to a (row, x) run of eight slots merge in L2) and the waves run decoupled.
  constexpr bool kDirect = PIRGPU_SCAN_DIRECT != 0;
  __shared__ __attribute__((aligned(16))) uint64_t stage[kDirect ? 1 : 2][kDirect ? 1 : 16][kDirect ? 1 : 16][NW + PIRGPU_SCAN_PAD];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int g = l >> 4, i16 = l & 15;
  constexpr int LOGNW = NW == 8 ? 3 : 2;
  const uint32_t nblocks = nslots >> LOGNW;
  const uint32_t nunits = nblocks * grp.n;
  uint32_t ch = 0;
  while (ch + 1 < plan.nchunks && blockIdx.x >= plan.first[ch + 1]) ++ch;
  const uint32_t wg_in_chunk = blockIdx.x - plan.first[ch], wgs_in_chunk = plan.first[ch + 1] - plan.first[ch];
  const uint32_t kg0 = ch * GC;                                   // GC <= 4 KS column groups per chunk
  const uint32_t gc = KG - kg0 < GC ? KG - kg0 : GC;              // column groups of this chunk
  const size_t slab = (size_t)RT * KG * TB;                       // database bytes of one slot
  const size_t chunk_base = (size_t)RT * kg0 * TB;                // this chunk inside a slot
  const size_t rt_stride = (size_t)gc * TB;
  const uint32_t lane16 = i16 * 16, lane8 = i16 * 8;              // the lane's bytes inside a full / a nibble tile

  v4i B[KS][L], A[KS][LF];
  v2i A4[KS];                           // TOP4: the top digit's tiles, packed
  // selector tiles of (local) slot jl of group gi
  auto load_B = [&](uint32_t gi, uint32_t jl) {
    const uint8_t* selp = grp.sel[gi];
    const uint32_t nx = 2u * grp.nq[gi];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t gl = ks * 4 + g, kg = kg0 + gl;
      const uint8_t* blk = selp + ((size_t)jl * KG + kg) * TB;
#pragma unroll
      for (int b = 0; b < L; ++b) {
        B[ks][b] = v4i{0, 0, 0, 0};   // columns beyond the group's queries stay zero and are neither packed nor read
        if (gl < gc && (uint32_t)i16 < nx) {
          if (TOP4 && b == L - 1) B[ks][b] = expand_top4(*reinterpret_cast<const v2i*>(blk + b * 256 + lane8));
          else B[ks][b] = *reinterpret_cast<const v4i*>(blk + b * 256 + lane16);
        }
      }
    }
  };
  // the L tiles of column group gl (inside the chunk) of one row tile, from `base` = that row tile's first byte
  auto load_A = [&](int ks, const uint8_t* base, uint32_t gl) {
    const uint8_t* blk = base + (size_t)gl * TB;
#pragma unroll
    for (int a = 0; a < LF; ++a) A[ks][a] = load_tile(blk + a * 256 + lane16);
    if constexpr (TOP4) A4[ks] = load_tile8(blk + (L - 1) * 256 + lane8);
  };

  // unit -> (group, slot block): group-major (u = group * nblocks + block: the launch sweeps the slots once per group) or
  // block-major (u = block * groups + group: the workgroups running side by side read the SAME database tiles for
  // different groups, so all but the first reader of a tile can be served by the memory-side cache)
  // (plain scalar arithmetic at each use: a helper taking references made the compiler keep the pair in scratch)
#define PIRGPU_UNIT_OF(uu, gi_, blk_)                                  \
  const uint32_t gi_ = blk_major ? (uu) % grp.n : (uu) / nblocks;      \
  const uint32_t blk_ = blk_major ? (uu) / grp.n : (uu) - gi_ * nblocks;
  uint32_t u = wg_in_chunk;
  if (u >= nunits) return;
  {
    PIRGPU_UNIT_OF(u, gi, blk)
    load_B(gi, blk * NW + w);
    const uint8_t* abase = dbp + (size_t)(blk * NW + w) * slab + chunk_base;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint32_t gl = ks * 4 + g;   // column group inside the chunk
#pragma unroll
      for (int a = 0; a < LF; ++a) A[ks][a] = v4i{0, 0, 0, 0};
      A4[ks] = v2i{0, 0};
      if (gl < gc) load_A(ks, abase, gl);
    }
  }

  uint32_t parity = 0;
  for (; u < nunits; u += wgs_in_chunk) {
    PIRGPU_UNIT_OF(u, gi, blk)
    const uint32_t j0 = blk * NW;            // local slot of wave 0
    const uint32_t j = slot0 + j0 + w;       // this wave's slot of the ring
    const uint32_t mi = j >> P->logN;
    const ModConst m = P->mod[mi];
    const uint32_t nx = 2u * grp.nq[gi];
    uint64_t* const obase = grp.out[gi] + ch * chunk_stride;
    // multiple of q that makes every 40-bit group positive: 2^57 <= bias < 2^58, |group| < 2^56.1
    const uint64_t bias = m.q << (58 - (64 - __builtin_clzll(m.q)));
    [[maybe_unused]] const F64Mod fm{P->tab[mi].qd, P->tab[mi].qinvd};
    [[maybe_unused]] const double fw0 = P->fold_w[mi][0], fw1 = P->fold_w[mi][1], fw2 = P->fold_w[mi][2];
    const uint8_t* abase = dbp + (size_t)(j0 + w) * slab + chunk_base;
    const uint32_t nu = u + wgs_in_chunk;
    const bool has_next = nu < nunits;
    PIRGPU_UNIT_OF(nu, ngi, nblk)
    const uint8_t* nbase = dbp + (size_t)(nblk * NW + w) * slab + chunk_base;

    for (uint32_t rt = 0; rt < RT; ++rt) {
      v4i T[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) T[s] = v4i{0, 0, 0, 0};
      const bool last = rt + 1 == RT;   // wave-uniform
      // ring of KS k-steps of A tiles: slot ks is refilled right after use with the same step of the next
      // row tile (or of the next unit's first row tile)
      const uint8_t* next_tile = last ? nbase : abase + (size_t)(rt + 1) * rt_stride;
      const bool refill = !last || has_next;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        [[maybe_unused]] v4i Atop;
        if constexpr (TOP4) Atop = expand_top4(A4[ks]);
        // the L*L digit products, ordered so that consecutive MFMAs accumulate into different diagonals
#pragma unroll
        for (int off = 0; off < L; ++off)
#pragma unroll
          for (int a = 0; a < L; ++a) {
            const int b = (a + off) % L;
            if (TOP4 && a == L - 1)
              T[a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Atop, B[ks][b], T[a + b], 0, 0, 0);
            else
              T[a + b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[ks][a < LF ? a : 0], B[ks][b], T[a + b], 0, 0, 0);
          }
        const uint32_t gl = ks * 4 + g;
        if (refill && gl < gc) load_A(ks, next_tile, gl);
      }
      if (last && has_next) load_B(ngi, nblk * NW + w);   // all MFMAs of this unit are issued: B is free
      // lane (g, i16) holds rows rt*16 + g*4 + i (i < 4) of column x = i16:  value = sum_s T[s] 2^(8 s)
      const int buf = parity;
      parity ^= 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        uint64_t r = 0;
        if constexpr (F64F) {
          // chunks of four diagonals, exact in a double: C_c = ((T[4c+3] 256 + T[4c+2]) 256 + T[4c+1]) 256 + T[4c]
          constexpr int NC = (NS + 3) / 4;
          double acc = 0.0;
#pragma unroll
          for (int c = NC - 1; c >= 0; --c) {
            double C = 0.0;
#pragma unroll
            for (int s = (4 * c + 3 < NS ? 4 * c + 3 : NS - 1); s >= 4 * c; --s) C = __builtin_fma(C, 256.0, (double)T[s][i]);
            if (c == 0) acc += f64_norm(C, fm);
            else acc += f64_mulmod(C, c == 1 ? fw0 : (c == 2 ? fw1 : fw2), fm);
          }
          r = f64_to_u64(f64_canon(f64_norm(acc, fm), fm));
        } else {
#pragma unroll
          for (int gq = NG - 1; gq >= 0; --gq) {
            int64_t G = 0;
#pragma unroll
            for (int s = gq * 5; s < gq * 5 + 5 && s < NS; ++s) G += (int64_t)T[s][i] << (8 * (s - gq * 5));
            if (gq == NG - 1 && NG > 1) {
              r = (uint64_t)(G + (int64_t)bias);   // top group: < 2^58, reduced together with the next one (bias = 0 mod q)
            } else {
              const u128 v = ((u128)r << 40) + (uint64_t)(G + (int64_t)bias);
              r = reduce128((uint64_t)v, (uint64_t)(v >> 64), m);
            }
          }
        }
        if constexpr (kDirect) {
          const uint32_t row = rt * 16 + g * 4 + i;
          if ((uint32_t)i16 < nx && row < rows)
            obase[(size_t)(i16 >> 1) * out_qstride + ((size_t)row * 2 + (i16 & 1)) * out_rstride + j0 + w] = r;
        } else {
          stage[buf][g * 4 + i][i16][w] = r;
        }
      }
      if constexpr (kDirect) continue;
      __syncthreads();
      // 256 (row, x) runs of NW slots = 8 NW bytes each; 64 NW threads x 16 B, two rounds
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        const int run = round * 128 + (threadIdx.x >> (LOGNW - 1));
        const int part = threadIdx.x & (NW / 2 - 1);
        const int r16 = run >> 4, x = run & 15;
        const uint32_t r = rt * 16 + r16;
        if (x < (int)nx && r < rows) {
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#if PIRGPU_SCAN_PAD == 2
          const u64x2 v = *reinterpret_cast<const u64x2*>(&stage[buf][r16][x][part * 2]);     // runs 80 bytes apart: one 16-byte read
#else
          const u64x2 v = {stage[buf][r16][x][part * 2], stage[buf][r16][x][part * 2 + 1]};   // two 8-byte LDS reads
#endif
          uint64_t* dst = obase + (size_t)(x >> 1) * out_qstride + ((size_t)r * 2 + (x & 1)) * out_rstride + j0 + part * 2;
          *reinterpret_cast<u64x2*>(dst) = v;
        }
      }
    }
  }
}

#undef PIRGPU_UNIT_OF

// Slot-sharded multi-GPU step: the row sums of a rank's own nq_total queries arrive from every rank h as
// [query][row, comp][slots of h] (the all-to-all's receive buffer; rank h's block starts at word nq_total * RC * cut[h]);
// this puts the queries q0 .. q0 + gridDim.z - 1 back into the [query][row, comp][k N] layout the inverse transform
// reads.  Two words per thread; cuts are multiples of 16.
__global__ void __launch_bounds__(256)
slots_assemble_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, SliceMap map, uint32_t RC, uint32_t kN,
                      uint64_t dst_qstride, uint32_t nq_total, uint32_t q0) {
  const uint32_t j = (blockIdx.x * 256 + threadIdx.x) * 2;
  const uint32_t rc = blockIdx.y, q = blockIdx.z;
  if (j >= kN) return;
  uint32_t h = 0;
  while (h + 1 < map.n && j >= map.cut[h + 1]) ++h;
  const uint32_t c0 = map.cut[h], width = map.cut[h + 1] - c0;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const uint64_t* s = src + (size_t)nq_total * RC * c0 + ((size_t)(q0 + q) * RC + rc) * width + (j - c0);
  *reinterpret_cast<u64x2*>(dst + (size_t)q * dst_qstride + (size_t)rc * kN + j) =
      __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(s));
}

// ------------------------------------------------------------------ host side

MfmaGeom mfma_geometry(const DevParams& hp, uint32_t rows, uint32_t cols, int wide_override, bool allow_top4) {
  MfmaGeom gm{};
  uint32_t bits = 0;
  for (uint32_t i = 0; i < hp.k; ++i) bits = std::max<uint32_t>(bits, 64 - (uint32_t)__builtin_clzll(hp.mod[i].q));
  // L balanced digits cover |v| <= 127 (256^L - 1) / 255 > 2^(8L - 2); centred residues are < 2^(bits - 1)
  if (bits <= 39) gm.L = 5;
  else if (bits <= 47) gm.L = 6;
  else if (bits <= 55) gm.L = 7;
  else return gm;                       // L = 0: not applicable
  if (rows < 1 || cols < 1) { gm.L = 0; return gm; }
  gm.RT = (rows + 15) / 16;
  gm.KG = (cols + 15) / 16;
  // k-steps (64 columns) whose selectors a wave keeps in registers: 3 (L <= 6) / 2 (L = 7) with two waves per SIMD
  // (NW = 8), 7 / 6 with one wave per SIMD and the whole register file (NW = 4, "wide").  Matrices that fit the narrow
  // budget keep it (cfg 3: 11 column groups); wider ones take the wide kernel so that they are scanned in one chunk
  // (cfg 4: 27 groups = 7 k-steps, cfg 5: 21 = 6) instead of three chunks with a quarter-full last k-step each.
  const uint32_t steps = (gm.KG + 3) / 4;
  const uint32_t narrow_ks = gm.L <= 6 ? 3 : 2, wide_ks = gm.L <= 6 ? 7 : 6;
  bool wide = steps > narrow_ks;
  if (wide_override >= 0) wide = wide_override != 0;
  gm.NW = wide ? 4 : 8;
  const uint32_t max_ks = wide ? wide_ks : narrow_ks;
  gm.KS = std::min(max_ks, steps);
  gm.nchunks = (steps + gm.KS - 1) / gm.KS;
  if (gm.nchunks > (uint32_t)kMaxScanChunks) { gm.L = 0; return gm; }   // wider than 16 x 4 KS x 16 columns: 64-bit kernels
  gm.GC = (gm.KG + gm.nchunks - 1) / gm.nchunks;   // equal chunks (<= 4 KS groups each)
  gm.KS = (gm.GC + 3) / 4;
  if (wide && gm.KS < 3) gm.NW = 8;                // the wide kernel is instantiated for 3..7 k-steps
  // top digit as a nibble when every residue is below 2^(8 (L - 1) + 4) (36 / 44 bits at L = 5 / 6: cfg 2, 3 and 4).
  // Not at L = 7 (cfg 5, 49 bits < 2^52 would allow it): the 4-wave kernel with 6 k-steps is out of registers there and
  // the unpacking temporaries spill (scan 9.4 against 8.6 ms) -- the L = 7 variants are not even instantiated.
  gm.top4 = allow_top4 && gm.L <= 6 && bits <= 8 * (gm.L - 1) + 4 ? 1 : 0;
  gm.tile_bytes = tile_bytes(gm.L, gm.top4 != 0);
  const size_t kN = (size_t)hp.k * hp.N;
  gm.db_bytes = kN * gm.RT * gm.KG * gm.tile_bytes;
  gm.sel_bytes = kN * gm.KG * gm.tile_bytes;
  return gm;
}

hipError_t launch_db_pack(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint64_t* db, uint8_t* dbp,
                          uint32_t rows, uint32_t cols, uint32_t kN, uint32_t slot0, uint32_t nslots) {
  if (nslots == 0) nslots = kN - slot0;
  if (slot0 % 16 || nslots % 16 || slot0 + nslots > kN) return hipErrorInvalidValue;
  const dim3 grid(nslots / 16, gm.RT, gm.KG);
#define PIRGPU_DBPACK(L_, T_) \
  hipLaunchKernelGGL((db_pack_kernel<L_, T_>), grid, dim3(256), 0, st, P, db, dbp, rows, cols, kN, gm.RT, gm.KG, gm.GC, slot0)
  switch (gm.L) {
    case 5:
      if (gm.top4) PIRGPU_DBPACK(5, true);
      else PIRGPU_DBPACK(5, false);
      break;
    case 6:
      if (gm.top4) PIRGPU_DBPACK(6, true);
      else PIRGPU_DBPACK(6, false);
      break;
    case 7: PIRGPU_DBPACK(7, false); break;
    default: return hipErrorInvalidValue;
  }
#undef PIRGPU_DBPACK
  return hipGetLastError();
}

hipError_t launch_db_unpack(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp, uint64_t* out,
                            uint32_t row, uint32_t col, uint32_t kN) {
  const dim3 grid((kN + 255) / 256);
  switch (gm.L) {
    case 5:
      if (gm.top4) hipLaunchKernelGGL((db_unpack_kernel<5, true>), grid, dim3(256), 0, st, P, dbp, out, row, col, kN, gm.RT, gm.KG, gm.GC);
      else hipLaunchKernelGGL((db_unpack_kernel<5, false>), grid, dim3(256), 0, st, P, dbp, out, row, col, kN, gm.RT, gm.KG, gm.GC);
      break;
    case 6:
      if (gm.top4) hipLaunchKernelGGL((db_unpack_kernel<6, true>), grid, dim3(256), 0, st, P, dbp, out, row, col, kN, gm.RT, gm.KG, gm.GC);
      else hipLaunchKernelGGL((db_unpack_kernel<6, false>), grid, dim3(256), 0, st, P, dbp, out, row, col, kN, gm.RT, gm.KG, gm.GC);
      break;
    case 7: hipLaunchKernelGGL((db_unpack_kernel<7, false>), grid, dim3(256), 0, st, P, dbp, out, row, col, kN, gm.RT, gm.KG, gm.GC); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_sel_pack(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const MfmaPtrs& sv, uint32_t nq,
                           uint8_t* selp, uint32_t cols, uint32_t kN, bool sel_f64, const SliceMap* map) {
  const int f = sel_f64 ? 1 : 0;
  const dim3 grid(kN / 16, gm.KG);
  SliceMap m{};
  if (map) {
    m = *map;
    if (m.n > (uint32_t)kMaxSlices || m.cut[0] != 0 || m.cut[m.n] != kN) return hipErrorInvalidValue;
    for (uint32_t r = 0; r <= m.n; ++r)
      if (m.cut[r] % 16) return hipErrorInvalidValue;
  }
#define PIRGPU_SELPACK(L_, T_) \
  hipLaunchKernelGGL((sel_pack_kernel<L_, T_>), grid, dim3(256), 0, st, P, sv, nq, selp, cols, kN, gm.KG, f, m)
  switch (gm.L) {
    case 5:
      if (gm.top4) PIRGPU_SELPACK(5, true);
      else PIRGPU_SELPACK(5, false);
      break;
    case 6:
      if (gm.top4) PIRGPU_SELPACK(6, true);
      else PIRGPU_SELPACK(6, false);
      break;
    case 7: PIRGPU_SELPACK(7, false); break;
    default: return hipErrorInvalidValue;
  }
#undef PIRGPU_SELPACK
  return hipGetLastError();
}

hipError_t launch_slots_assemble(hipStream_t st, const uint64_t* src, uint64_t* dst, const SliceMap& map, uint32_t RC,
                                 uint32_t kN, uint32_t nq, uint64_t dst_qstride, uint32_t nq_total, uint32_t q0) {
  if (map.n == 0 || map.n > (uint32_t)kMaxSlices || map.cut[0] != 0 || map.cut[map.n] != kN || q0 + nq > nq_total)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(slots_assemble_kernel, dim3((kN / 2 + 255) / 256, RC, nq), dim3(256), 0, st, src, dst, map, RC, kN,
                     dst_qstride, nq_total, q0);
  return hipGetLastError();
}

static uint32_t scan_wgs_all() {
  // persistent workgroups: one per CU and chunk (at most), each walking its share of the kN/8 slot blocks
  static const uint32_t wgs_all = [] {
    const char* v = pirgpu_env("PIRGPU_SCAN_MFMA_WGS");
    if (v && *v) return (uint32_t)strtoul(v, nullptr, 10);
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256u;
    return (uint32_t)prop.multiProcessorCount;
  }();
  return wgs_all;
}

template <int L, int KS, int NW, bool TOP4, bool F64F>
static void launch_scan_mfma_variant(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp,
                                     const ScanGroups& grp, uint32_t rows, uint64_t chunk_stride, uint32_t wgs_req,
                                     uint32_t slot0, uint32_t nslots, uint64_t out_qstride, uint32_t out_rstride,
                                     bool blk_major) {
  // wgs_req (batch pipeline): a workgroup takes a CU's whole register file, so a launch on fewer CUs leaves the others
  // to the VALU-bound kernels of the other lane -- the HBM-bound pass and the transforms then really overlap
  const uint32_t wgs = wgs_req ? std::min(wgs_req, scan_wgs_all()) : scan_wgs_all();
  // equal shares of the chip for the (equal) column chunks
  ChunkPlan plan{};
  plan.nchunks = gm.nchunks;
  const uint32_t units = nslots / NW * std::max<uint32_t>(grp.n, 1);
  const uint32_t share = std::min<uint32_t>(units, std::max<uint32_t>(1, wgs / std::min<uint32_t>(gm.nchunks, wgs)));
  for (uint32_t c = 0; c <= gm.nchunks; ++c) plan.first[c] = c * share;   // share >= 1: no chunk without workgroups
  hipLaunchKernelGGL((scan_mfma_kernel<L, KS, NW, TOP4, F64F>), dim3(gm.nchunks * share), dim3(NW * 64), 0, st, P, dbp, grp,
                     rows, gm.RT, gm.KG, chunk_stride, plan, gm.GC, slot0, nslots, out_qstride, out_rstride,
                     blk_major ? 1u : 0u);
}

hipError_t launch_scan_mfma_groups(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp,
                                   const ScanGroups& grp, uint32_t rows, uint64_t chunk_stride, uint32_t wgs, bool f64_fold,
                                   uint32_t slot0, uint32_t nslots, uint64_t out_qstride, uint32_t out_rstride, bool blk_major) {
  if (grp.n == 0 || grp.n > (uint32_t)kMaxScanGroups || nslots == 0 || nslots % gm.NW) return hipErrorInvalidValue;
#define PIRGPU_MFMA_ARGS st, P, gm, dbp, grp, rows, chunk_stride, wgs, slot0, nslots, out_qstride, out_rstride, blk_major
#define PIRGPU_MFMA_CASE(L_, KS_, NW_)                                                                    \
  if (gm.L == L_ && gm.KS == KS_ && gm.NW == NW_) {                                                                 \
    if constexpr (L_ <= 6) {                                                                                         \
      if (gm.top4) {                                                                                                 \
        if (f64_fold) launch_scan_mfma_variant<L_, KS_, NW_, true, true>(PIRGPU_MFMA_ARGS);                          \
        else launch_scan_mfma_variant<L_, KS_, NW_, true, false>(PIRGPU_MFMA_ARGS);                                  \
        return hipGetLastError();                                                                                    \
      }                                                                                                              \
    }                                                                                                                \
    if (gm.top4) return hipErrorInvalidValue;                                                                        \
    if (f64_fold) launch_scan_mfma_variant<L_, KS_, NW_, false, true>(PIRGPU_MFMA_ARGS);                             \
    else launch_scan_mfma_variant<L_, KS_, NW_, false, false>(PIRGPU_MFMA_ARGS);                                     \
    return hipGetLastError();                                                                                        \
  }
  PIRGPU_MFMA_CASE(5, 1, 8) PIRGPU_MFMA_CASE(5, 2, 8) PIRGPU_MFMA_CASE(5, 3, 8)
  PIRGPU_MFMA_CASE(6, 1, 8) PIRGPU_MFMA_CASE(6, 2, 8) PIRGPU_MFMA_CASE(6, 3, 8)
  PIRGPU_MFMA_CASE(7, 1, 8) PIRGPU_MFMA_CASE(7, 2, 8)
  PIRGPU_MFMA_CASE(5, 3, 4) PIRGPU_MFMA_CASE(5, 4, 4) PIRGPU_MFMA_CASE(5, 5, 4) PIRGPU_MFMA_CASE(5, 6, 4) PIRGPU_MFMA_CASE(5, 7, 4)
  PIRGPU_MFMA_CASE(6, 3, 4) PIRGPU_MFMA_CASE(6, 4, 4) PIRGPU_MFMA_CASE(6, 5, 4) PIRGPU_MFMA_CASE(6, 6, 4) PIRGPU_MFMA_CASE(6, 7, 4)
  PIRGPU_MFMA_CASE(7, 3, 4) PIRGPU_MFMA_CASE(7, 4, 4) PIRGPU_MFMA_CASE(7, 5, 4) PIRGPU_MFMA_CASE(7, 6, 4)
#undef PIRGPU_MFMA_CASE
#undef PIRGPU_MFMA_ARGS
  return hipErrorInvalidValue;
}

// One group of up to 8 queries over the whole ring (the single-GPU pass): query q writes to out + q * out_qstride.
hipError_t launch_scan_mfma(hipStream_t st, const DevParams* P, const MfmaGeom& gm, const uint8_t* dbp,
                            const uint8_t* selp, uint64_t* out, uint64_t out_qstride, uint32_t nq, uint32_t rows,
                            uint32_t kN, uint64_t chunk_stride, uint32_t wgs, bool f64_fold) {
  ScanGroups grp{};
  grp.n = 1;
  grp.sel[0] = selp;
  grp.out[0] = out;
  grp.nq[0] = (uint8_t)nq;
  return launch_scan_mfma_groups(st, P, gm, dbp, grp, rows, chunk_stride, wgs, f64_fold, 0, kN, out_qstride, kN, false);
}

}  // namespace pirgpu
