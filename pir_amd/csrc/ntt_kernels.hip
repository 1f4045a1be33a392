// ntt_kernels.hip -- every kernel that contains a negacyclic NTT, for ONE ring degree.
//
// Compiled once per supported degree with -DPIRGPU_LOGN=<11..14> (build.py), which keeps
// each translation unit small and lets the degrees build in parallel.  Each kernel
// exists in the three arithmetic flavours of ntt_core.h (integer / fp64 / wide fp64).
//
// Reference call sites replaced (paths relative to /root/reference):
//   ntt_batch_kernel          Evaluator::transform_to/from_ntt_inplace   database.cpp:190,222,252
//   ct_ntt_fwd_oop_kernel     transform_to_ntt_inplace(selection vector) database.cpp:188-191,221-224
//   db_encode_kernel          StringEncoder::encode + transform_to_ntt   database.cpp:100-106, string_encoder.cpp:58-122
//   ks_digit_kernel,
//   ks_mac_intt_kernel        Evaluator::apply_galois_inplace key switch server.cpp:71
//   upper_fused_kernel        CiphertextReencoder::Encode + plain NTT +  database.cpp:218-247, ct_reencoder.cpp:40-71
//                             multiply_plain + add_inplace (upper levels)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "device_params.h"
#include "kernels.h"
#include "ntt_core.h"

#ifndef PIRGPU_LOGN
#error "compile with -DPIRGPU_LOGN=<11..14>"
#endif

// -DPIRGPU_PACK_BYTES=<5|6|7> (default 5): bytes per residue of the PACKED form of the key-switch intermediates and of the
// tree between fused levels (the "40" in the names below is the 5-byte case).  The fp64 flavours store the unsigned
// integer x + q of a signed representative |x| <= q: 5 bytes for moduli below 2^39 (cfg 2 / 3), 6 below 2^47 (cfg 4:
// 43 / 44 bits), 7 below 2^55 (cfg 5: 48 / 49 bits) -- the wide levels are bound by these bytes.  One translation unit
// per (degree, width): the width is a constant of the whole file, not a template parameter of every kernel.
#ifndef PIRGPU_PACK_BYTES
#define PIRGPU_PACK_BYTES 5
#endif
#define PIRGPU_CAT2(a, b) a##b
#define PIRGPU_CAT(a, b) PIRGPU_CAT2(a, b)
#if PIRGPU_PACK_BYTES == 5
#define PIRGPU_OPS_NAME PIRGPU_CAT(ntt_ops_, PIRGPU_LOGN)
#define PIRGPU_DEG_NS PIRGPU_CAT(deg, PIRGPU_LOGN)
#else
#define PIRGPU_OPS_NAME PIRGPU_CAT(PIRGPU_CAT(ntt_ops_, PIRGPU_LOGN), PIRGPU_CAT(_p, PIRGPU_PACK_BYTES))
#define PIRGPU_DEG_NS PIRGPU_CAT(PIRGPU_CAT(deg, PIRGPU_LOGN), PIRGPU_CAT(p, PIRGPU_PACK_BYTES))
#endif

namespace pirgpu {
// every degree gets its own namespace: the kernels of the four translation units must
// not share mangled names (they differ only in the compile-time degree)
namespace PIRGPU_DEG_NS {

constexpr int LOGN = PIRGPU_LOGN;
constexpr int NT = Plan<LOGN>::NT;
constexpr int N = Plan<LOGN>::N;
constexpr int R_ = Plan<LOGN>::R;        // butterfly stages per register pass
constexpr int EPT = Plan<LOGN>::EPT;     // residues per thread: 16 (32 at N = 16384 with -DPIRGPU_LOG_EPT14=5, device_params.h)
constexpr size_t kLdsBytes = (size_t)Plan<LOGN>::LDS_WORDS * 8;
// twiddle prefetch one pass ahead costs 30 (EPT = 16) / 62 (EPT = 32) VGPRs: on up to N = 8192; the 1024-thread
// workgroups of N = 16384 have 128 VGPRs per wave and request a pass's twiddles between the previous pass's butterflies
// and the exchange instead (ntt_core.h).  (The 32-residue organisation of N = 16384 -- 512 threads, 256 VGPRs -- has the
// room; PIRGPU_PF14 = 0 switches the prefetch off there for the A/B of tools/experiments/r04_ab_ept.sh.)
#ifndef PIRGPU_PF14
#define PIRGPU_PF14 1
#endif
constexpr bool kPF = LOGN < 14 || (EPT == 32 && PIRGPU_PF14 != 0);   // 1024-thread workgroups (EPT = 16 at N = 16384): 128 VGPRs, no room
// the looped kernels (several transforms of one source per workgroup) hold the source across the transform: with the
// prefetch they need 138-142 VGPRs at N <= 8192 -- three waves per SIMD instead of four.  Capping them at 128 with the
// twiddles loaded between the previous pass's butterflies and the exchange instead (-DPIRGPU_PF_LOOP=0) was measured
// on one box, three alternating runs: cfg 3 5 331 / 5 350 / 5 405 against 5 376 / 5 431 / 5 442 queries/s with the
// prefetch, cfg 4 445 against 447 -- the prefetch at three waves wins (tools/experiments/r04_ab_loopregs.sh).
#ifndef PIRGPU_PF_LOOP
#define PIRGPU_PF_LOOP 1
#endif
constexpr bool kPFLoop = kPF && (EPT == 32 || PIRGPU_PF_LOOP != 0);
// ks_mac_combine_kernel with both of its paths (component 0 in the NTT domain, component 1 through the inverse transform)
// sits at 126 - 142 registers depending on the compiler's mood; its workgroups wait for loads (k digits + k key
// polynomials per transform), where the fourth wave per SIMD is worth more than a freer schedule (HISTORY section 9)
#ifndef PIRGPU_MC_FOUR_WAVES
#define PIRGPU_MC_FOUR_WAVES 1
#endif
#if PIRGPU_MC_FOUR_WAVES && PIRGPU_LOGN <= 12
#define PIRGPU_MC_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#else
#define PIRGPU_MC_WAVES
#endif
// The product kernels at N = 8192 with 6-byte digits come out at 132 registers: with 512-thread workgroups that is ONE
// workgroup per CU instead of two.  Held at 128 there (-DPIRGPU_PK_FOUR_WAVES=0: the compiler's choice).
#ifndef PIRGPU_PK_FOUR_WAVES
#define PIRGPU_PK_FOUR_WAVES 1
#endif
#if PIRGPU_PK_FOUR_WAVES && PIRGPU_LOGN == 13 && PIRGPU_PACK_BYTES != 5
#define PIRGPU_PK_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#else
#define PIRGPU_PK_WAVES
#endif
#ifndef PIRGPU_DIG13_FOUR_WAVES
#define PIRGPU_DIG13_FOUR_WAVES 0    // experiment: the looped digit kernel at N = 8192 held at 128 registers (two 512-thread workgroups per CU)
#endif
#if PIRGPU_PF_LOOP && !(PIRGPU_DIG13_FOUR_WAVES && PIRGPU_LOGN == 13)
#define PIRGPU_FOUR_WAVES
#else
#define PIRGPU_FOUR_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#endif
// The plain form of upper_fused_kernel (twiddles from global memory, 34 KB of LDS at N = 4096) requests the NEXT child's source
// words before this child's transform and reads sources and selectors through buffer resources: 1 = on (round 6: with it the
// plain form beats the LDS-twiddle form at N = 4096 -- 5 552 against 5 495 queries/s, four alternating runs each,
// profiles/r06_ab_upper_plain_prefetch.txt -- and is the default there; neutral at N = 8192), 2 = this child's first selector
// polynomial as well (32 more registers: +0.5 % instead of +1.0 %), 0 = the round-5 kernel.
#ifndef PIRGPU_UPPER_PLAIN_PF
#define PIRGPU_UPPER_PLAIN_PF 1
#endif
// upper_fused_kernel with the twiddle table in LDS (exchange buffer + N doubles of LDS per workgroup, ~245 VGPRs).
// N = 8192 spills there (four passes, more temporaries: 256 VGPRs + 76 bytes of scratch, cfg 4 363 -> 352 queries/s)
// and N = 16384 runs the split upper level anyway.
constexpr bool kUpperLdsTw = LOGN <= 12;
constexpr uint32_t kWideLevel = kKsWideLevel;  // nodes per launch from which the key-switch kernels use the XCD-aware 1-D grid

extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

// Packed polynomial storage for the key-switch intermediates: N low words (u32) followed by N * kHB high bytes = kPB N
// bytes per polynomial instead of 8 N (kPB = 5: all moduli < 2^39).  The wide expansion levels are bound by the traffic
// of these intermediates, not by the transforms.
constexpr int kPB = PIRGPU_PACK_BYTES;         // bytes per residue
constexpr int kHB = kPB - 4;                   // high bytes per residue: 1, 2, 3
static_assert(kPB >= 5 && kPB <= 7, "packed residues are 5, 6 or 7 bytes");
constexpr size_t kPoly40 = (size_t)kPB * N;
// integer flavour (5-byte form only, element-major: the host never asks the integer flavour for another width)
__device__ __forceinline__ uint64_t load40(const uint8_t* poly, uint32_t i) {
  return (uint64_t)reinterpret_cast<const uint32_t*>(poly)[i] | ((uint64_t)poly[4 * N + i] << 32);
}
__device__ __forceinline__ void store40(uint8_t* poly, uint32_t i, uint64_t v) {
  reinterpret_cast<uint32_t*>(poly)[i] = (uint32_t)v;
  poly[4 * N + i] = (uint8_t)(v >> 32);
}

// fp64 flavours: signed representative <-> offset storage (arith.h f64_pack40: the double 2^52 + x + q carries the integer
// in its low mantissa bits, so packing is one v_add_f64 and a store of the low kPB bytes).  The kernels below all hold
// element e * NT + tid in register e of thread tid, so the high bytes are stored THREAD-MAJOR: the kHB bytes of element e
// of thread tid at 4 N + kHB (EPT tid + e).  A thread then moves its kHB EPT high bytes with kHB 16-byte accesses (a wave:
// kHB KiB contiguous) instead of EPT narrow ones, and there are no partial-line byte stores.  (The integer flavour keeps
// the element-major load40 / store40 layout above; buffers are never shared between flavours.)
constexpr int kHW = kHB * EPT / 4;             // high words per thread
struct Hi16 {                                  // the high bytes of a thread's EPT residues
  uint32_t w[kHW];
};
__device__ __forceinline__ Hi16 load40f_hi(const uint8_t* poly, uint32_t tid) {
  Hi16 h;
#pragma unroll
  for (int g = 0; g < kHW / 4; ++g) {
    const uint4 v = *reinterpret_cast<const uint4*>(poly + 4 * N + (size_t)kHB * EPT * tid + 16 * g);
    h.w[4 * g] = v.x, h.w[4 * g + 1] = v.y, h.w[4 * g + 2] = v.z, h.w[4 * g + 3] = v.w;
  }
  return h;
}
// high word of the double 2^52 + u for element e: 0x43300000 | (u >> 32).  5 bytes: one v_perm_b32 (selector bytes: e & 3 of
// the word, 0, 0x30, 0x43); 6 bytes: one v_perm_b32 (two bytes of the word, 0x30, 0x43); 7 bytes: the three stored bytes
// (they may straddle two words; the third carries the 0x30 with bits 48.. of u) and an OR with 0x43000000.
__device__ __forceinline__ uint32_t hi_word(const Hi16& h, int e) {
  if constexpr (kHB == 1) {
    return __builtin_amdgcn_perm(0x43300000u, h.w[e >> 2], 0x07060c00u | (uint32_t)(e & 3));
  } else if constexpr (kHB == 2) {
    const uint32_t o = 2u * (uint32_t)(e & 1);
    return __builtin_amdgcn_perm(0x43300000u, h.w[e >> 1], 0x07060000u | ((o + 1u) << 8) | o);
  } else {
    const int b = 3 * e, wl = b >> 2;
    const uint32_t o = (uint32_t)(b & 3);
    const uint32_t nxt = h.w[wl + 1 < kHW ? wl + 1 : wl];
    return __builtin_amdgcn_perm(nxt, h.w[wl], 0x0c000000u | ((o + 2u) << 16) | ((o + 1u) << 8) | o) | 0x43000000u;
  }
}
// element e * NT + tid
__device__ __forceinline__ double load40f(const uint8_t* poly, const Hi16& h, int e, uint32_t tid, double magic) {
  const uint32_t lo = reinterpret_cast<const uint32_t*>(poly)[e * NT + tid];
  return __longlong_as_double((long long)(((uint64_t)hi_word(h, e) << 32) | lo)) - magic;
}
// the EPT high words' low kHB bytes -> the kHW words of a thread
__device__ __forceinline__ void pack_hi_bytes(const uint32_t (&hb)[EPT], uint32_t (&w)[kHW]) {
  if constexpr (kHB == 1) {
#pragma unroll
    for (int g = 0; g < EPT / 4; ++g) {
      const uint32_t ab = __builtin_amdgcn_perm(hb[4 * g + 1], hb[4 * g], 0x0c0c0400u);
      const uint32_t cd = __builtin_amdgcn_perm(hb[4 * g + 3], hb[4 * g + 2], 0x0c0c0400u);
      w[g] = __builtin_amdgcn_perm(cd, ab, 0x05040100u);
    }
  } else if constexpr (kHB == 2) {
#pragma unroll
    for (int g = 0; g < EPT / 2; ++g) w[g] = __builtin_amdgcn_perm(hb[2 * g + 1], hb[2 * g], 0x05040100u);
  } else {
    // byte B of the thread's 3 EPT bytes belongs to element B / 3, byte B % 3: every word draws from two neighbouring elements
#pragma unroll
    for (int mw = 0; mw < kHW; ++mw) {
      const int ea = (4 * mw) / 3;
      uint32_t sel = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int B = 4 * mw + i, el = B / 3, bi = B % 3;
        sel |= (uint32_t)(el == ea ? bi : 4 + bi) << (8 * i);
      }
      w[mw] = __builtin_amdgcn_perm(hb[ea + 1 < EPT ? ea + 1 : ea], hb[ea], sel);
    }
  }
}
__device__ __forceinline__ void store_hi_words(uint8_t* poly, uint32_t tid, const uint32_t (&w)[kHW]) {
#pragma unroll
  for (int g = 0; g < kHW / 4; ++g)
    *reinterpret_cast<uint4*>(poly + 4 * N + (size_t)kHB * EPT * tid + 16 * g) = uint4{w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]};
}
__device__ __forceinline__ void store40f(uint8_t* poly, uint32_t tid, const double (&x)[EPT], double magic) {
  uint32_t hb[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    uint32_t lo;
    f64_pack40(x[e], magic, lo, hb[e]);
    reinterpret_cast<uint32_t*>(poly)[e * NT + tid] = lo;
  }
  uint32_t w[kHW];
  pack_hi_bytes(hb, w);
  store_hi_words(poly, tid, w);
}

// Buffer-resource access to one polynomial of u64 words: element e * NT + tid is the thread's byte offset 8 tid (one
// VGPR for all 16 elements) plus a scalar offset 8 e NT -- against a 64-bit address pair per one or two elements with
// flat loads (their immediate offset reaches 4 KiB, the element stride is 8 NT bytes).  The base must be wave-uniform;
// readfirstlane makes that provable to the compiler.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t poly_rsrc(const uint64_t* base) {
  const uint64_t b = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (uint32_t)N * 8u,
                                           0x00020000);
}
__device__ __forceinline__ uint64_t poly_load_u64(__amdgpu_buffer_rsrc_t r, uint32_t tid, int e) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, tid * 8u, (uint32_t)e * (uint32_t)NT * 8u, 0);
  return ((uint64_t)v.y << 32) | v.x;
}

// The same loads through a buffer resource (one VGPR of offset per polynomial, scalar element offsets): a product loop
// built on flat loads needs an address pair per load, and with ~40 registers less the compiler keeps a digit's 33 loads in
// flight together instead of waiting for each one (ks_combine_c0_ntt; tests/test_isa_budget.py).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bytes_rsrc(const void* base, uint32_t bytes) {
  const uint64_t b = (uint64_t)base;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}
// 5-byte polynomial at `poly40` (kPoly40 bytes): this thread's EPT elements as signed representatives
__device__ __forceinline__ void load40f_rsrc(const uint8_t* poly40, uint32_t tid, double magic, double (&out)[EPT]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t r = bytes_rsrc(poly40, (uint32_t)kPoly40);
  Hi16 h;
#pragma unroll
  for (int g = 0; g < kHW / 4; ++g) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)kHB * EPT * tid + 16u * g, 4u * N, 0);
    h.w[4 * g] = v.x, h.w[4 * g + 1] = v.y, h.w[4 * g + 2] = v.z, h.w[4 * g + 3] = v.w;
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const uint32_t lo = __builtin_amdgcn_raw_buffer_load_b32(r, tid * 4u, (uint32_t)e * (uint32_t)NT * 4u, 0);
    out[e] = __longlong_as_double((long long)(((uint64_t)hi_word(h, e) << 32) | lo)) - magic;
  }
}
__device__ __forceinline__ void loadf64_rsrc(const double* poly, uint32_t tid, double (&out)[EPT]) {
  const __amdgpu_buffer_rsrc_t r = poly_rsrc(reinterpret_cast<const uint64_t*>(poly));
#pragma unroll
  for (int e = 0; e < EPT; ++e) out[e] = __longlong_as_double((long long)poly_load_u64(r, tid, e));
}

// Polynomial `poly` (index in units of polynomials) of an expansion-tree buffer, this thread's 16 elements: doubles, or
// (T40, wide levels of the fused expansion, every modulus < 2^39) the 5-byte offset form with modulus q.
template <bool T40>
__device__ __forceinline__ void tree_load(const uint64_t* tree_raw, size_t poly, uint32_t tid, double q, double (&out)[EPT]) {
  if constexpr (T40) {
    const uint8_t* pp = reinterpret_cast<const uint8_t*>(tree_raw) + poly * kPoly40;
    const Hi16 h = load40f_hi(pp, tid);
    const double magic = f64_pack_magic(q);
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e] = load40f(pp, h, e, tid, magic);
  } else {
    // buffer-resource loads: one VGPR of offset for all EPT elements (flat addressing spends an address pair per one or
    // two elements -- enough, in the digit kernels, to cross the 128-register line)
    const __amdgpu_buffer_rsrc_t r = poly_rsrc(tree_raw + poly * N);
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e] = __longlong_as_double((long long)poly_load_u64(r, tid, e));
  }
}

// SEAL position P -> pi_g(P) -> device slot
__device__ __forceinline__ uint32_t galois_ntt_slot(uint32_t P, uint32_t g) {
  const uint32_t r = __brev(P) >> (32 - LOGN);
  const uint32_t ex = ((2 * r + 1) * g) & (2 * N - 1);
  const uint32_t Pin = __brev(ex >> 1) >> (32 - LOGN);
  return (Pin & (uint32_t)(EPT - 1)) * NT + (Pin >> R_);
}

// sigma_g on NTT-form data as a TABLE (ctx.hip galois_perm_table): entry EPT * tid + e = padded LDS word index
// lds_idx(galois_ntt_slot(EPT * tid + e, g)), u16 -- a thread's EPT entries are 2 EPT contiguous bytes.  Computing the index
// costs two bit reversals, a quarter-rate 32-bit multiply and ~10 more integer operations per element (about a third
// of a transform's issue time per permuted polynomial); the table costs EPT / 8 16-byte loads and one extraction each.
#ifndef PIRGPU_PERM_TABLE
#define PIRGPU_PERM_TABLE 1
#endif
struct PermIdx {
  uint32_t w[EPT / 2];   // two u16 indices per word
};
__device__ __forceinline__ PermIdx load_perm(const uint16_t* __restrict__ perm, uint32_t tid) {
  PermIdx r;
#pragma unroll
  for (int g = 0; g < EPT / 8; ++g) {
    const uint4 v = *reinterpret_cast<const uint4*>(perm + (size_t)EPT * tid + 8 * g);
    r.w[4 * g] = v.x, r.w[4 * g + 1] = v.y, r.w[4 * g + 2] = v.z, r.w[4 * g + 3] = v.w;
  }
  return r;
}
// padded LDS word index of pi_g(EPT * tid + e)
__device__ __forceinline__ uint32_t perm_at(const PermIdx& pi, int e, uint32_t tid, uint32_t g) {
  if constexpr (PIRGPU_PERM_TABLE != 0) return (e & 1) ? pi.w[e >> 1] >> 16 : pi.w[e >> 1] & 0xffffu;
  else return lds_idx<R_>(galois_ntt_slot(EPT * tid + e, g));
}


// One workgroup per polynomial; modulus index = mod_base + (poly % mod_period).
// Forward: natural coefficients -> device NTT order; inverse: the reverse.  In place.
template <int MODE, bool INVERSE>
__global__ void __launch_bounds__(NT)
ntt_batch_kernel(const DevParams* __restrict__ P, uint64_t* __restrict__ data, uint32_t mod_period,
                 uint32_t mod_base) {
  using A = Arith<MODE>;
  const uint32_t tid = threadIdx.x;
  const int mi = mod_base + (blockIdx.x % mod_period);
  const typename A::Mod m = A::mod(P, mi);
  uint64_t* poly = data + (size_t)blockIdx.x * N;
  typename A::T x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) x[e] = A::in(poly[e * NT + tid], m);
  if constexpr (INVERSE)
    ntt_inverse<MODE, LOGN, kPF>(x, smem_raw, P, mi, tid);
  else
    ntt_forward<MODE, LOGN, kPF>(x, smem_raw, P, mi, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) poly[e * NT + tid] = A::out(x[e], m);
}

// Slot-sharded multi-GPU step (pirgpu_slots_finish): inverse NTT of the row sums of a rank's own queries, GATHERED out of
// the all-to-all's receive buffer -- rank h's block holds [query][row, comp][slots of h] (block h starts at word
// nq_total * RC * cut[h]), a polynomial's N slots lie in up to n of those blocks -- straight into the
// [query][row, comp][k][N] layout of the lane buffer: what transform_from_ntt_inplace (database.cpp:250-254) does on the
// row sums, without a separate assembly pass.  grid = nq * RC * k.  The cuts are multiples of the workgroup size (the host
// checks; otherwise it keeps the separate pass), so the NT consecutive slots a workgroup loads for one e lie in ONE
// piece: the piece lookup is scalar (a per-lane index into the by-value map made the compiler spill it to scratch and
// cost more than the pass it replaced).
template <int MODE>
__global__ void __launch_bounds__(NT)
ntt_inv_gather_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                      SliceMap map, uint32_t RC, uint32_t nq_total, uint32_t q0) {
  using A = Arith<MODE>;
  const uint32_t tid = threadIdx.x, k = P->k;
  const int mi = blockIdx.x % k;
  const uint32_t rc = (blockIdx.x / k) % RC, q = blockIdx.x / (k * RC);
  const typename A::Mod m = A::mod(P, mi);
  const size_t row = (size_t)(q0 + q) * RC + rc, per_slot = (size_t)nq_total * RC;
  typename A::T x[EPT];
  uint32_t h = 0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const uint32_t jb = (uint32_t)mi * N + e * NT;           // first slot of this load, workgroup-uniform
    while (h + 1 < map.n && jb >= map.cut[h + 1]) ++h;
    const uint32_t c0 = map.cut[h], width = map.cut[h + 1] - c0;
    const uint64_t* piece = src + per_slot * c0 + row * width + (jb - c0);
    x[e] = A::in(piece[tid], m);
  }
  ntt_inverse<MODE, LOGN, kPF>(x, smem_raw, P, mi, tid);
  uint64_t* poly = dst + (size_t)blockIdx.x * N;
#pragma unroll
  for (int e = 0; e < EPT; ++e) poly[e * NT + tid] = A::out(x[e], m);
}

// Out-of-place forward NTT of ciphertexts: src[ct][2][k][N] (coefficient form)
// -> dst[ct][2][k][N] (device NTT order); selection vector -> NTT form.  SRC_TREE: the source is the
// expansion tree (element type A::T: doubles holding signed representatives in the fp64 flavours).
template <int MODE, bool SRC_TREE>
__global__ void __launch_bounds__(NT)
ct_ntt_fwd_oop_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src, uint64_t* __restrict__ dst) {
  using A = Arith<MODE>;
  const uint32_t tid = threadIdx.x;
  const int mi = blockIdx.x % P->k;
  const typename A::Mod m = A::mod(P, mi);
  const uint64_t* in = src + (size_t)blockIdx.x * N;
  uint64_t* out = dst + (size_t)blockIdx.x * N;
  typename A::T x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    if constexpr (SRC_TREE) x[e] = reinterpret_cast<const typename A::T*>(in)[e * NT + tid];
    else x[e] = A::in(in[e * NT + tid], m);
  }
  ntt_forward<MODE, LOGN, kPF>(x, smem_raw, P, mi, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) out[e * NT + tid] = A::out(x[e], m);
}

// The same transform for B queries expanded together: source ciphertext index = slot * B + query
// (batched expansion keeps the queries interleaved), destination = that query's own selection vector.
template <int MODE>
__global__ void __launch_bounds__(NT)
ct_ntt_fwd_split_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src, MfmaPtrs dst, uint32_t B) {
  using A = Arith<MODE>;
  const uint32_t tid = threadIdx.x;
  const uint32_t k2 = 2 * P->k;
  const int mi = blockIdx.x % P->k;
  const typename A::Mod m = A::mod(P, mi);
  const uint32_t ct = blockIdx.x / k2, rem = blockIdx.x % k2;
  const uint32_t slot = ct / B, q = ct % B;
  const typename A::T* in = reinterpret_cast<const typename A::T*>(src) + (size_t)blockIdx.x * N;  // the expansion tree
  uint64_t* out = (uint64_t*)dst.p[q] + ((size_t)slot * k2 + rem) * N;
  typename A::T x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) x[e] = in[e * NT + tid];
  ntt_forward<MODE, LOGN, kPF>(x, smem_raw, P, mi, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) out[e * NT + tid] = A::out(x[e], m);
}

// grid = (n_pt, k).  Source is either pre-encoded coefficients (coeffs != null)
// or raw item bytes packed MSB-first into bits-wide coefficients
// (reference string_encoder.cpp:58-122); then plain lift + forward NTT.
template <int MODE>
__global__ void __launch_bounds__(NT)
db_encode_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ coeffs,
                 const uint8_t* __restrict__ bytes, uint64_t bytes_per_pt, uint64_t total_bytes, uint32_t bits,
                 uint64_t* __restrict__ db) {
  using A = Arith<MODE>;
  const uint32_t tid = threadIdx.x, k = P->k;
  const uint32_t j = blockIdx.y;
  const uint64_t pt = blockIdx.x;
  const ModConst mc = P->mod[j];
  const typename A::Mod m = A::mod(P, j);
  const uint64_t thr = P->plain_thr;
  const uint64_t inc = P->lift_inc[j] >= mc.q ? P->lift_inc[j] - mc.q : P->lift_inc[j];
  uint64_t L = 0;
  const uint8_t* src = nullptr;
  if (!coeffs) {
    uint64_t start = pt * bytes_per_pt;
    L = start >= total_bytes ? 0 : (total_bytes - start < bytes_per_pt ? total_bytes - start : bytes_per_pt);
    src = bytes + start;
  }
  typename A::T x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const uint32_t c = e * NT + tid;
    uint64_t v;
    if (coeffs) {
      v = coeffs[pt * N + c];
    } else {
      v = 0;
      uint64_t bitpos = (uint64_t)c * bits;
      uint64_t byte = bitpos >> 3;
      uint32_t off = (uint32_t)(bitpos & 7);
      int need = (int)bits;
      while (need > 0) {
        uint32_t B = byte < L ? src[byte] : 0u;
        int avail = 8 - (int)off;
        int take = avail < need ? avail : need;
        v = (v << take) | ((B >> (avail - take)) & ((1u << take) - 1u));
        need -= take;
        off = 0;
        ++byte;
      }
    }
    // Evaluator::transform_to_ntt_inplace(Plaintext): m >= (t+1)/2 ? m + (q_j - t) : m  
    uint64_t r = reduce64(v, mc);
    if (v >= thr) r = add_mod(r, inc, mc.q);
    x[e] = A::in(r, m);
  }
  ntt_forward<MODE, LOGN, kPF>(x, smem_raw, P, j, tid);
  uint64_t* out = db + (pt * k + j) * N;
#pragma unroll
  for (int e = 0; e < EPT; ++e) out[e * NT + tid] = A::out(x[e], m);
}

// A_0[n][j] = NTT_j(a_0 mod q_j) of tree ciphertext n (NTT-domain last expansion level, see ks_last_ntt_kernel),
// written into the unused data-residue slot prod[n][0][j] of the product buffer (signed representatives; 5-byte
// packing or doubles).  One workgroup; runs as extra workgroups of the last level's ks_digit_kernel launch (wide
// levels) or as tree_c0_ntt_kernel.
template <int MODE, bool P40, bool T40>
__device__ __forceinline__ void tree_c0_ntt_body(const DevParams* __restrict__ P, const uint64_t* __restrict__ tree_raw,
                                                 uint64_t* __restrict__ prod, uint32_t node, uint32_t j, uint32_t tid) {
  using A = Arith<MODE>;
  static_assert(MODE != kNttInt, "fp64 flavours only");
  const uint32_t k = P->k, km = k + 1;
  const typename A::Mod m = A::mod(P, j);
  double x[EPT];
  tree_load<T40>(tree_raw, (size_t)node * 2 * k + j, tid, m.q, x);
  ntt_forward<MODE, LOGN, kPF, /*CANON=*/false>(x, smem_raw, P, j, tid);
  const size_t opoly = (size_t)node * 2 * km + j;
  if constexpr (P40) {
    uint8_t* out = reinterpret_cast<uint8_t*>(prod) + opoly * kPoly40;
    store40f(out, tid, x, f64_pack_magic(m.q));
  } else {
    double* out = reinterpret_cast<double*>(prod) + opoly * N;
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e * NT + tid] = x[e];
  }
}

template <int MODE, bool P40>
__global__ void __launch_bounds__(NT)
tree_c0_ntt_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ tree_raw, uint64_t* __restrict__ prod) {
  tree_c0_ntt_body<MODE, P40, false>(P, tree_raw, prod, blockIdx.x / P->k, blockIdx.x % P->k, threadIdx.x);
}

// c0 of every tree ciphertext into NTT form, in place (doubles): the transition from the narrow levels (both
// polynomials in coefficient form, ks_combine_f64_kernel) to the fused levels, which keep c0 in NTT form for the rest of
// the tree (ks_combine_c0_ntt below).  grid = tree ciphertexts * k.
template <int MODE>
__global__ void __launch_bounds__(NT)
tree_c0_fwd_kernel(const DevParams* __restrict__ P, uint64_t* __restrict__ tree_raw) {
  static_assert(MODE != kNttInt, "fp64 flavours only");
  const uint32_t tid = threadIdx.x, k = P->k;
  const uint32_t node = blockIdx.x / k, j = blockIdx.x % k;
  double* pp = reinterpret_cast<double*>(tree_raw) + ((size_t)node * 2 * k + j) * N;
  double x[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) x[e] = pp[e * NT + tid];
  ntt_forward<MODE, LOGN, kPF, /*CANON=*/false>(x, smem_raw, P, j, tid);
#pragma unroll
  for (int e = 0; e < EPT; ++e) pp[e * NT + tid] = x[e];
}

// One level of the expansion tree, part 1a: for node n, key-level modulus I and
// RNS digit J:  dig[n][I][J] = NTT_I(sigma_g(c1)_J mod m_I)  (device NTT order, stored in
// the flavour's register type).  grid = (nodes, k+1, k).  T40: the tree is in the 5-byte form (tree_load).
//
// LOOPI (fp64 flavours, wide levels): one workgroup per (node, J) source polynomial runs the k+1 transforms itself, one
// after the other -- grid = nodes * k.  The source is loaded and permuted ONCE, and the store of transform I drains
// while transform I + 1 computes.  That is the overlap rings with several workgroups per CU get from their neighbours;
// at N = 16384 one polynomial of doubles fills the LDS, a CU holds ONE workgroup, and without the loop its load,
// transform and store phases run strictly one after the other (cfg 5).
template <int MODE, bool P40, bool T40 = false, bool LOOPI = false>
__global__ void __launch_bounds__(NT) PIRGPU_FOUR_WAVES   // 128 VGPRs: four waves per SIMD (the looped form would take 131-138)
ks_digit_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ res_in, uint32_t galois_elt,
                uint64_t* __restrict__ dig, uint64_t* __restrict__ c0_out, uint32_t digit_blocks) {
  using A = Arith<MODE>;
  uint64_t* s = reinterpret_cast<uint64_t*>(smem_raw);
  const uint32_t tid = threadIdx.x;
  const uint32_t k = P->k;
  if constexpr (MODE != kNttInt) {
    // last level in the NTT domain: the workgroups behind the digit workgroups transform the c0 polynomials
    if (blockIdx.x >= digit_blocks) {   // 1-D grids only (c0_out != nullptr)
      const uint32_t b = blockIdx.x - digit_blocks;
      tree_c0_ntt_body<MODE, P40, T40>(P, res_in, c0_out, b / k, b % k, tid);
      return;
    }
  }
  if constexpr (LOOPI) {
    static_assert(MODE != kNttInt, "fp64 flavours only");
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    const uint32_t J = t % k, node = (t / k) * 8 + xcd;
    double* sd = reinterpret_cast<double*>(smem_raw);
    const double qJd = P->tab[J].qd;
    const uint32_t raw0 = tid * galois_elt, rstep = (uint32_t)NT * galois_elt;
    double base[EPT];   // sigma_g(c1)_J, canonical in [0, q_J), in the transform's input layout
    {
      double c1[EPT];
      tree_load<T40>(res_in, ((size_t)node * 2 + 1) * k + J, tid, qJd, c1);
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const uint32_t raw = raw0 + (uint32_t)e * rstep;
        const uint64_t sign = (uint64_t)((raw << (31 - LOGN)) & 0x80000000u) << 32;
        double v = __longlong_as_double((long long)((uint64_t)__double_as_longlong(c1[e]) ^ sign));
        v = v < 0.0 ? v + qJd : v;  // -0.0 stays 0
        sd[lds_idx<R_>(raw & (N - 1))] = v;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < EPT; ++e) base[e] = sd[lds_lin_base<NT, R_>(tid) + lds_lin_off<NT, R_>(e)];
    }
#pragma clang loop unroll(disable)
    for (uint32_t I = 0; I <= k; ++I) {
      const typename A::Mod m = A::mod(P, I);
      const bool norm_in = MODE == kNttF64Wide && I != J;
      double x[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) x[e] = norm_in ? f64_norm(base[e], m) : base[e];
      __syncthreads();  // the permutation / the previous transform is done with the LDS words
      const size_t poly = ((size_t)node * (k + 1) + I) * k + J;
      // the thread index as the transform sees it is opaque per iteration: otherwise every LDS / twiddle address of
      // every pass is hoisted out of the loop and stays live across it (the 128-VGPR budget then spills)
      uint32_t tl = tid;
      asm volatile("" : "+v"(tl));
      if constexpr (P40) {
        ntt_forward<MODE, LOGN, kPFLoop, /*CANON=*/false>(x, smem_raw, P, I, tl);
        store40f(reinterpret_cast<uint8_t*>(dig) + poly * kPoly40, tl, x, f64_pack_magic(m.q));
      } else {
        ntt_forward<MODE, LOGN, kPFLoop>(x, smem_raw, P, I, tl);
        double* out = reinterpret_cast<double*>(dig) + poly * N;
#pragma unroll
        for (int e = 0; e < EPT; ++e) out[e * NT + tl] = x[e];
      }
    }
    return;
  }
  uint32_t node = blockIdx.x, I = blockIdx.y, J = blockIdx.z;
  if (gridDim.y == 1) {
    // wide levels, 1-D grid: the k+1 target moduli of one (node, J) source polynomial run back to back on
    // the same XCD (block b -> XCD b % 8), so the source is fetched from HBM once and re-read from L2
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    I = t % (k + 1);
    const uint32_t u = t / (k + 1);
    J = u % k;
    node = (u / k) * 8 + xcd;
  }
  const ModConst mI = P->mod[I];
  const typename A::Mod m = A::mod(P, I);
  [[maybe_unused]] const uint64_t qJ = P->mod[J].q;
  [[maybe_unused]] const uint64_t* src = res_in + (((size_t)node * 2 + 1) * k + J) * N;  // poly 1, residue J (u64 / doubles)
  typename A::T x[EPT];
  if constexpr (MODE == kNttInt) {
    // sigma_g in coefficient form (SEAL GaloisTool::apply_galois), then reduce mod m_I
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const uint32_t i = e * NT + tid;
      const uint32_t raw = i * galois_elt;
      uint64_t v = src[i];
      if ((raw >> LOGN) & 1) v = neg_mod(v, qJ);
      s[lds_idx<R_>(raw & (N - 1))] = reduce64(v, mI);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = A::in(s[lds_lin_base<NT, R_>(tid) + lds_lin_off<NT, R_>(e)], m);
  } else {
    // fp64 flavours: the tree holds doubles (signed representatives, |v| <= (1/2 + eps) q_J); sigma_g's sign
    // is the double's sign bit.  The RNS digit is the CANONICAL residue in [0, q_J) read as an integer and
    // reduced mod m_I (D_j mod m_i), so the representative is made canonical before it
    // changes modulus; as a butterfly input mod m_I it is then fine as it stands in the plain fp64 flavour
    // (every modulus < 2^46: inputs up to 2^52 are exact), the wide flavour normalises it.
    double* sd = reinterpret_cast<double*>(smem_raw);
    const bool norm_in = MODE == kNttF64Wide && I != J;
    const double qJd = P->tab[J].qd;
    const uint32_t raw0 = tid * galois_elt, rstep = (uint32_t)NT * galois_elt;
    double c1[EPT];
    tree_load<T40>(res_in, ((size_t)node * 2 + 1) * k + J, tid, qJd, c1);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const uint32_t raw = raw0 + (uint32_t)e * rstep;
      const uint64_t sign = (uint64_t)((raw << (31 - LOGN)) & 0x80000000u) << 32;
      double v = __longlong_as_double((long long)((uint64_t)__double_as_longlong(c1[e]) ^ sign));
      v = v < 0.0 ? v + qJd : v;  // -0.0 stays 0
      if (norm_in) v = f64_norm(v, m);
      sd[lds_idx<R_>(raw & (N - 1))] = v;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = sd[lds_lin_base<NT, R_>(tid) + lds_lin_off<NT, R_>(e)];
  }
  __syncthreads();  // the transform reuses the same LDS words with its own element type
  const size_t poly = ((size_t)node * (k + 1) + I) * k + J;
  if constexpr (P40 && MODE != kNttInt) {
    ntt_forward<MODE, LOGN, kPF, /*CANON=*/false>(x, smem_raw, P, I, tid);  // signed representatives, |x| <= q/2
    uint8_t* out = reinterpret_cast<uint8_t*>(dig) + poly * kPoly40;
    store40f(out, tid, x, f64_pack_magic(m.q));
  } else {
    ntt_forward<MODE, LOGN, kPF>(x, smem_raw, P, I, tid);
    if constexpr (P40) {
      uint8_t* out = reinterpret_cast<uint8_t*>(dig) + poly * kPoly40;
#pragma unroll
      for (int e = 0; e < EPT; ++e) store40(out, e * NT + tid, A::out(x[e], m));
    } else {
      typename A::T* out = reinterpret_cast<typename A::T*>(dig) + poly * N;
#pragma unroll
      for (int e = 0; e < EPT; ++e) out[e * NT + tid] = x[e];
    }
  }
}

// Part 1b: S[c][I] = sum_J dig[n][I][J] (.) K[J][c][I], then INTT_I.  The key is in device NTT order (and in the
// flavour's register type), so the dyadic products are formed directly in the register layout the inverse transform
// starts from.  Leaves x[e] = coefficient e * NT + tid: canonical for the integer flavour, a signed representative
// |x| <= (1/2 + eps) m_I for the fp64 flavours.
template <int MODE, bool P40>
__device__ __forceinline__ void ks_mac_core(const DevParams* __restrict__ P, const uint64_t* __restrict__ dig_raw,
                                            const uint64_t* __restrict__ key_raw, uint32_t node, uint32_t I,
                                            uint32_t comp, uint32_t tid, typename Arith<MODE>::T (&x)[EPT]) {
  using A = Arith<MODE>;
  using T = typename A::T;
  const uint32_t k = P->k, km = k + 1;
  const ModConst mI = P->mod[I];
  const typename A::Mod m = A::mod(P, I);
  const size_t poly0 = ((size_t)node * km + I) * k;
  const T* d0 = reinterpret_cast<const T*>(dig_raw) + poly0 * N;
  const uint8_t* d40 = reinterpret_cast<const uint8_t*>(dig_raw) + poly0 * kPoly40;
  const T* key = reinterpret_cast<const T*>(key_raw);
  if constexpr (MODE == kNttInt) {
    // digit J, element i
    auto digit = [&](uint32_t J, uint32_t i) -> T {
      if constexpr (P40) return A::in(load40(d40 + (size_t)J * kPoly40, i), m);
      else return d0[(size_t)J * N + i];
    };
    u128 acc[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) acc[e] = 0;
    for (uint32_t J = 0; J < k; ++J) {  // k <= 8 products of two residues < 2^61 fit 128 bits
      const T* kj = key + (((size_t)J * 2 + comp) * km + I) * N;
#pragma unroll
      for (int e = 0; e < EPT; ++e) acc[e] += (u128)digit(J, e * NT + tid) * kj[e * NT + tid];
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = reduce128((uint64_t)acc[e], (uint64_t)(acc[e] >> 64), mI);
    ntt_inverse<MODE, LOGN, kPF>(x, smem_raw, P, I, tid);
  } else {
    const double magic = f64_pack_magic(m.q);
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = 0.0;
    for (uint32_t J = 0; J < k; ++J) {
      const T* kj = key + (((size_t)J * 2 + comp) * km + I) * N;
      if constexpr (P40) {
        const uint8_t* dj = d40 + (size_t)J * kPoly40;
        const Hi16 h = load40f_hi(dj, tid);
#pragma unroll
        for (int e = 0; e < EPT; ++e) x[e] += f64_mulmod(load40f(dj, h, e, tid, magic), kj[e * NT + tid], m);
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) x[e] += f64_mulmod(d0[(size_t)J * N + e * NT + tid], kj[e * NT + tid], m);
      }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = f64_norm(x[e], m);
    ntt_inverse<MODE, LOGN, kPF, /*CANON=*/false>(x, smem_raw, P, I, tid);
  }
}

// Stores prod[n][c][I] (coefficient order): 5-byte packing (offset form for the fp64 flavours), doubles, or u64.
// grid = (nodes, I_count, 2) or the XCD-aware 1-D equivalent; I = I_base + the grid's modulus index -- the expansion
// runs it for the special prime alone (I_base = k, I_count = 1) below the last level, where the data residues go
// through ks_mac_combine_kernel instead.
template <int MODE, bool P40>
__global__ void __launch_bounds__(NT) PIRGPU_PK_WAVES
ks_mac_intt_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ dig_raw, KeyPtrs keys,
                   uint64_t* __restrict__ prod, uint32_t I_base, uint32_t I_count) {
  using A = Arith<MODE>;
  using T = typename A::T;
  const uint32_t tid = threadIdx.x;
  const uint32_t k = P->k, km = k + 1;
  uint32_t node = blockIdx.x, I = blockIdx.y, comp = blockIdx.z;
  if (gridDim.y == 1 && gridDim.z == 1) {
    // wide levels, 1-D grid: both components of one (node, I) run back to back on the same XCD, so the
    // second one finds the digits in L2 instead of re-reading them from HBM
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    comp = t & 1;
    const uint32_t u = t >> 1;
    I = u % I_count;
    node = (u / I_count) * 8 + xcd;
  }
  I += I_base;
  const typename A::Mod m = A::mod(P, I);
  T x[EPT];
  ks_mac_core<MODE, P40>(P, dig_raw, keys.p[node % keys.B], node, I, comp, tid, x);   // the key of this node's query
  const size_t opoly = ((size_t)node * 2 + comp) * km + I;
  if constexpr (P40 && MODE != kNttInt) {
    uint8_t* out = reinterpret_cast<uint8_t*>(prod) + opoly * kPoly40;
    store40f(out, tid, x, f64_pack_magic(m.q));
  } else if constexpr (MODE != kNttInt) {  // plain doubles (moduli >= 2^39): signed representatives as they are
    T* out = reinterpret_cast<T*>(prod) + opoly * N;
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e * NT + tid] = x[e];
  } else if constexpr (P40) {
    uint8_t* out = reinterpret_cast<uint8_t*>(prod) + opoly * kPoly40;
#pragma unroll
    for (int e = 0; e < EPT; ++e) store40(out, e * NT + tid, A::out(x[e], m));
  } else {
    uint64_t* out = prod + opoly * N;
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e * NT + tid] = A::out(x[e], m);
  }
}

// c0 of one expansion level with the tree's c0 polynomials kept in NTT form (fp64 flavours, fused levels).  c0 is never
// digit-decomposed: on its way to the leaves it is only permuted (sigma_g), added to and multiplied by monomials -- all
// of which commute with the transform -- and the leaves are consumed in NTT form (database.cpp:190,222).  So, with
// C0 = NTT_j(c0) and pi_g the permutation sigma_g induces on NTT positions (ks_last_ntt_kernel below),
//   G0  = (S_0j - NTT_j(lift(s_0))) p^-1         S_0j the dyadic key-switch product mod q_j (never inverse-transformed),
//                                                 s_0 the centred special-prime residue (coefficient form, ks_mac_intt)
//   sub = C0 o pi_g + G0                          NTT_j of apply_galois(ct)'s first polynomial (server.cpp:71)
//   lo  = C0 + sub,   hi = X (.) (C0 - sub)       X = NTT_j(x^(-2^level)) (server.cpp:97,137-141)
// One forward transform per (tree ciphertext, data modulus) instead of one inverse one: the same count as before, but
// the last level finds NTT(a_0) in the tree and tree_c0_ntt_kernel's k transforms per leaf pair disappear; lo / hi are
// plain contiguous stores (no index rotation for the monomial).  All arithmetic exact mod q_j: same canonical selectors.
template <int MODE, bool P40, bool TIN40, bool TOUT40>
__device__ __forceinline__ void ks_combine_c0_ntt(const DevParams* __restrict__ P, const uint64_t* __restrict__ dig_raw,
                                                  const uint64_t* __restrict__ key_raw, const uint64_t* __restrict__ prod,
                                                  const uint64_t* __restrict__ tree_in_raw, const double* __restrict__ xpow,
                                                  const uint16_t* __restrict__ perm, uint32_t galois_elt, uint32_t nodes,
                                                  uint32_t node, uint32_t j, uint32_t tid, uint64_t* __restrict__ tree_out_raw) {
  using A = Arith<MODE>;
  double* sd = reinterpret_cast<double*>(smem_raw);
  const uint32_t k = P->k, km = k + 1;
  const typename A::Mod m = A::mod(P, j);
  const double magic = f64_pack_magic(m.q);
  double x[EPT];
  {
    const double pf = P->p_f, half = P->p_half_f;
    const size_t spoly = ((size_t)node * 2 + 0) * km + k;
    [[maybe_unused]] Hi16 hs{};
    if constexpr (P40) hs = load40f_hi(reinterpret_cast<const uint8_t*>(prod) + spoly * kPoly40, tid);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      double sp;
      if constexpr (P40)
        sp = load40f(reinterpret_cast<const uint8_t*>(prod) + spoly * kPoly40, hs, e, tid, f64_pack_magic(pf));
      else
        sp = reinterpret_cast<const double*>(prod)[spoly * N + e * NT + tid];
      sp = sp > half ? sp - pf : sp;   // exact centring (ks_combine_f64_kernel)
      sp = sp < -half ? sp + pf : sp;
      x[e] = f64_norm(sp, m);
    }
  }
  ntt_forward<MODE, LOGN, kPF, /*CANON=*/false>(x, smem_raw, P, j, tid);
  {
    const size_t dpoly0 = ((size_t)node * km + j) * k;
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = -x[e];
    const double* key = reinterpret_cast<const double*>(key_raw);
    for (uint32_t J = 0; J < k; ++J) {
      // digit AND key words into arrays first: with the key word loaded inside the product expression the scheduler falls back
      // to one load + its unpacking + `s_waitcnt vmcnt(0)` at a time (33 dependent round trips per digit; the kernel then
      // took 102 us at 2 048 workgroups against 50 for its component-1 twin) -- 150 registers, three waves per SIMD
      double d[EPT], kv[EPT];
      if constexpr (P40) load40f_rsrc(reinterpret_cast<const uint8_t*>(dig_raw) + (dpoly0 + J) * kPoly40, tid, magic, d);
      else loadf64_rsrc(reinterpret_cast<const double*>(dig_raw) + (dpoly0 + J) * N, tid, d);
      loadf64_rsrc(key + (((size_t)J * 2 + 0) * km + j) * N, tid, kv);
#pragma unroll
      for (int e = 0; e < EPT; ++e) x[e] += f64_mulmod(d[e], kv[e], m);
    }
    const double pinv = P->p_inv_f[j];
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = f64_mulmod(f64_norm(x[e], m), pinv, m);
  }
  // (an opaque copy of the thread index for everything below: see ks_last_ntt_kernel)
  uint32_t tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  double v[EPT];
  tree_load<TIN40>(tree_in_raw, (size_t)node * 2 * k + j, tid_e, m.q, v);
  __syncthreads();  // the transform's last exchange is done with the LDS words
#pragma unroll
  for (int e = 0; e < EPT; ++e) sd[lds_lin_base<NT, R_>(tid_e) + lds_lin_off<NT, R_>(e)] = v[e];
  __syncthreads();
  {
    [[maybe_unused]] PermIdx pi{};
    if constexpr (PIRGPU_PERM_TABLE != 0) pi = load_perm(perm, tid_e);
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] += sd[perm_at(pi, e, tid_e, galois_elt)];
  }
  const double* X = xpow + (size_t)j * N;
  const size_t plo = (size_t)node * 2 * k + j, phi = ((size_t)node + nodes) * 2 * k + j;
  // lo first, then hi, through ONE temporary array: both outputs live together cost a fourth wave per SIMD
  double t[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) t[e] = f64_norm(v[e] + x[e], m);
  if constexpr (TOUT40) {
    store40f(reinterpret_cast<uint8_t*>(tree_out_raw) + plo * kPoly40, tid_e, t, magic);
  } else {
    double* tlo = reinterpret_cast<double*>(tree_out_raw) + plo * N;
#pragma unroll
    for (int e = 0; e < EPT; ++e) tlo[e * NT + tid_e] = t[e];
  }
  // (X's 16 loads hoisted above the lo stores cost 146 - 153 registers, three waves per SIMD: their address is opaque until
  // the lo stores are out)
  uint32_t tid_x = tid_e;   // (tid_e dies here)
  asm volatile("" : "+v"(tid_x) : : "memory");
#pragma unroll
  for (int e = 0; e < EPT; ++e) t[e] = f64_mulmod(f64_norm(v[e] - x[e], m), X[e * NT + tid_x], m);
  if constexpr (TOUT40) {
    store40f(reinterpret_cast<uint8_t*>(tree_out_raw) + phi * kPoly40, tid_x, t, magic);
  } else {
    double* thi = reinterpret_cast<double*>(tree_out_raw) + phi * N;
#pragma unroll
    for (int e = 0; e < EPT; ++e) thi[e * NT + tid_x] = t[e];
  }
}

// ks_combine_c0_ntt as a launch of its own (grid = nodes * k, XCD-aware for wide levels): the form in which the two
// components of a level do not share a kernel -- each keeps its own registers and its own 20 KB of code (the combined
// kernel is 39 KB of instructions next to the other lane's kernel in a 64 KB instruction cache shared by two CUs).
template <int MODE, bool P40, bool TIN40, bool TOUT40>
__global__ void __launch_bounds__(NT)
ks_c0_ntt_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ dig_raw, KeyPtrs keys,
                 const uint64_t* __restrict__ prod, const uint64_t* __restrict__ tree_in_raw, uint32_t galois_elt,
                 uint32_t nodes, uint64_t* __restrict__ tree_out_raw, const double* __restrict__ xpow,
                 const uint16_t* __restrict__ perm) {
  static_assert(MODE != kNttInt, "fp64 flavours only");
  const uint32_t k = P->k;
  uint32_t node = blockIdx.x, j = blockIdx.y;
  if (gridDim.y == 1) {
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    j = t % k;
    node = (t / k) * 8 + xcd;
  }
  ks_combine_c0_ntt<MODE, P40, TIN40, TOUT40>(P, dig_raw, keys.p[node % keys.B], prod, tree_in_raw, xpow, perm, galois_elt, nodes,
                                              node, j, threadIdx.x, tree_out_raw);
}

// Parts 1b + 2 for the DATA residues of one expansion level below the last (fp64 flavours): the workgroup of
// (tree ciphertext, data modulus j, component) forms S[c][j], runs its inverse transform and -- instead of storing
// the product -- applies ks_combine_f64_kernel's arithmetic while the polynomial is in registers: divide-and-round
// with the special-prime residue (computed before by ks_mac_intt_kernel with I_base = k and read back, the only
// product that still goes through HBM), + sigma_g(c0) through an LDS scatter, tree butterfly, and writes lo / hi.
// One transform per workgroup and nothing live across it (the fused variants with several transforms per workgroup
// lost to their register pressure); saves the data products' round trip and the separate
// HBM-bound combine pass.  grid = (nodes, k, 2) or the XCD-aware 1-D equivalent.
// C0NTT: the tree's c0 polynomials are in NTT form -- the workgroups of component 0 run ks_combine_c0_ntt instead
// (xpow = NTT_j(x^(-shift_pow)), [k][N] doubles in device order).
template <int MODE, bool P40, bool TIN40 = false, bool TOUT40 = false, bool C0NTT = false>
__global__ void __launch_bounds__(NT) PIRGPU_MC_WAVES PIRGPU_PK_WAVES
ks_mac_combine_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ dig_raw,
                      KeyPtrs keys, const uint64_t* __restrict__ prod,
                      const uint64_t* __restrict__ tree_in_raw, uint32_t galois_elt, uint32_t nodes, uint32_t shift_pow,
                      uint64_t* __restrict__ tree_out_raw, const double* __restrict__ xpow,
                      const uint16_t* __restrict__ perm, uint32_t comp1_only) {   // comp1_only: the grid covers component 1 alone
  using A = Arith<MODE>;
  static_assert(MODE != kNttInt, "fp64 flavours only");
  double* sd = reinterpret_cast<double*>(smem_raw);
  const uint32_t tid = threadIdx.x, k = P->k, km = k + 1;
  uint32_t node = blockIdx.x, j = blockIdx.y, comp = blockIdx.z;
  if (gridDim.y == 1 && gridDim.z == 1) {
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    comp = comp1_only ? 1 : t & 1;
    const uint32_t u = comp1_only ? t : t >> 1;
    j = u % k;
    node = (u / k) * 8 + xcd;
  } else if (comp1_only) {
    comp = 1;
  }
  if constexpr (C0NTT) {
    if (comp == 0) {   // uniform per workgroup
      ks_combine_c0_ntt<MODE, P40, TIN40, TOUT40>(P, dig_raw, keys.p[node % keys.B], prod, tree_in_raw, xpow, perm, galois_elt,
                                                  nodes, node, j, tid, tree_out_raw);
      return;
    }
  }
  const typename A::Mod m = A::mod(P, j);
  double g[EPT];
  ks_mac_core<MODE, P40>(P, dig_raw, keys.p[node % keys.B], node, j, comp, tid, g);
  {
    const double pf = P->p_f, half = P->p_half_f, pinv = P->p_inv_f[j];
    const size_t spoly = ((size_t)node * 2 + comp) * km + k;
    [[maybe_unused]] Hi16 hs{};
    if constexpr (P40) hs = load40f_hi(reinterpret_cast<const uint8_t*>(prod) + spoly * kPoly40, tid);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      double sp;
      if constexpr (P40) {
        sp = load40f(reinterpret_cast<const uint8_t*>(prod) + spoly * kPoly40, hs, e, tid, f64_pack_magic(pf));
      } else {
        sp = reinterpret_cast<const double*>(prod)[spoly * N + e * NT + tid];
      }
      sp = sp > half ? sp - pf : sp;   // exact centring (ks_combine_f64_kernel)
      sp = sp < -half ? sp + pf : sp;
      g[e] = f64_mulmod(g[e] - f64_norm(sp, m), pinv, m);
    }
  }
  if (!C0NTT && comp == 0) {  // + sigma_g(c0)
    const uint32_t raw0 = tid * galois_elt, rstep = (uint32_t)NT * galois_elt;
    double c0[EPT];
    tree_load<TIN40>(tree_in_raw, (size_t)node * 2 * k + j, tid, m.q, c0);
    __syncthreads();  // the inverse transform is done with the LDS words
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const uint32_t raw = raw0 + (uint32_t)e * rstep;
      const uint64_t sign = (uint64_t)((raw << (31 - LOGN)) & 0x80000000u) << 32;
      sd[lds_idx<R_>(raw & (N - 1))] = __longlong_as_double((long long)((uint64_t)__double_as_longlong(c0[e]) ^ sign));
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) g[e] += sd[lds_lin_base<NT, R_>(tid) + lds_lin_off<NT, R_>(e)];
  }
  const size_t opoly = (size_t)comp * k + j;
  double a[EPT];
  tree_load<TIN40>(tree_in_raw, (size_t)node * 2 * k + opoly, tid, m.q, a);
  if constexpr (TOUT40) {
    // 5-byte tree out (shift_pow < NT, checked by the host): lo keeps its index; hi = x^(-shift) (a - g) moves element
    // e NT + tid to (e NT + tid - shift) mod N, i.e. to thread tid' = (tid - shift) mod NT, same e -- or, for the
    // threads tid < shift, e - 1 with element 0 wrapping to EPT - 1 negated.  All EPT values of a thread land in ONE
    // destination thread, so its EPT high bytes are still one contiguous store (rotated by a byte when borrowing).
    const double magic = f64_pack_magic(m.q);
    uint8_t* plo = reinterpret_cast<uint8_t*>(tree_out_raw) + ((size_t)node * 2 * k + opoly) * kPoly40;
    uint8_t* phi = reinterpret_cast<uint8_t*>(tree_out_raw) + (((size_t)node + nodes) * 2 * k + opoly) * kPoly40;
    double lo[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) lo[e] = f64_norm(a[e] + g[e], m);
    store40f(plo, tid, lo, magic);
    const bool borrow = tid < shift_pow;
    const uint32_t tid_d = (tid - shift_pow) & (NT - 1);
    uint32_t hb[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      double d = f64_norm(a[e] - g[e], m);
      if (e == 0) d = borrow ? -d : d;
      uint32_t l32;
      f64_pack40(d, magic, l32, hb[e]);
      reinterpret_cast<uint32_t*>(phi)[(e * NT + tid - shift_pow) & (N - 1)] = l32;
    }
    uint32_t w[kHW];
    pack_hi_bytes(hb, w);
    uint32_t o[kHW];
#pragma unroll
    for (int q4 = 0; q4 < kHW; ++q4)   // borrowing threads: destination element e' holds source element e' + 1 (mod EPT): kHB bytes on
      o[q4] = borrow ? __builtin_amdgcn_alignbyte(w[(q4 + 1) % kHW], w[q4], kHB) : w[q4];
    store_hi_words(phi, tid_d, o);
  } else {
    const size_t off = opoly * N;
    double* tree_lo = reinterpret_cast<double*>(tree_out_raw) + (size_t)node * 2 * k * N + off;
    double* tree_hi = reinterpret_cast<double*>(tree_out_raw) + ((size_t)node + nodes) * 2 * k * N + off;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const uint32_t i = e * NT + tid;
      tree_lo[i] = f64_norm(a[e] + g[e], m);
      const double d = f64_norm(a[e] - g[e], m);
      const uint32_t sraw = i + (2 * N - shift_pow);
      tree_hi[sraw & (N - 1)] = (sraw & N) ? -d : d;
    }
  }
}

// Last level of the expansion tree fused with the selectors' forward NTT (fp64 flavours).  One workgroup per
// (tree ciphertext, component, data modulus, half): divide-and-round of the key-switch product, + sigma_g(c0), the
// tree butterfly lo = a + g (half 0) or hi = x^(-2^j) (a - g) (half 1) -- exactly ks_combine_f64_kernel's arithmetic
// -- but the output polynomial stays in registers / LDS and goes straight through the forward transform into the
// queries' selection vectors: the last level's ciphertexts are never written to HBM in coefficient form and never
// read back (reference server.cpp:137-141 followed by database.cpp:190,222).  Tree ciphertext index = slot * B +
// query; lo is selector `slot`, hi selector `slot + shift_pow`; selectors >= n_items are not produced
// (server.cpp:144).  The two halves recompute g (30 instructions per element) rather than keeping it across a
// transform: one transform per workgroup, no value live across it, 4 waves per SIMD.
template <int MODE, bool P40>
__global__ void __launch_bounds__(NT)
ks_last_level_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ tree_raw,
                     const uint64_t* __restrict__ prod, uint32_t galois_elt, uint32_t shift_pow, uint32_t n_items,
                     uint32_t B, MfmaPtrs dst) {
  using A = Arith<MODE>;
  static_assert(MODE != kNttInt, "fp64 flavours only");
  double* sd = reinterpret_cast<double*>(smem_raw);
  const uint32_t tid = threadIdx.x, k = P->k, km = k + 1;
  const uint32_t half_id = blockIdx.x & 1, bx = blockIdx.x >> 1;   // the two halves of a polynomial run back to back
  const uint32_t j = bx % k, comp = (bx / k) & 1, ct = bx / (2 * k);
  const uint32_t slot = ct / B, q = ct % B;
  const uint32_t out_slot = half_id ? slot + shift_pow : slot;
  if (out_slot >= n_items) return;  // uniform per workgroup
  const typename A::Mod m = A::mod(P, j);
  const double pf = P->p_f, half = P->p_half_f, pinv = P->p_inv_f[j];
  const double* tree = reinterpret_cast<const double*>(tree_raw) + (size_t)ct * 2 * k * N;
  // g = round(S / p) mod q_j, signed representative (ks_combine_f64_kernel)
  double g[EPT];
  [[maybe_unused]] Hi16 hs{}, hj{};
  if constexpr (P40) {
    const uint8_t* pr = reinterpret_cast<const uint8_t*>(prod) + ((size_t)ct * 2 + comp) * km * kPoly40;
    hs = load40f_hi(pr + (size_t)k * kPoly40, tid);
    hj = load40f_hi(pr + (size_t)j * kPoly40, tid);
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    [[maybe_unused]] const uint32_t i = e * NT + tid;
    double sp, dj;
    if constexpr (P40) {
      const uint8_t* pr = reinterpret_cast<const uint8_t*>(prod) + ((size_t)ct * 2 + comp) * km * kPoly40;
      sp = load40f(pr + (size_t)k * kPoly40, hs, e, tid, f64_pack_magic(pf));
      dj = load40f(pr + (size_t)j * kPoly40, hj, e, tid, f64_pack_magic(m.q));
    } else {
      const double* pr = reinterpret_cast<const double*>(prod) + ((size_t)ct * 2 + comp) * km * N;
      sp = pr[(size_t)k * N + i];
      dj = pr[(size_t)j * N + i];
    }
    sp = sp > half ? sp - pf : sp;
    sp = sp < -half ? sp + pf : sp;
    g[e] = f64_mulmod(dj - f64_norm(sp, m), pinv, m);
  }
  if (comp == 0) {  // + sigma_g(c0): scatter c0 through LDS (GaloisTool::apply_galois index map), read back in order
    const double* c0 = tree + (size_t)j * N;
    const uint32_t raw0 = tid * galois_elt, rstep = (uint32_t)NT * galois_elt;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const uint32_t raw = raw0 + (uint32_t)e * rstep;
      const uint64_t sign = (uint64_t)((raw << (31 - LOGN)) & 0x80000000u) << 32;
      sd[lds_idx<R_>(raw & (N - 1))] =
          __longlong_as_double((long long)((uint64_t)__double_as_longlong(c0[e * NT + tid]) ^ sign));
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) g[e] += sd[lds_lin_base<NT, R_>(tid) + lds_lin_off<NT, R_>(e)];
    __syncthreads();
  }
  const double* a = tree + ((size_t)comp * k + j) * N;
  typename A::T x[EPT];
  if (!half_id) {  // lo = a + g -> selector `slot`
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = f64_norm(a[e * NT + tid] + g[e], m);
  } else {         // hi = x^(-2^j) (a - g): negacyclic rotation by 2N - 2^j through LDS -> selector `slot + 2^j`
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const uint32_t i = e * NT + tid;
      const double d = f64_norm(a[i] - g[e], m);
      const uint32_t sraw = i + (2 * N - shift_pow);
      sd[lds_idx<R_>(sraw & (N - 1))] = (sraw & N) ? -d : d;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = sd[lds_lin_base<NT, R_>(tid) + lds_lin_off<NT, R_>(e)];
    __syncthreads();
  }
  ntt_forward<MODE, LOGN, kPF>(x, smem_raw, P, j, tid);
  uint64_t* out = (uint64_t*)dst.p[q] + ((size_t)out_slot * 2 * k + comp * k + j) * N;
#pragma unroll
  for (int e = 0; e < EPT; ++e) out[e * NT + tid] = A::out(x[e], m);
}

// ---- last expansion level in the NTT domain (fp64 flavours) ----
//
// The last level's outputs are only ever used in NTT form, and every step after the key-switch products is linear
// over Z_q[x]/(x^N+1), so it commutes with the transform:
//   NTT(lo) = A + G,   NTT(hi) = X (.) (A - G),        A = NTT(a), X = NTT(x^(-2^j)) (a table),
//   G = (S_j - NTT_j(lift(s))) p^-1 (+ NTT(sigma_g(a_0)) for component 0),
// with S_j the dyadic key-switch product mod q_j (never inverse-transformed) and s the centred special-prime
// residue of the product (coefficient form, from ks_mac_intt_kernel with I_base = k).  sigma_g acts on NTT-form data as the permutation
// pi_g(P) = br(((2 br(P) + 1) g mod 2N - 1) / 2) of the SEAL positions P (position P holds the evaluation at
// psi^(2 br(P) + 1)), so NTT(sigma_g(a_0)) = A_0 o pi_g, and A_1 is already there: the digit kernel's
// dig[n][j][j] = NTT_j(sigma_g(a_1) mod q_j) = A_1 o pi_g, hence A_1 = dig[n][j][j] o pi_(g^-1).  Per tree
// ciphertext that is k forward transforms (A_0, tree_c0_ntt_kernel) + 2 inverse (s) + 2k forward (lift(s)) after the
// digits, against 2(k+1) inverse + 4k forward for ks_mac_intt_kernel + ks_last_level_kernel: 8 instead of 14 at k = 2.
// All arithmetic is exact mod q_j, so the selectors are the same canonical residues (reference server.cpp:137-141
// followed by database.cpp:190,222).

// One workgroup per (tree ciphertext, data modulus j, component): one forward transform (of the lifted special
// residue), then the dyadic products and the epilogue above; writes selector `slot` and -- if < n_items --
// selector `slot + shift_pow` of query q (tree ciphertext index = slot * B + query).  xpow = X for the k data
// moduli, [k][N] doubles in device order.  grid = (nodes, k, 2) or the XCD-aware 1-D equivalent.
// C0T: where NTT(a_0) comes from -- 0: the product buffer's spare slot (tree_c0_ntt_kernel / the digit launch's extra
// workgroups), 1 / 2: the tree itself, whose c0 polynomials are in NTT form (doubles / 5-byte form; ks_combine_c0_ntt).
template <int MODE, bool P40, bool OUTF64 = false, int C0T = 0>
__global__ void __launch_bounds__(NT)
ks_last_ntt_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ dig_raw,
                   KeyPtrs keys, const uint64_t* __restrict__ prod,
                   const double* __restrict__ xpow, uint32_t galois_elt, uint32_t galois_inv, uint32_t shift_pow,
                   uint32_t n_items, uint32_t B, MfmaPtrs dst, const uint64_t* __restrict__ tree_raw,
                   const uint16_t* __restrict__ perm) {   // perm: [2][N], sigma_g then sigma_(g^-1) (galois_perm_table)
  using A = Arith<MODE>;
  static_assert(MODE != kNttInt, "fp64 flavours only");
  double* sd = reinterpret_cast<double*>(smem_raw);
  const uint32_t tid = threadIdx.x, k = P->k, km = k + 1;
  uint32_t node = blockIdx.x, j = blockIdx.y, comp = blockIdx.z;
  if (gridDim.y == 1 && gridDim.z == 1) {
    const uint32_t xcd = blockIdx.x & 7, t = blockIdx.x >> 3;
    comp = t & 1;
    const uint32_t u = t >> 1;
    j = u % k;
    node = (u / k) * 8 + xcd;
  }
  const typename A::Mod m = A::mod(P, j);
  const uint32_t slot = node / B, q = node % B;
  // t = NTT_j(lift(s))
  double x[EPT];
  {
    const double pf = P->p_f, half = P->p_half_f;
    const size_t spoly = ((size_t)node * 2 + comp) * km + k;
    [[maybe_unused]] Hi16 hs{};
    if constexpr (P40) hs = load40f_hi(reinterpret_cast<const uint8_t*>(prod) + spoly * kPoly40, tid);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      double sp;
      if constexpr (P40)
        sp = load40f(reinterpret_cast<const uint8_t*>(prod) + spoly * kPoly40, hs, e, tid, f64_pack_magic(pf));
      else
        sp = reinterpret_cast<const double*>(prod)[spoly * N + e * NT + tid];
      sp = sp > half ? sp - pf : sp;   // exact centring (ks_combine_f64_kernel)
      sp = sp < -half ? sp + pf : sp;
      x[e] = f64_norm(sp, m);
    }
  }
  ntt_forward<MODE, LOGN, kPF, /*CANON=*/false>(x, smem_raw, P, j, tid);
  // G = (S_j - t) p^-1
  const size_t dpoly0 = ((size_t)node * km + j) * k;
  const double magic = f64_pack_magic(m.q);
  // polynomial `poly` of a 5-byte / double buffer, this thread's 16 elements
  auto load_poly = [&](const uint64_t* base, size_t poly, double (&out)[EPT], uint32_t t) {
    if constexpr (P40) {
      const uint8_t* pp = reinterpret_cast<const uint8_t*>(base) + poly * kPoly40;
      const Hi16 h = load40f_hi(pp, t);
#pragma unroll
      for (int e = 0; e < EPT; ++e) out[e] = load40f(pp, h, e, t, magic);
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) out[e] = reinterpret_cast<const double*>(base)[poly * N + e * NT + t];
    }
  };
  {
    // the products accumulate onto -t in place: a second accumulator array next to x leaves too few registers for
    // the digit and key loads of one J to be in flight together (the MAC then costs more than the transform)
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = -x[e];
    const double* key = reinterpret_cast<const double*>(keys.p[node % keys.B]);   // the key of this node's query
    for (uint32_t J = 0; J < k; ++J) {
      const double* kj = key + (((size_t)J * 2 + comp) * km + j) * N;
      double d[EPT];
      load_poly(dig_raw, dpoly0 + J, d, tid);
#pragma unroll
      for (int e = 0; e < EPT; ++e) x[e] += f64_mulmod(d[e], kj[e * NT + tid], m);
    }
    const double pinv = P->p_inv_f[j];
#pragma unroll
    for (int e = 0; e < EPT; ++e) x[e] = f64_mulmod(f64_norm(x[e], m), pinv, m);
  }
  // Everything below is indexed through an opaque copy of the thread index: the compiler otherwise computes the
  // epilogue's 16 permuted LDS addresses and its global addresses BEFORE the product loop, and with those ~40
  // registers live it serialises the loop's loads (one s_waitcnt vmcnt(0) per load: the products then cost more
  // than the transform)
  uint32_t tid_e = tid;
  asm volatile("" : "+v"(tid_e));
  // A (and, for component 0, A_0 o pi_g) through an LDS permutation
  double a[EPT];
  {
    double v[EPT];
    if (comp == 0) {
      if constexpr (C0T == 0) load_poly(prod, (size_t)node * 2 * km + j, v, tid_e);
      else tree_load<C0T == 2>(tree_raw, (size_t)node * 2 * k + j, tid_e, m.q, v);
    } else {
      load_poly(dig_raw, dpoly0 + j, v, tid_e);
    }
    __syncthreads();  // the transform's last exchange is done with the LDS words
#pragma unroll
    for (int e = 0; e < EPT; ++e) sd[lds_lin_base<NT, R_>(tid_e) + lds_lin_off<NT, R_>(e)] = v[e];
    __syncthreads();
    const uint32_t gel = comp == 0 ? galois_elt : galois_inv;
    [[maybe_unused]] PermIdx pi{};
    if constexpr (PIRGPU_PERM_TABLE != 0) pi = load_perm(perm + (comp == 0 ? 0 : N), tid_e);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const double w = sd[perm_at(pi, e, tid_e, gel)];
      if (comp == 0) {
        a[e] = v[e];
        x[e] += w;
      } else {
        a[e] = w;
      }
    }
  }
  const size_t opoly = (size_t)comp * k + j;
  uint64_t* lo = (uint64_t*)dst.p[q] + ((size_t)slot * 2 * k + opoly) * N;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const double r = f64_canon(f64_norm(a[e] + x[e], m), m);
    if constexpr (OUTF64) reinterpret_cast<double*>(lo)[e * NT + tid_e] = r;   // lane-internal selectors: exact doubles
    else lo[e * NT + tid_e] = A::out(r, m);
  }
  if (slot + shift_pow < n_items) {
    uint64_t* hi = (uint64_t*)dst.p[q] + ((size_t)(slot + shift_pow) * 2 * k + opoly) * N;
    const double* X = xpow + (size_t)j * N;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const double r = f64_canon(f64_mulmod(f64_norm(a[e] - x[e], m), X[e * NT + tid_e], m), m);
      if constexpr (OUTF64) reinterpret_cast<double*>(hi)[e * NT + tid_e] = r;
      else hi[e * NT + tid_e] = A::out(r, m);
    }
  }
}

// Upper recursion level, fused: for output slot (row r, source ciphertext cc, Encode chunk
// e_idx, target residue jt) and a chunk of the row's children,
//   part[chunk][r][cc*E+e_idx][p][jt] = sum_{ii in chunk} sv[sv_first+ii][p][jt] (.)
//                                        NTT_jt(lift(Encode_e_idx(child[(r*n_dim+ii)*C+cc])))
// i.e. CiphertextReencoder::Encode + transform_to_ntt_inplace(Plaintext) + multiply_plain +
// add_inplace of database.cpp:218-247 without materialising the re-encoded plaintexts: the
// transformed plaintext stays in registers and is multiplied into both polynomials of the
// selector.  grid = (n_rows*C*n_chunks, E, k); partial sums are folded by reduce_splits_kernel, then ntt_batch_kernel (inverse).
template <int MODE, bool LDS_TW = false, bool SELF64 = false>
__global__ void __launch_bounds__(NT)
upper_fused_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src_all, MfmaPtrs svq,
                   uint64_t* __restrict__ part_all, uint32_t n_rows, uint32_t n_dim,
                   uint32_t n_children_total, uint32_t sv_first, uint32_t C, uint32_t chunk_len,
                   uint32_t n_chunks, uint64_t src_qstride, uint64_t part_qstride) {
  using A = Arith<MODE>;
  using T = typename A::T;
  const uint32_t tid = threadIdx.x, k = P->k, E = P->enc_count;
  // blockIdx.x = ((query * n_rows + r) * C + cc) * n_chunks + chunk: the queries of a group share one launch,
  // each with its own child ciphertexts (src_all + q * src_qstride), selectors (svq.p[q]) and partial sums
  const uint32_t per_query = n_rows * C * n_chunks;
  const uint32_t qi = blockIdx.x / per_query, bx = blockIdx.x % per_query;
  const uint64_t* src = src_all + (size_t)qi * src_qstride;
  const uint64_t* sv = reinterpret_cast<const uint64_t*>(svq.p[qi]);
  uint64_t* part = part_all + (size_t)qi * part_qstride;
  const uint32_t chunk = bx % n_chunks;
  const uint32_t cc = (bx / n_chunks) % C;
  const uint32_t r = bx / (n_chunks * C);
  const uint32_t e_idx = blockIdx.y, jt = blockIdx.z;
  const ModConst mc = P->mod[jt];
  const typename A::Mod m = A::mod(P, jt);
  const uint32_t sp = P->enc_poly[e_idx], sj = P->enc_res[e_idx], sh = P->enc_shift[e_idx];
  const uint64_t mask = (1ull << P->enc_bits) - 1;
  const uint64_t thr = P->plain_thr;
  const uint64_t inc = P->lift_inc[jt] >= mc.q ? P->lift_inc[jt] - mc.q : P->lift_inc[jt];
  const uint32_t child0 = r * n_dim;
  uint32_t nchild = n_children_total > child0 ? n_children_total - child0 : 0;
  if (nchild > n_dim) nchild = n_dim;
  const uint32_t ii0 = chunk * chunk_len;
  const uint32_t ii1 = ii0 + chunk_len < nchild ? ii0 + chunk_len : nchild;

  // lazy accumulators: fp64 sums of signed representatives (|term| <= 0.7 q, renormalised
  // every 8 terms) or exact 128-bit integer sums (folded every lazy_limit terms)
  T acc0[EPT], acc1[EPT];
  u128 wide0[MODE == kNttInt ? EPT : 1], wide1[MODE == kNttInt ? EPT : 1];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    acc0[e] = 0;
    acc1[e] = 0;
    if constexpr (MODE == kNttInt) wide0[e] = wide1[e] = 0;
  }
  // fp64 flavours, usual parameters (chunks of <= 31 bits, t < q_jt): the chunk fits a u32, converts with one
  // v_cvt, and the plain lift is "subtract t above the threshold" on a signed representative -- no 64-bit
  // integer reduction anywhere on the input side
  const bool fast_lift = MODE != kNttInt && P->enc_bits <= 31 && P->t < mc.q;
  const double td = (double)P->t;
  const uint32_t thr32 = (uint32_t)thr, mask32 = (uint32_t)mask;
  uint32_t since = 0;
  if constexpr (LDS_TW) {
    // Pipelined variant (fp64 flavours, rings whose twiddle table fits LDS next to the exchange buffer).  Getting a
    // child's 2 x 16 selector words into registers costs ~2 us of exposed HBM latency per child with only two waves per
    // SIMD to hide it (98 of the kernel's 349 us at cfg 3), and a prefetch across the transform
    // does not work while the transform loads its twiddles from global memory: vmcnt completes in order, so its first
    // twiddle wait would wait for the prefetch too.  Here the workgroup copies the modulus' forward twiddle table to
    // LDS once, the transform contains no vector memory instruction, and the selector words of child ii (and the
    // source words of child ii + 1) are requested BEFORE the transform of child ii and arrive under it.
    static_assert(MODE != kNttInt, "fp64 flavours only");
    double* ltw = reinterpret_cast<double*>(smem_raw + kLdsBytes);
    lds_twiddles_fill<LOGN>(ltw, A::tw(P, jt), tid);
    uint64_t in_raw[EPT];
    auto load_in = [&](uint32_t ii) {
      const __amdgpu_buffer_rsrc_t in = poly_rsrc(src + ((((size_t)(child0 + ii) * C + cc) * 2 + sp) * k + sj) * N);
#pragma unroll
      for (int e = 0; e < EPT; ++e) in_raw[e] = poly_load_u64(in, tid, e);
    };
    if (ii0 < ii1) load_in(ii0);
    for (uint32_t ii = ii0; ii < ii1; ++ii) {
      const __amdgpu_buffer_rsrc_t s0 = poly_rsrc(sv + (((size_t)(sv_first + ii) * 2 + 0) * k + jt) * N);
      const __amdgpu_buffer_rsrc_t s1 = poly_rsrc(sv + (((size_t)(sv_first + ii) * 2 + 1) * k + jt) * N);
      T x[EPT];
      if (fast_lift) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const uint32_t v = (uint32_t)(in_raw[e] >> sh) & mask32;
          const double d = (double)v;
          x[e] = v >= thr32 ? d - td : d;  // m >= (t+1)/2 -> m + q - t == m - t (mod q): SURVEY App. A.5
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          uint64_t v = (in_raw[e] >> sh) & mask;
          uint64_t rr = reduce64(v, mc);
          if (v >= thr) rr = add_mod(rr, inc, mc.q);
          x[e] = A::in(rr, m);
        }
      }
      uint64_t s0r[EPT], s1r[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        s0r[e] = poly_load_u64(s0, tid, e);
        s1r[e] = poly_load_u64(s1, tid, e);
      }
      if (ii + 1 < ii1) load_in(ii + 1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // previous iteration's transform may still be reading LDS (and, first time, the table copy)
      ntt_forward_tw<MODE, LOGN, false, /*CANON=*/false>(x, smem_raw, P, jt, tid, LdsTw{ltw});
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        // SELF64: the group's selectors were written as exact doubles by ks_last_ntt_kernel (no conversion)
        acc0[e] += f64_mulmod(x[e], SELF64 ? __longlong_as_double((long long)s0r[e]) : f64_from_u64(s0r[e]), m);
        acc1[e] += f64_mulmod(x[e], SELF64 ? __longlong_as_double((long long)s1r[e]) : f64_from_u64(s1r[e]), m);
      }
      if (++since == 8) {
        since = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          acc0[e] = f64_norm(acc0[e], m);
          acc1[e] = f64_norm(acc1[e], m);
        }
      }
    }
  } else if constexpr (MODE != kNttInt && PIRGPU_UPPER_PLAIN_PF != 0) {
    // Plain form with the NEXT child's source words requested before this child's transform (16 buffer loads, 32
    // registers across the transform): with one workgroup per CU (N = 8192: 204 registers, 512 threads) a child is three
    // exposed memory round trips -- source, twiddles, selectors -- and this takes the first one out of the chain.
    uint64_t in_raw[EPT];
    auto load_in = [&](uint32_t ii) {
      const __amdgpu_buffer_rsrc_t in = poly_rsrc(src + ((((size_t)(child0 + ii) * C + cc) * 2 + sp) * k + sj) * N);
#pragma unroll
      for (int e = 0; e < EPT; ++e) in_raw[e] = poly_load_u64(in, tid, e);
    };
    if (ii0 < ii1) load_in(ii0);
    for (uint32_t ii = ii0; ii < ii1; ++ii) {
      T x[EPT];
      if (fast_lift) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const uint32_t v = (uint32_t)(in_raw[e] >> sh) & mask32;
          const double d = (double)v;
          x[e] = v >= thr32 ? d - td : d;
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          uint64_t v = (in_raw[e] >> sh) & mask;
          uint64_t rr = reduce64(v, mc);
          if (v >= thr) rr = add_mod(rr, inc, mc.q);
          x[e] = A::in(rr, m);
        }
      }
      const __amdgpu_buffer_rsrc_t s0 = poly_rsrc(sv + (((size_t)(sv_first + ii) * 2 + 0) * k + jt) * N);
      const __amdgpu_buffer_rsrc_t s1 = poly_rsrc(sv + (((size_t)(sv_first + ii) * 2 + 1) * k + jt) * N);
      [[maybe_unused]] uint64_t s0r[EPT];
      if constexpr (PIRGPU_UPPER_PLAIN_PF >= 2) {   // ... and this child's first selector polynomial (32 more registers)
#pragma unroll
        for (int e = 0; e < EPT; ++e) s0r[e] = poly_load_u64(s0, tid, e);
      }
      if (ii + 1 < ii1) load_in(ii + 1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // previous iteration's transform may still be reading LDS
      ntt_forward<MODE, LOGN, false, /*CANON=*/false>(x, smem_raw, P, jt, tid);
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const uint64_t w0 = PIRGPU_UPPER_PLAIN_PF >= 2 ? s0r[e] : poly_load_u64(s0, tid, e), w1 = poly_load_u64(s1, tid, e);
        acc0[e] += f64_mulmod(x[e], SELF64 ? __longlong_as_double((long long)w0) : f64_from_u64(w0), m);
        acc1[e] += f64_mulmod(x[e], SELF64 ? __longlong_as_double((long long)w1) : f64_from_u64(w1), m);
      }
      if (++since == 8) {
        since = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          acc0[e] = f64_norm(acc0[e], m);
          acc1[e] = f64_norm(acc1[e], m);
        }
      }
    }
  } else
  for (uint32_t ii = ii0; ii < ii1; ++ii) {
    const uint64_t* in = src + ((((size_t)(child0 + ii) * C + cc) * 2 + sp) * k + sj) * N;
    T x[EPT];
    if constexpr (MODE != kNttInt) {
      if (fast_lift) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const uint32_t v = (uint32_t)(in[e * NT + tid] >> sh) & mask32;
          const double d = (double)v;
          x[e] = v >= thr32 ? d - td : d;  // m >= (t+1)/2 -> m + q - t == m - t (mod q): SURVEY App. A.5
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          uint64_t v = (in[e * NT + tid] >> sh) & mask;
          uint64_t rr = reduce64(v, mc);
          if (v >= thr) rr = add_mod(rr, inc, mc.q);
          x[e] = A::in(rr, m);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        uint64_t v = (in[e * NT + tid] >> sh) & mask;
        uint64_t rr = reduce64(v, mc);
        if (v >= thr) rr = add_mod(rr, inc, mc.q);
        x[e] = A::in(rr, m);
      }
    }
    __syncthreads();  // previous iteration's transform may still be reading LDS
    // no twiddle prefetch: the accumulators need the registers; signed representatives suffice for the products
    ntt_forward<MODE, LOGN, false, /*CANON=*/MODE == kNttInt>(x, smem_raw, P, jt, tid);
    const uint64_t* s0 = sv + (((size_t)(sv_first + ii) * 2 + 0) * k + jt) * N;
    const uint64_t* s1 = sv + (((size_t)(sv_first + ii) * 2 + 1) * k + jt) * N;
    if constexpr (MODE == kNttInt) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        wide0[e] += (u128)x[e] * s0[e * NT + tid];
        wide1[e] += (u128)x[e] * s1[e * NT + tid];
      }
      if (++since == P->lazy_limit) {
        since = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          wide0[e] = reduce128((uint64_t)wide0[e], (uint64_t)(wide0[e] >> 64), mc);
          wide1[e] = reduce128((uint64_t)wide1[e], (uint64_t)(wide1[e] >> 64), mc);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const uint64_t w0 = s0[e * NT + tid], w1 = s1[e * NT + tid];
        acc0[e] += f64_mulmod(x[e], SELF64 ? __longlong_as_double((long long)w0) : f64_from_u64(w0), m);
        acc1[e] += f64_mulmod(x[e], SELF64 ? __longlong_as_double((long long)w1) : f64_from_u64(w1), m);
      }
      if (++since == 8) {
        since = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          acc0[e] = f64_norm(acc0[e], m);
          acc1[e] = f64_norm(acc1[e], m);
        }
      }
    }
  }
  const size_t slot = (((size_t)chunk * n_rows + r) * C + cc) * E + e_idx;
  uint64_t* o0 = part + ((slot * 2 + 0) * k + jt) * N;
  uint64_t* o1 = part + ((slot * 2 + 1) * k + jt) * N;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    if constexpr (MODE == kNttInt) {
      o0[e * NT + tid] = reduce128((uint64_t)wide0[e], (uint64_t)(wide0[e] >> 64), mc);
      o1[e * NT + tid] = reduce128((uint64_t)wide1[e], (uint64_t)(wide1[e] >> 64), mc);
    } else {
      o0[e * NT + tid] = f64_to_u64(f64_canon(f64_norm(acc0[e], m), m));
      o1[e * NT + tid] = f64_to_u64(f64_canon(f64_norm(acc1[e], m), m));
    }
  }
}

// Upper recursion level for ring degrees whose fused kernel cannot keep two accumulator sets in registers
// (N = 16384: 1024-thread workgroups, 128 VGPRs -> upper_fused_kernel spills 128 registers), part 1 of 2: re-encode
// chunk e_idx of one child ciphertext, plain lift, forward NTT for target modulus jt, and STORE the transformed
// plaintext (doubles, signed representatives) instead of multiplying it at once; upper_mac_kernel (kernels.hip)
// then forms the products with the selectors elementwise.  Costs a round trip of the transformed plaintexts through
// HBM (bounded by processing the children in blocks) and saves the spills: fp64 flavours only.
// grid = (queries * n_rows * C * blk, E, k); child ii = b0 + (blockIdx.x % blk) of row r.
// LOOP: one workgroup per (child, SOURCE polynomial) -- grid.y = 2 k, grid.z = 1 -- loads the source once and runs the
// transforms of all its Encode chunks under all k target moduli one after the other (3 x 4 = 12 at cfg 5); the store of
// one drains under the next (see ks_digit_kernel, LOOPI: the overlap a ring with one workgroup per CU does not get
// from neighbouring workgroups).
template <int MODE, bool LOOP = false>
__global__ void __launch_bounds__(NT) PIRGPU_FOUR_WAVES
upper_ntt_kernel(const DevParams* __restrict__ P, const uint64_t* __restrict__ src_all, double* __restrict__ scratch,
                 uint32_t n_rows, uint32_t n_dim, uint32_t n_children_total, uint32_t C, uint32_t b0, uint32_t blk,
                 uint64_t src_qstride) {
  using A = Arith<MODE>;
  static_assert(MODE != kNttInt, "fp64 flavours only");
  const uint32_t tid = threadIdx.x, k = P->k, E = P->enc_count;
  const uint32_t iib = blockIdx.x % blk;
  const uint32_t cc = (blockIdx.x / blk) % C;
  const uint32_t r = (blockIdx.x / (blk * C)) % n_rows;
  const uint32_t qi = blockIdx.x / (blk * C * n_rows);
  const uint32_t ii = b0 + iib;
  const uint32_t child0 = r * n_dim;
  uint32_t nchild = n_children_total > child0 ? n_children_total - child0 : 0;
  if (nchild > n_dim) nchild = n_dim;
  // chunks [e_lo, e_hi) x target moduli [jt_lo, jt_hi) of this workgroup; every chunk of the range reads the same source
  uint32_t e_lo = blockIdx.y, e_hi = blockIdx.y + 1, jt_lo = blockIdx.z, jt_hi = blockIdx.z + 1;
  if constexpr (LOOP) {
    const uint32_t want_p = blockIdx.y / k, want_j = blockIdx.y % k;
    e_lo = E, e_hi = 0;
    for (uint32_t e = 0; e < E; ++e)   // Encode order is (polynomial, residue, chunk): one contiguous range
      if (P->enc_poly[e] == want_p && P->enc_res[e] == want_j) {
        e_lo = e < e_lo ? e : e_lo;
        e_hi = e + 1;
      }
    jt_lo = 0, jt_hi = k;
    if (e_lo >= e_hi) return;
  }
  double* out0 = scratch + ((((size_t)qi * n_rows + r) * C + cc) * blk + iib) * E * k * N;   // [chunk][target modulus][N]
  if (ii >= nchild) {  // beyond the database: contributes nothing (uniform per workgroup)
    for (uint32_t e_idx = e_lo; e_idx < e_hi; ++e_idx)
      for (uint32_t jt = jt_lo; jt < jt_hi; ++jt) {
        double* out = out0 + ((size_t)e_idx * k + jt) * N;
#pragma unroll
        for (int e = 0; e < EPT; ++e) out[e * NT + tid] = 0.0;
      }
    return;
  }
  const uint32_t sp = P->enc_poly[e_lo], sj = P->enc_res[e_lo];
  const uint64_t mask = (1ull << P->enc_bits) - 1;
  const uint64_t thr = P->plain_thr;
  const double td = (double)P->t;
  const uint32_t thr32 = (uint32_t)thr, mask32 = (uint32_t)mask;
  const uint64_t* in = src_all + (size_t)qi * src_qstride + ((((size_t)(child0 + ii) * C + cc) * 2 + sp) * k + sj) * N;
  uint64_t in_raw[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) in_raw[e] = in[e * NT + tid];
  bool first = true;
  // target modulus outside, chunk inside: the chunk extraction then changes every iteration and stays a few
  // instructions in front of the transform instead of being hoisted into 32 more live registers
#pragma clang loop unroll(disable)
  for (uint32_t jt = jt_lo; jt < jt_hi; ++jt) {
    const ModConst mc = P->mod[jt];
    const typename A::Mod m = A::mod(P, jt);
    const uint64_t inc = P->lift_inc[jt] >= mc.q ? P->lift_inc[jt] - mc.q : P->lift_inc[jt];
    const bool fast_lift = P->enc_bits <= 31 && P->t < mc.q;
#pragma clang loop unroll(disable)
    for (uint32_t e_idx = e_lo; e_idx < e_hi; ++e_idx) {
      const uint32_t sh = P->enc_shift[e_idx];
      typename A::T x[EPT];
      if (fast_lift) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const uint32_t v = (uint32_t)(in_raw[e] >> sh) & mask32;
          const double d = (double)v;
          x[e] = v >= thr32 ? d - td : d;
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          uint64_t v = (in_raw[e] >> sh) & mask;
          uint64_t rr = reduce64(v, mc);
          if (v >= thr) rr = add_mod(rr, inc, mc.q);
          x[e] = A::in(rr, m);
        }
      }
      if (LOOP && !first) __syncthreads();  // the previous transform may still be reading LDS
      first = false;
      uint32_t tl = tid;
      if constexpr (LOOP) asm volatile("" : "+v"(tl));   // see ks_digit_kernel, LOOPI: no address hoisting across the loop
      ntt_forward<MODE, LOGN, LOOP ? kPFLoop : kPF, /*CANON=*/false>(x, smem_raw, P, jt, tl);
      double* out = out0 + ((size_t)e_idx * k + jt) * N;
#pragma unroll
      for (int e = 0; e < EPT; ++e) out[e * NT + tl] = x[e];
    }
  }
}

// ------------------------------------------------------------------ host side

#define PIRGPU_BY_MODE(mode, EXPR)                                \
  switch (mode) {                                                 \
    case kNttInt: { constexpr int MODE = kNttInt; EXPR; } break;  \
    case kNttF64: { constexpr int MODE = kNttF64; EXPR; } break;  \
    default: { constexpr int MODE = kNttF64Wide; EXPR; } break;   \
  }

template <int MODE>
static hipError_t configure_mode() {
  const int bytes = (int)kLdsBytes;  // beyond 64 KiB for N = 16384 (136 KiB of the CU's 160 KiB)
  hipError_t e;
#define PIRGPU_SET(K) \
  if ((e = hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, bytes))) return e
  PIRGPU_SET((ntt_batch_kernel<MODE, false>));
  PIRGPU_SET((ntt_batch_kernel<MODE, true>));
  PIRGPU_SET(ntt_inv_gather_kernel<MODE>);
  PIRGPU_SET((ct_ntt_fwd_oop_kernel<MODE, false>));
  PIRGPU_SET((ct_ntt_fwd_oop_kernel<MODE, true>));
  PIRGPU_SET(ct_ntt_fwd_split_kernel<MODE>);
  PIRGPU_SET(db_encode_kernel<MODE>);
  PIRGPU_SET((ks_digit_kernel<MODE, false>));
  PIRGPU_SET((ks_digit_kernel<MODE, true>));
  PIRGPU_SET((ks_mac_intt_kernel<MODE, false>));
  PIRGPU_SET((ks_mac_intt_kernel<MODE, true>));
  PIRGPU_SET(upper_fused_kernel<MODE>);
  if constexpr (MODE != kNttInt && kUpperLdsTw) {
    if ((e = hipFuncSetAttribute((const void*)upper_fused_kernel<MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 bytes + N * 8)))
      return e;
    if ((e = hipFuncSetAttribute((const void*)upper_fused_kernel<MODE, true, true>,
                                 hipFuncAttributeMaxDynamicSharedMemorySize, bytes + N * 8)))
      return e;
  }
  if constexpr (MODE != kNttInt) {
    PIRGPU_SET(upper_ntt_kernel<MODE>);
    PIRGPU_SET((ks_mac_combine_kernel<MODE, false>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, false, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, true, false>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, true, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, false, false, false, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, false, false, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, false, true, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, true, false, true>));
    PIRGPU_SET((ks_mac_combine_kernel<MODE, true, true, true, true>));
    PIRGPU_SET(tree_c0_fwd_kernel<MODE>);
    PIRGPU_SET((ks_c0_ntt_kernel<MODE, false, false, false>));
    PIRGPU_SET((ks_c0_ntt_kernel<MODE, true, false, false>));
    PIRGPU_SET((ks_c0_ntt_kernel<MODE, true, false, true>));
    PIRGPU_SET((ks_c0_ntt_kernel<MODE, true, true, false>));
    PIRGPU_SET((ks_c0_ntt_kernel<MODE, true, true, true>));
    PIRGPU_SET((ks_digit_kernel<MODE, true, true>));
    PIRGPU_SET((ks_digit_kernel<MODE, false, false, true>));
    PIRGPU_SET((ks_digit_kernel<MODE, true, false, true>));
    PIRGPU_SET((ks_digit_kernel<MODE, true, true, true>));
    PIRGPU_SET((upper_ntt_kernel<MODE, true>));
    PIRGPU_SET((ks_last_level_kernel<MODE, false>));
    PIRGPU_SET((ks_last_level_kernel<MODE, true>));
    PIRGPU_SET((tree_c0_ntt_kernel<MODE, false>));
    PIRGPU_SET((tree_c0_ntt_kernel<MODE, true>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, false>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, true>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, false, true>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, true, true>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, false, false, 1>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, false, true, 1>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, true, false, 1>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, true, true, 1>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, true, false, 2>));
    PIRGPU_SET((ks_last_ntt_kernel<MODE, true, true, 2>));
    PIRGPU_SET((upper_fused_kernel<MODE, false, true>));
  }
#undef PIRGPU_SET
  return hipSuccess;
}

static hipError_t op_configure(int mode) {
  PIRGPU_BY_MODE(mode, return configure_mode<MODE>());
  return hipSuccess;
}

static hipError_t op_ntt_batch(hipStream_t st, int mode, const DevParams* P, uint64_t* data, uint64_t n_polys,
                               uint32_t mod_period, uint32_t mod_base, bool inverse) {
  if (inverse) {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ntt_batch_kernel<MODE, true>), dim3((uint32_t)n_polys), dim3(NT),
                                            kLdsBytes, st, P, data, mod_period, mod_base));
  } else {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ntt_batch_kernel<MODE, false>), dim3((uint32_t)n_polys), dim3(NT),
                                            kLdsBytes, st, P, data, mod_period, mod_base));
  }
  return hipGetLastError();
}

static hipError_t op_ntt_inv_gather(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* src,
                                    uint64_t* dst, const SliceMap& map, uint32_t RC, uint32_t nq, uint32_t nq_total,
                                    uint32_t q0) {
  if (map.n == 0 || map.n > (uint32_t)kMaxSlices || map.cut[0] != 0 || map.cut[map.n] != k * (uint32_t)N || q0 + nq > nq_total)
    return hipErrorInvalidValue;
  for (uint32_t r = 0; r <= map.n; ++r)
    if (map.cut[r] % NT) return hipErrorInvalidValue;   // the caller falls back to launch_slots_assemble + ntt_batch
  PIRGPU_BY_MODE(mode, hipLaunchKernelGGL(ntt_inv_gather_kernel<MODE>, dim3(nq * RC * k), dim3(NT), kLdsBytes, st, P, src, dst,
                                          map, RC, nq_total, q0));
  return hipGetLastError();
}

static hipError_t op_ct_ntt_fwd_oop(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* src,
                                    uint64_t* dst, uint64_t n_cts, bool src_is_tree) {
  const dim3 grid((uint32_t)(n_cts * 2 * k));
  if (src_is_tree) {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ct_ntt_fwd_oop_kernel<MODE, true>), grid, dim3(NT), kLdsBytes, st, P, src, dst));
  } else {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ct_ntt_fwd_oop_kernel<MODE, false>), grid, dim3(NT), kLdsBytes, st, P, src, dst));
  }
  return hipGetLastError();
}

static hipError_t op_ct_ntt_fwd_split(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* src,
                                      const MfmaPtrs& dst, uint32_t B, uint64_t n_cts_total) {
  PIRGPU_BY_MODE(mode, hipLaunchKernelGGL(ct_ntt_fwd_split_kernel<MODE>, dim3((uint32_t)(n_cts_total * 2 * k)),
                                          dim3(NT), kLdsBytes, st, P, src, dst, B));
  return hipGetLastError();
}

static hipError_t op_db_encode(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* coeffs,
                               const uint8_t* bytes, uint64_t bytes_per_pt, uint64_t total_bytes, uint32_t bits,
                               uint64_t n_pt, uint64_t* db) {
  PIRGPU_BY_MODE(mode, hipLaunchKernelGGL(db_encode_kernel<MODE>, dim3((uint32_t)n_pt, k), dim3(NT), kLdsBytes, st,
                                          P, coeffs, bytes, bytes_per_pt, total_bytes, bits, db));
  return hipGetLastError();
}

// c0_out (fp64 flavours, wide levels -- kernels.h ks_digit_takes_c0): also transform the nodes' c0 polynomials into
// the product buffer c0_out (what tree_c0_ntt_kernel does), as extra workgroups of the same launch
static hipError_t op_ks_digit(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* res_in,
                              uint32_t galois_elt, uint32_t nodes, uint64_t* dig, bool pack40, uint64_t* c0_out,
                              bool tree40, bool loop_targets) {
  const bool wide = ks_digit_takes_c0(nodes);
  if (c0_out && (!wide || mode == kNttInt)) return hipErrorInvalidValue;
  if (tree40 && (!pack40 || mode == kNttInt)) return hipErrorInvalidValue;
  // (the caller asks for it from ~4 workgroups per CU on: below that -- a lone query's last levels at k = 2 -- the
  // k + 1 transforms of a source in parallel finish sooner than in a row; ctx.hip, loop_min_sources)
  if (loop_targets && wide && mode != kNttInt) {
    // one workgroup per source polynomial, the k + 1 target moduli in a loop (ks_digit_kernel, LOOPI)
    const uint32_t digit_blocks = nodes * k;
    const dim3 grid(digit_blocks + (c0_out ? nodes * k : 0));
#define PIRGPU_KSD_LOOP(M, P40_, T40_)                                                                              \
  hipLaunchKernelGGL((ks_digit_kernel<M, P40_, T40_, true>), grid, dim3(NT), kLdsBytes, st, P, res_in, galois_elt, dig, \
                     c0_out, digit_blocks)
    if (mode == kNttF64) {
      if (tree40) PIRGPU_KSD_LOOP(kNttF64, true, true);
      else if (pack40) PIRGPU_KSD_LOOP(kNttF64, true, false);
      else PIRGPU_KSD_LOOP(kNttF64, false, false);
    } else {
      if (tree40) PIRGPU_KSD_LOOP(kNttF64Wide, true, true);
      else if (pack40) PIRGPU_KSD_LOOP(kNttF64Wide, true, false);
      else PIRGPU_KSD_LOOP(kNttF64Wide, false, false);
    }
#undef PIRGPU_KSD_LOOP
    return hipGetLastError();
  }
  const uint32_t digit_blocks = wide ? nodes * (k + 1) * k : 0xffffffffu;
  const dim3 grid = wide ? dim3(digit_blocks + (c0_out ? nodes * k : 0)) : dim3(nodes, k + 1, k);
  if (tree40) {
    if (mode == kNttF64)
      hipLaunchKernelGGL((ks_digit_kernel<kNttF64, true, true>), grid, dim3(NT), kLdsBytes, st, P, res_in, galois_elt, dig,
                         c0_out, digit_blocks);
    else
      hipLaunchKernelGGL((ks_digit_kernel<kNttF64Wide, true, true>), grid, dim3(NT), kLdsBytes, st, P, res_in, galois_elt,
                         dig, c0_out, digit_blocks);
  } else if (pack40) {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ks_digit_kernel<MODE, true>), grid, dim3(NT), kLdsBytes, st, P, res_in,
                                            galois_elt, dig, c0_out, digit_blocks));
  } else {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ks_digit_kernel<MODE, false>), grid, dim3(NT), kLdsBytes, st, P, res_in,
                                            galois_elt, dig, c0_out, digit_blocks));
  }
  return hipGetLastError();
}

static hipError_t op_ks_mac_intt(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* dig,
                                 const KeyPtrs& key, uint32_t nodes, uint64_t* prod, bool pack40, uint32_t I_base,
                                 uint32_t I_count) {
  const dim3 grid = nodes >= kWideLevel && nodes % 8 == 0 ? dim3(nodes * I_count * 2) : dim3(nodes, I_count, 2);
  if (pack40) {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ks_mac_intt_kernel<MODE, true>), grid, dim3(NT), kLdsBytes, st, P, dig,
                                            key, prod, I_base, I_count));
  } else {
    PIRGPU_BY_MODE(mode, hipLaunchKernelGGL((ks_mac_intt_kernel<MODE, false>), grid, dim3(NT), kLdsBytes, st, P, dig,
                                            key, prod, I_base, I_count));
  }
  return hipGetLastError();
}

// data residues of a level below the last, fused with the combine step (fp64 flavours)
// xpow != nullptr: the tree's c0 polynomials are (and stay) in NTT form, xpow = NTT_j(x^(-shift_pow)) (ks_combine_c0_ntt)
static hipError_t op_ks_mac_combine(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* dig,
                                    const KeyPtrs& key, const uint64_t* prod, const uint64_t* tree_in,
                                    uint32_t galois_elt, uint32_t nodes, uint32_t shift_pow, uint64_t* tree_out,
                                    bool pack40, bool tin40, bool tout40, const uint64_t* xpow, const uint16_t* perm,
                                    bool c0_split) {
  const dim3 grid = nodes >= kWideLevel && nodes % 8 == 0 ? dim3(nodes * k * 2) : dim3(nodes, k, 2);
  if (mode != kNttF64 && mode != kNttF64Wide) return hipErrorInvalidValue;
  if ((tin40 || tout40) && !pack40) return hipErrorInvalidValue;
  if (tout40 && shift_pow >= (uint32_t)NT) return hipErrorInvalidValue;   // the 5-byte hi store assumes shift < NT
  const double* X = reinterpret_cast<const double*>(xpow);
  if (X && PIRGPU_PERM_TABLE != 0 && !perm) return hipErrorInvalidValue;
  if (X && c0_split) {
    // component 0 (NTT domain) and component 1 (inverse transform) as two launches
    const bool wide = nodes >= kWideLevel && nodes % 8 == 0;
    const dim3 g1 = wide ? dim3(nodes * k) : dim3(nodes, k, 1);
#define PIRGPU_C0K(M, P40, TI, TO)                                                                                  \
  do {                                                                                                              \
    hipLaunchKernelGGL((ks_c0_ntt_kernel<M, P40, TI, TO>), g1, dim3(NT), kLdsBytes, st, P, dig, key, prod, tree_in, \
                       galois_elt, nodes, tree_out, X, perm);                                                       \
    hipLaunchKernelGGL((ks_mac_combine_kernel<M, P40, TI, TO, false>), g1, dim3(NT), kLdsBytes, st, P, dig, key,    \
                       prod, tree_in, galois_elt, nodes, shift_pow, tree_out, X, perm, 1u);                         \
  } while (0)
#define PIRGPU_C0K_MODE(M)                                      \
  do {                                                          \
    if (!pack40) PIRGPU_C0K(M, false, false, false);            \
    else if (tin40 && tout40) PIRGPU_C0K(M, true, true, true);  \
    else if (tin40) PIRGPU_C0K(M, true, true, false);           \
    else if (tout40) PIRGPU_C0K(M, true, false, true);          \
    else PIRGPU_C0K(M, true, false, false);                     \
  } while (0)
    if (mode == kNttF64) PIRGPU_C0K_MODE(kNttF64);
    else PIRGPU_C0K_MODE(kNttF64Wide);
#undef PIRGPU_C0K_MODE
#undef PIRGPU_C0K
    return hipGetLastError();
  }
#define PIRGPU_MC(M, P40, TI, TO, C0)                                                                               \
  hipLaunchKernelGGL((ks_mac_combine_kernel<M, P40, TI, TO, C0>), grid, dim3(NT), kLdsBytes, st, P, dig, key, prod, \
                     tree_in, galois_elt, nodes, shift_pow, tree_out, X, perm, 0u)
#define PIRGPU_MC_FMT(M, C0)                                    \
  do {                                                          \
    if (!pack40) PIRGPU_MC(M, false, false, false, C0);         \
    else if (tin40 && tout40) PIRGPU_MC(M, true, true, true, C0);   \
    else if (tin40) PIRGPU_MC(M, true, true, false, C0);        \
    else if (tout40) PIRGPU_MC(M, true, false, true, C0);       \
    else PIRGPU_MC(M, true, false, false, C0);                  \
  } while (0)
#define PIRGPU_MC_MODE(M)                 \
  do {                                    \
    if (X) PIRGPU_MC_FMT(M, true);        \
    else PIRGPU_MC_FMT(M, false);         \
  } while (0)
  if (mode == kNttF64) PIRGPU_MC_MODE(kNttF64);
  else PIRGPU_MC_MODE(kNttF64Wide);
#undef PIRGPU_MC_MODE
#undef PIRGPU_MC_FMT
#undef PIRGPU_MC
  return hipGetLastError();
}

// c0 of `cts` tree ciphertexts (doubles) into NTT form, in place (fp64 flavours)
static hipError_t op_tree_c0_fwd(hipStream_t st, int mode, const DevParams* P, uint32_t k, uint64_t* tree, uint32_t cts) {
  if (mode == kNttF64) hipLaunchKernelGGL(tree_c0_fwd_kernel<kNttF64>, dim3(cts * k), dim3(NT), kLdsBytes, st, P, tree);
  else if (mode == kNttF64Wide) hipLaunchKernelGGL(tree_c0_fwd_kernel<kNttF64Wide>, dim3(cts * k), dim3(NT), kLdsBytes, st, P, tree);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// fp64 flavours only (the caller keeps the unfused ks_combine + forward NTT for the integer flavour)
static hipError_t op_ks_last_level(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* tree,
                                   const uint64_t* prod, uint32_t galois_elt, uint32_t shift_pow, uint32_t n_items,
                                   uint32_t B, const MfmaPtrs& dst, uint32_t tree_cts, bool pack40) {
  const dim3 grid(tree_cts * 2 * k * 2);
  if (mode == kNttF64) {
    if (pack40)
      hipLaunchKernelGGL((ks_last_level_kernel<kNttF64, true>), grid, dim3(NT), kLdsBytes, st, P, tree, prod, galois_elt,
                         shift_pow, n_items, B, dst);
    else
      hipLaunchKernelGGL((ks_last_level_kernel<kNttF64, false>), grid, dim3(NT), kLdsBytes, st, P, tree, prod, galois_elt,
                         shift_pow, n_items, B, dst);
  } else if (mode == kNttF64Wide) {
    if (pack40)
      hipLaunchKernelGGL((ks_last_level_kernel<kNttF64Wide, true>), grid, dim3(NT), kLdsBytes, st, P, tree, prod,
                         galois_elt, shift_pow, n_items, B, dst);
    else
      hipLaunchKernelGGL((ks_last_level_kernel<kNttF64Wide, false>), grid, dim3(NT), kLdsBytes, st, P, tree, prod,
                         galois_elt, shift_pow, n_items, B, dst);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// fp64 flavours: A_0 into the product buffer, then the NTT-domain last level
// c0_tree: 0 = NTT(a_0) comes from the product buffer (computed here unless c0_done), 1 / 2 = the tree's c0 polynomials
// are in NTT form already (doubles / 5-byte form)
static hipError_t op_ks_last_ntt(hipStream_t st, int mode, const DevParams* P, uint32_t k, const uint64_t* tree,
                                 const uint64_t* dig, const KeyPtrs& key, uint64_t* prod, const uint64_t* xpow,
                                 uint32_t galois_elt, uint32_t galois_inv, uint32_t shift_pow, uint32_t n_items,
                                 uint32_t B, const MfmaPtrs& dst, uint32_t nodes, bool pack40, bool out_f64,
                                 bool c0_done, int c0_tree, const uint16_t* perm) {
  const dim3 g0(nodes * k);
  const dim3 grid = nodes >= kWideLevel && nodes % 8 == 0 ? dim3(nodes * k * 2) : dim3(nodes, k, 2);
  const double* X = reinterpret_cast<const double*>(xpow);
  if (c0_tree == 2 && !pack40) return hipErrorInvalidValue;
  if (PIRGPU_PERM_TABLE != 0 && !perm) return hipErrorInvalidValue;
#define PIRGPU_LAST_NTT_K(M, P40, OF, CT)                                                                          \
  hipLaunchKernelGGL((ks_last_ntt_kernel<M, P40, OF, CT>), grid, dim3(NT), kLdsBytes, st, P, dig, key, prod, X,    \
                     galois_elt, galois_inv, shift_pow, n_items, B, dst, tree, perm)
#define PIRGPU_LAST_NTT(M, P40)                                                                                    \
  do {                                                                                                             \
    if (c0_tree == 0 && !c0_done)                                                                                  \
      hipLaunchKernelGGL((tree_c0_ntt_kernel<M, P40>), g0, dim3(NT), kLdsBytes, st, P, tree, prod);                \
    if (c0_tree == 2) {                                                                                            \
      if constexpr (P40) {                                                                                         \
        if (out_f64) PIRGPU_LAST_NTT_K(M, P40, true, 2);                                                           \
        else PIRGPU_LAST_NTT_K(M, P40, false, 2);                                                                  \
      }                                                                                                            \
    } else if (c0_tree == 1) {                                                                                     \
      if (out_f64) PIRGPU_LAST_NTT_K(M, P40, true, 1);                                                             \
      else PIRGPU_LAST_NTT_K(M, P40, false, 1);                                                                    \
    } else {                                                                                                       \
      if (out_f64) PIRGPU_LAST_NTT_K(M, P40, true, 0);                                                             \
      else PIRGPU_LAST_NTT_K(M, P40, false, 0);                                                                    \
    }                                                                                                              \
  } while (0)
  if (mode == kNttF64) {
    if (pack40) PIRGPU_LAST_NTT(kNttF64, true);
    else PIRGPU_LAST_NTT(kNttF64, false);
  } else if (mode == kNttF64Wide) {
    if (pack40) PIRGPU_LAST_NTT(kNttF64Wide, true);
    else PIRGPU_LAST_NTT(kNttF64Wide, false);
  } else {
    return hipErrorInvalidValue;
  }
#undef PIRGPU_LAST_NTT
#undef PIRGPU_LAST_NTT_K
  return hipGetLastError();
}

static hipError_t op_upper_ntt(hipStream_t st, int mode, const DevParams* P, uint32_t k, uint32_t enc_count,
                               const uint64_t* src, uint64_t* scratch, uint32_t n_rows, uint32_t n_dim,
                               uint32_t n_children_total, uint32_t C, uint32_t b0, uint32_t blk, uint32_t n_queries,
                               uint64_t src_qstride, bool loop_source) {
  double* out = reinterpret_cast<double*>(scratch);
  if (loop_source) {   // one workgroup per (child, source polynomial): all its chunks x target moduli in a loop
    const dim3 lgrid(n_queries * n_rows * C * blk, 2 * k, 1);
    if (mode == kNttF64)
      hipLaunchKernelGGL((upper_ntt_kernel<kNttF64, true>), lgrid, dim3(NT), kLdsBytes, st, P, src, out, n_rows, n_dim,
                         n_children_total, C, b0, blk, src_qstride);
    else if (mode == kNttF64Wide)
      hipLaunchKernelGGL((upper_ntt_kernel<kNttF64Wide, true>), lgrid, dim3(NT), kLdsBytes, st, P, src, out, n_rows, n_dim,
                         n_children_total, C, b0, blk, src_qstride);
    else
      return hipErrorInvalidValue;
    return hipGetLastError();
  }
  const dim3 grid(n_queries * n_rows * C * blk, enc_count, k);
  if (mode == kNttF64)
    hipLaunchKernelGGL(upper_ntt_kernel<kNttF64>, grid, dim3(NT), kLdsBytes, st, P, src, out, n_rows, n_dim,
                       n_children_total, C, b0, blk, src_qstride);
  else if (mode == kNttF64Wide)
    hipLaunchKernelGGL(upper_ntt_kernel<kNttF64Wide>, grid, dim3(NT), kLdsBytes, st, P, src, out, n_rows, n_dim,
                       n_children_total, C, b0, blk, src_qstride);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

static hipError_t op_upper_fused(hipStream_t st, int mode, const DevParams* P, uint32_t k, uint32_t enc_count,
                                 const uint64_t* src, const MfmaPtrs& svq, uint64_t* part, uint32_t n_rows,
                                 uint32_t n_dim, uint32_t n_children_total, uint32_t sv_first, uint32_t C,
                                 uint32_t chunk_len, uint32_t n_chunks, uint32_t n_queries, uint64_t src_qstride,
                                 uint64_t part_qstride, bool sel_f64) {
  const dim3 grid(n_queries * n_rows * C * n_chunks, enc_count, k);
#define PIRGPU_UF_ARGS P, src, svq, part, n_rows, n_dim, n_children_total, sv_first, C, chunk_len, n_chunks, src_qstride, part_qstride
  if (sel_f64 && mode == kNttInt) return hipErrorInvalidValue;
  if constexpr (kUpperLdsTw) {
    // (round 6: off by default -- the plain form with the source prefetch is 1 % faster; PIRGPU_UPPER_LDS_TW=1 selects it)
    static const bool lds_tw = pirgpu_env("PIRGPU_UPPER_LDS_TW") && atoi(pirgpu_env("PIRGPU_UPPER_LDS_TW")) != 0;
    if (lds_tw && mode != kNttInt) {
      const size_t lds = kLdsBytes + (size_t)N * 8;
      if (mode == kNttF64 && sel_f64) hipLaunchKernelGGL((upper_fused_kernel<kNttF64, true, true>), grid, dim3(NT), lds, st, PIRGPU_UF_ARGS);
      else if (mode == kNttF64) hipLaunchKernelGGL((upper_fused_kernel<kNttF64, true>), grid, dim3(NT), lds, st, PIRGPU_UF_ARGS);
      else if (sel_f64) hipLaunchKernelGGL((upper_fused_kernel<kNttF64Wide, true, true>), grid, dim3(NT), lds, st, PIRGPU_UF_ARGS);
      else hipLaunchKernelGGL((upper_fused_kernel<kNttF64Wide, true>), grid, dim3(NT), lds, st, PIRGPU_UF_ARGS);
      return hipGetLastError();
    }
  }
  if (sel_f64) {
    if (mode == kNttF64) hipLaunchKernelGGL((upper_fused_kernel<kNttF64, false, true>), grid, dim3(NT), kLdsBytes, st, PIRGPU_UF_ARGS);
    else hipLaunchKernelGGL((upper_fused_kernel<kNttF64Wide, false, true>), grid, dim3(NT), kLdsBytes, st, PIRGPU_UF_ARGS);
    return hipGetLastError();
  }
  PIRGPU_BY_MODE(mode, hipLaunchKernelGGL(upper_fused_kernel<MODE>, grid, dim3(NT), kLdsBytes, st, PIRGPU_UF_ARGS));
#undef PIRGPU_UF_ARGS
  return hipGetLastError();
}

}  // namespace PIRGPU_DEG_NS

// host-only accessor (a namespace-scope const object would also be emitted for the device)
const NttOps* PIRGPU_OPS_NAME() {
  using namespace PIRGPU_DEG_NS;
  static const NttOps ops = {op_configure, op_ntt_batch,   op_ct_ntt_fwd_oop, op_ct_ntt_fwd_split, op_db_encode,
                             op_ks_digit,  op_ks_mac_intt, op_upper_fused,     op_ks_last_level,
                             op_upper_ntt, op_ks_mac_combine, op_ks_last_ntt, op_ntt_inv_gather, op_tree_c0_fwd};
  return &ops;
}

}  // namespace pirgpu
