// ntt_core.h -- register-resident negacyclic NTT for one workgroup per polynomial.
//
// One workgroup of N/EPT threads transforms one polynomial; every thread keeps EPT = 2^R
// residues in registers and runs up to R butterfly stages per pass: EPT = 16 (R = 4) for every
// degree -- a 4096-point transform is 3 register passes with 2 LDS exchanges, a 16384-point
// one 4 passes with 3 exchanges in a 1024-thread workgroup.  The code is written for any R:
// EPT = 32 (R = 5) at N = 16384 -- 512 threads whose waves get 256 VGPRs, 14 = 5 + 5 + 4 stages
// in 3 passes -- was built and measured 7 % slower (device_params.h, ntt_log_ept).  A pass with
// window LB owns, per thread, the EPT residues
//     idx = (outer << (LB+R)) | (e << LB) | inner,      tid = (outer << LB) | inner.
//
//  * LDS is padded by one word per EPT (lds_idx), which makes all three access
//    patterns (stride N/EPT, stride EPT, contiguous EPT) bank-conflict free;
//  * NTT-domain data lives in HBM in "device order": SEAL's bit-reversed position
//    pos = EPT*tid + e is stored at slot e*(N/EPT) + tid.  Every dyadic operation is
//    order-agnostic, both transforms read and write global memory fully
//    coalesced, and the register layout on the NTT side is exactly that order --
//    key-switch products are formed between a forward and an inverse transform
//    without leaving registers;
//  * the butterflies come in two exact flavours (Arith<MODE>): 64-bit integer
//    Harvey/Shoup for any modulus < 2^61, and error-free fp64 for moduli < 2^49
//    (arith.h).  Outputs are canonical residues either way, so the bits are the
//    ones SEAL's ntt_negacyclic_harvey / inverse_ntt_negacyclic_harvey produce
//    (reference database.cpp:190,222,252).
#pragma once
#include "arith.h"

namespace pirgpu {

template <int LOGN>
struct Plan {
  static constexpr int N = 1 << LOGN;
  static constexpr int R = ntt_log_ept(LOGN);   // butterfly stages per register pass
  static constexpr int EPT = 1 << R;            // residues per thread
  static constexpr int NT = N / EPT;
  static constexpr int LDS_WORDS = N + N / EPT;
};

// padded LDS position of element i (one pad word per 2^R)
template <int R>
__device__ __forceinline__ uint32_t lds_idx(uint32_t i) { return i + (i >> R); }

// Twiddle tables are reached through pointers stored in DevParams; typing them as global
// (address space 1) makes hipcc emit global_load instead of flat_load for them.
#define PIRGPU_GLOBAL __attribute__((address_space(1)))

// ------------------------------------------------------------------ arithmetic policies

template <int MODE>
struct Arith;

template <>
struct Arith<kNttInt> {
  using T = uint64_t;
  using TW = Twiddle;
  struct Mod {
    uint64_t q, q2;
  };
  static __device__ __forceinline__ Mod mod(const DevParams* P, int mi) {
    const uint64_t q = P->mod[mi].q;
    return Mod{q, q << 1};
  }
  using TWPtr = const PIRGPU_GLOBAL TW*;
  static __device__ __forceinline__ TW load_tw(TWPtr p, uint32_t i) { return TW{p[i].w, p[i].ws}; }
  static __device__ __forceinline__ TWPtr tw(const DevParams* P, int mi) { return (TWPtr)P->tab[mi].tw; }
  static __device__ __forceinline__ TWPtr itw(const DevParams* P, int mi) { return (TWPtr)P->tab[mi].itw; }
  static __device__ __forceinline__ TW ninv(const DevParams* P, int mi) { return P->tab[mi].ninv; }
  static __device__ __forceinline__ TW iw1n(const DevParams* P, int mi) { return P->tab[mi].iw1n; }
  // canonical residue < q  <->  register element
  static __device__ __forceinline__ T in(uint64_t v, const Mod&) { return v; }
  static __device__ __forceinline__ uint64_t out(T v, const Mod&) { return v; }
  // Harvey lazy butterflies: forward keeps values < 4q, inverse < 2q
  static __device__ __forceinline__ void fwd(T& a, T& b, const TW& w, const Mod& m) {
    T X = a >= m.q2 ? a - m.q2 : a;
    const T t = mul_shoup_lazy(b, w, m.q);
    a = X + t;
    b = X - t + m.q2;
  }
  static __device__ __forceinline__ void inv(T& a, T& b, const TW& w, const Mod& m) {
    const T u = a + b, d = a - b + m.q2;
    a = u >= m.q2 ? u - m.q2 : u;
    b = mul_shoup_lazy(d, w, m.q);
  }
  static __device__ __forceinline__ void inv_last(T& a, T& b, const TW& ninv, const TW& iw1n, const Mod& m) {
    const T u = a + b, d = a - b + m.q2;
    a = mul_shoup_lazy(u, ninv, m.q);
    b = mul_shoup_lazy(d, iw1n, m.q);
  }
  static __device__ __forceinline__ T canon_fwd(T v, const Mod& m) {
    v = v >= m.q2 ? v - m.q2 : v;
    return v >= m.q ? v - m.q : v;
  }
  static __device__ __forceinline__ T canon_inv(T v, const Mod& m) { return v >= m.q ? v - m.q : v; }
  static __device__ __forceinline__ T signed_fwd(T v, const Mod& m) { return canon_fwd(v, m); }
  static __device__ __forceinline__ T pass_norm(T v, const Mod&) { return v; }
};

template <int MODE>
struct ArithF64 {
  using T = double;
  using TW = double;
  using Mod = F64Mod;
  static __device__ __forceinline__ Mod mod(const DevParams* P, int mi) {
    return Mod{P->tab[mi].qd, P->tab[mi].qinvd, MODE == kNttF64 && P->f64_lazy_inv != 0};
  }
  using TWPtr = const PIRGPU_GLOBAL TW*;
  // any pointer type: the table in global memory (TWPtr) or a copy of it in LDS (upper_fused_kernel)
  template <typename PT>
  static __device__ __forceinline__ TW load_tw(PT p, uint32_t i) { return p[i]; }
  static __device__ __forceinline__ TWPtr tw(const DevParams* P, int mi) { return (TWPtr)P->tab[mi].twf; }
  static __device__ __forceinline__ TWPtr itw(const DevParams* P, int mi) { return (TWPtr)P->tab[mi].itwf; }
  static __device__ __forceinline__ TW ninv(const DevParams* P, int mi) { return P->tab[mi].ninv_f; }
  static __device__ __forceinline__ TW iw1n(const DevParams* P, int mi) { return P->tab[mi].iw1n_f; }
  static __device__ __forceinline__ T in(uint64_t v, const Mod&) { return f64_from_u64(v); }
  static __device__ __forceinline__ uint64_t out(T v, const Mod&) { return f64_to_u64(v); }
  // forward: |a| grows by at most 0.6 q per stage from < q (<= 9.4 q < 2^53 after 14 stages)
  static __device__ __forceinline__ void fwd(T& a, T& b, const TW& w, const Mod& m) {
    const T y = MODE == kNttF64Wide ? f64_norm(b, m) : b;
    const T t = f64_mulmod(y, w, m);
    b = a - t;
    a = a + t;
  }
  // inverse: sums double per stage.  kNttF64 (q < 2^46) lets them grow through the 4 stages of a
  // pass (|values| <= 16 q < 2^50, quotient estimates stay exact to < 0.1) and renormalises once
  // per pass (pass_norm); kNttF64Wide renormalises every stage, |values| <= 0.7 q throughout.
  static constexpr bool kNormPerPass = MODE == kNttF64;
  static __device__ __forceinline__ void inv(T& a, T& b, const TW& w, const Mod& m) {
    const T u = a + b, d = a - b;
    a = kNormPerPass ? u : f64_norm(u, m);
    b = f64_mulmod(d, w, m);
  }
  static __device__ __forceinline__ T pass_norm(T v, const Mod& m) {
    return kNormPerPass && !m.lazy_inv ? f64_norm(v, m) : v;  // lazy_inv is uniform: a scalar branch per pass
  }
  static __device__ __forceinline__ void inv_last(T& a, T& b, const TW& ninv, const TW& iw1n, const Mod& m) {
    const T u = a + b, d = a - b;
    a = f64_mulmod(u, ninv, m);
    b = f64_mulmod(d, iw1n, m);
  }
  static __device__ __forceinline__ T canon_fwd(T v, const Mod& m) { return f64_canon(f64_norm(v, m), m); }
  static __device__ __forceinline__ T signed_fwd(T v, const Mod& m) { return f64_norm(v, m); }
  static __device__ __forceinline__ T canon_inv(T v, const Mod& m) { return f64_canon(v, m); }
};

template <>
struct Arith<kNttF64> : ArithF64<kNttF64> {};
template <>
struct Arith<kNttF64Wide> : ArithF64<kNttF64Wide> {};

// ------------------------------------------------------------------ passes

// Padded LDS position of the EPT = 2^R residues a thread owns in a pass with window LB: lds_idx(base | (e << LB)) is
// lds_base<LB, R>(tid) + lds_off<LB, R>(e) with a compile-time second term, so the accesses of a pass are one
// address computation plus immediate offsets (written as `idx + (idx >> R)` per element the compiler spends
// four VALU instructions on every access).
//   LB >= R: (idx >> R) = (base >> R) | (e << (LB - R)), the two terms occupy disjoint bits;
//   LB <  R: (idx >> R) = (outer << LB) | (e >> (R - LB)), since (e << LB) | inner < 2^(LB + R).
template <int LB, int R>
__device__ __forceinline__ uint32_t lds_base(uint32_t tid) {
  const uint32_t inner = tid & ((1u << LB) - 1u), outer = tid >> LB;
  const uint32_t base = (outer << (LB + R)) | inner;
  if constexpr (LB >= R) return base + (base >> R);
  else return base + (outer << LB);
}
template <int LB, int R>
__device__ __forceinline__ constexpr uint32_t lds_off(int e) {
  if constexpr (LB >= R) return (uint32_t)e * ((1u << LB) + (1u << (LB - R)));
  else return ((uint32_t)e << LB) + ((uint32_t)e >> (R - LB));
}

template <int LB, int R, typename T>
__device__ __forceinline__ void lds_store_pass(T* s, const T (&x)[1 << R], uint32_t tid) {
  T* p = s + lds_base<LB, R>(tid);
#pragma unroll
  for (int e = 0; e < (1 << R); ++e) p[lds_off<LB, R>(e)] = x[e];
}

template <int LB, int R, typename T>
__device__ __forceinline__ void lds_load_pass(const T* s, T (&x)[1 << R], uint32_t tid) {
  const T* p = s + lds_base<LB, R>(tid);
#pragma unroll
  for (int e = 0; e < (1 << R); ++e) x[e] = p[lds_off<LB, R>(e)];
}

// Padded LDS position of element e * NT + tid (the layout a thread's residues have in global memory), NT a
// multiple of EPT: tid + (tid >> R) + e * (NT + NT / EPT).
template <int NT_, int R>
__device__ __forceinline__ uint32_t lds_lin_base(uint32_t tid) { return tid + (tid >> R); }
template <int NT_, int R>
__device__ __forceinline__ constexpr uint32_t lds_lin_off(int e) { return (uint32_t)e * (NT_ + (NT_ >> R)); }

// A forward twiddle table copied to LDS (upper_fused_kernel: a transform free of vector memory instructions lets
// loads issued before it complete under it -- vmcnt is in order).  Stored pass by pass, slot-major: the twiddle of
// (pass window LB, slot w = (8 >> rb) - 1 + g, outer = tid >> LB) sits at
//   offset(LB) + (w - w0(LB)) * (NT >> LB) + outer,
// so that the lanes of a wave read consecutive words (or the same word) -- the table's natural order has the last
// pass's eight stage-0 twiddles of a thread 64 bytes from the next thread's: four-way bank conflicts.
template <int LOGN>
struct TwSoA {
  static constexpr int NT = 1 << (LOGN - 4);
  static constexpr int next_lb(int LB) { return LB >= 4 ? LB - 4 : 0; }
  static constexpr int rhi(int LB) {  // highest relative bit of the pass with window LB (fwd_chain's sequence)
    int lb = LOGN - 4, r = 3;
    while (lb != LB) {
      r = lb >= 4 ? 3 : lb - 1;
      lb = next_lb(lb);
    }
    return r;
  }
  static constexpr int slots(int r) { return 16 - (8 >> r); }  // sum over rb <= r of (8 >> rb)
  static constexpr int w0(int LB) { return (8 >> rhi(LB)) - 1; }
  static constexpr int offset(int LB) {
    int lb = LOGN - 4, off = 0;
    while (lb != LB) {
      off += slots(rhi(lb)) * (NT >> lb);
      lb = next_lb(lb);
    }
    return off;
  }
  static constexpr int kWords = offset(0) + slots(rhi(0)) * NT;  // = N - 1
};

struct LdsTw {
  const double* p;
};

// twiddle of (window LB, relative bit rb, g, outer): table index 2^(LOGN-1-(LB+rb)) + (outer << (R-1-rb)) + g
template <typename A, int LOGN, int LB, typename TP>
__device__ __forceinline__ typename A::TW tw_at(TP tw, int rb, int g, uint32_t outer) {
  constexpr int R = Plan<LOGN>::R;
  const uint32_t mm = 1u << (LOGN - 1 - (LB + rb));
  return A::load_tw(tw, mm + (outer << (R - 1 - rb)) + g);
}
template <typename A, int LOGN, int LB>
__device__ __forceinline__ typename A::TW tw_at(LdsTw tw, int rb, int g, uint32_t outer) {
  static_assert(Plan<LOGN>::R == 4, "the LDS twiddle table is laid out for 16 residues per thread");
  using S = TwSoA<LOGN>;
  return tw.p[S::offset(LB) + ((8 >> rb) - 1 + g - S::w0(LB)) * (S::NT >> LB) + outer];
}

// Fills the LDS copy (TwSoA<LOGN>::kWords doubles) from the table in global memory; all threads of the workgroup.
template <int LOGN, int LB, typename SRC>
__device__ __forceinline__ void lds_twiddles_fill_pass(double* dst, SRC src, uint32_t tid) {
  using S = TwSoA<LOGN>;
  constexpr int W = S::NT >> LB;  // distinct outer values
#pragma unroll
  for (int rb = S::rhi(LB); rb >= 0; --rb) {
    const uint32_t mm = 1u << (LOGN - 1 - (LB + rb));
#pragma unroll
    for (int g = 0; g < (8 >> rb); ++g) {
      double* row = dst + S::offset(LB) + ((8 >> rb) - 1 + g - S::w0(LB)) * W;
      for (uint32_t o = tid; o < (uint32_t)W; o += S::NT) row[o] = src[mm + (o << (3 - rb)) + g];
    }
  }
  if constexpr (LB > 0) lds_twiddles_fill_pass<LOGN, S::next_lb(LB)>(dst, src, tid);
}
template <int LOGN, typename SRC>
__device__ __forceinline__ void lds_twiddles_fill(double* dst, SRC src, uint32_t tid) {
  lds_twiddles_fill_pass<LOGN, LOGN - 4>(dst, src, tid);
}

// The EPT - 1 twiddles of one R-stage pass, in the order the stages consume them: relative bit rb
// (R-1..0) owns slots (EPT/2 >> rb) - 1 ... ; loaded one pass AHEAD of their use so that their L2
// latency hides under the previous pass's butterflies and LDS exchange.
template <typename A, int LOGN, int LB, int RHI, int RLO, typename TP>
__device__ __forceinline__ void load_twiddles(typename A::TW (&W)[Plan<LOGN>::EPT - 1], TP tw, uint32_t outer) {
  constexpr int H = Plan<LOGN>::EPT / 2;
#pragma unroll
  for (int rb = RHI; rb >= RLO; --rb) {
#pragma unroll
    for (int g = 0; g < (H >> rb); ++g) W[(H >> rb) - 1 + g] = tw_at<A, LOGN, LB>(tw, rb, g, outer);
  }
}

// Cooley-Tukey stages on one window for relative bits RHI..RLO (high to low) over EPT = 2^R residues.
template <typename A, int R, int RHI, int RLO>
__device__ __forceinline__ void fwd_stages(typename A::T (&x)[1 << R], const typename A::TW (&W)[(1 << R) - 1],
                                           const typename A::Mod& m) {
  constexpr int H = 1 << (R - 1);
#pragma unroll
  for (int rb = RHI; rb >= RLO; --rb) {
#pragma unroll
    for (int g = 0; g < (H >> rb); ++g) {
#pragma unroll
      for (int l = 0; l < (1 << rb); ++l) {
        const int e0 = (g << (rb + 1)) | l, e1 = e0 | (1 << rb);
        A::fwd(x[e0], x[e1], W[(H >> rb) - 1 + g], m);
      }
    }
  }
}

// The LDS exchange between two passes.  The threads that trade residues when a pass leaves window LB for the next
// window (forward; the inverse runs the same exchanges the other way round) differ only in the bits [next window, LB)
// of their index: with LB <= 6 they are lanes of ONE 64-wide wavefront, whose LDS instructions execute in order --
// the exchange then needs no workgroup barrier, only the compiler kept from reordering it.  At N = 16384 that is two
// of the three exchanges of a transform (windows 10 -> 6 -> 2 -> 0), at N = 4096 one of two (8 -> 4 -> 0): after the
// first exchange the waves of a workgroup drift apart and overlap each other's LDS, twiddle and butterfly phases --
// which matters most where a CU holds a single workgroup (N = 16384) and a barrier idles the whole CU.
#ifndef PIRGPU_WAVE_LOCAL_EXCHANGE
#define PIRGPU_WAVE_LOCAL_EXCHANGE 1
#endif
template <int WINDOW>
__device__ __forceinline__ void exchange_sync() {
  if constexpr (PIRGPU_WAVE_LOCAL_EXCHANGE != 0 && WINDOW <= 6) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// Runs the pass whose twiddles are in Wcur on window LB (relative bits RHI..0), then the rest.
// PF: load the next pass's twiddles before this pass's butterflies (hides their L2 latency, costs
// 30-60 registers); without PF they are loaded between this pass's butterflies and the exchange.
template <typename A, int LOGN, int LB, int RHI, bool PF, typename TP>
__device__ __forceinline__ void fwd_chain(typename A::T (&x)[Plan<LOGN>::EPT], typename A::T* s, TP tw,
                                          const typename A::Mod& m, uint32_t tid,
                                          const typename A::TW (&Wcur)[Plan<LOGN>::EPT - 1]) {
  constexpr int R = Plan<LOGN>::R;
  if constexpr (LB > 0) {
    constexpr int NLB = LB >= R ? LB - R : 0;
    constexpr int NRHI = LB >= R ? R - 1 : LB - 1;
    typename A::TW Wnext[Plan<LOGN>::EPT - 1];
    if constexpr (PF) {
      load_twiddles<A, LOGN, NLB, NRHI, 0>(Wnext, tw, tid >> NLB);
      __builtin_amdgcn_sched_barrier(0);
    }
    fwd_stages<A, R, RHI, 0>(x, Wcur, m);
    // without PF the next pass's twiddles are requested here, once this pass's are dead: their latency hides under
    // the exchange instead of following it
    if constexpr (!PF) load_twiddles<A, LOGN, NLB, NRHI, 0>(Wnext, tw, tid >> NLB);
    lds_store_pass<LB, R>(s, x, tid);
    exchange_sync<LB>();
    lds_load_pass<NLB, R>(s, x, tid);
    fwd_chain<A, LOGN, NLB, NRHI, PF>(x, s, tw, m, tid, Wnext);
  } else {
    fwd_stages<A, R, RHI, 0>(x, Wcur, m);
  }
}

// Forward NTT.  In: x[e] = coefficient e*NT + tid (A::in of a canonical residue).
// Out: x[e] = SEAL NTT position EPT*tid + e == device-order slot e*NT + tid, canonical
// representative (A::out gives the residue).  The caller guarantees nobody still reads
// `s` (barrier) when this is entered.
// CANON = false (fp64 flavours only): leave signed representatives |x| <= (1/2 + eps) q instead of canonical ones.
// `tw`: the forward twiddle table of modulus mi, in global memory (A::tw) or copied to LDS by the caller.
template <int MODE, int LOGN, bool PF, bool CANON, typename TP>
__device__ __forceinline__ void ntt_forward_tw(typename Arith<MODE>::T (&x)[Plan<LOGN>::EPT], void* lds, const DevParams* P,
                                               int mi, uint32_t tid, TP tw) {
  using A = Arith<MODE>;
  constexpr int R = Plan<LOGN>::R;
  typename A::T* s = reinterpret_cast<typename A::T*>(lds);
  const typename A::Mod m = A::mod(P, mi);
  typename A::TW W0[Plan<LOGN>::EPT - 1];
  load_twiddles<A, LOGN, LOGN - R, R - 1, 0>(W0, tw, 0u);
  fwd_chain<A, LOGN, LOGN - R, R - 1, PF>(x, s, tw, m, tid, W0);
#pragma unroll
  for (int e = 0; e < Plan<LOGN>::EPT; ++e) {
    if constexpr (CANON) x[e] = A::canon_fwd(x[e], m);
    else x[e] = A::signed_fwd(x[e], m);
  }
}

template <int MODE, int LOGN, bool PF = true, bool CANON = true>
__device__ __forceinline__ void ntt_forward(typename Arith<MODE>::T (&x)[Plan<LOGN>::EPT], void* lds, const DevParams* P,
                                            int mi, uint32_t tid) {
  ntt_forward_tw<MODE, LOGN, PF, CANON>(x, lds, P, mi, tid, Arith<MODE>::tw(P, mi));
}

// Gentleman-Sande stages for relative bits RLO..R-1 (low to high) with preloaded twiddles; with
// LAST the final stage multiplies by N^-1 (folded into both outputs, its own twiddle unused).
template <typename A, int R, int RLO, bool LAST>
__device__ __forceinline__ void inv_stages(typename A::T (&x)[1 << R], const typename A::TW (&W)[(1 << R) - 1],
                                           const typename A::TW& ninv, const typename A::TW& iw1n,
                                           const typename A::Mod& m) {
  constexpr int H = 1 << (R - 1);
#pragma unroll
  for (int rb = RLO; rb <= R - 1; ++rb) {
    if (LAST && rb == R - 1) {
#pragma unroll
      for (int l = 0; l < H; ++l) A::inv_last(x[l], x[l | H], ninv, iw1n, m);
    } else {
#pragma unroll
      for (int g = 0; g < (H >> rb); ++g) {
#pragma unroll
        for (int l = 0; l < (1 << rb); ++l) {
          const int e0 = (g << (rb + 1)) | l, e1 = e0 | (1 << rb);
          A::inv(x[e0], x[e1], W[(H >> rb) - 1 + g], m);
        }
      }
    }
  }
}

// Runs the pass on window LB (relative bits RLO..R-1) whose twiddles are in Wcur; D = index bits
// done once this pass completes.
template <typename A, int LOGN, int LB, int RLO, bool PF>
__device__ __forceinline__ void inv_chain(typename A::T (&x)[Plan<LOGN>::EPT], typename A::T* s,
                                          typename A::TWPtr itw, const typename A::TW& ninv,
                                          const typename A::TW& iw1n, const typename A::Mod& m, uint32_t tid,
                                          const typename A::TW (&Wcur)[Plan<LOGN>::EPT - 1]) {
  constexpr int R = Plan<LOGN>::R;
  constexpr int D = LB + R;
  constexpr bool LAST = (D == LOGN);
  if constexpr (!LAST) {
    constexpr int NLB = (LOGN - D >= R) ? D : LOGN - R;
    constexpr int NRLO = D - NLB;
    typename A::TW Wnext[Plan<LOGN>::EPT - 1];
    if constexpr (PF) {  // one pass ahead
      load_twiddles<A, LOGN, NLB, R - 1, NRLO>(Wnext, itw, tid >> NLB);
      __builtin_amdgcn_sched_barrier(0);
    }
    inv_stages<A, R, RLO, false>(x, Wcur, ninv, iw1n, m);
#pragma unroll
    for (int e = 0; e < Plan<LOGN>::EPT; ++e) x[e] = A::pass_norm(x[e], m);
    if constexpr (!PF) load_twiddles<A, LOGN, NLB, R - 1, NRLO>(Wnext, itw, tid >> NLB);
    lds_store_pass<LB, R>(s, x, tid);
    exchange_sync<NLB>();
    lds_load_pass<NLB, R>(s, x, tid);
    inv_chain<A, LOGN, NLB, NRLO, PF>(x, s, itw, ninv, iw1n, m, tid, Wnext);
  } else {
    inv_stages<A, R, RLO, true>(x, Wcur, ninv, iw1n, m);
  }
}

// Inverse NTT.  In: x[e] = NTT position EPT*tid + e (any representative the flavour
// accepts: < 2q for integers, |v| <= 4q for fp64).  Out: x[e] = coefficient e*NT + tid,
// canonical, scaled by N^-1.
template <int MODE, int LOGN, bool PF = true, bool CANON = true>
__device__ __forceinline__ void ntt_inverse(typename Arith<MODE>::T (&x)[Plan<LOGN>::EPT], void* lds, const DevParams* P,
                                            int mi, uint32_t tid) {
  static_assert(LOGN >= 8, "at least two passes expected");
  using A = Arith<MODE>;
  typename A::T* s = reinterpret_cast<typename A::T*>(lds);
  const typename A::Mod m = A::mod(P, mi);
  const typename A::TWPtr itw = A::itw(P, mi);
  const typename A::TW ninv = A::ninv(P, mi), iw1n = A::iw1n(P, mi);
  typename A::TW W0[Plan<LOGN>::EPT - 1];
  load_twiddles<A, LOGN, 0, Plan<LOGN>::R - 1, 0>(W0, itw, tid);
  inv_chain<A, LOGN, 0, 0, PF>(x, s, itw, ninv, iw1n, m, tid, W0);
  if constexpr (CANON) {
#pragma unroll
    for (int e = 0; e < Plan<LOGN>::EPT; ++e) x[e] = A::canon_inv(x[e], m);
  }  // else: the last stage's products already are signed representatives
}

}  // namespace pirgpu
