// device_params.h -- parameter block shared by host code and gfx950 kernels.
#pragma once
#include <stdint.h>

namespace pirgpu {

constexpr int kMaxPrimes = 8;            // data primes
constexpr int kMaxMod = kMaxPrimes + 1;  // + special prime
constexpr int kMaxEnc = 64;              // max 2 * ExpansionRatio
constexpr int kMaxScanQueries = 4;       // queries sharing one database pass in batch mode
// NTT workgroup = N / EPT threads, EPT = 2^ntt_log_ept residues per thread in registers.  This also defines the DEVICE
// ORDER of NTT-form data: SEAL position EPT * t + e lives at slot e * (N / EPT) + t.  16 residues per thread for every
// degree; the kernels are written for any power of two, and N = 16384 was measured with 32 (512-thread workgroups whose
// waves get 256 VGPRs: twiddle prefetch on, three passes instead of four) against 16 (1024-thread workgroups, 128
// VGPRs) on one box, round 4: 104.4 against 112.7 queries/s at cfg 5, single query 19.1 against 18.1 ms -- with one
// workgroup per CU either way (a 16384-point polynomial fills the LDS), sixteen waves hide more latency than eight with
// twice the registers (-DPIRGPU_LOG_EPT14=5 rebuilds the other organisation, tools/experiments/r04_ab_ept.sh).
#ifndef PIRGPU_LOG_EPT14
#define PIRGPU_LOG_EPT14 4
#endif
#ifdef PIRGPU_LOG_EPT_ALL   // prototypes only (tools/ntt_short_proto.hip): every degree with 2^PIRGPU_LOG_EPT_ALL residues per thread
constexpr int ntt_log_ept(int) { return PIRGPU_LOG_EPT_ALL; }
#else
constexpr int ntt_log_ept(int logN) { return logN >= 14 ? PIRGPU_LOG_EPT14 : 4; }
#endif

#include "env_gate.h"

// One RNS modulus with the Barrett ratio floor(2^128 / q) (SEAL Modulus::const_ratio).
struct ModConst {
  uint64_t q;
  uint64_t br_lo, br_hi;
};

// {w, floor(w * 2^64 / q)}: one 16-byte load per twiddle.
struct alignas(16) Twiddle {
  uint64_t w, ws;
};

// Negacyclic NTT tables for one modulus (reference
// database.cpp:190,252 -> Evaluator::transform_to/from_ntt_inplace):
//   tw[i]  = psi^bitrev(i),  itw[i] = psi^-bitrev(i)   (psi = minimal primitive 2N-th root)
// and the constants of the last inverse stage with N^-1 folded in.
struct NttTable {
  const Twiddle* tw;
  const Twiddle* itw;
  Twiddle ninv;   // N^-1
  Twiddle iw1n;   // psi^-bitrev(1) * N^-1
  // exact-fp64 flavour (moduli < 2^49): the same tables as signed doubles in (-q/2, q/2]
  const double* twf;
  const double* itwf;
  double ninv_f, iw1n_f;
  double qd, qinvd;
};

// Arithmetic flavour of the NTT kernels, chosen per context from the largest modulus:
//   kNttInt      64-bit integer Shoup/Harvey butterflies, any modulus < 2^61
//   kNttF64      exact fp64 butterflies, all moduli < 2^46
//   kNttF64Wide  exact fp64 with an extra normalisation per forward butterfly, all moduli < 2^49
enum NttMode : int { kNttInt = 0, kNttF64 = 1, kNttF64Wide = 2 };

struct DevParams {
  uint32_t N, logN, k, pad0;
  ModConst mod[kMaxMod];   // [0..k-1] data primes, [k] special prime
  NttTable tab[kMaxMod];
  // key-switch mod-down constants
  uint64_t p_half;                  // floor(p / 2)
  uint64_t p_half_mod[kMaxPrimes];  // floor(p/2) mod q_j
  uint64_t p_inv[kMaxPrimes];       // p^-1 mod q_j
  uint64_t p_inv_s[kMaxPrimes];     // Shoup quotient of p_inv
  // the same for the fp64 flavours: p, (p-1)/2 and the centred p^-1 mod q_j as doubles
  double p_f, p_half_f;
  double p_inv_f[kMaxPrimes];
  // plain lift
  uint64_t t, plain_thr;
  uint64_t lift_inc[kMaxPrimes];    // q_j - (t mod q_j)
  // CiphertextReencoder::Encode order (reference ct_reencoder.cpp:49-69)
  uint32_t enc_count;               // 2 * ExpansionRatio
  uint32_t enc_bits;                // floor(log2 t)
  uint8_t enc_poly[kMaxEnc], enc_res[kMaxEnc], enc_shift[kMaxEnc];
  // how many 128-bit products of two residues may be summed before reduction
  uint32_t lazy_limit;
  int32_t ntt_mode;  // NttMode
  // fp64 flavours: 1 when (bits of the largest modulus) + log2 N <= 52 -- the sums of a whole inverse transform
  // (they at most double per stage) then stay below 2^52 without the per-pass renormalisation
  uint32_t f64_lazy_inv;
  // MFMA scan, fp64 fold (all data moduli < 2^50): 2^32, 2^64, 2^96 mod q_j as centred doubles
  double fold_w[kMaxPrimes][3];
};

}  // namespace pirgpu
