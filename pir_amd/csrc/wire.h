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
#ifdef __cplusplus
}
#endif
