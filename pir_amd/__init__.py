"""pir_amd -- MI355X-native server query path of OpenMined/PIR.

Python front-end over the C ABI of ``libpirgpu.so`` (``include/pirgpu.h``): the
hand-written gfx950 kernels do all arithmetic; this package only marshals numpy
buffers and mirrors the reference's ``PIRServer`` / ``PIRDatabase`` interface
(reference pir/cpp/server.h, database.h).  There is no CPU fallback: every call
fails loudly when the extension or the GPU is missing.
"""
from __future__ import annotations

from . import capi
from .parameters import (BFV_DEFAULT, EncryptionParams, PIRParameters, create_pir_parameters,
                         generate_encryption_params, generate_galois_elts, next_power_two,
                         plain_modulus_batching, coeff_modulus_create)
from .server import PIRDatabase, PIRServer, PirGpuError, StatusCode
from .client import PIRClient

__all__ = ["capi", "BFV_DEFAULT", "EncryptionParams", "PIRParameters", "create_pir_parameters",
           "generate_encryption_params", "generate_galois_elts", "next_power_two", "plain_modulus_batching",
           "coeff_modulus_create", "PIRDatabase", "PIRServer", "PIRClient", "PirGpuError", "StatusCode"]
