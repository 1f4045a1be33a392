"""Helpers shared by the GPU parity tests: oracle PirParams <-> product PIRParameters."""
import numpy as np

import oracle
import pir_amd
from pir_amd.parameters import EncryptionParams, PIRParameters


def to_product_params(p: "oracle.PirParams") -> PIRParameters:
    enc = EncryptionParams(p.N, list(p.moduli), p.t)
    return PIRParameters(num_items=p.num_items, num_pt=p.num_pt, dimensions=list(p.dimensions),
                         encryption_parameters=enc, bytes_per_item=p.bytes_per_item,
                         items_per_plaintext=p.items_per_plaintext, bits_per_coeff=p.bits_per_coeff,
                         use_ciphertext_multiplication=p.use_ciphertext_multiplication)


def random_ct(orc, rng, n=1):
    out = np.empty((n, 2, orc.k, orc.N), dtype=np.uint64)
    for j in range(orc.k):
        out[:, :, j, :] = rng.integers(0, orc.moduli[j], size=(n, 2, orc.N), dtype=np.uint64)
    return out


def random_key(orc, rng):
    key = np.empty((orc.k, 2, orc.k + 1, orc.N), dtype=np.uint64)
    for i in range(orc.k + 1):
        key[:, :, i, :] = rng.integers(0, orc.moduli[i], size=(orc.k, 2, orc.N), dtype=np.uint64)
    return key


def all_to_all_in_process(recvs, sends, recv_splits, send_splits):
    """What torch.distributed.all_to_all_single does, between in-process 'ranks' (1-D tensors, element splits)."""
    G = len(sends)
    for dst in range(G):
        ro = 0
        for src in range(G):
            so = sum(send_splits[src][:dst])
            n = send_splits[src][dst]
            assert n == recv_splits[dst][src]
            recvs[dst][ro:ro + n].copy_(sends[src][so:so + n])
            ro += n
    # the copies run on torch's current stream, the library's kernels on its own non-blocking streams: nothing orders the
    # two but the host (a product caller uses Comm, which waits, or the entry points' after / then streams)
    if recvs and recvs[0].is_cuda:
        import torch
        torch.cuda.synchronize()
