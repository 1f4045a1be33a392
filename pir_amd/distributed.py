"""Multi-GPU glue (one process per GPU, torch.distributed over RCCL): row sharding of the
database and the sum of per-shard partial replies.  Not in the reference (single-threaded,
single-process); see DESIGN.md section 7."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_top: int, rank: int, world: int) -> Tuple[int, int]:
    """Top-level index range [begin, end) of dimension 0 held by `rank` (contiguous, balanced)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return (n_top * rank) // world, (n_top * (rank + 1)) // world


def all_reduce_reply(server, reply_tensor, dist) -> None:
    """Sum the ranks' partial replies in place and reduce mod q_j.

    reply_tensor: int64 CUDA tensor [reply_cts, 2, k, N] owned by the caller.  Partial replies
    are canonical residues (< q_j < 2^61), so the integer sum over <= 8 ranks cannot overflow.
    """
    import torch
    if not reply_tensor.is_cuda:
        raise RuntimeError("all_reduce_reply needs a CUDA tensor (RCCL); there is no CPU path")
    server.reply_copy_to_device(reply_tensor.data_ptr())     # waits for this rank's kernels
    dist.all_reduce(reply_tensor, op=dist.ReduceOp.SUM)      # RCCL over xGMI
    torch.cuda.current_stream().synchronize()
    server.reduce_fixup_device(reply_tensor.data_ptr())      # x mod q_j on the GPU


def all_reduce_batch_replies(server, reply_tensor, dist) -> None:
    """Batch-mode counterpart of all_reduce_reply: reply_tensor int64 CUDA [count, reply_cts, 2, k, N]."""
    import torch
    if not reply_tensor.is_cuda:
        raise RuntimeError("all_reduce_batch_replies needs a CUDA tensor (RCCL); there is no CPU path")
    server.batch_reply_copy_to_device(reply_tensor.data_ptr())   # waits for every worker stream
    dist.all_reduce(reply_tensor, op=dist.ReduceOp.SUM)
    torch.cuda.current_stream().synchronize()
    server.reduce_fixup_device_n(reply_tensor.data_ptr(), reply_tensor.shape[0] * reply_tensor.shape[1])


def owned_queries(count: int, rank: int, world: int):
    """Queries of a batch this rank expands: the contiguous block [begin, end) (count % world == 0)."""
    if count % world:
        raise ValueError("batch size must be a multiple of the world size for query-parallel expansion")
    per = count // world
    return rank * per, (rank + 1) * per


def run_batch_query_parallel(server, sv_all, replies, dist, rank: int, world: int) -> None:
    """One step over a staged batch on `world` GPUs holding row shards of the database:

      1. every rank expands only its own block of the batch's queries (oblivious expansion is the
         part of the path that does not shard by rows) straight into its slice of `sv_all`;
      2. one RCCL all-gather makes every query's NTT-form selection vector available everywhere;
      3. every rank multiplies all queries against its row shard (shared database passes);
      4. the partial replies are summed with one RCCL all-reduce and reduced mod q_j.

    sv_all:  int64 CUDA tensor [count, dim_sum, 2, k, N]   (all-gather buffer)
    replies: int64 CUDA tensor [count, reply_cts, 2, k, N] (all-reduce buffer)
    """
    import torch
    count = sv_all.shape[0]
    lo, hi = owned_queries(count, rank, world)
    server.batch_expand(lo, hi - lo, sv_all[lo].data_ptr())          # synchronous
    if world > 1:
        dist.all_gather_into_tensor(sv_all.view(-1), sv_all[lo:hi].reshape(-1))
        torch.cuda.current_stream().synchronize()
    server.batch_run_selectors(sv_all.data_ptr(), count)
    all_reduce_batch_replies(server, replies, dist)
