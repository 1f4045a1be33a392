#!/usr/bin/env python3
"""bench.py -- PIR server query path throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (PIRServer::ProcessRequest's query loop, reference
server.cpp:60-63 -> processQuery :173-195: oblivious expansion -> database scan ->
recursive re-encode / multiply-accumulate) over one batch of `--batch` independent
queries (default 16, different synthetic query ciphertexts), with the encoded database,
the Galois keys and the query ciphertexts already resident in HBM.  Groups of up to 8
queries are expanded together and share one pass over the database (the digit-sliced
int8-MFMA scan serves 8 queries per pass); two groups alternate on two lanes so the
bandwidth-bound scan of one overlaps the compute-bound expansion of the other; every
query does the full work.  `value` = queries/s over the timed steps; the single-query latency
(`--batch 1` behaviour, what benchmark.cpp:71-79 times per request) is measured as well
and reported as `latency_ms_single_query`, and the scan kernel's roofline comes from
those single-query runs (one scan launch per query, HIP events on the library's stream).
Workload = BASELINE.json configs[2]: N=4096, 2 RNS primes, DB = 2^20 x 288 B, d=2 (the
reference's own benchmark parameters, benchmark.cpp:17-23).

Multi-GPU (launched by torch.distributed.run, one rank per GPU), two modes:
  --dist-mode queries (what the default `auto` resolves to while the database fits one GPU): queries are independent, so they are the unit that is sharded --
      every GPU holds the whole packed database (1.27 GB of 288 GB at this workload) and serves
      its own `--batch` queries per step; no collective on the data path (the barrier and the
      max-over-ranks timing are the only communication); per-GPU work is fixed as N grows ->
      "scaling": "weak", value = all ranks' queries / time.
  --dist-mode rows: the database is row-sharded across ranks (for databases larger than one
      GPU), every rank expands its share of the batch, the selection vectors are all-gathered,
      every rank scans its rows and the per-shard reply ciphertexts are summed with one RCCL
      all-reduce + a mod-q fix-up.  Total work is fixed as N grows -> "scaling": "strong".
DESIGN.md section 7 explains why `queries` is the default at this database size.

The GPU leg uses synthetic inputs of the right shape (uniform residues); the
cpu_baseline leg (rank 0, N=1 only) times the CPU oracle -- the restatement of the
reference algorithm, built natively on this host -- on the same database / keys /
query and checks the GPU reply against it bit for bit.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PMC_FILE = "r01_pmc_scan_mfma_traffic.json"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def synthetic_inputs(pp, seed=42, n_queries=1, query_seed=None):
    """DB bytes, Galois keys and n_queries queries: uniform random of the right shape.
    query_seed (per rank in query-sharded multi-GPU runs) changes the queries only."""
    enc = pp.encryption_parameters
    N, q = enc.poly_modulus_degree, enc.coeff_modulus
    k = len(q) - 1
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 256, size=(pp.num_items, pp.bytes_per_item), dtype=np.uint8)
    nq = pp.dim_sum // N + 1
    query = np.empty((n_queries, nq, 2, k, N), dtype=np.uint64)
    qrng = rng if query_seed is None else np.random.default_rng(query_seed)
    for j in range(k):
        query[:, :, :, j, :] = qrng.integers(0, q[j], size=(n_queries, nq, 2, N), dtype=np.uint64)
    keys = {}
    import pir_amd
    for g in pir_amd.generate_galois_elts(N):
        key = np.empty((k, 2, k + 1, N), dtype=np.uint64)
        for i in range(k + 1):
            key[:, :, i, :] = rng.integers(0, q[i], size=(k, 2, N), dtype=np.uint64)
        keys[g] = key
    return raw, keys, query


def cpu_baseline(pp, raw, keys, query, gpu_reply, budget_s=25.0):
    """Times the CPU oracle (single thread, like the reference + SEAL 3.5.6) on the same inputs."""
    import oracle
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    enc = pp.encryption_parameters
    orc = oracle.Oracle(enc.poly_modulus_degree, enc.coeff_modulus, enc.plain_modulus, native=native)
    bits = pp.bits_per_coeff or oracle.bits_per_coeff(enc.plain_modulus)
    t0 = time.perf_counter()
    rc, db_ntt = orc.db_encode(raw.tobytes(), pp.num_items, pp.bytes_per_item, pp.items_per_plaintext, bits,
                               pp.num_pt)
    assert rc == 0
    t_encode = time.perf_counter() - t0
    times = []
    reply = None
    while True:
        t0 = time.perf_counter()
        rc, reply = orc.process_query(db_ntt, pp.dimensions, query, keys)
        times.append(time.perf_counter() - t0)
        assert rc == 0
        if sum(times) + times[-1] > budget_s or len(times) >= 8:
            break
    match = bool(np.array_equal(reply, gpu_reply))
    sec = float(np.median(times))
    # all host cores: as many independent single-threaded queries as there are cores, run concurrently (the
    # reference and SEAL 3.5.6 are single-threaded, so a CPU server scales over queries, not inside one);
    # ctypes releases the GIL during the call, the oracle context and the database are read-only
    all_cores = None
    try:
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(len(os.sched_getaffinity(0)), 64))
        if threads > 1:
            def one(_):
                rc_, rep_ = orc.process_query(db_ntt, pp.dimensions, query, keys)
                return rc_ == 0 and bool(np.array_equal(rep_, gpu_reply))
            t0 = time.perf_counter()
            with ThreadPoolExecutor(max_workers=threads) as ex:
                oks = list(ex.map(one, range(threads)))
            dt = time.perf_counter() - t0
            all_cores = {"value": threads / dt, "unit": "queries/s", "cores": threads,
                         "speedup_over_one_core": (threads / dt) * sec,   # < cores when a CPU quota caps the box
                         "sample": "%d concurrent single-threaded queries (one per core), wall %.2f s; all bit-exact=%s"
                                   % (threads, dt, all(oks))}
    except Exception as e:   # reported extra only
        all_cores = {"error": repr(e)}
    return {
        "value": 1.0 / sec, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": "%d full queries of the same workload (same DB, keys, query) on the CPU oracle, median; "
                  "ms_per_query=%.1f; db_encode_s=%.1f; native_build=%s; gpu_reply_bit_exact=%s"
                  % (len(times), sec * 1e3, t_encode, native, match),
        "ms_per_query": sec * 1e3, "bit_exact_vs_gpu": match, "all_cores": all_cores,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="queries per step (per request)")
    ap.add_argument("--workers", type=int, default=0, help="queries in flight on the GPU (0 = min(batch, 16): two "
                                                           "groups of 8 share one database pass each and overlap)")
    ap.add_argument("--latency-runs", type=int, default=30, help="single-query runs for latency + roofline")
    ap.add_argument("--log-items", type=int, default=20, help="database = 2^log_items x 288 B")
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-mode", choices=["auto", "queries", "rows"],
                    default=os.environ.get("PIRGPU_DIST_MODE", "auto"),
                    help="multi-GPU: 'queries' = every GPU holds the whole database and serves its own batch "
                         "(independent queries, no data-path collective, weak scaling); 'rows' = the database is "
                         "row-sharded, replies summed with an RCCL all-reduce (strong scaling; for databases "
                         "larger than one GPU); 'auto' = queries while the encoded database (both layouts) takes "
                         "less than half of one GPU's memory, else rows")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json config (1-based as in SURVEY.md section 8): 3 = the headline workload; "
                         "2/4/5 are reference points (other ring degrees / database shapes)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: anything native libraries print there (RCCL's version banner)
    # is sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    import pir_amd

    dist = None
    # PIRGPU_FORCE_DIST=1 / =queries: run the multi-GPU code path (rows / queries mode) with a single rank
    force = os.environ.get("PIRGPU_FORCE_DIST", "")
    use_dist = world > 1 or force in ("1", "rows", "queries")
    if force in ("1", "rows"):
        args.dist_mode = "rows"
    elif force == "queries":
        args.dist_mode = "queries"
    dist_mode_requested = args.dist_mode
    row_sharded = use_dist and args.dist_mode == "rows"   # 'auto' is resolved once the database size is known
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload: BASELINE.json configs[2] (benchmark.cpp:17-23 parameters)
    item_bytes = 288
    if args.config == 2:      # N=4096, 2 primes, DB = 2^16 x 288 B, d=1
        enc = pir_amd.generate_encryption_params(4096, 24)
        args.log_items, args.dims = 16, 1
    elif args.config == 4:    # N=8192, 3 data primes (first 3 of BFVDefault + its last as special), 2^22 x 1 KB, d=2
        m = pir_amd.BFV_DEFAULT[8192]
        enc = pir_amd.generate_encryption_params(8192, 24, coeff_modulus=m[:3] + [m[4]])
        args.log_items, args.dims, item_bytes = 22, 2, 1024
    elif args.config == 5:    # N=16384, 4 data primes (first 4 of BFVDefault + its last as special), 2^24 x 288 B, d=2
        m = pir_amd.BFV_DEFAULT[16384]
        enc = pir_amd.generate_encryption_params(16384, 24, coeff_modulus=m[:4] + [m[8]])
        args.log_items, args.dims = 24, 2
    else:
        enc = pir_amd.generate_encryption_params(4096, 24)
    pp = pir_amd.create_pir_parameters(1 << args.log_items, item_bytes, args.dims, enc)
    if args.dist_mode == "auto":
        # u64 staging copy + operand-layout copy (at most 7/8 of it, plus tile padding)
        db_bytes = pp.num_pt * (len(enc.coeff_modulus) - 1) * enc.poly_modulus_degree * 8 * 2
        hbm = torch.cuda.get_device_properties(local_rank).total_memory
        args.dist_mode = "rows" if db_bytes > hbm // 2 else "queries"
        row_sharded = use_dist and args.dist_mode == "rows"
    batch = max(1, args.batch)
    workers = args.workers if args.workers > 0 else min(batch, 16)
    # query-sharded runs: same database and keys everywhere, every rank draws its own queries
    raw, keys, queries = synthetic_inputs(pp, n_queries=batch,
                                          query_seed=1000 + rank if use_dist and not row_sharded else None)
    query = queries[0]

    from pir_amd.distributed import (all_reduce_batch_replies, all_reduce_reply, run_batch_query_parallel,
                                     shard_range)
    shard = shard_range(pp.dimensions[0], rank, world) if world > 1 and row_sharded else None
    db = pir_amd.PIRDatabase.Create(pp, device=local_rank, shard=shard)
    t0 = time.perf_counter()
    db.populate(raw)
    t_populate = time.perf_counter() - t0
    srv = pir_amd.PIRServer(db, pp) if shard else pir_amd.PIRServer.Create(db, pp)
    srv.set_galois_keys(keys)

    reply_cts = db.reply_ct_count()
    k, N = srv.k, srv.N
    dev = "cuda:%d" % local_rank
    red1 = redb = sv_all = None
    # multi-GPU: query-parallel expansion + all-gather of the selection vectors when the batch
    # divides evenly over the ranks; otherwise every rank expands every query (replicated)
    query_parallel = row_sharded and batch % world == 0 and os.environ.get("PIRGPU_REPLICATED_EXPANSION") != "1"
    if row_sharded:
        red1 = torch.empty((reply_cts, 2, k, N), dtype=torch.int64, device=dev)
        redb = torch.empty((batch, reply_cts, 2, k, N), dtype=torch.int64, device=dev)
        if query_parallel:
            sv_all = torch.empty((batch, pp.dim_sum, 2, k, N), dtype=torch.int64, device=dev)

    def barrier():
        srv.sync()                      # the library's own streams (not torch's current stream)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- (1) single-query latency + scan-kernel roofline (one scan launch per query)
    srv.stage_query(query)
    for _ in range(3):
        srv.run_staged()
        if row_sharded:
            all_reduce_reply(srv, red1, dist)
    srv.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.latency_runs):
        srv.run_staged()
        if row_sharded:
            all_reduce_reply(srv, red1, dist)
    barrier()
    latency_ms = (time.perf_counter() - t0) / args.latency_runs * 1e3
    timings = srv.last_timings()
    srv.set_profiling(False)
    single_reply = srv.fetch_reply() if world == 1 else None

    # ---- (2) throughput: `batch` queries per step, `workers` in flight
    srv.set_concurrency(workers)
    srv.stage_batch(queries)

    def step():
        if query_parallel:
            run_batch_query_parallel(srv, sv_all, redb, dist, rank, world)
        else:
            srv.run_batch()
            if row_sharded:
                all_reduce_batch_replies(srv, redb, dist)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # query-sharded: every rank served its own `batch` queries per step
        total_batch = batch * world if use_dist and not row_sharded else batch
        qps = args.steps * total_batch / elapsed
        scan_bytes = srv.scan_bytes()                   # bytes one database pass must read
        info = srv.scan_info()
        u64_bytes = pp.num_pt * (len(enc.coeff_modulus) - 1) * enc.poly_modulus_degree * 8
        scan_ms = timings["scan_ms"]
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per scan launch from the committed PMC passes (profiles/), same workload only
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            if args.config == 3 and args.log_items == 20 and args.dims == 2 and world == 1 and info["single_query_mfma"]:
                traffic = pm["traffic_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "PIR queries/sec (ms/query in ms_per_step), N=%d DB=2^%d x %dB d=%d"
                      % (enc.poly_modulus_degree, args.log_items, item_bytes, args.dims),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if row_sharded or world == 1 else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "N=%d, %d RNS data primes (%s bit | special %d bit), t=24 bit, DB=2^%d x %dB, "
                                   "d=%d, dims=%s, num_pt=%d, %d queries/step, %d in flight (BASELINE.json configs[%d])"
                                   % (enc.poly_modulus_degree, len(enc.coeff_modulus) - 1,
                                      ",".join(str(q.bit_length()) for q in enc.coeff_modulus[:-1]),
                                      enc.coeff_modulus[-1].bit_length(), args.log_items, item_bytes, args.dims,
                                      pp.dimensions, pp.num_pt, batch, workers, args.config - 1),
                       "queries_per_step": total_batch, "queries_per_step_per_gpu": batch, "workers": workers,
                       "dist_mode": "%s (requested: %s)" % (args.dist_mode, dist_mode_requested),
                       "parallelism": ("single GPU" if world == 1 else
                                       ("rows sharded over %d GPU(s), %s, RCCL all-reduce of replies"
                                        % (world, "query-parallel expansion + RCCL all-gather of selection vectors"
                                           if query_parallel else "replicated expansion")) if row_sharded else
                                       ("queries sharded over %d GPUs: every GPU holds the whole database and "
                                        "serves its own %d queries per step, no data-path collective" % (world, batch)))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes, profiles/" + PMC_FILE,
                         "kernel": ("scan_mfma_kernel<%d digits, %d k-steps> (int8 MFMA digit products, 1 query; "
                                    "the same pass serves up to 8)" % (info["digits"], info["ksteps"])) if info["single_query_mfma"]
                         else ("scan_mq_kernel<4 rows/wave, 1 query>" if args.dims > 1
                               else "scan_kernel + reduce_splits (column split)"),
                         "kernel_ms": scan_ms, "algorithmic_bytes": scan_bytes,
                         "launches_averaged": timings["runs"],
                         # SURVEY 8(d) prices the scan in u64 residues (num_pt*k*N*8 per query) whatever the
                         # stored layout: reported for comparison only -- these are NOT bytes the kernel moves
                         "survey_8d_u64_equivalent": {
                             "B_q": u64_bytes,
                             "single_query_launch_GBps": u64_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0,
                             "batched_B_q_times_qps_GBps": u64_bytes * qps / max(world, 1) / 1e9,
                             "note": "equivalent rates; can exceed the HBM peak because the packed layout is "
                                     "smaller than u64 and one pass serves up to 8 queries"}},
            "latency_ms_single_query": round(latency_ms, 4),
            "single_query_qps": round(1e3 / latency_ms, 1),
            "phases_ms_single_query": {kk: round(v, 4) for kk, v in timings.items() if kk.endswith("_ms")},
            "db_populate_s": round(t_populate, 2),
        }
        if world == 1 and args.config == 3:
            # wire-level ProcessRequest (what benchmark.cpp:71-79 times): serialized pir.Request in host
            # memory -> serialized pir.Response, incl. parsing, H2D of keys + query, D2H, serialisation
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import seal_wire as W
                mods = enc.coeff_modulus
                gk = W.save_galois_keys(keys, N, W.parms_id(N, mods, enc.plain_modulus))
                req = W.save_request([query], gk, W.parms_id(N, mods[:-1], enc.plain_modulus))
                srv.set_concurrency(1)
                wt = []
                for _ in range(6):
                    t0 = time.perf_counter()
                    resp = srv.ProcessRequest(req)
                    wt.append((time.perf_counter() - t0) * 1e3)
                same = bool(np.array_equal(W.load_response(resp)[0], single_reply))
                out["wire_process_request_ms"] = {"first_request_with_key_upload": round(wt[0], 3),
                                                  "repeat_client_keys_cached": round(float(np.median(wt[1:])), 3),
                                                  "request_bytes": len(req), "response_bytes": len(resp),
                                                  "response_equals_residue_path": same}
            except Exception as e:   # measurement extra only
                out["wire_process_request_ms"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            batch_replies = srv.fetch_batch()
            out["batch_reply0_equals_single_query_reply"] = bool(np.array_equal(batch_replies[0], single_reply))
            out["cpu_baseline"] = cpu_baseline(pp, raw, keys, query, single_reply)
            out["speedup_vs_cpu_baseline"] = qps / out["cpu_baseline"]["value"]
        print(json.dumps(out), file=result_out)
        result_out.flush()
    if use_dist:
        if world == 1 and rank == 0 and row_sharded:   # forced single-rank run: the reduced replies must equal the plain ones
            ok = bool(np.array_equal(redb.cpu().numpy().view(np.uint64), srv.fetch_batch()))
            print("forced-dist check: all-reduced batch replies equal plain replies: %s" % ok, file=sys.stderr)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
