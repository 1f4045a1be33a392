#!/usr/bin/env python3
"""bench.py -- PIR server query path throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (PIRServer::ProcessRequest's query loop, reference
server.cpp:60-63 -> processQuery :173-195: oblivious expansion -> database scan ->
recursive re-encode / multiply-accumulate) over one batch of `--batch` independent
queries (default 64 for the WHOLE job, different synthetic query ciphertexts), with the encoded
database, the Galois keys and the query ciphertexts already resident in HBM.  Groups of up to 8
queries are expanded together and share one pass over the database (the digit-sliced
int8-MFMA scan serves 8 queries per pass); two groups alternate on two lanes so the
bandwidth-bound scan of one overlaps the compute-bound expansion of the other; every
query does the full work.  `value` = queries/s over the timed steps (defaults: 200 steps of 64 queries,
about 2.3 s of GPU time); the single-query latency (`--batch 1` behaviour, what benchmark.cpp:71-79 times per
request) is measured as well and reported as `latency_ms_single_query`, and the scan kernel's roofline comes from
those single-query runs (one scan launch per query, HIP events on the library's stream).
Workload = BASELINE.json configs[2]: N=4096, 2 RNS primes, DB = 2^20 x 288 B, d=2 (the
reference's own benchmark parameters, benchmark.cpp:17-23).

Multi-GPU (launched by torch.distributed.run, one rank per GPU).  The headline `value` is ALWAYS the
north-star mode: the database SHARDED across the GPUs ("scaling": "strong" -- the same 64 queries per
step whatever the GPU count) -- by row (`rows`) or by NTT slot (`slots`); every form is timed over the full W + K steps
and the fastest is the headline (`exchange_autotune`, `config.exchange`):
  slots    every rank holds 1/G of the NTT slots of EVERY plaintext (the base case of PIRDatabase::multiply is a dyadic
           product, independent per slot), expands batch/G of the step's queries, and the step is two all-to-alls: the
           packed column selectors' slot slices out (a rank receives 1/G of each query's), the row sums back to the
           rank that expanded the query, which runs the upper level itself -- no row-selector exchange, no reduce
           (DESIGN.md section 7.1);
  rows     every rank holds 1/G of the rows, expands batch/G of the step's queries (the expansion does not
           shard by rows, so it is partitioned by query), packs their column selectors into the scan's
           operand layout and lays their row selectors out by owner; one RCCL all-gather (packed column
           selectors) and one all-to-all (each rank receives only its own rows' selectors); every rank scans
           its shard once per group of 8 queries and runs the upper level on its rows; one reduce-scatter
           sums the per-shard reply ciphertexts (rank r ends with the replies of the queries it expanded),
           then x mod q_j ("packed").  A second form moves no selectors at all: every rank expands all queries
           itself on its shard and only the reduce-scatter remains ("replicated").  With several ranks BOTH forms
           are timed over the full W + K steps and the faster one is the headline (`exchange_autotune`,
           `config.exchange`): which one wins depends on what the links of the machine sustain.
           `--exchange u64` all-gathers whole selection vectors instead (any d).
  queries  (reference point, reported in the same JSON line as `replicas_reference` when --dist-mode both,
           the default): every GPU holds the whole database and serves batch/G of the same queries, no
           data-path collective.
DESIGN.md section 7 has the byte counts of the exchange and what bounds each mode.

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
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PMC_GLOB = "r*_pmc_scan_traffic.json"     # newest round's PMC passes (tools/pmc_scan_traffic.sh), per config
VALU_GLOB = "r*_valu_roofline.json"
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


def cpu_baseline_reference_case(pp, raw, keys, query_ct):
    """cpu_baseline leg of the reference sweep: the CPU oracle's processQuery (one core) on one case's database, keys
    and query ciphertext; returns (median ms, reply residues)."""
    import oracle
    try:
        oracle.build(native=True)
        native = True
    except Exception:
        native = False
    enc = pp.encryption_parameters
    orc = oracle.Oracle(enc.poly_modulus_degree, enc.coeff_modulus, enc.plain_modulus, native=native)
    bits = pp.bits_per_coeff or oracle.bits_per_coeff(enc.plain_modulus)
    rc, db_ntt = orc.db_encode(raw.tobytes(), pp.num_items, pp.bytes_per_item, pp.items_per_plaintext, bits, pp.num_pt)
    assert rc == 0
    times, reply = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        rc, reply = orc.process_query(db_ntt, pp.dimensions, query_ct, keys)
        times.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    return float(np.median(times)), reply


def c_abi_request_timer(srv, request: bytes):
    """pirgpu_process_request timed AT THE C ABI: serialized pir.Request bytes in, malloc'd serialized pir.Response out,
    released with pirgpu_free -- what benchmark.cpp:71-79 times around PIRServer::ProcessRequest.  (The Python mirror
    PIRServer.ProcessRequest additionally copies the megabyte of response into a bytes object: the binding's cost.)
    Returns a function that serves the request once and returns (milliseconds, response length)."""
    import ctypes as C
    lib, handle = srv.lib, srv.db.handle
    buf = np.frombuffer(request, dtype=np.uint8)
    ptr = buf.ctypes.data_as(C.POINTER(C.c_uint8))
    n = len(request)

    def once():
        resp, rlen = C.c_void_p(), C.c_size_t(0)
        t0 = time.perf_counter()
        rc = lib.pirgpu_process_request(handle, ptr, n, C.byref(resp), C.byref(rlen))
        dt = (time.perf_counter() - t0) * 1e3
        if rc:
            raise RuntimeError("pirgpu_process_request failed: %d %s" % (rc, lib.pirgpu_last_error(handle).decode()))
        lib.pirgpu_free(resp)
        return dt, int(rlen.value)
    once.keepalive = buf
    return once


def reference_sweep(pir_amd, device=0, log_sizes=(8, 10, 12, 14, 16), reps=12, with_cpu=True):
    """The reference's own benchmark at the reference's own sizes (benchmark.cpp:17-23, 56-107): 2^8 .. 2^16 items of
    288 bytes, d = 2, N = 4096, 24-bit t, ONE query per request.  Per size the four registered cases -- SetupDb,
    ClientCreateRequest, ServerProcessRequest (wire level, seed-compressed keys like the reference client sends: the
    first request of a client, and with that client's keys resident), ClientProcessResponse -- the CPU oracle's
    processQuery on one core beside the GPU's, whether the two replies are the same bits, and whether the client gets
    its item back (at these sizes the parameters leave a positive noise budget: it must)."""
    enc = pir_amd.generate_encryption_params(4096, 24)
    rows = []
    for li in log_sizes:
        pp = pir_amd.create_pir_parameters(1 << li, 288, 2, enc)
        rng = np.random.default_rng(1000 + li)
        raw = rng.integers(0, 256, size=(pp.num_items, pp.bytes_per_item), dtype=np.uint8)
        idx = int(rng.integers(0, pp.num_items))
        t_db = []
        db = None
        for _ in range(3):
            if db is not None:
                db.close()
            t0 = time.perf_counter()
            db = pir_amd.PIRDatabase.Create(pp, device=device)
            db.populate(raw)
            db.finalize(release_staging=False)
            db.lib.pirgpu_sync(db.handle)
            t_db.append((time.perf_counter() - t0) * 1e3)
        srv = pir_amd.PIRServer.Create(db, pp)
        cl = pir_amd.PIRClient.Create(pp, seed=b"reference-sweep-%d" % li)
        t_req = []
        for _ in range(5):
            t0 = time.perf_counter()
            req = cl.CreateRequest([idx])
            t_req.append((time.perf_counter() - t0) * 1e3)
        t_srv = []
        t0 = time.perf_counter()
        resp = srv.ProcessRequest(req)                    # the client's FIRST request (keys parsed, re-sampled, uploaded)
        t_srv.append((time.perf_counter() - t0) * 1e3)
        once = c_abi_request_timer(srv, req)
        for _ in range(reps):
            t_srv.append(once()[0])
        t_rsp = []
        for _ in range(5):
            t0 = time.perf_counter()
            items = cl.ProcessResponse([idx], resp)
            t_rsp.append((time.perf_counter() - t0) * 1e3)
        # device-resident processQuery (residues in, residues out) with the same client's keys, for the latency floor
        q_ct = cl.create_query_for(idx)
        slot = srv.install_keyset(b"reference-sweep-residues-%d" % li, cl.galois_keys())
        srv.use_keyset(slot)
        srv.stage_query(q_ct)
        for _ in range(3):
            srv.run_staged()
        srv.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            srv.run_staged()
        srv.sync()
        dev_ms = (time.perf_counter() - t0) / reps * 1e3
        gpu_reply = srv.fetch_reply()
        srv.use_keyset(0)
        row = {"items": 1 << li, "log2_items": li, "num_pt": pp.num_pt, "dimensions": list(pp.dimensions),
               "expansion_nodes": 1 << max(0, (pp.dim_sum - 1).bit_length()),
               "SetupDb_ms": round(float(np.median(t_db)), 3),
               "ClientCreateRequest_ms": round(float(np.median(t_req)), 3),
               "ServerProcessRequest_ms": {"first_request_seeded_keys": round(t_srv[0], 3),
                                           "keys_resident_median": round(float(np.median(t_srv[1:])), 3),
                                           "keys_resident_min": round(float(np.min(t_srv[1:])), 3),
                                           "device_resident_query": round(dev_ms, 3)},
               "ClientProcessResponse_ms": round(float(np.median(t_rsp)), 3),
               "request_bytes": len(req), "response_bytes": len(resp),
               "item_recovered": bool(items[0] == raw[idx].tobytes()),
               "scan": "int8-MFMA" if srv.scan_info()["mfma"] else "64-bit multiply-accumulate (fewer than 8 rows)"}
        if with_cpu:
            # a query ciphertext is fresh randomness every time: both sides get THIS one
            q2 = cl.create_query_for(idx)
            cpu_ms, cpu_reply = cpu_baseline_reference_case(pp, raw, cl.galois_keys(), q2)
            srv.use_keyset(slot)
            same = bool(np.array_equal(srv.process_query(q2), cpu_reply))
            srv.use_keyset(0)
            row["cpu_oracle_ServerProcessRequest_ms_one_core"] = round(cpu_ms, 2)
            row["bit_exact"] = same
            row["gpu_over_cpu_one_core"] = round(cpu_ms / row["ServerProcessRequest_ms"]["keys_resident_median"], 1)
        srv.release_keyset(slot)
        db.close()
        rows.append(row)
    return {"rows": rows,
            "note": "benchmark.cpp's four cases at the sizes the reference registers them for (2^8 .. 2^16 items, 288 B, "
                    "d = 2, N = 4096, 24-bit t, QUERIES_PER_REQUEST = 1). ServerProcessRequest is timed at the C ABI "
                    "(pirgpu_process_request: serialized pir.Request -> malloc'd serialized pir.Response; seed-compressed "
                    "keys, parsing, key compare, PCIe both ways, serialisation inside; the first request through the "
                    "Python mirror); the CPU figure is the oracle's processQuery on residues (no parsing), one core."}


def build_workload(args, pir_amd):
    """BASELINE.json configs (1-based as in SURVEY.md section 8): parameters of the selected one."""
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
    else:                     # the headline workload: benchmark.cpp:17-23 parameters at 2^20 items
        enc = pir_amd.generate_encryption_params(4096, 24)
    pp = pir_amd.create_pir_parameters(1 << args.log_items, item_bytes, args.dims, enc)
    return enc, pp, item_bytes


BLOCK_LOG = []           # per measurement: the per-block elapsed seconds behind the reported median


def timed_steps(step, barrier, steps, warmup, dist, use_dist, torch, dev, min_seconds=1.0, max_blocks=9):
    """The bench contract: W untimed steps, then exactly K steps between barrier + synchronise, MAX over ranks.
    When that timed block is shorter than `min_seconds` (the driver's --steps 20 is 0.24 s at the headline config, and the
    same kernel measures 0.195 / 0.207 ms in consecutive runs), the block of K steps is REPEATED -- every block
    bracketed the same way -- and the median block is reported; the block count follows from the first block's
    all-reduced time, so every rank runs the same number."""
    def block():
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(warmup):
        step()
    times = [block()]
    want = 1 if times[0] >= min_seconds else min(max_blocks, int(np.ceil(min_seconds / max(times[0], 1e-6))))
    if want > 1 and want % 2 == 0:
        want += 1                      # odd: the median is a measured block
    while len(times) < want:
        times.append(block())
    BLOCK_LOG.append([round(t, 6) for t in times])
    return float(np.median(times))


def batch_launch_block(bs, scan_bytes, info, k, N):
    """roofline.batch_launch: the scan launch as it runs inside the batched step (pirgpu_batch_scan_timings)."""
    if "error" in bs or not bs.get("launches"):
        return bs
    row_tiles = (info["rows"] + 15) // 16
    sel = scan_bytes // row_tiles                      # packed selectors of a group: the database's bytes of ONE row tile
    outb = bs["queries"] * info["rows"] * 2 * k * N * 8
    moved = scan_bytes + sel + outb
    sec = bs["mean_ms"] * 1e-3
    return {**bs, "stored_database_bytes": scan_bytes, "packed_selector_bytes": sel, "row_sum_bytes": outb,
            "achieved_GBps": moved / sec / 1e9, "frac": moved / sec / 1e9 / HBM_PEAK_GBS,
            "frac_database_bytes_only": scan_bytes / sec / 1e9 / HBM_PEAK_GBS,
            "note": "HIP events on the lane's stream around the scan launches of 6 more steps after the timed region: "
                    "`workgroups` persistent workgroups (half the chip) beside the other lane's transform kernels, 8 queries "
                    "per pass -- bound per CU (loads in flight), not by HBM; mean = as it shares the chip, min = its least "
                    "disturbed instance"}


def scan_source_sha16():
    """What the PMC traffic file is stamped with (tools/pmc_scan_traffic.sh): the scan kernel's source as it was profiled."""
    import hashlib
    with open(os.path.join(ROOT, "pir_amd", "csrc", "scan_mfma.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


KERNEL_SOURCES = ("ntt_kernels.hip", "ntt_core.h", "kernels.hip", "arith.h")     # tools/valu_roofline.py stamps these


def kernel_sources_sha16():
    """What the VALU table is stamped with (tools/valu_roofline.py): the transform kernels' sources as they were profiled."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "pir_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def hip_runtime_in_use():
    """Path of the libamdhip64 this process has mapped (torch's bundled copy or the system's: whichever was loaded first)."""
    try:
        libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
        return libs[0] if len(libs) == 1 else libs
    except Exception:
        return None


def recorded_compute():
    """roofline_compute: the VALU-issue table of the transform kernels as RECORDED by the newest committed PMC passes
    (tools/pmc_kernels.sh + tools/valu_roofline.py).  `stale`: False when the kernels' sources are the ones that were
    profiled, True when they have changed since -- the table is then WITHHELD (only the stamp and the file name are
    reported: per-kernel numbers of other kernels are not evidence) --, None when the file carries no stamp."""
    import glob
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", VALU_GLOB)))[-1]
        table = json.load(open(path))
    except Exception:
        return None
    stamp = table.get("kernel_sources_sha16")
    if not stamp:
        return {**table, "stale": None, "file": os.path.basename(path),
                "stale_note": "no kernel_sources_sha16 stamp in this file: whether the kernels have changed since cannot be told"}
    have = kernel_sources_sha16()
    if stamp != have:
        return {"stale": True, "file": os.path.basename(path), "kernel_sources_sha16_profiled": stamp,
                "kernel_sources_sha16_running": have, "commit_profiled": table.get("commit"),
                "note": "the transform kernels' sources (%s) have changed since these PMC passes: table withheld; re-run "
                        "tools/pmc_kernels.sh + tools/valu_roofline.py at this commit" % ", ".join(KERNEL_SOURCES)}
    return {**table, "stale": False, "file": os.path.basename(path)}


def recorded_traffic(config, log_items, world, single_query_mfma):
    """roofline.traffic: HBM bytes per single-query scan launch as RECORDED by the newest committed PMC passes of this
    workload (rocprofv3 --pmc cannot run inside the bench); `stale` says whether the scan kernel's source has changed
    since those passes (True), is the one that was profiled (False), or cannot be told (None: no stamp in the file)."""
    import glob
    try:
        pmf = sorted(glob.glob(os.path.join(ROOT, "profiles", PMC_GLOB)))[-1]
        pm = json.load(open(pmf))
        ent = pm["configs"].get("cfg%d" % config)
        default_shape = log_items == {2: 16, 3: 20, 4: 22, 5: 24}[config]
        if not (ent and "traffic_bytes_per_launch" in ent and default_shape and world == 1 and single_query_mfma):
            return None, None, None
        stamp = pm.get("scan_source_sha16")
        stale = None if not stamp else stamp != scan_source_sha16()
        src = "RECORDED, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gfx950 x2 correction " \
              "on FETCH_SIZE) of this workload at commit %s, profiles/%s" % (pm.get("commit"), os.path.basename(pmf))
        return ent["traffic_bytes_per_launch"], src, stale
    except Exception:
        return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="queries per step: the WHOLE job's, whatever the GPU count "
                                                         "(strong scaling); a multiple of 8 x gpus keeps every rank's "
                                                         "expansion groups full")
    ap.add_argument("--workers", type=int, default=0, help="queries in flight on one GPU (0 = min(batch, 16): two "
                                                           "groups of 8 share one database pass each and overlap)")
    ap.add_argument("--latency-runs", type=int, default=30, help="single-query runs for latency + roofline")
    ap.add_argument("--log-items", type=int, default=20, help="database = 2^log_items x 288 B")
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-mode", choices=["both", "rows", "queries"],
                    default=os.environ.get("PIRGPU_DIST_MODE", "both"),
                    help="multi-GPU: 'rows' = the database is row-sharded across the GPUs, every rank expands its "
                         "share of the batch, selectors are exchanged (packed), replies summed over RCCL -- the "
                         "north-star mode and always the headline `value`; 'queries' = every GPU holds the whole "
                         "database and serves its own share of the batch (replicas, no data-path collective); "
                         "'both' (default) = time rows for `value` and replicas as the named extra `replicas_reference`")
    ap.add_argument("--exchange", choices=["auto", "packed", "u64", "replicated", "slots"],
                    default=os.environ.get("PIRGPU_EXCHANGE", "auto"),
                    help="rows mode: what is exchanged per query -- packed = column selectors in the scan's operand "
                         "layout (all-gather) + each rank's own row selectors (all-to-all); u64 = whole NTT-form "
                         "selection vectors (all-gather); replicated = nothing: every rank expands every query itself "
                         "and only the partial replies are reduced; slots = the database is sharded by NTT SLOT instead of by "
                         "row (every rank holds 1/N of the slots of every plaintext): all-to-all of each query's packed "
                         "column selectors' slot slices, all-to-all of the row sums back to the query's owner, no reduce; "
                         "auto = with several GPUs, replicated, packed and slots are all run for the full W + K "
                         "steps on this machine's links and the fastest one is the headline (`exchange_autotune` in the line; "
                         "PIRGPU_EXCHANGE_AUTOTUNE=0: replicated at 2 GPUs -- one xGMI link --, packed beyond); u64 when a "
                         "shard cannot take the packed path (d != 2, no MFMA scan)")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="BASELINE.json config (1-based as in SURVEY.md section 8): 3 = the headline workload; "
                         "2/4/5 are reference points (other ring degrees / database shapes)")
    ap.add_argument("--reference-sweep", action="store_true",
                    help="only the reference's own benchmark at its own sizes (benchmark.cpp:102-104: 2^8 .. 2^16 items, "
                         "one query per request) with the CPU oracle beside it; prints one JSON line")
    ap.add_argument("--keep-staging", action="store_true",
                    help="keep the u64 staging copy of the database next to the operand-layout copy (default: "
                         "released for d >= 2, one copy of the database in HBM)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # `python bench.py --gpus N` without a launcher environment: start the N ranks ourselves (fresh child processes,
    # before anything here touches a GPU), relay rank 0's JSON line, fail if any rank fails
    # PIRGPU_BENCH_SHARE_GPU=1 (a test facility for one-GPU boxes): all ranks use device 0 and the collectives go
    # over gloo (RCCL refuses two ranks on one device) -- every line of the multi-rank flow except the RCCL calls;
    # the line then says backend gloo / rccl_ranks 0 and is NOT a multi-GPU measurement
    share_gpu = os.environ.get("PIRGPU_BENCH_SHARE_GPU", "") == "1"
    from pir_amd import launcher
    if args.gpus > 1 and not launcher.launched_by_a_launcher():
        sys.exit(launcher.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                      check_devices=not share_gpu))

    # stdout carries exactly one JSON line: anything native libraries print there (RCCL's version banner)
    # is sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    # Which HIP runtime serves the library.  This image's torch wheel bundles its own libamdhip64 / libhsa-runtime64 (ROCm
    # 7.0); the system has ROCm 7.2.  Whichever is loaded FIRST serves the whole process (same soname), and torch.cuda
    # cannot initialise on the system's copy ("No HIP GPUs are available").  Under torch's copy the wire path's group-wise
    # reply downloads run as blit KERNELS (__amd_rocclr_copyBuffer: 256 workgroups x 512 threads sitting on every CU for
    # the PCIe transfer), under the system runtime -- what a C++ server linking libpirgpu.so gets -- as SDMA copies
    # (profiles/r06_wire_copy_engines.txt, profiles/r06_d2h_probe.txt).  Measured on one box, three alternating runs each
    # (profiles/r06_ab_hip_runtime.txt): headline 5 390 against 5 389 queries/s, 64 clients' requests through the wire
    # 5 329 against 5 310, one synchronous caller 4 655 against 4 628, a new client's first request 2.3 - 3.1 against
    # 1.95 - 2.0 ms -- the engine that carries the downloads does not move the step.  The default therefore stays torch
    # first (the bracket around the timed steps is torch.cuda.synchronize(), as the bench contract words it);
    # PIRGPU_BENCH_TORCH_FIRST=0 runs a single-GPU bench WITHOUT torch in the process: the library loads the system
    # runtime and the bracket is pirgpu_device_synchronize() (hipDeviceSynchronize of that runtime).
    launcher_world = int(os.environ.get("WORLD_SIZE", "1"))
    torch_first = (os.environ.get("PIRGPU_BENCH_TORCH_FIRST", "1") == "1" or launcher_world > 1
                   or bool(os.environ.get("PIRGPU_FORCE_DIST")))
    if torch_first:
        import torch
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible -- one process per GPU needs %d devices"
                             % (rank, local_rank, torch.cuda.device_count(), world))
        device_synchronize = torch.cuda.synchronize
    else:
        torch = None
        from pir_amd import capi as _capi
        _capi.load()                                  # pulls in the system's libamdhip64
        sync_servers = []                             # the server whose context the bracket goes through (set below)

        def device_synchronize():
            sync_servers[-1].device_synchronize()     # pirgpu_device_synchronize: hipDeviceSynchronize of that runtime
    import pir_amd
    from pir_amd import distributed as D

    dist = None
    # PIRGPU_FORCE_DIST=1: run the multi-GPU code path (process group, collectives, fix-up) with a single rank
    force_env = os.environ.get("PIRGPU_FORCE_DIST", "")     # "both": also run the replica leg with the single rank
    force = force_env in ("1", "rows", "queries", "both")
    use_dist = world > 1 or force
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = "cuda:%d" % local_rank

    if args.reference_sweep:
        sweep = reference_sweep(pir_amd, device=local_rank, reps=24)
        print(json.dumps({"reference_sweep": sweep}), file=result_out)
        result_out.flush()
        return

    enc, pp, item_bytes = build_workload(args, pir_amd)
    k, N = len(enc.coeff_modulus) - 1, enc.poly_modulus_degree
    batch = max(1, args.batch)
    if use_dist and batch % world:
        batch += world - batch % world
    per_rank = batch // world if use_dist else batch
    # two groups of 8 alternate on two lanes: 16 working sets, whatever the number of queries this rank expands
    workers = args.workers if args.workers > 0 else (16 if use_dist else min(batch, 16))
    run_rows = (not use_dist) or args.dist_mode in ("both", "rows")
    run_replicas = use_dist and (world > 1 or force_env == "both") and args.dist_mode in ("both", "queries")
    # same database, keys and queries on every rank (fixed seeds): query i of the batch is the same everywhere
    raw, keys, queries = synthetic_inputs(pp, n_queries=batch)
    query = queries[0]
    if use_dist:
        D.check_sum_fits(max(enc.coeff_modulus[:-1]), world)

    def make_server(shard):
        db = pir_amd.PIRDatabase.Create(pp, device=local_rank, shard=shard)
        t0 = time.perf_counter()
        db.populate(raw)
        if args.dims > 1 and not args.keep_staging:
            db.finalize(release_staging=True)      # one copy of the database: the scan's operand layout
        t_pop = time.perf_counter() - t0
        srv = pir_amd.PIRServer(db, pp) if shard else pir_amd.PIRServer.Create(db, pp)
        srv.set_galois_keys(keys)
        return db, srv, t_pop

    pipes = []                              # the pipelined rows step (multi-GPU): drained before every barrier

    def barrier_for(srv):
        if not torch_first:
            sync_servers.append(srv)

        def barrier():
            for pp_ in pipes:
                pp_.flush()                 # multiply + reduce of the last submitted step, then wait
            srv.sync()                      # the library's own streams (not torch's current stream)
            device_synchronize()            # torch.cuda.synchronize(), or hipDeviceSynchronize() of the runtime in use
            if use_dist:
                dist.barrier()
                device_synchronize()
        return barrier

    out_extra = {}
    comm = D.Comm(dist, world) if use_dist else None

    # =========================== rows mode (single GPU: the plain path) -> headline ===========================
    shard = D.shard_range(pp.dimensions[0], rank, world) if world > 1 else None
    db, srv, t_populate = make_server(shard)
    reply_cts = db.reply_ct_count()
    barrier = barrier_for(srv)
    exchange = "none"
    bufs = sv_all = redb = red1 = None
    if use_dist:
        red1 = torch.empty((reply_cts, 2, k, N), dtype=torch.int64, device=dev)
        packed_ok = args.exchange != "u64" and D.packed_exchange_supported(srv, dist, world, comm, torch, dev)
        if args.exchange == "packed" and not packed_ok:
            raise SystemExit("--exchange packed needs d = 2 and an int8-MFMA-scanned shard (>= 8 rows) on every rank")
        exchange = "packed" if packed_ok else "u64"
        if args.exchange == "replicated":
            exchange = "replicated"
        # auto with several ranks and both forms available: measured below, not assumed (`autotune`)
        autotune = args.exchange == "auto" and world > 1 and packed_ok and os.environ.get("PIRGPU_EXCHANGE_AUTOTUNE", "1") != "0"
        if args.exchange == "auto" and world == 2 and not autotune:
            exchange = "replicated"

    # ---- (1) single-query latency + scan-kernel roofline (one scan launch per query)
    srv.stage_query(query)
    for _ in range(3):
        srv.run_staged()
        if use_dist:
            D.all_reduce_reply(srv, red1, dist, comm)
    srv.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.latency_runs):
        srv.run_staged()
        if use_dist:
            D.all_reduce_reply(srv, red1, dist, comm)
    barrier()
    latency_ms = (time.perf_counter() - t0) / args.latency_runs * 1e3
    timings = srv.last_timings()
    srv.set_profiling(False)
    single_reply = srv.fetch_reply() if world == 1 else None

    # The line is assembled by emit(): at the very end normally, or by the watchdog below if a reference leg stalls
    emit_lock, emitted = threading.Lock(), [False]

    def emit(note=None):
        if rank != 0:
            return
        with emit_lock:       # a second caller waits until the line is out, then returns
            if not emitted[0]:
                emitted[0] = True
                emit_line(note)

    skip_wire = os.environ.get("PIRGPU_BENCH_SKIP_WIRE", "") == "1"     # A/B sweeps of the device path only

    def emit_line(note):
        ms_per_step = elapsed / args.steps * 1e3
        u64_bytes = pp.num_pt * k * N * 8          # B_q = num_pt * k * N * 8
        scan_ms = timings["scan_ms"]
        achieved = scan_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        import glob
        traffic, traffic_src, traffic_stale = recorded_traffic(args.config, args.log_items, world, info["single_query_mfma"])
        compute = recorded_compute()   # VALU-issue roofline of the transform kernels (recorded; withheld when stale)
        if world == 1:
            parallelism = "single GPU" + (" (collective code path forced with one rank)" if use_dist else "")
        elif exchange == "slots":
            parallelism = ("database slot-sharded over %d GPUs (every rank holds 1/%d of the NTT slots of every plaintext, "
                           "dyadic base case of database.cpp:185-194); every rank expands %d of the %d queries of a step; "
                           "all-to-all of the packed column selectors' slot slices, scan of all rows on the rank's slots, "
                           "all-to-all of the row sums back to the query's owner, upper level there (RCCL; no reduce)"
                           % (world, world, per_rank, batch))
        else:
            parallelism = ("database row-sharded over %d GPUs; every rank expands %d of the %d queries of a step; "
                           % (world, batch if exchange == "replicated" else per_rank, batch)) + \
                          ("all-gather of packed column selectors + all-to-all of row selectors, reduce-scatter of replies (RCCL)"
                           if exchange == "packed" else
                           "no selector exchange (every rank expands all %d queries itself), reduce-scatter of replies (RCCL)" % batch
                           if exchange == "replicated" else "all-gather of u64 selection vectors, all-reduce of replies (RCCL)")
        out = {
            "metric": "PIR queries/sec (ms/query in ms_per_step), N=%d DB=2^%d x %dB d=%d"
                      % (N, args.log_items, item_bytes, args.dims),
            "value": qps, "unit": "queries/s", "n_gpus": world,
            # ranks of the RCCL process group the collectives of this run went through (null: no process group)
            "rccl_ranks": (dist.get_world_size() if dist.get_backend() == "nccl" else 0) if use_dist else None,
            **({"backend": "gloo (ranks share one GPU: test facility, not a multi-GPU measurement)"} if share_gpu else {}),
            "hip_runtime": hip_runtime_in_use(),
            "steps": args.steps, "warmup": args.warmup,
            # blocks of K timed steps behind ms_per_step (median block; one block when K steps last >= 1 s)
            "timed_blocks": len(headline_blocks), "timed_block_seconds": headline_blocks,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "N=%d, %d RNS data primes (%s bit | special %d bit), t=24 bit, DB=2^%d x %dB, "
                                   "d=%d, dims=%s, num_pt=%d, %d queries/step (whole job), %d in flight per GPU "
                                   "(BASELINE.json configs[%d])"
                                   % (N, k, ",".join(str(q.bit_length()) for q in enc.coeff_modulus[:-1]),
                                      enc.coeff_modulus[-1].bit_length(), args.log_items, item_bytes, args.dims,
                                      pp.dimensions, pp.num_pt, batch, workers, args.config - 1),
                       "queries_per_step": batch, "queries_per_step_per_gpu": per_rank, "workers": workers,
                       "dist_mode": "rows" if world > 1 or use_dist else "single",
                       "exchange": exchange, "parallelism": parallelism,
                       "database_copies_in_hbm": "operand layout only (u64 staging released)"
                       if args.dims > 1 and not args.keep_staging else "u64 staging" + (" + operand layout" if info["mfma"] else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_definition": "bytes the kernel must read (packed operand layout) / HIP-event duration / 8 TB/s",
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         # true: scan_mfma.hip has changed since the PMC passes were taken -- the figure describes an
                         # older kernel; null: the PMC file carries no source stamp
                         "traffic_stale": traffic_stale,
                         "kernel": ("scan_mfma_kernel<%d digits, %d k-steps> (int8 MFMA digit products, 1 query; "
                                    "the same pass serves up to 8)" % (info["digits"], info["ksteps"])) if info["single_query_mfma"]
                         else ("scan_mq_kernel<4 rows/wave, 1 query>" if args.dims > 1
                               else "scan_kernel + reduce_splits (column split)"),
                         "kernel_ms": scan_ms, "algorithmic_bytes": scan_bytes,
                         # stored bytes per database residue: L signed base-256 digits, the top one a nibble when the
                         # moduli allow (36-bit residues: 4.5 instead of 5 -- round 3: 10 % fewer bytes per pass, the
                         # single-query pass 5-10 % shorter on the same box)
                         **({"bytes_per_residue": info["digits"] - (0.5 if info.get("top_digit_nibble") else 0.0)}
                            if info["single_query_mfma"] else {}),
                         "algorithmic_bytes_definition": "packed operand-layout bytes of this GPU's shard = what one "
                                                         "launch must read (DESIGN.md section 5)",
                         "launches_averaged": timings["runs"],
                         # the launch that serves the headline step: `workgroups` persistent workgroups beside the other
                         # lane's transform kernels, 8 queries per pass -- bound per CU (loads in flight), not by HBM;
                         # bytes = the same stored database + the group's packed selectors + its row sums
                         **({"batch_launch": batch_launch_block(batch_scan, scan_bytes, info, k, N)} if batch_scan else {}),
                         # d = 1: the selection vector (one ciphertext per plaintext = 2x the database in u64) is read
                         # by the same launch; SURVEY 8(d)'s B_q leaves it out, so `frac` is a lower bound there
                         **({"d1_selector_bytes": pp.num_pt * 2 * k * N * 8,
                             "frac_incl_selectors": (scan_bytes + pp.num_pt * 2 * k * N * 8) / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                             if scan_ms > 0 else 0.0} if args.dims == 1 else {}),
                         # SURVEY 8(d) prices the scan in u64 residues (num_pt*k*N*8 per query) whatever the stored
                         # layout; both figures side by side, named
                         "frac_survey_8d_u64": (u64_bytes / max(world, 1)) / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if scan_ms > 0 else 0.0,
                         "survey_8d_u64_equivalent": {
                             "B_q": u64_bytes,
                             "single_query_launch_GBps": (u64_bytes / max(world, 1)) / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0,
                             "batched_B_q_times_qps_per_gpu_GBps": u64_bytes * qps / max(world, 1) / 1e9,
                             "batched_frac_of_one_read_per_query_bound": u64_bytes * qps / max(world, 1) / 1e9 / HBM_PEAK_GBS,
                             "note": "u64-equivalent rates (SURVEY 8(d) formula); they can exceed the packed-bytes "
                                     "figure because the operand layout stores L <= 7 bytes per residue and one "
                                     "pass serves up to 8 queries"}},
            "roofline_compute": compute,
            "latency_ms_single_query": round(latency_ms, 4),
            "single_query_qps": round(1e3 / latency_ms, 1),
            "phases_ms_single_query": {kk: round(v, 4) for kk, v in timings.items() if kk.endswith("_ms")},
            "db_populate_s": round(t_populate, 2),
        }
        out.update(out_extra)
        if forced_check is not None:
            out["forced_dist_replies_equal_plain"] = forced_check
        if use_dist and exchange == "slots" and slots_details:
            out["exchange_bytes_received_per_query_per_gpu"] = slots_details["exchange_bytes_received_per_query_per_gpu"]
            out["slots_step"] = slots_details
        if use_dist and bufs is not None and exchange == "packed":
            out["exchange_bytes_received_per_query_per_gpu"] = bufs.exchange_bytes_per_query(world)
            out["rows_step"] = {"pipelined": pipe is not None,
                                "phases_ms_serial": serial_phases,
                                "serial_sum_ms": round(sum(serial_phases.values()), 4) if serial_phases else None,
                                "note": "phases_ms_serial: expand / exchange (all-gather + all-to-all) / multiply / "
                                        "reduce (reduce-scatter + mod q) of one step with a host wait after every phase; "
                                        "the timed steps run them pipelined (exchange of step s under the multiply of "
                                        "step s-1 and the expansion of step s+1, no host waits): ms_per_step"}
        if world == 1 and not use_dist and args.config == 3 and not skip_wire:
            # wire-level ProcessRequest (what benchmark.cpp:71-79 times): serialized pir.Request in host
            # memory -> serialized pir.Response, incl. parsing, H2D of keys + query, D2H, serialisation.  The requests
            # come from the product client library (pir_amd.PIRClient, CPU): real keys, a real query ciphertext.
            try:
                cl = pir_amd.PIRClient.Create(pp, seed=b"bench-wire-client-0")
                q_ct = cl.create_query_for(123457 % pp.num_items)
                cl.set_seeded_keys(False)                 # expanded key objects (4.9 MB), as server_test.cpp builds them
                req = cl.SaveRequest([q_ct])
                srv.set_concurrency(1)
                wt = []
                for _ in range(25):
                    t0 = time.perf_counter()
                    resp = srv.ProcessRequest(req)
                    wt.append((time.perf_counter() - t0) * 1e3)
                once = c_abi_request_timer(srv, req)      # the same request timed at the C ABI itself
                wc = [once()[0] for _ in range(25)]
                # the same ciphertext through the residue-level entry point with the same client's keys
                slot = srv.install_keyset(b"bench-wire-client-0-residues", cl.galois_keys())
                srv.use_keyset(slot)
                same = bool(np.array_equal(cl.LoadResponse(resp)[0], srv.process_query(q_ct)))
                srv.use_keyset(0)
                srv.release_keyset(slot)
                # the reference client's DEFAULT request: seed-compressed Serializable<GaloisKeys> (client.cpp:47-54),
                # every key's uniform half re-sampled on the server (BLAKE2Xb + rejection sampling, on the pool's threads)
                cl.set_seeded_keys(True)
                req_s = cl.SaveRequest([q_ct])
                ws = []
                for _ in range(13):
                    t0 = time.perf_counter()
                    resp_s = srv.ProcessRequest(req_s)
                    ws.append((time.perf_counter() - t0) * 1e3)
                same_s = resp_s == resp                   # same keys, same query: the same bytes come back
                # NEW clients' first requests on the warm context (keys parsed / re-sampled, validated, uploaded)
                new_exp, new_seed = [], []
                for i in range(3):
                    c2 = pir_amd.PIRClient.Create(pp, seed=b"bench-wire-new-%d" % i)
                    q2 = c2.create_query_for((1000003 * (i + 1)) % pp.num_items)
                    for seeded, acc in ((False, new_exp), (True, new_seed)):
                        c3 = pir_amd.PIRClient.Create(pp, seed=b"bench-wire-new-%d-%d" % (i, seeded))
                        c3.set_seeded_keys(seeded)
                        r3 = c3.SaveRequest([q2])
                        t0 = time.perf_counter()
                        srv.ProcessRequest(r3)
                        acc.append((time.perf_counter() - t0) * 1e3)
                out["wire_process_request_ms"] = {"first_request_with_key_upload": round(wt[0], 3),
                                                  "new_client_first_request_on_warm_context": round(float(np.median(new_exp)), 3),
                                                  "new_client_seeded_keys_ms": round(float(np.median(new_seed)), 3),
                                                  "repeat": round(float(np.median(wt[1:])), 3),
                                                  "repeat_at_c_abi": round(float(np.median(wc[1:])), 3),
                                                  "repeat_at_c_abi_min": round(float(np.min(wc[1:])), 3),
                                                  "repeat_client_keys_cached_median_of_24": round(float(np.median(wt[1:])), 3),
                                                  "repeat_min": round(float(np.min(wt[1:])), 3),
                                                  "repeat_seeded_keys_median_of_12": round(float(np.median(ws[1:])), 3),
                                                  "request_bytes": len(req), "request_bytes_seeded_keys": len(req_s),
                                                  "response_bytes": len(resp),
                                                  "response_equals_residue_path": same,
                                                  "seeded_response_equals_expanded_response": same_s,
                                                  "timing": "`repeat*` through the Python mirror PIRServer.ProcessRequest (adds one "
                                                            "copy of the response into a bytes object); `repeat_at_c_abi*` "
                                                            "around pirgpu_process_request + pirgpu_free alone",
                                                  "requests_from": "pir_amd.PIRClient (product client library): expanded "
                                                                   "key objects for the first block, seed-compressed ones "
                                                                   "(the reference client's default) for the seeded figures"}
            except Exception as e:   # measurement extra only
                out["wire_process_request_ms"] = {"error": repr(e)}
            # the reference's OTHER three benchmarks at this size (benchmark.cpp:56-69, 81-95): SetupDb on the device,
            # ClientCreateRequest / ClientProcessResponse on the host with the product client library
            try:
                idx = 123457 % pp.num_items
                t_req, t_rsp = [], []
                for _ in range(5):
                    t0 = time.perf_counter()
                    r1 = cl.CreateRequest([idx])
                    t_req.append((time.perf_counter() - t0) * 1e3)
                rsp1 = srv.ProcessRequest(r1)
                for _ in range(5):
                    t0 = time.perf_counter()
                    items = cl.ProcessResponse([idx], rsp1)
                    t_rsp.append((time.perf_counter() - t0) * 1e3)
                t_db = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    db2 = pir_amd.PIRDatabase.Create(pp, device=local_rank)
                    db2.populate(raw)
                    db2.finalize(release_staging=True)
                    db2.lib.pirgpu_sync(db2.handle)
                    t_db.append((time.perf_counter() - t0) * 1e3)
                    db2.close()
                out["reference_benchmarks_ms"] = {
                    "SetupDb": round(float(np.median(t_db)), 2),
                    "ClientCreateRequest": round(float(np.median(t_req)), 3),
                    "ServerProcessRequest": out["wire_process_request_ms"].get("repeat_seeded_keys_median_of_12"),
                    "ClientProcessResponse": round(float(np.median(t_rsp)), 3),
                    "item_recovered_not_guaranteed_at_this_size": bool(items[0] == raw[idx].tobytes()),
                    "note": "benchmark.cpp's four cases at THIS configuration (the reference registers them for 2^8..2^16 "
                            "items, QUERIES_PER_REQUEST = 1 -- `reference_sweep` below has those sizes): SetupDb = "
                            "PIRDatabase::Create + populate from host bytes (H2D, encode, NTT, operand layout) on the GPU; "
                            "the two Client cases run on the host in libpirclient.so (single thread); ServerProcessRequest "
                            "= the seeded-keys request above.  At 2^20 items the reference's parameters leave no "
                            "worst-case noise budget (DESIGN.md section 2): the reply is bit-exact against the CPU path, "
                            "but that the item decrypts is luck of the noise, not a property -- it is one at the sizes "
                            "of `reference_sweep`"}
            except Exception as e:   # measurement extra only
                out["reference_benchmarks_ms"] = {"error": repr(e)}
            # the reference's benchmark at the reference's OWN sizes (2^8 .. 2^16 items), CPU oracle beside it
            if os.environ.get("PIRGPU_BENCH_SKIP_SWEEP", "") != "1":
                try:
                    t0 = time.perf_counter()
                    out["reference_sweep"] = reference_sweep(pir_amd, device=local_rank, reps=12,
                                                             with_cpu=not args.no_cpu_baseline)
                    out["reference_sweep"]["seconds"] = round(time.perf_counter() - t0, 2)
                except Exception as e:   # measurement extra only
                    out["reference_sweep"] = {"error": repr(e)}
        if world == 1 and not use_dist and args.config == 3 and not skip_wire:
            # several CLIENTS at once (the reference's keys are per request, server.cpp:46-48): `batch` clients with
            # different Galois keys, one query each, through the same batch pipeline -- every group of 8 holds 8
            # different clients' queries, each switched with its own client's resident key set
            try:
                mods = enc.coeff_modulus
                n_cl = batch
                srv.set_keyset_capacity(max(64, n_cl))
                slots = []
                for cidx in range(n_cl):
                    ck = {}
                    for g, key in keys.items():
                        kk = key.copy()
                        for i in range(k + 1):
                            kk[:, :, i, :] = (kk[:, :, i, :] + np.uint64(cidx + 1)) % np.uint64(mods[i])
                        ck[g] = kk
                    slots.append(srv.install_keyset(b"bench-client-%d" % cidx, ck))
                srv.set_concurrency(workers)
                srv.stage_batch(queries)
                srv.set_batch_keysets(slots)
                mc_steps = max(5, min(args.steps, 50))
                for _ in range(2):
                    srv.run_batch()
                srv.sync()
                t0 = time.perf_counter()
                for _ in range(mc_steps):
                    srv.run_batch()
                srv.sync()
                dt = time.perf_counter() - t0
                mc_replies = srv.fetch_batch()
                # same query + same keys as the single-client batch only for a client whose keys were not shifted:
                # check instead that query 0 under client 0's keys equals the single-query path with that key set
                srv.use_keyset(slots[0])
                same = bool(np.array_equal(srv.process_query(queries[0]), mc_replies[0]))
                srv.use_keyset(0)
                out["multi_client_qps"] = {"value": mc_steps * n_cl / dt, "unit": "queries/s", "clients": n_cl,
                                           "queries_per_client": 1, "steps": mc_steps,
                                           "ms_per_step": dt / mc_steps * 1e3,
                                           "reply0_equals_single_query_with_that_clients_keys": same,
                                           "keysets": srv.keyset_stats(),
                                           "note": "device-resident: %d clients' key sets in HBM, every query switched "
                                                   "with its own client's keys inside mixed groups of 8" % n_cl}
            except Exception as e:   # measurement extra only
                out["multi_client_qps"] = {"error": repr(e)}
            # the same through the WIRE: n_w clients' serialized pir.Request protos (the product client's CreateRequest:
            # seed-compressed keys + one query each) served by pirgpu_process_requests -- parsing, key fingerprint + byte
            # compare against the resident sets, H2D of the queries, the batch pipeline, D2H and serialisation of the
            # responses, all inside the timed region.  `value` = the server under sustained load: `callers` threads, each
            # sending its n_w clients' requests in a loop (two request windows in flight on the context, so parsing /
            # staging of one window and the download / serialisation of another run beside the GPU's work on a third);
            # `single_caller` = one thread, one call after the other (every call fills and drains the pipeline alone).
            try:
                srv.unstage_batch()       # (the device-resident leg's staged batch is not run again)
                n_w = int(os.environ.get("PIRGPU_BENCH_WIRE_CLIENTS", "64"))
                callers = max(1, int(os.environ.get("PIRGPU_BENCH_WIRE_CALLERS", "2")))
                srv.set_keyset_capacity(int(os.environ.get("PIRGPU_BENCH_WIRE_CAPACITY", str(max(64, 2 * n_w)))))
                wcl = [pir_amd.PIRClient.Create(pp, seed=b"bench-wire-mc-%d" % i) for i in range(n_w)]
                reqs = [c.CreateRequest([(7919 * i + 13) % pp.num_items]) for i, c in enumerate(wcl)]
                first = srv.ProcessRequests(reqs)                 # installs the key sets
                # timed through the C ABI itself (ctypes call, responses freed unread): the Python mirror's copies of
                # a megabyte per response are the binding's cost, not the library's
                import ctypes as C
                lib, handle = srv.lib, srv.db.handle
                req_views = [np.frombuffer(r, dtype=np.uint8) for r in reqs]
                ptrs = (C.c_void_p * n_w)(*[b.ctypes.data for b in req_views])
                lens = (C.c_size_t * n_w)(*[len(r) for r in reqs])

                def call_state():
                    return (C.c_void_p * n_w)(), (C.c_size_t * n_w)(), (C.c_int * n_w)()

                def one_call(state, keep=False):
                    resp, rlen, status = state
                    lib.pirgpu_process_requests(handle, n_w, ptrs, lens, resp, rlen, status)
                    good = all(status[i] == 0 for i in range(n_w))
                    kept = C.string_at(resp[0], rlen[0]) if keep and good else None
                    for i in range(n_w):
                        if status[i] == 0:
                            lib.pirgpu_free(resp[i])
                    return good, kept

                st0 = call_state()
                for _ in range(2):
                    one_call(st0)
                w_steps = 20
                ok = True
                t0 = time.perf_counter()
                for _ in range(w_steps):
                    ok = one_call(st0)[0] and ok
                dt1 = time.perf_counter() - t0
                # ONE calling thread with two calls in flight (pirgpu_process_requests_begin / _end): call i + 1 is handed
                # over before call i is waited for, so its parsing / staging / queueing run under call i's tail
                def begin(state):
                    resp, rlen, status = state
                    call = C.c_void_p()
                    rc = lib.pirgpu_process_requests_begin(handle, n_w, ptrs, lens, resp, rlen, status, C.byref(call))
                    if rc:
                        raise RuntimeError("pirgpu_process_requests_begin: %d" % rc)
                    return call

                def end(call, state):
                    resp, rlen, status = state
                    lib.pirgpu_process_requests_end(call)
                    good = all(status[i] == 0 for i in range(n_w))
                    for i in range(n_w):
                        if status[i] == 0:
                            lib.pirgpu_free(resp[i])
                    return good

                sts = [call_state(), call_state()]
                ok2 = end(begin(sts[0]), sts[0])
                t0 = time.perf_counter()
                pending = begin(sts[0])
                for i in range(1, w_steps):
                    nxt = begin(sts[i & 1])
                    ok2 = end(pending, sts[(i - 1) & 1]) and ok2
                    pending = nxt
                ok2 = end(pending, sts[(w_steps - 1) & 1]) and ok2
                dt2 = time.perf_counter() - t0
                # sustained load: `callers` threads (ctypes releases the interpreter lock for the duration of a call)
                oks = [True] * callers
                gate = threading.Barrier(callers + 1)

                def caller(ci):
                    st = call_state()
                    one_call(st)
                    gate.wait()
                    for _ in range(w_steps):
                        oks[ci] = one_call(st)[0] and oks[ci]
                    gate.wait()

                ths = [threading.Thread(target=caller, args=(ci,)) for ci in range(callers)]
                for th in ths:
                    th.start()
                gate.wait()
                t0 = time.perf_counter()
                gate.wait()
                dtn = time.perf_counter() - t0
                for th in ths:
                    th.join()
                good, kept = one_call(st0, keep=True)
                ok = ok and good and all(oks) and all(st == 0 for st, _ in first)
                same = ok and kept == first[0][1]
                out["wire_multi_client_qps"] = {"value": callers * w_steps * n_w / dtn, "unit": "queries/s", "clients": n_w,
                                                "callers": callers, "calls_per_caller": w_steps,
                                                "ms_per_call": dtn / w_steps * 1e3,
                                                "single_caller": {"value": w_steps * n_w / dt1, "unit": "queries/s",
                                                                  "ms_per_call": dt1 / w_steps * 1e3},
                                                "single_caller_two_calls_in_flight": {
                                                    "value": w_steps * n_w / dt2, "unit": "queries/s",
                                                    "ms_per_call": dt2 / w_steps * 1e3, "all_ok": ok2,
                                                    "note": "one calling thread, pirgpu_process_requests_begin(i + 1) before "
                                                            "pirgpu_process_requests_end(i)"},
                                                "all_ok": ok, "repeatable": same, "request_bytes_each": len(reqs[0]),
                                                "keys": "seed-compressed (product client default)",
                                                "keysets": srv.keyset_stats(),
                                                "note": "pirgpu_process_requests on %d clients' serialized requests per "
                                                        "call (keys resident after the first call): wire parsing, key "
                                                        "lookup + byte compare, PCIe both ways and response "
                                                        "serialisation inside the timed region, timed at the C ABI; value = "
                                                        "%d calling threads in a loop (sustained load, two request windows "
                                                        "in flight), single_caller = one thread, call after call (every call "
                                                        "fills and drains the pipeline alone), single_caller_two_calls_in_flight "
                                                        "= one thread using the begin / end form" % (n_w, callers)}
            except Exception as e:   # measurement extra only
                out["wire_multi_client_qps"] = {"error": repr(e)}
        if world == 1 and not use_dist and not args.no_cpu_baseline:
            out["batch_reply0_equals_single_query_reply"] = bool(np.array_equal(batch_replies[0], single_reply))
            out["cpu_baseline"] = cpu_baseline(pp, raw, keys, query, single_reply)
            out["speedup_vs_cpu_baseline"] = qps / out["cpu_baseline"]["value"]
        if note:
            out["extras_aborted"] = note
        print(json.dumps(out), file=result_out)
        result_out.flush()

    # Everything that runs AFTER a complete headline measurement exists (the second candidate form of the rows step, the
    # replicas and hybrid reference legs) must never cost it: an exception is caught per leg, and a part that STALLS (a
    # collective that never completes on some fabric) is cut off by a watchdog armed as soon as the first complete
    # measurement is in hand -- rank 0 prints the line with what it has, every rank leaves.
    watchdog = []
    aborting = threading.Event()

    deadline = [None]

    cut_off_note = ("what runs after the first complete headline measurement (second candidate form / reference legs) "
                    "exceeded its time budget and was cut off; the headline above is complete; every rank exits with "
                    "code 3 (a collective that never completed is a failed job, whatever was measured before it)")
    STALLED_EXIT = 3        # the watchdog's exit code: rank 0 prints its line first, then every rank leaves non-zero

    def past_deadline():
        return aborting.is_set() or (deadline[0] is not None and time.monotonic() > deadline[0])

    def park_if_aborting():
        # past the watchdog's deadline the job is being ended: whatever fails after that is its doing (a rank whose
        # main thread sat in a wait that holds the interpreter lock may get here before its own timer could run)
        if past_deadline():
            emit(cut_off_note)
            os._exit(STALLED_EXIT)

    stages_after_headline = [1]     # how many stages share PIRGPU_EXTRAS_TIMEOUT_S (set once the candidates are known)

    def arm_watchdog(stage="extras"):
        """(Re-)arms the watchdog for the next stage after a complete headline measurement: every later candidate form
        and the reference legs get their OWN slice of PIRGPU_EXTRAS_TIMEOUT_S (the budget divided by the number of such
        stages), counted from the moment the stage starts -- a form that stalls is cut off at the end of its slice, it
        cannot eat the time of the stages before it, and the line then names the stage that stalled."""
        if not (use_dist and world > 1):
            return
        for w_ in watchdog:
            w_.cancel()
        del watchdog[:]
        budget = float(os.environ.get("PIRGPU_EXTRAS_TIMEOUT_S", "300")) / max(1, stages_after_headline[0])

        def _abort():
            aborting.set()
            if rank == 0:
                emit(cut_off_note + " [stage cut off: %s, after %.0f s]" % (stage, budget))
                time.sleep(6.0)   # the other ranks leave on their own timers (5 s later) while this one still answers
            os._exit(STALLED_EXIT)

        deadline[0] = time.monotonic() + budget
        watchdog.append(threading.Timer(budget + (0.0 if rank == 0 else 5.0), _abort))   # rank 0 prints first
        watchdog[0].daemon = True
        watchdog[0].start()

    # tests only: PIRGPU_TEST_STALL=<form> puts a long sleep into every measured step of that candidate form -- what a
    # collective that never completes on some fabric looks like to the job (tests/test_gpu_distributed.py)
    stall_form = os.environ.get("PIRGPU_TEST_STALL", "")
    stall_s = float(os.environ.get("PIRGPU_TEST_STALL_S", "3600"))

    def maybe_stall(form):
        if stall_form and stall_form == form:
            time.sleep(stall_s)


    # ---- (2) throughput: `batch` queries per step (the whole job's), `workers` in flight per GPU
    srv.set_concurrency(workers)
    srv.stage_batch(queries)
    pipe = rpipe = None
    serial_phases = None
    forced_check = None
    batch_replies = None
    batch_scan = None        # (read by the line's assembly: must exist before the watchdog can print a line)
    scan_bytes = srv.scan_bytes()
    info = srv.scan_info()
    if use_dist:
        D.sync_zero_plaintexts(srv, dist, world, comm, torch, dev)   # the transparent-ciphertext decision is collective

    def step_rows():
        maybe_stall(active[0])
        if not use_dist:
            srv.run_batch()
        elif pipe is not None:
            pipe.submit()
        elif rpipe is not None:
            rpipe.submit()
        elif active[0] == "packed":
            D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm)
        else:
            D.run_batch_query_parallel(srv, sv_all, redb, dist, rank, world, comm)

    active = [exchange]     # the form being built / timed; `exchange` is the one the line reports

    def measure(form):
        """Builds what `form` of the step needs and times the contract's W + K steps with it."""
        nonlocal bufs, sv_all, redb, pipe, rpipe, serial_phases
        pipe = rpipe = None
        active[0] = form
        if use_dist and form == "packed":
            bufs = D.PackedBuffers(srv, batch, rank, world, torch, dev)
            # serial phase times from the synchronous form of the step (every phase followed by a host wait) ...
            for _ in range(2):
                D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm)
            acc = None
            for _ in range(5):
                ph = D.run_batch_rows_packed(srv, bufs, dist, rank, world, comm)
                acc = ph if acc is None else {kk: acc[kk] + v for kk, v in ph.items()}
            serial_phases = {kk: round(v / 5, 4) for kk, v in acc.items()}
            # ... and the pipelined form for the timed steps: exchange of step s under the multiply of step s - 1 and
            # the expansion of step s + 1, streams ordered by events, no host waits (PIRGPU_ROWS_PIPELINE=0: synchronous)
            if os.environ.get("PIRGPU_ROWS_PIPELINE", "1") != "0":
                pipe = D.RowsPipeline(srv, batch, rank, world, dist, torch, dev)
                pipes.append(pipe)
        elif use_dist and form == "replicated":
            rpipe = D.RowsReplicatedPipeline(srv, batch, rank, world, dist, torch, dev)
            pipes.append(rpipe)
        elif use_dist:
            sv_all = torch.empty((batch, pp.dim_sum, 2, k, N), dtype=torch.int64, device=dev)
            redb = torch.empty((batch, reply_cts, 2, k, N), dtype=torch.int64, device=dev)
        el = timed_steps(step_rows, barrier, args.steps, args.warmup, dist, use_dist, torch, dev)
        if rpipe is not None:
            rpipe.close()   # later plain batches write to the context's own reply buffer again
        pipes.clear()
        return el

    slots_details = None
    slots_replies = [None]
    slots_blocks = [None]

    def measure_slots():
        """The slot-sharded step: its own context (this rank's 1 / world of the NTT slots of every plaintext), the
        synchronous form for serial phase times, the pipelined form for the contract's W + K steps."""
        nonlocal slots_details
        cuts = D.slot_cuts(k * N, world)
        sdb = pir_amd.PIRDatabase.Create(pp, device=local_rank, slots=(cuts[rank], cuts[rank + 1]) if world > 1 else None)
        try:
            sdb.populate(raw)
            sdb.finalize(release_staging=True)
            ssrv = pir_amd.PIRServer(sdb, pp)
            ssrv.set_galois_keys(keys)
            ssrv.set_concurrency(workers)
            ssrv.stage_batch(queries)
            sbarrier = barrier_for(ssrv)
            sb = D.SlotsBuffers(ssrv, batch, rank, world, torch, dev)
            for _ in range(2):
                D.run_batch_slots(ssrv, sb, dist, rank, world, comm)
            acc = None
            for _ in range(5):
                ph = D.run_batch_slots(ssrv, sb, dist, rank, world, comm)
                acc = ph if acc is None else {kk: acc[kk] + v for kk, v in ph.items()}
            phases = {kk: round(v / 5, 4) for kk, v in acc.items()}
            per_q = sb.exchange_bytes_per_query(world)
            del sb
            # The timed steps: the pipelined form, with the row sums crossing the links as u64 or in 5 bytes per residue
            # (-37 % of the second all-to-all for two packing passes).  Which pays depends on the links: at 2 / 4 GPUs
            # (one / three xGMI links per GPU) the per-link arithmetic of DESIGN.md section 7.1 says the links bind, so
            # the 5-byte form is measured FIRST there and u64 is the second pass; at 8 GPUs the other way round.  The
            # second pass only runs inside the watchdog's slice; PIRGPU_SLOTS_PACK40 set: only that form.
            can40 = world > 1 and ssrv.pack40_supported()
            forced40 = os.environ.get("PIRGPU_SLOTS_PACK40")
            if forced40 is not None:
                forms = ["5 bytes per residue" if (forced40 == "1" and can40) else "u64"]
            elif can40:
                forms = ["5 bytes per residue", "u64"] if world <= 4 else ["u64", "5 bytes per residue"]
            else:
                forms = ["u64"]
            rowsums = {"order": forms}
            el = None

            def stalled_submit(sp):
                def f():
                    maybe_stall("slots")
                    sp.submit()
                return f

            for fi, form in enumerate(forms):
                if fi > 0 and past_deadline():
                    break
                saved = os.environ.get("PIRGPU_SLOTS_PACK40")
                os.environ["PIRGPU_SLOTS_PACK40"] = "1" if form != "u64" else "0"
                try:
                    spipe = D.SlotsPipeline(ssrv, batch, rank, world, dist, torch, dev)
                finally:
                    if saved is None:
                        del os.environ["PIRGPU_SLOTS_PACK40"]
                    else:
                        os.environ["PIRGPU_SLOTS_PACK40"] = saved
                pipes.append(spipe)
                el_f = timed_steps(stalled_submit(spipe), sbarrier, args.steps, args.warmup, dist, use_dist, torch, dev)
                pipes.clear()
                rowsums["u64_ms_per_step" if form == "u64" else "packed_5_byte_ms_per_step"] = round(el_f / args.steps * 1e3, 4)
                if el is None or el_f < el:
                    el = el_f
                    rowsums["form"] = form
                    per_q = spipe.sets[0].exchange_bytes_per_query(world)
                    slots_replies[0] = spipe.replies(spipe.step - 1).cpu().numpy().view(np.uint64).copy()
                    slots_blocks[0] = BLOCK_LOG[-1]
                del spipe
            slots_details = {"pipelined": True, "phases_ms_serial": phases, "serial_sum_ms": round(sum(phases.values()), 4),
                             "row_sums_on_the_links": rowsums,
                             "exchange_bytes_received_per_query_per_gpu": per_q,
                             "database_bytes_per_gpu": ssrv.scan_bytes(), "slots_per_gpu": cuts[rank + 1] - cuts[rank],
                             "note": "every rank holds 1/%d of the NTT slots of EVERY plaintext; per step it receives its "
                                     "slots of every query's packed column selectors, scans all rows, returns the row sums "
                                     "to the rank that expanded the query; that rank runs the upper level with the row "
                                     "selectors it kept -- no row-selector exchange, no reduce" % world}
            return el
        finally:
            pipes.clear()
            sdb.close()

    # (decided from the whole matrix, not from this rank's row shard: a slot shard holds ALL rows; a configuration the
    # step cannot take -- matrices wider than one column chunk, moduli of 55 bits and more -- fails in measure_slots,
    # which the autotune reports as an error of that candidate)
    slots_ok = use_dist and args.dims == 2 and pp.dimensions[0] >= 8 and (k * N) % (16 * world) == 0
    if use_dist and args.exchange == "slots":
        if not slots_ok:
            raise SystemExit("--exchange slots needs d = 2 and the int8-MFMA scan in one column chunk")
        exchange = "slots"
        elapsed = measure_slots()
        qps = args.steps * batch / elapsed
        headline_blocks = slots_blocks[0]
    elif use_dist and autotune:
        # Which form of the sharded step is fastest depends on what the links between THESE GPUs sustain (DESIGN.md
        # section 7), so it is measured, not assumed: every candidate runs the contract's W + K steps (same barrier +
        # max-over-ranks timing, every rank sees the same numbers) and the fastest one is the headline.  ORDER: the
        # slot-sharded step FIRST -- it is the candidate whose per-rank budget meets north_star (1.49 ms per rank-step
        # at 8 GPUs against 7.92 for replicated) --, then `replicated` (its only collective is a reduce-scatter), then
        # `packed` (all-gather + all-to-all + reduce-scatter: the form with the most to go wrong on a fabric never seen
        # before).  As soon as ONE candidate is in hand the watchdog is armed, re-armed per later candidate with that
        # candidate's own slice of the budget: a form that hangs or crawls costs itself, never the headline.
        import traceback as _tb
        order = (["slots"] if slots_ok and os.environ.get("PIRGPU_AUTOTUNE_SLOTS", "1") != "0" else []) + ["replicated", "packed"]
        stages_after_headline[0] = len(order)      # the later candidates + the reference legs
        tune = {"ms_per_step": {}, "order": order, "wall_s": {}, "steps_each": args.steps,
                "watchdog_slice_s": round(float(os.environ.get("PIRGPU_EXTRAS_TIMEOUT_S", "300")) / len(order), 1),
                "note": "every form of the sharded step (slot shards; rows + replicated expansion; rows + packed selector "
                        "exchange) timed over the full W + K steps on this machine's links, in `order`; the fastest one "
                        "is the headline; wall_s = seconds each candidate took including its set-up"}
        out_extra["exchange_autotune"] = tune     # in the line from now on, whatever happens to a later candidate
        elapsed = None
        packed_details = None
        for cand_i, form in enumerate(order):
            t_c = time.monotonic()
            if elapsed is not None:
                arm_watchdog("candidate " + form)
            try:
                el_c = measure_slots() if form == "slots" else measure(form)
                blocks_c = slots_blocks[0] if form == "slots" else BLOCK_LOG[-1]
            except Exception as e:     # noqa: BLE001 -- a candidate that fails must not cost the others
                park_if_aborting()
                _tb.print_exc()
                tune["ms_per_step"][form] = None
                tune[form + "_error"] = repr(e)
                tune["wall_s"][form] = round(time.monotonic() - t_c, 3)
                if form == "packed":
                    bufs = pipe = serial_phases = None
                if elapsed is None and cand_i + 1 == len(order):
                    raise
                continue
            tune["wall_s"][form] = round(time.monotonic() - t_c, 3)
            tune["ms_per_step"][form] = round(el_c / args.steps * 1e3, 4)
            if form == "packed":
                packed_details = {"exchange_bytes_received_per_query_per_gpu": bufs.exchange_bytes_per_query(world),
                                  "phases_ms_serial": serial_phases}
            if elapsed is None or el_c < elapsed:
                exchange = form
                elapsed, qps = el_c, args.steps * batch / el_c
                headline_blocks = blocks_c
            tune["chosen"] = exchange
        if exchange != "packed":
            if packed_details:
                tune["packed_details"] = packed_details
            bufs = pipe = serial_phases = None
        if exchange != "slots" and slots_details:
            tune["slots_details"] = slots_details
        tune["chosen"] = exchange
    else:
        elapsed = measure(exchange)
        qps = args.steps * batch / elapsed
        headline_blocks = BLOCK_LOG[-1]
    if use_dist and world == 1:   # forced single-rank run: the reduced replies must equal the plain ones
        got = slots_replies[0] if exchange == "slots" else \
              ((pipe.replies(pipe.step - 1) if pipe is not None else bufs.replies) if exchange == "packed"
               else (rpipe.replies(rpipe.step - 1) if rpipe is not None else redb)).cpu().numpy().view(np.uint64)
        srv.stage_batch(queries)
        srv.run_batch()
        forced_check = bool(np.array_equal(got, srv.fetch_batch()))
        print("forced-dist check: replies through the collective path equal plain replies: %s" % forced_check, file=sys.stderr)
    batch_replies = srv.fetch_batch() if world == 1 and not use_dist else None
    # the database pass as it runs INSIDE the step just timed (roofline.kernel_ms above is the single-query launch on the
    # whole chip): HIP events around the batch pipeline's scan launches over a few more steps, outside the timed region
    batch_scan = None
    if world == 1 and not use_dist and info["mfma"]:
        try:
            srv.set_profiling(True)
            for _ in range(6):
                srv.run_batch()
            batch_scan = srv.batch_scan_timings()
            srv.set_profiling(False)
        except Exception as e:     # noqa: BLE001 -- measurement extra only
            batch_scan = {"error": repr(e)}
    arm_watchdog("reference legs")

    # =========================== replicas (reference point, multi-GPU only) ===========================
    # the two reference legs below must never cost the headline: a failure (the same on every rank: they run the same
    # program on the same inputs) is reported in the extra's place
    import traceback

    def replicas_leg():
        nonlocal db, srv, barrier
        if shard is not None:
            db.close()
            db, srv, _ = make_server(None)
            barrier = barrier_for(srv)
        # every GPU holds the whole database and serves its own share of the same global batch
        srv.set_concurrency(min(max(per_rank, 1), 16))
        lo, hi = D.owned_queries(batch, rank, world)
        srv.stage_batch(queries[lo:hi])
        el2 = timed_steps(srv.run_batch, barrier, args.steps, args.warmup, dist, use_dist, torch, dev)
        out_extra["replicas_reference"] = {
            "value": args.steps * batch / el2, "unit": "queries/s", "ms_per_step": el2 / args.steps * 1e3,
            "scaling": "strong", "queries_per_step": batch, "queries_per_step_per_gpu": per_rank,
            "note": "every GPU holds the whole database and serves batch/gpus queries of the same global batch; no "
                    "data-path collective (barrier + max-over-ranks timing only). Reference point, not the headline."}

    if run_replicas:
        try:
            replicas_leg()
        except Exception as e:     # noqa: BLE001
            park_if_aborting()
            traceback.print_exc()
            out_extra["replicas_reference"] = {"error": repr(e)}

    # ============ hybrid (reference point, multi-GPU only): R replica groups x S row shards ============
    # "report a 2 x 4 hybrid -- 4-way rows inside 2 replica groups -- as a named extra, not the
    # headline".  Every group of S ranks holds the whole database row-sharded S ways and serves batch / R of the
    # step's queries with the same pipelined step, its collectives confined to the group's own process group.
    hyb_R = int(os.environ.get("PIRGPU_HYBRID_GROUPS", "2"))
    run_hybrid = (use_dist and world > 1 and args.dist_mode == "both" and args.dims == 2 and hyb_R >= 1
                  and world % hyb_R == 0 and batch % world == 0
                  and (world // hyb_R >= 2 or os.environ.get("PIRGPU_HYBRID_FORCE") == "1"))
    def hybrid_leg():
        nonlocal db, srv, barrier
        gi, gr, S, groups = D.hybrid_layout(rank, world, hyb_R)
        pgs = [dist.new_group(g) for g in groups]          # every rank creates every group, in the same order
        db.close()
        db, srv, _ = make_server(D.shard_range(pp.dimensions[0], gr, S) if S > 1 else None)
        barrier = barrier_for(srv)
        hcomm = D.Comm(dist, S, host_sync=True, group=pgs[gi])
        D.sync_zero_plaintexts(srv, dist, S, hcomm, torch, dev)
        mine_ok = D.packed_exchange_supported(srv, dist, S, hcomm, torch, dev)
        t_ok = torch.tensor([1 if mine_ok else 0], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t_ok)                                # every group must be able to take the packed path
        if int(t_ok.item()) == world:
            bpg = batch // hyb_R
            srv.set_concurrency(workers)
            srv.stage_batch(queries)
            hpipe = D.RowsPipeline(srv, bpg, gr, S, dist, torch, dev, comm=D.Comm(dist, S, host_sync=False, group=pgs[gi]))
            pipes.append(hpipe)
            el3 = timed_steps(lambda: hpipe.submit(first=gi * bpg), barrier, args.steps, args.warmup, dist, use_dist, torch, dev)
            pipes.clear()
            out_extra["hybrid_rows_reference"] = {
                "value": args.steps * batch / el3, "unit": "queries/s", "ms_per_step": el3 / args.steps * 1e3,
                "replica_groups": hyb_R, "row_shards_per_group": S, "scaling": "strong", "queries_per_step": batch,
                "queries_per_step_per_group": bpg,
                "exchange_bytes_received_per_query_per_gpu": hpipe.sets[0].exchange_bytes_per_query(S) * bpg / batch,
                "note": "%d replica groups x %d row shards: every group holds the whole database sharded %d ways and "
                        "serves %d of the %d queries of a step with the pipelined rows step inside its own process "
                        "group. Reference point, not the headline." % (hyb_R, S, S, bpg, batch)}

    if run_hybrid:
        try:
            hybrid_leg()
        except Exception as e:     # noqa: BLE001
            park_if_aborting()
            traceback.print_exc()
            pipes.clear()
            out_extra["hybrid_rows_reference"] = {"error": repr(e)}

    late = past_deadline()
    for w_ in watchdog:
        w_.cancel()
    emit()
    if late:
        os._exit(STALLED_EXIT)   # past the watchdog's deadline ranks may already have left: no final barrier
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
