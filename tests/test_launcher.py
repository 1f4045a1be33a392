"""`python bench.py --gpus N` without a launcher environment starts its own N ranks (pir_amd/launcher.py): spawn and
relay logic on CPU -- two gloo ranks running the row-sharded step with the oracle-backed server, a failing rank, and
the fail-fast when more GPUs are requested than are visible."""
import json
import os
import subprocess
import sys
import time

from pir_amd import launcher

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CHILD = os.path.join(HERE, "launcher_child.py")


def test_spawn_two_ranks_relays_rank0_json(tmp_path):
    out = tmp_path / "out.txt"
    with open(out, "w") as fo:
        rc = launcher.spawn_ranks([sys.executable, CHILD, "rows"], 2, check_devices=False, stdout=fo, timeout_s=600)
    assert rc == 0
    lines = [l for l in open(out).read().splitlines() if l.strip()]
    assert len(lines) == 1                                  # exactly rank 0's line, nothing from rank 1
    j = json.loads(lines[0])
    assert j == {"n_gpus": 2, "ranks": 2, "backend": "gloo", "all_ranks_ok": True}


def test_spawn_eight_ranks(tmp_path):
    """The launch shape of `bench.py --gpus 8`: eight rank processes, rank 0's single line relayed."""
    out = tmp_path / "out.txt"
    with open(out, "w") as fo:
        rc = launcher.spawn_ranks([sys.executable, CHILD, "rows8"], 8, check_devices=False, stdout=fo, timeout_s=900)
    assert rc == 0
    lines = [l for l in open(out).read().splitlines() if l.strip()]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"n_gpus": 8, "ranks": 8, "backend": "gloo", "all_ranks_ok": True}


def test_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    out = tmp_path / "out.txt"
    t0 = time.monotonic()
    with open(out, "w") as fo, open(tmp_path / "err.txt", "w") as fe:
        rc = launcher.spawn_ranks([sys.executable, CHILD, "fail"], 2, check_devices=False, stdout=fo, stderr=fe)
    assert rc == 3
    assert time.monotonic() - t0 < 30                       # rank 0 (sleeping 60 s) was terminated, not waited for
    assert open(out).read().strip() == ""
    assert "rank 1 exited with code 3" in open(tmp_path / "err.txt").read()


def test_bench_refuses_more_gpus_than_visible():
    """No GPU in the CPU test environment: --gpus 2 must fail fast with a clear message instead of running one rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                          # also on a GPU box: nothing visible
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2
    assert p.stdout.strip() == ""
    assert "--gpus 2 requested but only 0 GPU(s) are visible" in p.stderr


def test_launcher_environment_detection():
    assert launcher.launched_by_a_launcher({"RANK": "0", "WORLD_SIZE": "2"})
    assert not launcher.launched_by_a_launcher({"WORLD_SIZE": "2"})
    assert not launcher.launched_by_a_launcher({})


def test_bench_output_stage_does_not_shadow_the_measurement_state():
    """bench.py assembles its JSON line in a nested function that reads the measurement's variables of main() through
    its closure.  An assignment to one of those names inside that function makes the name LOCAL to it and the line fails
    with UnboundLocalError -- only in the runs that reach the affected branch (round 3: `bufs`, read only by multi-rank
    runs with the packed exchange, i.e. never on the builder's one-GPU boxes).  Static check: the function assigns none
    of the names main() assigns."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")

    def targets(node, skip_nested):
        names = set()

        class V(ast.NodeVisitor):
            def visit_FunctionDef(self, n):
                if not skip_nested:
                    self.generic_visit(n)

            def _add(self, t):
                for x in ast.walk(t):
                    if isinstance(x, ast.Name):
                        names.add(x.id)

            def visit_Assign(self, n):
                for t in n.targets:
                    self._add(t)
                self.generic_visit(n)

            def visit_AugAssign(self, n):
                self._add(n.target)
                self.generic_visit(n)

            def visit_For(self, n):
                self._add(n.target)
                self.generic_visit(n)

            def visit_With(self, n):
                for it in n.items:
                    if it.optional_vars is not None:
                        self._add(it.optional_vars)
                self.generic_visit(n)

        v = V()
        for child in node.body:
            v.visit(child)
        return names

    emit_line = next(n for n in ast.walk(main) if isinstance(n, ast.FunctionDef) and n.name == "emit_line")
    shadowed = targets(emit_line, skip_nested=False) & targets(main, skip_nested=True)
    assert shadowed <= {"t0", "_"}, shadowed      # t0: a timer both use locally; _: loop dummies -- never read across
    # ... and every name of main() that the line reads exists BEFORE the watchdog can be armed: the watchdog prints the
    # line from a timer thread in the middle of a later candidate (round 6: `batch_scan` was first assigned after the
    # candidates -- a stalled candidate would have died with NameError instead of printing the headline it had)
    first = {}
    for node in main.body:
        for x in ast.walk(node):
            if isinstance(x, ast.FunctionDef) and x is not node:
                continue
            if isinstance(x, (ast.Assign, ast.AugAssign, ast.For)):
                tgts = x.targets if isinstance(x, ast.Assign) else [x.target]
                for t in tgts:
                    for nm in ast.walk(t):
                        if isinstance(nm, ast.Name):
                            first.setdefault(nm.id, x.lineno)
    arm = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "arm_watchdog")
    reads = {x.id for x in ast.walk(emit_line) if isinstance(x, ast.Name) and isinstance(x.ctx, ast.Load)}
    own = targets(emit_line, skip_nested=False)
    late = sorted(n for n in reads if n in first and first[n] > arm and n not in own)
    assert late == [], late


def test_bench_traffic_staleness_and_timed_blocks(tmp_path, monkeypatch):
    """roofline.traffic is a RECORDED figure -- bench.py says whether the scan kernel's source is
    still the one the PMC passes profiled (`traffic_stale`), and a timed block shorter than a second is repeated and the
    median block reported."""
    import hashlib
    import types
    sys.path.insert(0, ROOT)
    import bench
    traffic, src, stale = bench.recorded_traffic(3, 20, 1, True)
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_scan_traffic.json"))[-1]
    pm = json.load(open(os.path.join(ROOT, "profiles", newest)))
    assert traffic == pm["configs"]["cfg3"]["traffic_bytes_per_launch"] and newest in src
    have = hashlib.sha256(open(os.path.join(ROOT, "pir_amd", "csrc", "scan_mfma.hip"), "rb").read()).hexdigest()[:16]
    assert stale == (pm["scan_source_sha16"] != have)          # committed state: False; a changed kernel: True
    monkeypatch.setattr(bench, "scan_source_sha16", lambda: "0" * 16)
    assert bench.recorded_traffic(3, 20, 1, True)[2] is True
    assert bench.recorded_traffic(3, 19, 1, True) == (None, None, None)        # another shape: nothing recorded
    # timed blocks: 10 ms steps x 5 = 50 ms per block -> repeated up to 9 times (odd), median reported
    calls = []
    t = bench.timed_steps(lambda: (calls.append(1), time.sleep(0.01)), lambda: None, 5, 1, None, False, None, "cpu")
    assert len(bench.BLOCK_LOG[-1]) == 9 and len(calls) == 1 + 9 * 5
    assert 0.045 < t < 0.2 and abs(t - sorted(bench.BLOCK_LOG[-1])[4]) < 1e-5
    # a block of a second or more is not repeated
    t = bench.timed_steps(lambda: time.sleep(0.26), lambda: None, 4, 0, None, False, None, "cpu")
    assert len(bench.BLOCK_LOG[-1]) == 1 and t >= 1.0


def test_bench_withholds_a_stale_valu_table(tmp_path, monkeypatch):
    """`roofline_compute` is a RECORDED table (PMC passes cannot run inside the bench).  The file is
    stamped with the sha256 of the transform kernels' sources (tools/valu_roofline.py); bench.py reports the table only
    while the sources it runs are the ones that were profiled -- with a doctored hash the field says `stale: true` and the
    per-kernel numbers are withheld; tools/check_pmc_fresh.py refuses the same file."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import check_pmc_fresh
    import valu_roofline
    assert bench.kernel_sources_sha16() == valu_roofline.kernel_sources_sha16() == check_pmc_fresh.sha16(check_pmc_fresh.KERNEL_SOURCES)
    prof = tmp_path / "profiles"
    prof.mkdir()
    table = {"peak": {}, "kernels": {"deg12::ks_digit_kernel<1, true, true, true> grid=2097152": {"valu_issue_frac": 0.68}},
             "kernel_sources_sha16": bench.kernel_sources_sha16(), "commit": "unknown"}
    f = prof / "r99_valu_roofline.json"
    f.write_text(json.dumps(table))
    real = bench.kernel_sources_sha16()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_sha16", lambda: real)          # (ROOT moved: keep hashing the real sources)
    fresh = bench.recorded_compute()
    assert fresh["stale"] is False and fresh["kernels"] == table["kernels"] and fresh["file"] == "r99_valu_roofline.json"
    assert check_pmc_fresh.check(str(f)) in (0, 1)        # (1 only if the sources have uncommitted edits right now)
    # a kernel change after the passes = another hash: table withheld
    table["kernel_sources_sha16"] = "0" * 16
    f.write_text(json.dumps(table))
    stale = bench.recorded_compute()
    assert stale["stale"] is True and "kernels" not in stale and stale["kernel_sources_sha16_profiled"] == "0" * 16
    assert check_pmc_fresh.check(str(f)) == 1
    # an older file without a stamp: reported, flagged as untellable
    del table["kernel_sources_sha16"]
    f.write_text(json.dumps(table))
    assert bench.recorded_compute()["stale"] is None
