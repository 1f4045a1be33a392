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
