"""Rank process for tests/test_launcher.py, started by pir_amd.launcher.spawn_ranks (environment as torch.distributed.run
sets it).  Modes: `rows` = the row-sharded step over gloo with the oracle-backed server (tests/test_distributed_gloo.py),
rank 0 prints ONE JSON line; `rows8` = the same at whatever world size was spawned with the uneven 13-row shards of
tests/test_distributed_world8.py; `fail` = rank 1 exits 3 at once while rank 0 would block for a minute."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "fail":
        if rank == 1:
            sys.exit(3)
        time.sleep(60)
        print(json.dumps({"should": "never be printed"}))
        return
    # like bench.py: stdout carries exactly one JSON line; what native libraries print there (gloo's banner) -> stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if mode == "rows8":
            from test_distributed_world8 import world8_check
            ok = world8_check(rank, world, 13, 1, "packed")
        else:
            from test_distributed_gloo import rows_step_check
            ok = rows_step_check(rank, world, 2, 300, 288, 2)
        import torch
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks": dist.get_world_size(), "backend": dist.get_backend(),
                              "all_ranks_ok": int(t.item()) == world}), file=result_out)
            result_out.flush()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
