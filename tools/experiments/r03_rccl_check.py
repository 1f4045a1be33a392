"""Single-rank RCCL process group: which collective of the row-sharded step changes the data it should only copy?"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.stdout.flush(); out = os.fdopen(os.dup(1), "w"); os.dup2(2, 1)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
import pir_amd
from pir_amd import distributed as D
import bench
class A: pass
args = A(); args.config = 3; args.log_items = int(sys.argv[1]) if len(sys.argv) > 1 else 14; args.dims = 2
enc, pp, _ = bench.build_workload(args, pir_amd)
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
raw, keys, queries = bench.synthetic_inputs(pp, n_queries=batch)
db = pir_amd.PIRDatabase.Create(pp, device=0); db.populate(raw)
if len(sys.argv) > 3 and sys.argv[3] == "release":
    db.finalize(release_staging=True)
srv = pir_amd.PIRServer.Create(db, pp); srv.set_galois_keys(keys); srv.set_concurrency(16)
srv.stage_batch(queries); srv.run_batch(); plain = srv.fetch_batch()
comm = D.Comm(dist, 1)
print("device_native", comm.device_native, file=out)
bufs = D.PackedBuffers(srv, batch, 0, 1, torch, "cuda:0")
srv.stage_batch(queries)
srv.batch_expand_packed(0, batch, bufs.packed[0].data_ptr(), bufs.rows_send.data_ptr(), bufs.cuts)
p0 = bufs.packed.clone(); r0 = bufs.rows_send.clone()
comm.all_gather_inplace(bufs.packed, 0)
print("all_gather keeps data:", bool(torch.equal(p0, bufs.packed)), file=out)
comm.all_to_all(bufs.rows_recv, bufs.rows_send, bufs.recv_splits, bufs.send_splits, units=bufs.per)
print("all_to_all copies data:", bool(torch.equal(r0, bufs.rows_recv)), file=out)
srv.batch_run_packed(bufs.packed.data_ptr(), 1, bufs.per, bufs.rows_recv.data_ptr())
srv.batch_reply_copy_to_device(bufs.partial.data_ptr())
part = bufs.partial.cpu().numpy().view(np.uint64)
print("partial == plain:", bool(np.array_equal(part, plain)), file=out)
comm.reduce_scatter_sum(bufs.replies, bufs.partial, 0)
print("reduce_scatter copies data:", bool(torch.equal(bufs.replies.view(-1), bufs.partial.view(-1))), file=out)
srv.reduce_fixup_device_n(bufs.replies.data_ptr(), bufs.replies.shape[0] * bufs.replies.shape[1])
got = bufs.replies.cpu().numpy().view(np.uint64)
print("replies == plain:", bool(np.array_equal(got, plain)), [i for i in range(batch) if not np.array_equal(got[i], plain[i])][:20], file=out)
bad = [i for i in range(batch) if not np.array_equal(part[i], plain[i])]
print("partial differs at queries:", bad[:20], file=out)
# pipelined
pipe = D.RowsPipeline(srv, batch, 0, 1, dist, torch, "cuda:0")
for t in range(4):
    pipe.submit()
pipe.flush()
for t in (2, 3):
    print("pipelined step", t, "== plain:", bool(np.array_equal(pipe.replies(t).cpu().numpy().view(np.uint64), plain)), file=out)
out.flush()
dist.destroy_process_group()
