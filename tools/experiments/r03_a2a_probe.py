"""RCCL single-rank all_to_all_single / all_gather / reduce_scatter self-copy at growing sizes: where does it stop copying?"""
import os, sys
import torch, torch.distributed as dist
sys.stdout.flush(); out = os.fdopen(os.dup(1), "w"); os.dup2(2, 1)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
for mb in (64, 256, 512, 640, 768, 1024, 1100, 1360, 2048, 3000):
    n = mb * (1 << 20) // 8
    send = torch.arange(n, dtype=torch.int64, device="cuda")
    recv = torch.zeros(n, dtype=torch.int64, device="cuda")
    dist.all_to_all_single(recv, send, [n], [n])
    torch.cuda.synchronize()
    eq = recv == send
    bad = int((~eq).sum().item())
    first = int((~eq).nonzero()[0].item()) if bad else -1
    print("all_to_all %5d MB: wrong words %d, first wrong word %d (byte %d)" % (mb, bad, first, first * 8), file=out)
    del send, recv, eq
    # split in <= 256 MB pieces
    send = torch.arange(n, dtype=torch.int64, device="cuda")
    recv = torch.zeros(n, dtype=torch.int64, device="cuda")
    step = (256 << 20) // 8
    for o in range(0, n, step):
        e = min(n, o + step)
        dist.all_to_all_single(recv[o:e], send[o:e], [e - o], [e - o])
    torch.cuda.synchronize()
    print("   in 256 MB pieces: ok=%s" % bool(torch.equal(recv, send)), file=out)
    del send, recv
out.flush()
dist.destroy_process_group()
