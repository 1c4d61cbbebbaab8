"""Probe (GPU box, one rank): can RCCL collectives and a child-graph launch be captured into a HIP graph here?"""
import os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.ones(13 * 1024 * 1024 // 4, device=dev); s = torch.ones(4, device=dev)
dist.all_reduce(x); dist.all_reduce(s); torch.cuda.synchronize()
# graph A: stands for the next step's prefix
a_in = torch.zeros(1 << 20, device=dev); a_out = torch.zeros(1 << 20, device=dev)
ga = torch.cuda.CUDAGraph()
with torch.cuda.graph(ga, capture_error_mode="thread_local"):
    a_out.copy_(torch.sin(a_in) + 1)
ga.replay(); torch.cuda.synchronize()
for child in (False, True):
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            y = x * 2
            dist.all_reduce(s)
            z = y + s[0]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if child:
                    ga.replay()          # child graph node?
                else:
                    w = torch.sin(y)
            dist.all_reduce(x)
            torch.cuda.current_stream().wait_stream(side)
            out = z + x[0] + a_out[0]
        a_in.fill_(0.5)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        print("capture OK (child graph launch inside: %s); replay %.1f us; a_out[0]=%.4f (expect %.4f)" % (child, (time.perf_counter() - t0) / 50 * 1e6, float(a_out[0]), 1 + float(torch.sin(torch.tensor(0.5)))))
    except Exception as e:
        print("capture FAILED (child graph launch inside: %s): %s %s" % (child, type(e).__name__, str(e)[:300]))
        torch.cuda.synchronize()
dist.destroy_process_group()
