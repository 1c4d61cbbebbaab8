import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x=torch.ones(13*1024*1024//4, device=dev); s=torch.ones(4, device=dev)
dist.all_reduce(x); dist.all_reduce(s); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
side=torch.cuda.Stream()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        y=x*2
        dist.all_reduce(s)
        z=y+s[0]
        # fork: collective on main, other work on side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            w=torch.sin(y)
        dist.all_reduce(x)
        torch.cuda.current_stream().wait_stream(side)
        out=w+x[0]
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print("in-graph RCCL capture OK; replay %.1f us; x[0]=%g s[0]=%g" % ((time.perf_counter()-t0)/50*1e6, float(x[0]), float(s[0])))
except Exception as e:
    print("in-graph RCCL capture FAILED:", type(e).__name__, str(e)[:300])
dist.destroy_process_group()
