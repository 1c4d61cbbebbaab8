"""Can HIP events recorded INSIDE a captured hipGraph (event-record nodes) time a kernel of the replayed graph?
torch.cuda.Event(enable_timing=True, external=True) -> hipEventRecordWithFlags(hipEventRecordExternal) while capturing.
Prints the elapsed time between a pair recorded around a known kernel on a forked branch of the graph, next to the same
kernel timed eagerly.  Used to decide how bench.py measures the roofline kernel where it actually runs (in the step).
(Every step is flushed: a CUDAGraph destroyed while a capture is under way aborts the process on ROCm -- ~CUDAGraph calls
hipDeviceSynchronize -- so a failed variant ends its capture before anything is dropped.)"""
import gc
import sys

import torch


def say(*a):
    print(*a, flush=True)


dev = torch.device("cuda:0")
x = torch.randn(64 << 20, device=dev)
z = torch.empty_like(x)


def work():  # ~256 MB read + 256 MB written
    torch.mul(x, 2.0, out=z)


for _ in range(3):
    work()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); work(); e1.record(); torch.cuda.synchronize()
say("eager: %.3f ms" % e0.elapsed_time(e1))
gc.disable()
keep = []
for external in (True, False):
    kw = {"external": True} if external else {}
    ev = [torch.cuda.Event(enable_timing=True, **kw) for _ in range(4)]
    side, branch = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    keep.append((g, ev, side, branch))
    y = torch.empty(1 << 20, device=dev)
    side.wait_stream(torch.cuda.current_stream())
    err = None
    with torch.cuda.stream(side):
        g.capture_begin(capture_error_mode="thread_local")
        try:
            y.mul_(1.0001)
            branch.wait_stream(side)
            with torch.cuda.stream(branch):
                ev[0].record(branch)
                work()
                ev[1].record(branch)
            ev[2].record(side)
            for _ in range(20):
                y.mul_(1.0001)
            ev[3].record(side)
        except Exception as e:  # noqa: BLE001
            err = "%s: %s" % (type(e).__name__, str(e)[:200])
        try:
            side.wait_stream(branch)
            g.capture_end()
        except Exception as e:  # noqa: BLE001
            err = (err or "") + " | capture_end: %s" % str(e)[:200]
    if err:
        say("external=%s: capture FAILED %s" % (external, err))
        continue
    try:
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        say("external=%s: kernel on the branch %.3f ms, 20 small kernels on the main chain %.3f ms, branch start -> main end %.3f ms" % (
            external, ev[0].elapsed_time(ev[1]), ev[2].elapsed_time(ev[3]), ev[0].elapsed_time(ev[3])))
    except Exception as e:  # noqa: BLE001
        say("external=%s: replay / read-out FAILED %s: %s" % (external, type(e).__name__, str(e)[:300]))
say("done")
