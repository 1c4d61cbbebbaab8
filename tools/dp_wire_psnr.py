#!/usr/bin/env python3
"""Does a 16-bit wire for the ray-DP gradient exchange (PVD_DP_WIRE=f16|bf16) cost PSNR?  Two ranks (gloo, sharing cuda:0 -- the
boxes have one GPU), 2048 rays each = the bench's 4096-ray step, the bench's staged distillation schedule (bench.py: psnr_run), then
student-vs-teacher PSNR on the held-out views; one run per wire format, same seeds.  VERDICT r4 "next" 3(b).
Round 6: the same question for a hash student's table gradient crossing as the half-precision table the scatter wrote
(PVD_DP_HASH_WIRE=f16, the default) against widened to fp32:  python tools/dp_wire_psnr.py 3000,500,1500,6000 f32,f16,f32,f16 hash PVD_DP_HASH_WIRE"""
import json
import os
import socket
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, wire, sched, out_path, student="vm", knob="PVD_DP_WIRE"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ[knob] = wire
    sys.path[:0] = [REPO, os.path.join(REPO, "aaai2023-pvd_amd")]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bench
    from pvd.trainer import RayDP
    dp = RayDP()
    torch.cuda.manual_seed(1234 + rank)
    ts, s1, s2, tot = sched
    res = bench.psnr_run(torch.device("cuda:0"), student, ts, s1, s2, tot, oracle_check=False, dp=dp, num_rays=2048)
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


def _port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


if __name__ == "__main__":
    sched = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3000,500,1500,6000").split(","))
    # python tools/dp_wire_psnr.py [schedule] [values] [student] [knob]   e.g.  ... 3000,500,1500,6000 f32,f16 hash PVD_DP_HASH_WIRE
    student = sys.argv[3] if len(sys.argv) > 3 else "vm"
    knob = sys.argv[4] if len(sys.argv) > 4 else "PVD_DP_WIRE"
    print("student %s, %s, two ranks (gloo) sharing one GPU, 2048 rays each" % (student, knob), flush=True)
    for wire in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["f32", "f16", "bf16"]):
        out = "/tmp/dp_wire_%s.json" % wire
        mp.spawn(_worker, args=(2, _port(), wire, sched, out, student, knob), nprocs=2, join=True)
        r = json.load(open(out))
        print("wire %-4s: student vs teacher %.2f dB, student vs ground truth %.2f dB (teacher %.2f dB); %d steps, distillation %.1f s"
              % (wire, r["student_vs_teacher_heldout_db"], r["student_vs_gt_db"], r["teacher_vs_gt_db"], r["steps"], r["distill_s"]), flush=True)
