#!/usr/bin/env python3
"""bench.py -- train rays/s of the hash->vm distillation step on the synthetic chair (BASELINE.json
metric "train rays/s + PSNR, hash->vm chair distill", configs[2]) on N MI355X.

One "step" = one full optimisation step of the student on one batch of 4096 rays per GPU:
ray generation -> occupancy-grid march -> student (VM) forward -> teacher (hash) forward on the
inherited samples -> compositing x2 -> distillation losses -> backward -> AdamW.  Inputs (poses,
tables, occupancy bitfield) are resident in HBM when the timed region starts.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 5

Multi-GPU = ray data parallel, weak scaling (4096 rays per GPU), one RCCL all-reduce of the flat
student gradient per step (only the table rows the occupancy grid lets a sample touch: 13 of 69 MB).  Rank 0 prints ONE
JSON line.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (the host driver does not support the legacy mode); the
# environment exports it already -- keep it if the launcher dropped it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The replayed step is a graph with two parallel chains (the next step's prefix on a forked branch).  How
# the HIP runtime spreads the graph's internal streams over hardware queues decides whether the chains share the chip or
# trip over each other: with 8 queues every run is slow (0.48-0.66 ms/step), with the default 4 one process in ~15 is, with 2
# none was in 25 runs and the step is ~1 % shorter (profiles/r02_hw_queues_ab.txt).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "aaai2023-pvd_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md

# algorithmic bytes per sample of the grid-encoder forward (SURVEY.md section 8d / BASELINE.md section 2.4):
#   4*D (position) + L*2^D*C*T (corner gathers) + L*C*T (output)
def grid_fwd_bytes_per_sample(D, C, L, T):
    return 4 * D + L * (2 ** D) * C * T + L * C * T


def kernel_source_sha16():
    """Hash of the sources of the roofline kernel: a PMC traffic record is only quoted for the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("fusedhead.hip", "grid_lookup.h", "gridencoder.hip"):
        h.update(open(os.path.join(REPO, "aaai2023-pvd_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _host_cpu():
    """(model string, physical cores, logical cpus) of the box (/proc/cpuinfo)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    ncpu = os.cpu_count() or 1
    return model, (len(cores) or ncpu), ncpu


def _cpu_protocol(step, num_rays, steps, what):
    """SURVEY section 8(d)'s protocol on a bounded sample: 5 warm-up steps, `steps` (>= 20) steps timed one by one, median rays/s;
    thread count = the best of a climb up to all physical cores (a job this small loses to oversubscription on a many-core host:
    256 threads measured 34 s/step where 8 take < 1 s -- so the climb's times are reported, and the all-physical-cores figure
    always, from one step if the climb stopped below), a 1-thread figure, the CPU's model string.  step(i) runs training step i."""
    import oracle
    model, physical, ncpu = _host_cpu()
    sweep, best_t, best_n, k = {}, None, 1, 0
    for n in sorted({c for c in (4, 8, 16, 32, 64, 128) if c < physical} | {physical}):
        torch.set_num_threads(n)
        oracle.set_num_threads(n)
        t0 = time.perf_counter()
        step(k)
        t = time.perf_counter() - t0
        k += 1
        sweep[n] = num_rays / t
        if best_t is None or t < best_t:
            best_t, best_n = t, n
        elif t > 1.3 * best_t:
            break
    if physical not in sweep:  # the climb stopped below all physical cores: that figure is part of the protocol, one step of it
        torch.set_num_threads(physical)
        oracle.set_num_threads(physical)
        t0 = time.perf_counter()
        step(k)
        sweep[physical] = num_rays / (time.perf_counter() - t0)
        k += 1
    torch.set_num_threads(best_n)
    oracle.set_num_threads(best_n)
    for _ in range(max(0, 5 - k)):  # (the climb's steps count as warm-up)
        step(k)
        k += 1
    times = []
    for _ in range(max(int(steps), 1)):
        t0 = time.perf_counter()
        step(k)
        times.append(time.perf_counter() - t0)
        k += 1
    torch.set_num_threads(1)
    oracle.set_num_threads(1)
    one = []
    for _ in range(2):
        t0 = time.perf_counter()
        step(k)
        one.append(time.perf_counter() - t0)
        k += 1
    torch.set_num_threads(best_n)
    oracle.set_num_threads(best_n)
    med = sorted(times)[len(times) // 2]
    return {"value": num_rays / med, "unit": "rays/s", "cores": int(best_n), "kind": "port",
            "protocol": "5 warm-up + %d steps timed one by one, median" % len(times),
            "cpu_model": model, "physical_cores": int(physical), "logical_cpus": int(ncpu),
            "one_thread_rays_per_s": num_rays / min(one),
            "all_physical_cores_rays_per_s": sweep.get(physical),  # (one step; the climb's own figure when it got there)
            "thread_climb_rays_per_s": {str(n): v for n, v in sweep.items()},
            "sample": what % (len(times), num_rays)}


def cpu_baseline(workload, steps, num_rays):
    """The same distillation step on the host cores through the CPU oracle (fp32), on a bounded
    sample: `steps` steps of `num_rays` rays with the GPU run's weights and occupancy grid."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.workload import DistillWorkload

    opt = PVDConfig(**{**workload.opt.__dict__, "fp16": False, "num_rays": num_rays})
    cw = DistillWorkload(oracle_ops(), "cpu", opt, teacher_pretrain_steps=0, seed=0)
    cw.tea.load_state_dict({k: v.detach().float().cpu() for k, v in workload.tea.state_dict().items()})
    cw.stu.load_state_dict({k: v.detach().float().cpu() for k, v in workload.stu.state_dict().items()})
    cw.tea.mean_count = cw.stu.mean_count = int(workload.stu.mean_count * num_rays / workload.opt.num_rays)
    return _cpu_protocol(lambda i: cw.step(), num_rays, steps,
                         "%d distillation steps x %d rays, fp32, oracle C kernels (OpenMP) + PyTorch-CPU MLP/autograd/AdamW, "
                         "same weights and occupancy grid as the GPU run")


def cpu_baseline_teacher(tea_gpu, opt, topt, steps, num_rays):
    """configs[1] on the host cores: `steps` training steps of the hash teacher (`num_rays` rays each, fp32) through the CPU
    oracle, starting from the GPU run's weights and occupancy grid (no grid update inside the sample)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle
    from oracle_ops import oracle_ops
    from pvd.config import PVDConfig
    from pvd.scene import BLENDER_INTRINSICS, get_rays
    from pvd.trainer import TeacherTrainer
    from pvd.workload import DistillWorkload

    copt = PVDConfig(**{**opt.__dict__, "fp16": False, "num_rays": num_rays})
    cw = DistillWorkload(oracle_ops(), "cpu", copt, teacher_pretrain_steps=0, seed=0)
    tea = cw.tea
    tea.load_state_dict({k: v.detach().float().cpu() for k, v in tea_gpu.state_dict().items()})
    ctopt = PVDConfig(**{**topt.__dict__, "fp16": False, "num_rays": num_rays, "update_extra_interval": 10 ** 9})
    tea.teacher_variant = True
    tea.requires_grad_(True).train()
    tea.args = tea.opt = ctopt
    tea.mean_count = int(tea_gpu.mean_count * num_rays / opt.num_rays)
    tr = TeacherTrainer(ctopt, tea, "cpu", fp16=False)
    batches = []
    for it in range(4):
        r = get_rays(cw.poses[it % len(cw.poses)][None], BLENDER_INTRINSICS, 800, 800, num_rays, generator=cw.gen)
        bg = torch.rand(1, num_rays, 3, generator=cw.gen)
        batches.append((r["rays_o"], r["rays_d"], cw.target(r["rays_o"], r["rays_d"], bg), bg))
    tr.global_step = 1  # (not a multiple of the update interval)
    return _cpu_protocol(lambda i: tr.train_step(*batches[i % 4]), num_rays, steps,
                         "%d teacher training steps x %d rays, fp32, oracle C kernels (OpenMP) + PyTorch-CPU MLP/autograd/AdamW, same weights "
                         "and occupancy grid as the GPU run, no grid update inside the sample")


def gpu_reference_step(opt, steps=10):
    """(reported baseline, next to cpu_baseline) the metric's step through the REFERENCE'S OWN native code + PyTorch on this GPU: oracle/_ref =
    raymarching.cu / shencoder.cu of the reference built for gfx950 (oracle/build_ref.py), bound under this repo's restatement of the
    reference's host code in its generic form (the reference's autograd wrappers, F.grid_sample, nn.Linear under autocast, torch AdamW +
    GradScaler, eager launches); the hash teacher's lookup is this repo's generic encoder kernel (gridencoder.cu does not build on HIP).
    None when oracle/_ref did not travel with the tree."""
    try:
        sys.path.insert(0, os.path.join(REPO, "tools"))
        from oracle.build_ref import available
        if set(available()) != {"_raymarching_ref", "_shencoder_ref"}:
            return None
        from bench_reference_kernels_step import reference_kernel_ops
        from pvd.config import PVDConfig
        from pvd.workload import DistillWorkload
        dev = torch.device("cuda", torch.cuda.current_device())
        ropt = PVDConfig(**{**opt.__dict__})
        torch.manual_seed(0)
        w = DistillWorkload(reference_kernel_ops(), dev, ropt, teacher_pretrain_steps=0, seed=0)
        for _ in range(3):
            w.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return {"value": opt.num_rays / dt, "unit": "rays/s", "ms_per_step": dt * 1e3, "kind": "reference kernels (hipcc build of the reference's raymarching.cu / "
                "shencoder.cu) + PyTorch-ROCm eager, generic host code; hash lookup: this repo's generic encoder kernel",
                "sample": "%d eager steps x %d rays after 3 warm-up steps, same scene / models / AMP as the timed run" % (steps, opt.num_rays)}
    except Exception as e:  # noqa: BLE001  (never lose the line to a reported baseline)
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def psnr_run(dev, student, teacher_steps, stage1, stage2, steps, oracle_check=True, dp=None, num_rays=None):
    """The metric's second half, OUTSIDE the timed region: a whole (short) distillation run through the three stages of the
    reference schedule (main_distill_mutual.py:387-396, scaled: stage 1 feature loss only, stage 2 + sigma / colour, stage 3 + RGB)
    from a teacher trained on the analytic scene, then PSNR on held-out views rendered with the inference path
    (distill_mutual/utils.py:491-529 PSNRMeter, :1246-1265 evaluate_one_epoch): student vs teacher, student vs the analytic
    ground truth, teacher vs ground truth -- and the SAME rays through the HIP path and through the CPU oracle path (fp32, same
    weights): the render-level parity figure."""
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, get_rays, synthetic_poses
    from pvd.trainer import psnr
    from pvd.workload import DistillWorkload

    opt = PVDConfig(model_type=student, iters=steps, stage_iters={"stage1": stage1, "stage2": stage2}, **({"num_rays": num_rays} if num_rays else {}))
    t0 = time.perf_counter()
    w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=teacher_steps, start_stage="stage1", dp=dp)
    if dp is not None and dp.enabled:  # (tools/dp_wire_psnr.py) replicas start bit-identical, as in main()
        import pvd_hip
        for m in (w.tea, w.stu):
            for t in list(m.parameters()) + list(m.buffers()):
                d = t.data
                if not d.is_contiguous():
                    d = d.permute(0, 2, 3, 1) if d.dim() == 4 else d.permute(0, 2, 3, 4, 1)
                dist.broadcast(d, src=0)
            pvd_hip.note_weights_changed(list(m.parameters()))
    torch.cuda.synchronize()
    t_teacher = time.perf_counter() - t0
    tr = w.trainer

    def spg():  # stage 3 has no further boundary: ten steps per graph launch (a stage mark may be overshot by < 10 steps)
        return 10 if tr._stage_of(tr.global_step + 3) == 3 else 1

    t1 = time.perf_counter()
    while tr.global_step < steps:
        if tr._stage_of(tr.global_step) != getattr(tr, "_captured_stage", None) or not getattr(w, "_graph", False):
            w.enable_graph(steps_per_graph=spg())
        w.step()
    torch.cuda.synchronize()
    t_distill = time.perf_counter() - t1
    held_out = torch.from_numpy(synthetic_poses(np.random.RandomState(123))[:4]).to(dev)
    res = 200
    k = res / 800.0
    intr = tuple(v * k for v in BLENDER_INTRINSICS)
    rows = []
    for m in (w.stu, w.tea):
        m.eval()
    with torch.no_grad():
        for pose in held_out:
            r = get_rays(pose[None], intr, res, res, -1)
            with torch.autocast("cuda", dtype=torch.float16):
                s_img = w.stu.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
                t_img = w.tea.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"]
            gt = w.target(r["rays_o"], r["rays_d"], torch.ones(1, res * res, 3, device=dev))
            rows.append((float(psnr(s_img, t_img)), float(psnr(s_img, gt)), float(psnr(t_img, gt))))
    st, sg, tg = (float(v) for v in np.mean(np.array(rows), axis=0))
    out = {"student_vs_teacher_heldout_db": st, "student_vs_gt_db": sg, "teacher_vs_gt_db": tg, "steps": int(tr.global_step),
           "schedule": "teacher %d steps on the analytic scene; distillation stage 1 to %d, stage 2 to %d, stage 3 to %d (the reference "
                       "schedule scaled); 4 held-out %dx%d views, inference path (march_rays / composite_rays / compact_rays), fp16 AMP"
                       % (teacher_steps, stage1, stage2, int(tr.global_step), res, res),
           "teacher_train_s": t_teacher, "distill_s": t_distill}
    if oracle_check:
        # the same rays through libpvd_hip.so (GPU, fp32) and through the oracle operator set (CPU, fp32), same weights
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from oracle_ops import oracle_ops
        from pvd.workload import make_model
        res_o = 64
        intr_o = tuple(v * res_o / 800.0 for v in BLENDER_INTRINSICS)
        r = get_rays(held_out[:1], intr_o, res_o, res_o, -1)
        par = {}
        with torch.no_grad():
            for name, m in (("student", w.stu), ("teacher", w.tea)):
                g_img = m.render(r["rays_o"], r["rays_d"], staged=True, bg_color=1, perturb=False, max_steps=1024)["image"].float().cpu()
                cm = make_model(oracle_ops(), opt, m.model_type, name == "teacher", "cpu").eval()
                cm.load_state_dict({kk: v.detach().float().cpu() for kk, v in m.state_dict().items()})
                cm.mean_count = m.mean_count
                c_img = cm.render(r["rays_o"].cpu(), r["rays_d"].cpu(), staged=True, bg_color=1, perturb=False, max_steps=1024)["image"].float()
                par[name] = (float(psnr(g_img, c_img)), float((g_img - c_img).abs().max()))
        out["hip_vs_oracle_same_rays_db"] = min(par["student"][0], par["teacher"][0])
        out["hip_vs_oracle_detail"] = {"student_db": par["student"][0], "student_max_abs": par["student"][1], "teacher_db": par["teacher"][0],
                                       "teacher_max_abs": par["teacher"][1],
                                       "what": "one held-out %dx%d view, fp32, HIP operators on the GPU vs oracle operators on the CPU, same weights" % (res_o, res_o)}
    for m in (w.stu, w.tea):
        m.train()
    return out


def teacher_workload(args, dev):
    """BASELINE.json configs[1]: training step of the hash (INGP) teacher on the synthetic chair, 4096 rays/batch, incl. the
    occupancy-grid update every 16 steps.  Eager launches (the sample budget changes with every grid update), so the
    hash-lookup roofline is timed with HIP events INSIDE the timed region."""
    import pvd_hip
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.scene import BLENDER_INTRINSICS, get_rays
    from pvd.trainer import TeacherTrainer, psnr
    from pvd.workload import DistillWorkload, measure_mean_count

    if not args.eager:  # whole blocks of 16 steps (one graph launch each), at least two blocks of warm-up
        args.steps = max(16, args.steps // 16 * 16)
        args.warmup = max(32, (args.warmup + 15) // 16 * 16)
    opt = PVDConfig(num_rays=args.rays, fp16=not args.fp32)
    w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=0)
    topt = PVDConfig(**{**opt.__dict__, "model_type": opt.teacher_type, "iters": 30000, "stage_iters": {"stage1": -1, "stage2": -1}})
    tea = w.tea
    tea.teacher_variant = True
    tea.requires_grad_(True).train()
    tea.args = tea.opt = topt
    tr = TeacherTrainer(topt, tea, dev, fp16=not args.fp32)
    tea.mean_count = measure_mean_count(tea, w.poses, opt, generator=w.gen)
    batches = []
    data = "synthetic (analytic chair-like scene; no dataset offline)"
    if args.data_root:
        # a Blender-format scene on disk through the reference-shaped reader (pvd/provider.py <- distill_mutual/provider.py:133-326):
        # frames + cameras from transforms_train.json, ground truth = the PNGs' pixels blended over a random background by their alpha
        from pvd.provider import BlenderScene, training_target
        scene = BlenderScene(args.data_root, "train", scale=opt.scale, device=dev, num_rays=opt.num_rays, error_map=args.error_map)
        if args.error_map:
            # the reference's --error_map loop (utils.py:1120-1129): every step draws its pixels by the frame's current map and feeds the
            # per-ray error back -- the batch depends on the previous step's result, so these steps run eagerly, outside the timed blocks
            args.eager = True
            em_losses = []
            for it in range(32):
                b = scene.batch([it % len(scene)], generator=w.gen)
                gt, bg = training_target(b["images"], generator=w.gen)
                l, _ = tr.train_step(b["rays_o"], b["rays_d"], gt, bg, error_sink=lambda e, b=b: scene.update_error(b, e))
                em_losses.append(float(l))
            data_note = "; --error_map: 32 eager steps drew their pixels by the per-frame error map (map range %.3g .. %.3g after them)" % (
                float(scene.error_map.min()), float(scene.error_map.max()))
        else:
            data_note = ""
        for it in range(16):
            b = scene.batch([it % len(scene)], generator=w.gen)
            gt, bg = training_target(b["images"], generator=w.gen)
            batches.append((b["rays_o"], b["rays_d"], gt, bg))
        data = "Blender-format scene read by pvd/provider.py from %s: %d train views of %dx%d (tools/make_blender_scene.py: the synthetic chair written to disk)" % (
            args.data_root, len(scene), scene.W, scene.H) + data_note
    else:
        for it in range(16):
            r = get_rays(w.poses[it % len(w.poses)][None], BLENDER_INTRINSICS, 800, 800, opt.num_rays, generator=w.gen)
            bg = torch.rand(1, opt.num_rays, 3, device=dev, generator=w.gen)
            batches.append((r["rays_o"], r["rays_d"], w.target(r["rays_o"], r["rays_d"], bg), bg))
    # (the lookup of a training step: since round 6 the positions are mapped inside the kernel and the head's weight image rides on the launch)
    names = {"pvd_grid_encode_forward", "pvd_grid_encode_forward_affine", "pvd_grid_encode_forward_affine_pack"}
    block = (not args.eager) and args.steps % 16 == 0 and args.warmup % 16 == 0 and args.warmup >= 32
    launch = "eager"
    if block:
        # graph mode: one eager block first (lazy initialisations, a measured mean_count), then 16 steps = one graph launch,
        # the occupancy-grid update (eager) between the launches
        for it in range(16):
            tr.train_step(*batches[it])
        try:
            tr.capture_block(batches)
            for _ in range(args.warmup // 16 - 1):
                tr.train_block()
            launch = "hipGraph replay, 16 steps (one block between occupancy-grid updates) per graph launch; %d sample rows allocated" % tea.sample_alloc
        except Exception:
            import traceback
            traceback.print_exc(file=sys.stderr)
            torch.cuda.synchronize()
            block = False
            tea.sample_alloc = None
    if not block:
        for it in range(args.warmup):
            tr.train_step(*batches[it % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if block:
        for _ in range(args.steps // 16):
            loss, pred = tr.train_block()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # the lookup of one more step, eagerly, between HIP events (it sits inside a replayed graph in the timed region)
        with pvd_hip.KernelTimer(names) as kt:
            for it in range(8):
                tr.train_step(*batches[1 + it])  # (not a multiple of 16: no grid update)
            torch.cuda.synchronize()
    else:
        with pvd_hip.KernelTimer(names) as kt:
            for it in range(args.steps):
                loss, pred = tr.train_step(*batches[it % 16])
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    name = max(sorted(names), key=kt.launches)
    ms = kt.mean_ms(name)
    B, D, C, L, dt_code = kt.meta[name][-1]
    T = 2 if dt_code == 1 else 4
    bps = grid_fwd_bytes_per_sample(D, C, L, T)
    achieved = bps * B / (ms * 1e-3) / 1e9
    out = {
        "metric": "train rays/s (hash teacher training step)", "value": args.steps * args.rays / elapsed, "unit": "rays/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.fp32 else "f16 tables+MLP (AMP) / f32 marcher+compositor",
        "data": data,
        "config": {"workload": "train hash teacher, synthetic chair, %d rays/step, occupancy update every %d steps"
                               % (args.rays, topt.update_extra_interval), "rays_per_gpu": args.rays, "parallelism": "single GPU",
                   "launch": launch, "mean_count": int(tea.mean_count), "psnr_vs_analytic_gt_db": float(psnr(pred.detach(), batches[(args.steps - 1) % 16][2])),
                   "loss": float(loss)},
        "roofline": {"kernel": "k_grid_fwd_lps<2> / k_grid_fwd<%s,3,2> (%s), HIP events around eager launches%s"
                               % ("f16" if T == 2 else "f32", name, " right after the timed region" if block else " inside the timed region"),
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes_per_launch": bps * B, "bytes_per_sample": bps, "samples_per_launch": B,
                     "us_per_launch": ms * 1e3, "launches": kt.launches(name)},
        "cpu_baseline": None,
    }
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline_teacher(tea, opt, topt, max(2, args.cpu_steps // 2), args.rays)
        except Exception as e:  # noqa: BLE001  (never lose the line to the baseline)
            import traceback
            traceback.print_exc(file=sys.stderr)
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    print(json.dumps(out))


def _spawn_ranks(n):
    """Re-execute this command line as n ranks of one node (what the driver's `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` does).  The children inherit stdout / stderr; the exit
    code is theirs."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")  # (torch.distributed.run sets it anyway and says so on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--student", type=str, default="vm")
    ap.add_argument("--teacher-pretrain", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=20, help="timed steps of the CPU baseline (after 5 warm-up steps; SURVEY 8d)")
    ap.add_argument("--fp32", action="store_true", help="disable AMP (the reference forces fp16 on)")
    ap.add_argument("--eager", action="store_true", help="do not capture the step into HIP graphs")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --rays is the GLOBAL batch, split evenly across the ranks (default: weak, --rays per GPU)")
    ap.add_argument("--bound", type=float, default=1.0, help="scene bound (> 1: more than one occupancy cascade, as for Tanks&Temples, configs[4])")
    ap.add_argument("--dt-gamma", type=float, default=0.0, help="dt_gamma of the marcher (the reference uses 1/256 for unbounded scenes)")
    ap.add_argument("--data-type", choices=["synthetic", "llff", "tank"], default="synthetic",
                    help="random-camera generator of the distillation (get_rand_poses, distill_mutual/utils.py:100-197): configs[3] uses llff, configs[4] tank")
    ap.add_argument("--teacher", type=str, default="hash", help="teacher model type (configs[3]: mlp)")
    ap.add_argument("--data-root", type=str, default=None,
                    help="--workload teacher: train from a Blender-format scene on disk (transforms_train.json + PNGs) through pvd/provider.py")
    ap.add_argument("--error-map", action="store_true",
                    help="--workload teacher --data-root: sample pixels by the per-image error map and update it every step (the reference's --error_map; eager steps)")
    ap.add_argument("--scene-scale", type=float, default=1.0, help="scale of the synthetic scene (with --bound > 1)")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="after the timed region: this many more steps in one synchronised window (`sustained`); 0 = skip")
    ap.add_argument("--no-psnr", action="store_true", help="skip the staged distillation run + held-out PSNR (`psnr`, ~12 s, outside the timed region)")
    ap.add_argument("--psnr-schedule", type=str, default="3000,500,1500,6000", help="teacher steps, end of stage 1, end of stage 2, total distillation steps")
    ap.add_argument("--occupancy", type=int, choices=[1, 5, 15], default=5,
                    help="SURVEY 8(d)'s occupancy sweep: ~1 / ~5 (the metric's scene) / ~15 %% of the 128^3 cells occupied (ChairScene(thicken = 0 / 0.08 / 0.2))")
    ap.add_argument("--workload", choices=["distill", "teacher"], default="distill",
                    help="distill = BASELINE.json's metric (configs[2]); teacher = hash teacher training step (configs[1], single GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started the way the driver starts `--gpus 1`: spawn the N ranks here (one process per GPU through
        # torch.distributed.run on a free local port) and hand their output through -- rank 0 prints the ONE JSON line
        return _spawn_ranks(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    # PVD_DIST_BACKEND=gloo + fewer GPUs than ranks: the N > 1 code path (two-graph capture, compact exchange) can be
    # exercised on a single-GPU box; the driver's runs use one GPU per rank over RCCL
    backend = os.environ.get("PVD_DIST_BACKEND", "nccl")
    assert backend != "nccl" or world <= torch.cuda.device_count(), \
        "--gpus %d over RCCL needs one GPU per rank (%d visible); PVD_DIST_BACKEND=gloo lets the ranks share a GPU" % (world, torch.cuda.device_count())
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # PVD_DP_FORCE=1: run the ray-DP code path (collectives, segmented capture, compact exchange) in a world of ONE rank,
    # so that RCCL itself (its streams and watchdog thread next to graph capture) is exercised on a single-GPU box
    force_dp = os.environ.get("PVD_DP_FORCE") == "1"
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, "--gpus must match WORLD_SIZE"
    global_rays = args.rays * (1 if args.strong else world)
    if args.strong:  # SURVEY section 8e's partition: one global batch of pixels, rank r renders its contiguous slice
        assert args.rays % world == 0, "--strong needs --rays divisible by the number of GPUs"
        args.rays //= world
    if args.workload == "teacher":
        assert world == 1, "the teacher workload is single-GPU"
        return teacher_workload(args, dev)

    import pvd_hip
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.trainer import RayDP, psnr
    from pvd.workload import DistillWorkload

    opt = PVDConfig(num_rays=args.rays, model_type=args.student, teacher_type=args.teacher, fp16=not args.fp32, bound=args.bound, dt_gamma=args.dt_gamma,
                    data_type=args.data_type)
    dp = RayDP()
    w = DistillWorkload(hip_ops(), dev, opt, teacher_pretrain_steps=args.teacher_pretrain, seed=0, dp=dp, scene_scale=args.scene_scale,
                        thicken={1: 0.0, 5: 0.08, 15: 0.2}[args.occupancy])
    if dp.enabled:  # replicas must start bit-identical (teacher pre-training uses float atomics)
        for m in (w.tea, w.stu):
            for t in list(m.parameters()) + list(m.buffers()):
                d = t.data
                if not d.is_contiguous():  # channels-last VM factors: broadcast the dense [H][W][R] view
                    d = d.permute(0, 2, 3, 1) if d.dim() == 4 else d.permute(0, 2, 3, 4, 1)
                    assert d.is_contiguous()
                dist.broadcast(d, src=0)
            pvd_hip.note_weights_changed(list(m.parameters()))  # .data writes do not bump autograd versions: drop derived caches

    torch.cuda.manual_seed(1234 + rank)  # in-graph ray / background sampling: different rays on every rank
    launch_mode = "eager"

    def pick(n):
        return next((k for k in (20, 10, 5, 4, 2) if n % k == 0), 1)

    def record(spg):
        """The step as HIP graph(s) of spg steps each (launch-bound eagerly: ~200 kernels of a few us); returns how it is launched."""
        try:
            # (roofline, in_step) every recorded launch of the teacher's lookup + head leaves its own extent behind: read after the timed region
            w.trainer.record_fused_spans = not dp.enabled
            w.enable_graph(steps_per_graph=spg)
            return "hipGraph replay" + (" (%d steps per graph launch%s)" % (
                w.steps_per_call, (", the next replay's marches + teacher forwards on ONE forked branch per graph" if getattr(w.trainer, "pipeline_fork", "") == "graph" else
                                           ", next step's march + teacher forward on a forked branch (fork at %s)" % getattr(w.trainer, "pipeline_fork", "mid"))
                if getattr(w.trainer, "pipelined_ingraph", False) else "")
                if w.steps_per_call > 1 else "")
        except Exception as e:  # never lose the measurement to a capture problem: fall back to eager launches
            import traceback
            traceback.print_exc(file=sys.stderr)
            torch.cuda.synchronize()
            try:  # a failed capture leaves its error in the runtime's last-error slot: consume it with a throw-away launch
                pvd_hip.check_finite(torch.zeros(4, device=dev), torch.zeros(1, device=dev))
            except Exception:
                pass
            w._graph = False
            w._eager_device_batches = True  # keep using the one-kernel batch generator (no torch generator involved)
            return "eager (graph capture failed: %s: %s)" % (type(e).__name__, str(e)[:120])

    # steps per graph launch: the largest of 20, 10, 5, 4, 2 that divides the timed step count (exactly K timed steps).  A warm-up count it
    # does not divide (the driver's 5 before 20) is replayed from a recording of its own first -- single GPU only: under ray-DP one
    # recording serves both, sized to divide both
    forced = os.environ.get("PVD_STEPS_PER_GRAPH", "")
    untimed_extra = 0  # steps run before the timed window beyond --warmup and beyond the three eager steps every recording starts with
    spg = int(forced) if forced else pick(args.steps)
    two_recordings = not args.eager and not forced and not dp.enabled and args.warmup > 0 and args.warmup % spg != 0
    if not forced and not two_recordings and args.warmup % spg != 0:
        spg = next((k for k in (20, 10, 5, 4, 2) if args.steps % k == 0 and args.warmup % k == 0), 1)
    if two_recordings:
        launch_mode = record(pick(args.warmup))
        if launch_mode.startswith("hipGraph"):
            assert args.warmup % w.steps_per_call == 0
            for _ in range(args.warmup // w.steps_per_call):
                w.step()
            launch_mode = record(spg)
            if launch_mode.startswith("hipGraph"):
                # the timed recording's FIRST launch uploads the graph (measured: 0.2785 ms/step in the window against 0.270-0.271 on later
                # launches): one untimed replay takes that out of the window, as the warm-up replays do for a single recording
                w.step()
                untimed_extra = w.steps_per_call
            launch_mode += "; the %d warm-up steps replayed from a recording of %d steps, then one untimed replay of the timed recording" % (
                args.warmup, pick(args.warmup))
        else:  # (eager fall-back: the warm-up steps run below)
            two_recordings = False
    elif not args.eager:
        launch_mode = record(spg)
    spc = w.steps_per_call
    assert args.steps % spc == 0
    if not two_recordings:
        assert args.warmup % spc == 0
        for _ in range(args.warmup // spc):
            w.step()

    if dp.enabled:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps // spc):  # exactly args.steps optimisation steps
        loss, info, pred_stu, pred_tea = w.step()
    if dp.enabled:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dp.enabled and dist.get_world_size() > 1:  # the slowest rank's clock (the barriers make them agree to a few us anyway)
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    # ---- a longer window behind the timed one (the driver's K = 20 steps are 6 ms: nothing outside this process can see them)
    sustained = None
    if args.sustained_steps > 0:
        n_calls = max(1, args.sustained_steps // spc)
        if dp.enabled:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_calls):
            w.step()
        if dp.enabled:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        sustained = {"steps": n_calls * spc, "ms_per_step": dt / (n_calls * spc) * 1e3, "window_s": dt,
                     "vs_timed": (dt / (n_calls * spc)) / (elapsed / args.steps)}

    # ---- the roofline kernel WHERE IT RUNS, live: a few more replays behind the measured windows; every launch of the teacher's
    # lookup + head inside the replayed graph writes {first workgroup's start, last workgroup's end} (the device's 100 MHz counter)
    # into its own record (pvd_hash_head_forward_fused_span), reset here before every replay
    in_step_live = None
    spans = getattr(w.trainer, "fused_spans", None)
    if spans is not None and not args.eager:
        try:
            import pvd_hip
            init = torch.tensor([list(pvd_hip.FUSED_SPAN_INIT)] * spans.shape[0], dtype=torch.int64, device=spans.device)
            vals = []
            for _ in range(max(2, 100 // spc)):
                spans.copy_(init)
                w.step()
                torch.cuda.synchronize()
                vals.extend(float(v) for v in pvd_hip.fused_span_us(spans) if v == v)
            if vals:
                in_step_live = {"us_per_launch": float(np.median(vals)), "mean_us": float(np.mean(vals)), "min_us": float(np.min(vals)),
                                "max_us": float(np.max(vals)), "launches": len(vals)}
        except Exception as e:  # noqa: BLE001
            in_step_live = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}

    # ---- roofline of the hash-grid lookup (the kernel north_star names).  HIP events cannot sit inside the replayed step,
    # so right after the timed region the SAME kernel is launched on the samples of one more step: the frozen teacher's
    # forward is ONE launch (pvd_hash_head_forward_fused: 14-level lookup + sigma/colour head).  100 launches, captured 20
    # to a HIP graph so that the host is out of the picture, between one pair of HIP events on the launch stream; the mean
    # therefore includes one kernel boundary (~1.5 us) per launch.
    import fusedhead
    roof = None
    try:
        if getattr(w.tea, "model_type", None) != "hash":  # (configs[3]: an mlp teacher has no hash-grid lookup)
            raise LookupError("the roofline kernel (hash-grid lookup) does not run in this configuration: teacher is %r" % getattr(w.tea, "model_type", None))
        per_graph, reps = 20, 5
        n_launch = per_graph * reps
        side = torch.cuda.Stream()

        def samples_of_pose(idx):
            """the sample rows the step generates for camera `idx` of the epoch (the in-graph batch generator's pose counter)"""
            st = getattr(w, "_batch_state", None)
            if st is not None and idx is not None:
                st[0] = int(idx) % len(w.poses)
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=opt.fp16):
                rays_o, rays_d, bg, *_ = w.device_batch()
                out_stu = w.stu.render(rays_o, rays_d, staged=False, bg_color=bg, perturb=True, force_all_rays=False, dt_gamma=opt.dt_gamma,
                                       max_steps=opt.max_steps)
            return out_stu["inherited_params"][0], out_stu["inherited_params"][1]

        def time_alone(xyzs, dirs):
            with torch.no_grad():
                for _ in range(3):
                    fusedhead.hash_head_infer(w.tea, xyzs, dirs)
                torch.cuda.synchronize()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        for _ in range(per_graph):
                            fusedhead.hash_head_infer(w.tea, xyzs, dirs)
                    g.replay()
                    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev_a.record(side)
                    for _ in range(reps):
                        g.replay()
                    ev_b.record(side)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
            return ev_a.elapsed_time(ev_b) / n_launch * 1e3

        def sol_probe(xyzs):
            """The lookup's gather-only speed of light on THESE sample rows (tools/probes/hash_sol.hip, sol_gather variant 1): the product
            kernel's index code, lane mapping and therefore address stream, all 14 levels' gathers in flight, no blend, no head, 4 B
            written per lane -- what the memory system charges for the gathers alone at this launch size."""
            import ctypes
            lib = ctypes.CDLL(os.path.join(REPO, "tools", "probes", "libhash_sol.so"))
            enc = w.tea.encoder
            emb16 = enc.embeddings.detach().half().contiguous()
            x01 = ((xyzs.float() + opt.bound) / (2 * opt.bound)).contiguous()
            offs = enc.offsets.cpu().numpy().astype(np.int32)
            S = np.float32(np.log2(enc.per_level_scale))
            scales = (np.exp2(np.arange(14, dtype=np.float32) * S) * np.float32(enc.base_resolution) - np.float32(1)).astype(np.float32)
            sink = torch.zeros(2 * x01.shape[0], dtype=torch.int32, device=dev)

            def run():
                rc = lib.sol_gather(ctypes.c_int(1), ctypes.c_void_p(x01.data_ptr()), ctypes.c_void_p(emb16.data_ptr()), offs.ctypes.data_as(ctypes.c_void_p),
                                    scales.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(x01.shape[0]), ctypes.c_void_p(sink.data_ptr()),
                                    ctypes.c_uint32(0xFFFFFFFF), ctypes.c_uint32(0), ctypes.c_uint32(0),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    for _ in range(per_graph):
                        run()
                g.replay()
                ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev_a.record(side)
                for _ in range(reps):
                    g.replay()
                ev_b.record(side)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            return ev_a.elapsed_time(ev_b) / n_launch * 1e3

        # WHICH cameras: the launch's duration depends on the batch's camera (the hash keeps x-neighbours in one cache line, so
        # rays that run along x coalesce and rays that run along y or z do not: 21.7 us vs 32 us on the same box,
        # tools/probe_alone_vs_history.py).  `alone` = cameras of the TIMED REGION (what "the kernel's average launch duration
        # over the timed region" is an average over); `alone_epoch` = cameras spread over the whole epoch of 312.
        st = getattr(w, "_batch_state", None)
        pose_now = int(st[0].item()) if st is not None else None
        saved = st.clone() if st is not None else None
        P = len(w.poses)
        if pose_now is not None:
            span = args.steps + (args.sustained_steps // spc) * spc if args.sustained_steps > 0 else args.steps
            first = (pose_now - span) % P  # the pose counter advanced once per step since the timed region began (+ prefix prefetch)
            n_t = min(args.steps, 10)
            timed_cams = sorted({(first + (args.steps * i) // n_t) % P for i in range(n_t)})
            epoch_cams = [(P * i) // 8 for i in range(8)]
            next_cam = (first + args.steps) % P  # the batch right behind the timed region: what rounds 1-3 measured on (one camera)
        else:
            timed_cams, epoch_cams, next_cam = [None], [], None
        per_cam, sol_cam, sol_err = {}, {}, None
        B = None
        for cam in list(timed_cams) + [c for c in epoch_cams + ([next_cam] if next_cam is not None else []) if c not in timed_cams]:
            xyzs, dirs = samples_of_pose(cam)
            per_cam[cam] = time_alone(xyzs, dirs)
            B = int(xyzs.shape[0])
            if cam in timed_cams and sol_err is None and opt.fp16:
                try:
                    sol_cam[cam] = sol_probe(xyzs)
                except Exception as e:  # noqa: BLE001  (the probe library is a tool: never lose the line to it)
                    sol_err = "%s: %s" % (type(e).__name__, str(e)[:160])
        if saved is not None:
            st.copy_(saved)
        us_alone = float(np.mean([per_cam[c] for c in timed_cams]))
        us_epoch = float(np.mean([per_cam[c] for c in epoch_cams])) if epoch_cams else None
        fused = bool(getattr(fusedhead, "FUSED_LOOKUP", False)) and opt.fp16
        # ALGORITHMIC bytes per sample: SURVEY.md section 8(d)'s figure for the f16 lookup -- position 12 + 14 levels x 8 corners x 4 B
        # (f16 pair) = 448 gathered + 14 x 2 x 2 = 56 of features = 516 B/sample.  (The fused launch does not write the 56 B of
        # features to HBM and instead moves what the head reads and writes -- dirs 12 + sigma 4 + rgb 12 + feature_sigma_color 64:
        # 552 B/sample -- kept as `bytes_per_sample_fused` for reference; `achieved` is priced at 516.)
        bps = grid_fwd_bytes_per_sample(3, 2, 14, 2 if opt.fp16 else 4)
        gbs = lambda us_: bps * B / (us_ * 1e-6) / 1e9  # noqa: E731
        alone = {"us_per_launch": us_alone, "launches": n_launch * len(timed_cams), "achieved": gbs(us_alone), "frac": gbs(us_alone) / HBM_PEAK_GBS,
                 "cameras": [c for c in timed_cams], "us_per_camera": [per_cam[c] for c in timed_cams],
                 "timing": "HIP events on the launch stream around %d back-to-back launches (HIP graphs of %d) per camera, nothing else on the "
                           "chip, right after the timed region, on the sample rows of %d cameras of the timed region" % (n_launch, per_graph, len(timed_cams))}
        # the speed of light of the lookup at THIS launch size on THESE rows (VERDICT r5 "next" 3): `frac` = fraction of 8 TB/s the
        # gather-only probe reaches; the product kernel's fraction OF that figure is `alone.frac_of_sol` / roofline.frac_of_sol
        sol = {"error": sol_err} if sol_err else None
        if sol_cam and sol_err is None:
            us_sol = float(np.mean([sol_cam[c] for c in timed_cams]))
            sol = {"us_per_launch": us_sol, "achieved": gbs(us_sol), "frac": gbs(us_sol) / HBM_PEAK_GBS, "cameras": [c for c in timed_cams],
                   "us_per_camera": [sol_cam[c] for c in timed_cams],
                   "what": "gather-only probe (tools/probes/hash_sol.hip, all 14 levels in flight): the product kernel's index code, lane mapping "
                           "and address stream on the same sample rows and the same f16 table -- no blend, no LDS tile, no head; timed like `alone`",
                   "why_below_peak": "every fine-level corner pair is its own 128-byte line (8 useful bytes) and each of the 8 non-coherent XCD L2s "
                                     "pulls its own copy of every 2 MB level through the fabric: DESIGN.md (the roofline kernel), profiles/r06_hash_sol_table.txt"}
            alone["frac_of_sol"] = us_sol / us_alone
        alone_epoch = None
        if us_epoch is not None:
            alone_epoch = {"us_per_launch": us_epoch, "achieved": gbs(us_epoch), "frac": gbs(us_epoch) / HBM_PEAK_GBS, "cameras": epoch_cams,
                           "us_per_camera": [per_cam[c] for c in epoch_cams], "us_min": min(per_cam[c] for c in epoch_cams),
                           "us_max": max(per_cam[c] for c in epoch_cams),
                           "why": "same kernel, same launch size; the camera decides how many cache lines a wave's gathers touch (x-neighbours "
                                  "share a line, y / z neighbours do not): the mean over 8 cameras spread over the epoch of %d" % P}
        # ---- the same kernel WHERE IT RUNS: inside the replayed step it sits on the forked branch of the graph next to the student's
        # head backward and is stretched by sharing the chip.  No host-side event can be placed inside a replayed hipGraph on this
        # runtime (torch refuses external events on ROCm, plain event-record nodes return hipErrorInvalidHandle from
        # hipEventElapsedTime: tools/probe_graph_events.py, profiles/r03_graph_events_probe.txt) -- since round 4 the launch stamps
        # its own extent (in_step_live above: the headline).  The rocprofv3 record of the driver's command, committed under
        # profiles/, is kept next to it as the outside view of the same quantity when it was taken on this build of the kernel
        # (under the profiler the graph's kernels overlap less, so that figure is the shorter one).
        in_step = None
        rec_path = next((q for q in (os.path.join(REPO, "profiles", n) for n in ("r06_in_step.json", "r05_in_step.json", "r04_in_step.json", "r03_in_step.json")) if os.path.exists(q)), None)
        if fused and rec_path:
            rec = json.load(open(rec_path))
            if rec.get("source_sha16") == kernel_source_sha16():
                us_in, Bi = float(rec["median_us"]), int(rec["samples_per_launch"])
                in_step = {"us_per_launch": us_in, "mean_us": rec.get("mean_us"), "launches": rec.get("launches"), "samples_per_launch": Bi,
                           "achieved": bps * Bi / (us_in * 1e-6) / 1e9, "frac": bps * Bi / (us_in * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "timing": "NOT live: rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 5` on the builder's lease, launches that "
                                     "overlap the student's backward / scatter (profiles/%s), same kernel source hash"
                                     % os.path.basename(rec_path).replace("in_step.json", "kernel_populations.txt")}
        # ... and LIVE since round 4: the launch stamps its own extent (in_step_live above); the rocprofv3 record stays in the object
        # as the outside corroboration of the same quantity (`in_step_rocprof`)
        in_step_rocprof = in_step
        if fused and in_step_live is not None and "error" not in in_step_live:
            us_in = in_step_live["us_per_launch"]
            in_step = dict(in_step_live, samples_per_launch=B, achieved=gbs(us_in), frac=gbs(us_in) / HBM_PEAK_GBS,
                           timing="LIVE: every launch of the kernel inside the replayed graphs writes {min start, max end} over its workgroups in "
                                  "s_memrealtime ticks (10 ns) into its own record; %d replays behind the timed and the sustained window, "
                                  "median over their launches" % max(2, 100 // spc))
        # headline: the in-step figure (the figure where the kernel actually runs), else the live alone figure; both are
        # always in the object
        head = in_step if in_step is not None else alone
        traffic, traffic_note = None, "no PMC pass recorded for this build of the kernel"
        pmc_path = next((q for q in (os.path.join(REPO, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"))
                         if os.path.exists(q)), None)
        if pmc_path:  # FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 --pmc passes of THIS kernel source
            pmc = json.load(open(pmc_path))
            if pmc.get("source_sha16") == kernel_source_sha16() and fused:
                traffic = (pmc["fetch_kb"] + pmc["write_kb"]) * 1024.0 / pmc["samples_per_launch"] * head.get("samples_per_launch", B)
                traffic_note = "bytes/launch, rocprofv3 FETCH_SIZE + WRITE_SIZE (separate --pmc passes, tools/pmc_teacher_fwd.py; %s), scaled by " \
                               "samples; gather pattern, uncorrected (the guide's x2 FETCH_SIZE correction is for wide coalesced streams)" % os.path.basename(pmc_path)
            else:
                traffic_note = "%s was taken on a different build of the kernel (source hash differs)" % os.path.basename(pmc_path)
        Bh = head.get("samples_per_launch", B)
        roof = {"kernel": ("k_hash_fwd_fused (pvd_hash_head_forward_fused: hash-grid lookup f16 3x2x14 + sigma/colour head, one launch)" if fused
                           else "pvd_grid_encode_forward_affine + pvd_head_forward (two launches)"),
                "bound": "hbm", "achieved": head["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["frac"],
                "where": ("inside the replayed step, next to the student's backward (%s, see in_step); alone on "
                          "the chip, live: see alone / alone_epoch" % ("live: the launch's own start / end stamps" if in_step is not in_step_rocprof
                                                                     else "quoted rocprofv3 record of this build") if in_step is not None else
                          "alone on the chip (live HIP events, cameras of the timed region); no rocprofv3 record of this build for the in-step figure"),
                "traffic": traffic, "traffic_note": traffic_note, "algorithmic_bytes_per_launch": bps * Bh, "bytes_per_sample": bps,
                "bytes_per_sample_fused": 552 if fused else None,
                "samples_per_launch": Bh, "us_per_launch": head["us_per_launch"], "launches": head.get("launches"),
                "sol": sol, "frac_of_sol": (sol["us_per_launch"] / head["us_per_launch"]) if (sol and "us_per_launch" in sol) else None,
                "alone": alone, "alone_epoch": alone_epoch,
                "alone_next_batch": (None if next_cam is None else {"camera": next_cam, "us_per_launch": per_cam[next_cam], "frac": gbs(per_cam[next_cam]) / HBM_PEAK_GBS,
                                                                    "what": "rounds 1-3's protocol: ONE camera, the batch right behind the timed region"}),
                "in_step": in_step, "in_step_rocprof": in_step_rocprof if in_step_rocprof is not in_step else None,
                "rederive": "python tools/roofline_from_profile.py  (profiles/r06_kernel_populations.txt + r06_bench_profiled_line.json + r06_kernel_stats.csv)"}
    except LookupError as e:
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "note": str(e)}
    except Exception as e:  # noqa: BLE001  (never lose the throughput line to the roofline measurement)
        roof = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}

    samples = int(w.stu.step_counter[:, 0].float().mean().item())
    bits = w.stu.density_bitfield
    occupied = float(sum(int(((bits >> k) & 1).sum()) for k in range(8))) / float(bits.numel() * 8)
    total_rays = args.steps * global_rays
    out = {
        "metric": "train rays/s (%s->%s chair distillation step)" % (opt.teacher_type, opt.model_type),
        "value": total_rays / elapsed,
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.fp32 else "f16 tables+MLP (AMP, as the reference forces) / f32 marcher+compositor",
        "data": "synthetic (analytic chair-like scene, 800x800 Blender-style cameras at r=3.2; no dataset offline)",
        "config": {"workload": "distill %s->%s, synthetic chair, %s cameras, stage 3 (rgb + feature/sigma/colour losses), %d rays/GPU/step, "
                               "occupancy 128^3 %.1f%% occupied, max_steps 1024, teacher pre-trained %d steps%s" % (
                                   args.teacher, args.student, args.data_type, args.rays, 100.0 * occupied, args.teacher_pretrain,
                                   "" if (args.bound == 1.0 and args.dt_gamma == 0.0) else "; bound %g (%d cascades), dt_gamma %g, scene scale %g"
                                   % (args.bound, w.stu.cascade, args.dt_gamma, args.scene_scale)),
                   "rays_per_gpu": args.rays, "parallelism": "ray-dp%d" % world, "launch": launch_mode,
                   "update": {"late": "AdamW in two launches: rows the backward can write behind the scatter; the L1-only rows one step later on the forked "
                                      "branch (same bits; the reported loss counts their L1 value one step late)",
                              "1": "AdamW in two launches (L1-only rows at the start of the next step's branch)"}.get(
                       getattr(w.trainer, "adamw_split", "0"), "AdamW in one launch"),
                   # steps that ran before the timed window beyond --warmup (each is a real optimisation step): the three eager steps every
                   # recording starts with, and the untimed first replay of a timed recording that the warm-up count does not divide
                   "untimed_steps_beyond_warmup": (0 if args.eager else 3 * (2 if two_recordings else 1)) + untimed_extra,
                   "capture_fallback": (not args.eager) and not launch_mode.startswith("hipGraph replay"),  # True = the step fell back to eager launches (~5x the ms)
                   "samples_per_step_per_gpu": samples,
                   # rays/s depends on the scene through the samples a ray generates: the transferable figure is samples/s
                   "occupied_fraction": occupied, "samples_per_ray": samples / float(args.rays),
                   "samples_per_s": samples * world * args.steps / elapsed,
                   "padded_rows_per_step": int(w.stu.mean_count) + 128 - int(w.stu.mean_count) % 128,
                   "teacher_psnr_db": w.teacher_psnr,
                   "psnr_student_vs_teacher_db": float(psnr(pred_stu.detach(), pred_tea.detach())) if pred_stu is not None else None,
                   "loss": float(loss)},
        "roofline": roof,
        "sustained": sustained,
    }
    if dp.enabled:  # what one step's gradient exchange moved (informational; never let it cost the line)
        try:
            tr = w.trainer
            c = getattr(tr, "_compactor", None)
            compact = c is not None and getattr(c, "agreed", True) and c.fraction < 0.7
            total = tr.flat.flat.numel() * 4 / 1e6
            lay = getattr(tr, "_xlayouts", None)
            form = "all-reduce"
            if lay is not None:  # round 6: the gather zeroes / checks, the buffer carries the inf flag, part B of the update reads it directly
                form = ("reduce_scatter -> AdamW on the rank's rows -> all_gather (sharded update)" if lay[1].pbuf is not None else
                        "all-reduce feeding the update directly (no scatter / check launches)")
            emb = getattr(getattr(w.stu, "encoder", None), "embeddings", None)
            half_table = not compact and lay is None and emb is not None and getattr(emb, "_pvd_half_grad_taker", None) is not None
            if half_table:  # a hash student: the table's gradient crosses as the half-precision table the scatter wrote, the heads' in fp32
                moved, what = emb.numel() * 2 / 1e6 + (total - emb.numel() * 4 / 1e6), "the hash table's gradient in half precision as the scatter wrote it + the heads' fp32 gradients, instead"
            else:
                moved, what = (c.idx.numel() * 4 / 1e6 if compact else total), ("touched rows only" if compact else "dense")
            out["config"]["exchange"] = "%s of %.1f MB (%s of %.1f MB of fp32 gradients)%s + 16 B of loss sums%s" % (
                form, moved, what, total,
                ", next step's prefix replayed underneath" if getattr(tr, "_g_prefix", None) is not None else "",
                " between the two compositing launches (objective riding on them)" if getattr(tr, "dp_objective_rides", False) else "")
            out["config"]["exchange"] += ("; collectives recorded into the step's HIP graph (1 graph launch per step)" if dp.ingraph else
                                          "; collectives eager between %d graphs" % len(getattr(tr, "_cap").graphs))
        except Exception as e:  # noqa: BLE001
            out["config"]["exchange"] = "unknown (%s)" % type(e).__name__
        # the exchange's collective ALONE on a buffer of the step's size, every rank in step (barrier, then 20 back-to-back calls between
        # two synchronisations; MAX over ranks): what the wire costs next to ms_per_step(N) - ms_per_step(1).  Informational.
        try:
            if world > 1:
                c = getattr(w.trainer, "_compactor", None)
                compact = c is not None and getattr(c, "agreed", True) and c.fraction < 0.7
                n_el = int(c.idx.numel()) if compact else int(w.trainer.flat.flat.numel())
                emb = getattr(getattr(w.stu, "encoder", None), "embeddings", None)
                half_table = not compact and emb is not None and getattr(emb, "_pvd_half_grad_taker", None) is not None and getattr(w.trainer, "_xlayouts", None) is None
                scratch = torch.zeros(n_el, dtype=torch.float32, device=dev)
                if half_table:  # (the hash table's half-precision gradient: the bulk of a hash student's exchange)
                    n_el = int(emb.numel())
                    scratch = torch.zeros(n_el, dtype=torch.float16, device=dev)
                reps = 20
                for _ in range(3):
                    dp.all_reduce_sum_(scratch)
                dist.barrier()
                torch.cuda.synchronize()
                tx = time.perf_counter()
                for _ in range(reps):
                    dp.all_reduce_sum_(scratch)
                torch.cuda.synchronize()
                us = torch.tensor([(time.perf_counter() - tx) / reps * 1e6], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(us, op=dist.ReduceOp.MAX)
                out["config"]["exchange_alone"] = {
                    "collective": "sharded pair (reduce_scatter + all_gather)" if os.environ.get("PVD_DP_EXCHANGE", "allreduce") == "sharded" else "all_reduce",
                    "backend": backend, "bytes": n_el * scratch.element_size(), "us_per_call": float(us[0]),
                    "bus_GBps": 2.0 * (world - 1) / world * n_el * scratch.element_size() / (float(us[0]) * 1e-6) / 1e9}
        except Exception as e:  # noqa: BLE001
            out["config"]["exchange_alone"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w, args.cpu_steps, args.rays)
        out["gpu_reference"] = gpu_reference_step(opt)
    else:
        out["cpu_baseline"] = None
        out["gpu_reference"] = None
    out["psnr"] = None
    if rank == 0 and world == 1 and not args.no_psnr and not args.eager and args.student == "vm" and args.bound == 1.0:
        try:
            del w  # (its graphs and pools)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            ts, s1, s2, tot = (int(v) for v in args.psnr_schedule.split(","))
            out["psnr"] = psnr_run(dev, args.student, ts, s1, s2, tot, oracle_check=not args.no_cpu_baseline)
        except Exception as e:  # noqa: BLE001  (never lose the throughput line to the quality run)
            import traceback
            traceback.print_exc(file=sys.stderr)
            out["psnr"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST thing on stdout: the communication library writes its version banner through the C
        # runtime's buffer, which would otherwise be flushed at exit, after Python's
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
