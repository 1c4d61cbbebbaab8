"""bench.py's contract, end to end, on the one GPU of the box: the JSON line of the single-GPU run (metric, config, roofline
and -- when asked for -- cpu_baseline objects), and the N > 1 launch exactly as the driver issues it
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`) with both ranks sharing the GPU
over gloo (PVD_DIST_BACKEND=gloo): ray-DP capture, gradient exchange, max-over-ranks timing, weak and strong scaling.
Short runs of a small batch -- what is checked is the plumbing and the line, not the number."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(cmd, env=None, timeout=600):
    p = subprocess.run(cmd, cwd=REPO, env=dict(os.environ, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, (p.returncode, out[-2000:], p.stderr.decode(errors="replace")[-4000:])
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]  # ONE JSON line (rank 0 only)
    return json.loads(lines[0])


COMMON = ["--steps", "10", "--warmup", "5", "--rays", "1024", "--teacher-pretrain", "20", "--sustained-steps", "40"]


def test_single_gpu_line_carries_the_contract():
    d = _line([sys.executable, "bench.py", *COMMON, "--cpu-steps", "1", "--psnr-schedule", "300,40,80,200"])
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "rays/s" and d["vs_baseline"] is None and d["data"].startswith("synthetic")
    assert abs(d["value"] - 10 * 1024 / (d["ms_per_step"] * 10 / 1e3)) <= 1e-6 * d["value"]
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg and cfg["capture_fallback"] is False and cfg["launch"].startswith("hipGraph replay")
    assert math.isfinite(cfg["loss"]) and cfg["samples_per_step_per_gpu"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["bytes_per_sample"] == 516  # SURVEY section 8(d)'s algorithmic figure for the f16 lookup
    assert abs(r["achieved"] - r["bytes_per_sample"] * r["samples_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    assert r["alone"]["us_per_launch"] > 0 and r["alone"]["launches"] >= 20
    # the headline figure is the kernel's duration INSIDE the replayed step, stamped by the launch itself (live, not quoted)
    assert r["in_step"]["timing"].startswith("LIVE") and r["in_step"]["launches"] >= 20 and r["in_step"]["us_per_launch"] > 0
    assert abs(r["us_per_launch"] - r["in_step"]["us_per_launch"]) < 1e-9 and "live" in r["where"]
    # the lookup's speed of light on the same rows (gather-only probe), and the product kernel's fraction of it
    assert "error" not in r["sol"], r["sol"]
    assert 0 < r["sol"]["frac"] < 1 and r["sol"]["us_per_launch"] > 0 and 0 < r["frac_of_sol"] < 1.5
    assert abs(r["alone"]["frac_of_sol"] - r["sol"]["us_per_launch"] / r["alone"]["us_per_launch"]) < 1e-9
    assert 0.01 < cfg["occupied_fraction"] < 0.2 and cfg["samples_per_ray"] > 1 and cfg["samples_per_s"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "rays/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    gr = d["gpu_reference"]  # the reference's own kernels + PyTorch on this GPU (None when oracle/_ref did not travel)
    assert gr is None or ("error" not in gr and 0 < gr["value"] < d["value"]), gr
    su = d["sustained"]  # a second, longer synchronised window behind the timed one
    assert su["steps"] == 40 and su["ms_per_step"] > 0 and abs(su["vs_timed"] - su["ms_per_step"] / d["ms_per_step"]) < 1e-9
    q = d["psnr"]  # the metric's second half, outside the timed region: staged distillation run + held-out views
    assert "error" not in q, q
    assert q["steps"] >= 200 and all(math.isfinite(q[k]) for k in ("student_vs_teacher_heldout_db", "student_vs_gt_db", "teacher_vs_gt_db"))
    assert q["hip_vs_oracle_same_rays_db"] > 80.0, q["hip_vs_oracle_detail"]  # render-level parity: same rays, HIP vs CPU oracle, fp32


def test_teacher_workload_line_has_a_cpu_baseline():
    d = _line([sys.executable, "bench.py", "--workload", "teacher", "--steps", "16", "--warmup", "32", "--rays", "1024", "--cpu-steps", "2"])
    assert d["metric"].startswith("train rays/s (hash teacher") and d["roofline"]["bytes_per_sample"] == 516
    c = d["cpu_baseline"]
    assert c and "error" not in c and c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1


def test_teacher_workload_from_a_blender_scene_on_disk(tmp_path):
    """configs[1] fed by pvd/provider.py (the reference's NeRFDataset, provider.py:133-326) from a scene written by
    tools/make_blender_scene.py: transforms_train.json + RGBA PNGs -> rays, alpha-blended targets -> the same training block."""
    root = str(tmp_path / "chair")
    p = subprocess.run([sys.executable, "tools/make_blender_scene.py", root, "--views", "8", "--res", "64"], cwd=REPO, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    assert os.path.exists(os.path.join(root, "transforms_train.json")) and os.path.exists(os.path.join(root, "train", "r_0.png"))
    d = _line([sys.executable, "bench.py", "--workload", "teacher", "--steps", "16", "--warmup", "32", "--rays", "1024", "--no-cpu-baseline",
               "--data-root", root])
    assert d["data"].startswith("Blender-format scene read by pvd/provider.py") and "8 train views of 64x64" in d["data"]
    assert d["value"] > 0 and math.isfinite(d["config"]["loss"]) and d["config"]["psnr_vs_analytic_gt_db"] > 5.0


@pytest.mark.parametrize("strong", [False, True])
def test_two_ranks_as_the_driver_launches_them(strong):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", *COMMON] + (["--strong"] if strong else [])
    d = _line(cmd, env={"PVD_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == ("strong" if strong else "weak")
    per_gpu = 512 if strong else 1024
    assert d["config"]["rays_per_gpu"] == per_gpu and d["config"]["parallelism"] == "ray-dp2"
    total = 10 * (1024 if strong else 2048)  # whole-job rays over the timed steps
    assert abs(d["value"] - total / (d["ms_per_step"] * 10 / 1e3)) <= 1e-6 * d["value"]
    assert "all-reduce" in d["config"]["exchange"] and math.isfinite(d["config"]["loss"]) and d["config"]["capture_fallback"] is False
    assert d["cpu_baseline"] is None and d["psnr"] is None  # rank 0 at N = 1 only
    assert d["sustained"]["steps"] == 40
    xa = d["config"]["exchange_alone"]  # the collective alone on a buffer of the step's size (MAX over ranks): the wire's share of a step
    assert "error" not in xa, xa
    assert xa["collective"] == "all_reduce" and xa["backend"] == "gloo" and xa["bytes"] > 0 and xa["us_per_call"] > 0 and xa["bus_GBps"] > 0


def test_gpus_flag_alone_spawns_the_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the way the driver starts `--gpus 1`): bench.py starts its two ranks itself
    through torch.distributed.run and rank 0 prints the ONE line (VERDICT r5 missing #2: this used to die on an assertion)."""
    env = {"PVD_DIST_BACKEND": "gloo"}
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *COMMON], cwd=REPO, env=dict(clean, **env), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, (p.returncode, out[-2000:], p.stderr.decode(errors="replace")[-4000:])
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "ray-dp2" and d["scaling"] == "weak" and d["value"] > 0


@pytest.mark.parametrize("name,extra", [
    ("configs[3] mlp->tensors, llff cameras", ["--teacher", "mlp", "--student", "tensors", "--data-type", "llff", "--teacher-pretrain", "0"]),
    ("configs[4] hash->hash, bound 2, tank cameras", ["--student", "hash", "--data-type", "tank", "--bound", "2", "--dt-gamma", "0.00390625", "--scene-scale", "1.9"]),
])
def test_the_eight_gpu_configurations_in_their_two_rank_form(name, extra):
    """BASELINE configs[3] and configs[4] are quoted on 8 GPUs with ray-DP: their N > 1 recording (exchange of the Plenoxel volume's /
    the hash table's gradient -- the latter arrives in half precision and is widened for the exchange --, loss sums, MAX over ranks,
    one line) on two ranks sharing the GPU over gloo, as the driver launches them."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", *COMMON, *extra]
    d = _line(cmd, env={"PVD_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "ray-dp2" and d["scaling"] == "weak", name
    assert d["value"] > 0 and math.isfinite(d["config"]["loss"]) and d["config"]["capture_fallback"] is False, name
    assert "error" not in d["config"]["exchange_alone"] and d["config"]["exchange_alone"]["bytes"] > 0, name
    assert abs(d["value"] - 10 * 2048 / (d["ms_per_step"] * 10 / 1e3)) <= 1e-6 * d["value"]
