"""hipGraph capture of the whole distillation step: replays must train exactly like eager steps do."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _workload(seed=0):
    from pvd.config import PVDConfig
    from pvd.ops import hip_ops
    from pvd.workload import DistillWorkload
    opt = PVDConfig(num_rays=1024, resolution0=64, iters=300)
    return DistillWorkload(hip_ops(), torch.device("cuda:0"), opt, teacher_pretrain_steps=30, seed=seed)


def test_graph_replay_trains_and_advances_device_state():
    w = _workload()
    torch.cuda.manual_seed(7)
    before = [p.detach().clone() for p in w.stu.parameters()]
    w.enable_graph()
    lr0 = float(w.trainer.optimizer.param_groups[0]["lr"])
    losses, idx = [], []
    for _ in range(40):
        loss, info, ps, pt = w.step()
        losses.append(float(info["rgb"]))
        idx.append(int(w._batch_state[0]))
    assert all(np.isfinite(losses))
    assert idx[1] == idx[0] + 1  # the pose index lives on the device and advances inside the graph
    assert np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), losses
    assert float(w.trainer.optimizer.param_groups[0]["lr"]) < lr0  # cosine schedule reaches the captured optimizer
    after = list(w.stu.parameters())
    assert any((a - b).abs().max() > 0 for a, b in zip(after, before))
    assert ps.shape == (1, 1024, 3) and torch.isfinite(ps).all()
    # teacher untouched
    assert all(p.grad is None for p in w.tea.parameters())


def test_graph_and_eager_agree_on_the_same_batches():
    """Same seeds, same device-side batch generator: one eager step vs one replayed step from identical
    states give the same loss (float atomics in the VM backward only perturb the update, not this loss)."""
    wa, wb = _workload(3), _workload(3)
    wb.stu.load_state_dict(wa.stu.state_dict()); wb.tea.load_state_dict(wa.tea.state_dict())
    wb.stu.mean_count = wa.stu.mean_count
    torch.cuda.manual_seed(11)
    wa.enable_graph()          # runs 3 warm-up steps; the capture pass itself only records
    la = [float(wa.step()[0]) for _ in range(3)]
    torch.cuda.manual_seed(11)
    lb = []
    for _ in range(3 + 3):
        lb.append(float(wb.trainer.train_step(*wb.device_batch())[0]))
    # the first replay corresponds to eager step index 3 (after the 3 warm-up steps)
    assert np.allclose(la, lb[3:6], rtol=1e-3), (la, lb)


def test_several_steps_per_graph_launch_train_like_single_step_replays():
    """enable_graph(steps_per_graph=3): one graph launch = three consecutive optimisation steps (batch generation, schedule,
    loss-rate decay and loss scale all live on the device, so the second and third step of a replay see their own state).
    Same seeds -> the same loss trajectory as single-step replays."""
    wa, wb = _workload(5), _workload(5)
    wb.stu.load_state_dict(wa.stu.state_dict()); wb.tea.load_state_dict(wa.tea.state_dict())
    wb.stu.mean_count = wa.stu.mean_count
    torch.cuda.manual_seed(21)
    wa.enable_graph()
    la = [float(wa.step()[0]) for _ in range(6)]
    torch.cuda.manual_seed(21)
    wb.enable_graph(steps_per_graph=3)
    assert wb.steps_per_call == 3 and wa.steps_per_call == 1
    # several steps per graph: the next step's prefix rides on a forked branch, the last one feeds the next replay
    assert getattr(wb.trainer, "pipelined_ingraph", False) and not getattr(wa.trainer, "pipelined_ingraph", False)
    g0 = wb.trainer.global_step
    lb = [float(wb.step()[0]) for _ in range(2)]
    assert wb.trainer.global_step == g0 + 6 and wb.trainer.scheduler.last_epoch == wa.trainer.scheduler.last_epoch
    assert np.allclose([la[2], la[5]], lb, rtol=2e-3), (la, lb)
    assert float(wa.trainer.optimizer.param_groups[0]["lr"]) == pytest.approx(float(wb.trainer.optimizer.param_groups[0]["lr"]), rel=1e-6)


def test_two_part_update_in_the_pipelined_graph_leaves_the_single_launch_bits():
    """Multi-step graph, fork at the start of the step: AdamW runs in two parts -- behind the scatter only what the backward can
    have written; the L1-only / still-decaying rows (`_warm_A`) one step later on the forked branch, from the scalars the step
    recorded (FlatAdamW.two_part, pvd_adamw_extras.snapshot / replay).  Those rows see no atomics, so against the same run with
    the single launch (PVD_ADAMW_SPLIT=0) their parameters and both moments must agree BIT FOR BIT after several replays; the
    other rows and the loss agree to the scatter's rounding.
    Part A of step k is launched at the END of step k + 1's branch (next to the backward), the objective does not wait for it,
    and the part A of a graph's LAST step is carried to the next replay's first branch (FlatAdamW.carry_last): after the replays
    it is still owed, and flush() / an eager step runs it exactly once."""
    import os
    mode = "late"
    runs = {}
    for split in (mode, "0"):
        old = os.environ.get("PVD_ADAMW_SPLIT")
        os.environ["PVD_ADAMW_SPLIT"] = split
        try:
            w = _workload(9)
            torch.cuda.manual_seed(31)
            w.enable_graph(steps_per_graph=4)
            tr, o = w.trainer, w.trainer.optimizer
            assert getattr(tr, "pipelined_ingraph", False) and tr.pipeline_fork == "start"
            assert o._graph_is_two_part == (split != "0") and o._part_a_owed is None and not o.two_part
            losses = [float(w.step()[0]) for _ in range(3)]
            assert (o._part_a_owed is not None and o._owed_is_carried) == (split == "late")  # the last step's part A rides on the next replay
            st = o._l1_track  # the L1 value the next objective would add = the L1 term of the parameters as they are in memory now
            torch.cuda.synchronize()
            assert abs(float(st["buf"].sum()) - float(o.l1_value(st["scale"]))) <= 2e-5 * abs(float(o.l1_value(st["scale"]))), split
            a = (o._warm_A.long()[:, None] * 4 + torch.arange(4, device=o.flat_p.device)).reshape(-1)
            assert a.numel() > 10000
            o.flush()
            l1 = float(o.l1_value(1.0))
            # an eager step after the replays goes back to the single launch (and re-bases the L1 partial sums)
            le = float(tr.train_step(*w.device_batch())[0])
            # ... and a replay after that: the part A the flush ran must not be applied again by the replay's first branch
            losses.append(float(w.step()[0]))
            o.flush()
            runs[split] = (o.flat_p[a].clone(), o.flat_m[a].clone(), o.flat_v[a].clone(), losses, l1, le, float(o.step_count[0]))
        finally:
            if old is None:
                os.environ.pop("PVD_ADAMW_SPLIT", None)
            else:
                os.environ["PVD_ADAMW_SPLIT"] = old
    (pa, ma, va, la, l1a, lea, sa), (pb, mb, vb, lb, l1b, leb, sb) = runs[mode], runs["0"]
    assert sa == sb and sa >= 12  # (the same steps were applied / skipped by the loss scaler in both runs)
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert (ma != 0).any()  # (the L1 term does drive these rows)
    assert np.allclose(la, lb, rtol=2e-3) and abs(l1a - l1b) <= 1e-4 * abs(l1b) and abs(lea - leb) <= 2e-3 * abs(leb), (la, lb, l1a, l1b, lea, leb)


def test_a_forked_recording_that_fails_falls_back_to_back_to_back_steps_that_train(monkeypatch):
    """with the update in two parts by default, a forked recording that raises half way left the L1 partial sums laid
    out for two parts; the back-to-back fall-back (one launch per step, recorded without warm-up) then tripped over
    `the form of the update must be settled before a capture begins`.  Force the failure: the fall-back must record, replay
    and train like an undisturbed run, and the L1 value a replayed step adds must be the L1 term of the parameters in memory
    (the stale two-region layout over-counted it)."""
    ref = _workload(13)
    torch.cuda.manual_seed(41)
    ref.enable_graph(steps_per_graph=4)
    assert getattr(ref.trainer, "pipelined_ingraph", False) and ref.trainer.optimizer._graph_is_two_part
    lr = [float(ref.step()[0]) for _ in range(3)]
    monkeypatch.setenv("PVD_TEST_FAIL_IN_CAPTURE", "forked")
    w = _workload(13)
    torch.cuda.manual_seed(41)
    w.enable_graph(steps_per_graph=4)
    monkeypatch.delenv("PVD_TEST_FAIL_IN_CAPTURE")
    tr, o = w.trainer, w.trainer.optimizer
    assert getattr(tr, "capture_fallback", None) == "back-to-back" and not getattr(tr, "pipelined_ingraph", False)
    assert not o._graph_is_two_part and o._l1_layout == "one" and o._part_a_owed is None and not o.two_part
    lw = [float(w.step()[0]) for _ in range(3)]
    # (not the same batches: the failed recording's eager prologue -- the first replayed step's prefix -- drew one; the same training)
    assert np.all(np.isfinite(lw)) and lw[-1] < lw[0] and abs(lw[-1] - lr[-1]) <= 0.1 * lr[-1], (lw, lr)
    st = o._l1_track
    torch.cuda.synchronize()
    assert abs(float(st["buf"].sum()) - float(o.l1_value(st["scale"]))) <= 2e-5 * abs(float(o.l1_value(st["scale"])))


def test_what_the_host_believes_about_the_gradients_after_a_replay_is_what_the_graph_did():
    """Inside a multi-step graph the update zeroes the gradients it has read (touched-set / warm-list form), so only the first
    step of the graph launches a zero_grad; after a replay the touched set IS clean and the next zero_grad has nothing to
    launch.  When the update cannot take that form (here: PVD_ADAMW_LAZY=0, no warm list) every recorded step zeroes for itself
    and the host must not assume a clean buffer afterwards."""
    import os
    for lazy in ("1", "0"):
        old = os.environ.get("PVD_ADAMW_LAZY")
        os.environ["PVD_ADAMW_LAZY"] = lazy
        try:
            w = _workload(13)
            torch.cuda.manual_seed(41)
            w.enable_graph(steps_per_graph=3)
            tr, o = w.trainer, w.trainer.optimizer
            assert o._zeroed_by_step is False  # (a recording runs nothing)
            w.step()
            torch.cuda.synchronize()
            clean = not bool(o.flat_g[o.touched.idx].any())
            assert tr._graph_zeroes == (lazy == "1") and o._zeroed_by_step == (lazy == "1") and clean == (lazy == "1")
            loss = tr.train_step(*w.device_batch())[0]  # an eager step after the replay starts from zero either way ...
            assert np.isfinite(float(loss)) and o._zeroed_by_step is False
            g_eager = o.flat_g[o.touched.idx].clone()
            assert bool(g_eager.any())  # ... and leaves its gradients in place for whoever wants to look at them
            o.zero_grad()
            assert not bool(o.flat_g[o.touched.idx].any())
        finally:
            if old is None:
                os.environ.pop("PVD_ADAMW_LAZY", None)
            else:
                os.environ["PVD_ADAMW_LAZY"] = old


_GARBAGE_DURING_CAPTURE = r'''
import gc, os, sys
sys.path[:0] = [%(repo)r, %(pkg)r]
import torch
from pvd.trainer import SegmentedCapture
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
class Holder:
    pass
def make_garbage():
    h = Holder()
    h.me = h  # a reference cycle: only the cyclic collector frees it (a trainer and its captured step are one, too)
    h.cap = SegmentedCapture(dev)
    with h.cap:
        x.add_(1)
    h.cap.replay()
gc.disable()
make_garbage()
torch.cuda.synchronize()
cap = SegmentedCapture(dev)
with cap:
    x.add_(1)
    gc.collect()  # without the guard in SegmentedCapture.__enter__ the old graph is destroyed HERE, inside the capture
    x.add_(1)
cap.replay()
torch.cuda.synchronize()
print("SURVIVED %%g" %% float(x[0]), flush=True)
'''


def test_a_graph_collected_as_garbage_cannot_die_inside_a_capture():
    """Round 2's SIGABRT, pinned: on ROCm `~CUDAGraph` synchronises the device (HIPGraph.cpp), which is not permitted while
    a stream of the thread is capturing -- the error is thrown from a destructor, i.e. std::terminate.  A captured step of an
    EARLIER trainer that is unreachable but not yet collected (reference cycle) dies whenever the cyclic collector happens
    to run; if that is inside the next trainer's capture the process is gone.  SegmentedCapture therefore collects before it
    begins and keeps the collector off while recording (as torch.cuda.graph does).  The control run (PVD_CAPTURE_GC=0 = the
    old behaviour) documents the mechanism: it aborts with exactly that message."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    repo = os.path.dirname(here)
    script = _GARBAGE_DURING_CAPTURE % {"repo": repo, "pkg": os.path.join(repo, "aaai2023-pvd_amd")}
    env = {k: v for k, v in os.environ.items() if k != "PVD_CAPTURE_GC"}
    p = subprocess.run([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and b"SURVIVED 3" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    c = subprocess.run([sys.executable, "-c", script], env=dict(env, PVD_CAPTURE_GC="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    if c.returncode != 0:  # (a runtime that no longer aborts here would make the guard redundant, not wrong)
        assert c.returncode == -6 and b"stream is capturing" in c.stderr, (c.returncode, c.stderr[-2000:])
    print("control without the guard: rc=%d" % c.returncode)
