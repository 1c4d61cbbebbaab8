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
