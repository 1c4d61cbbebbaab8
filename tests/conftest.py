"""Test-suite plumbing.

Two things beyond the marker registration, both there so that one bad test cannot hide the rest of the GPU suite
(round 2 lost every oracle-vs-HIP result to a single abort in a graph-capture test that happened to sort first):

* ORDER.  `-m gpu` runs the oracle parity files first (kernel level, then step / workload level), everything that records
  hipGraphs last.  `pytest -x` therefore stops, if it must, after the parity evidence is on record.
* ISOLATION.  The files that record hipGraphs (`ISOLATED`) run in a child interpreter, one child per file; the parent
  reports every test of the file under its own node id from the child's per-test records.  A child that dies (SIGABRT
  inside the HIP runtime cannot be caught in-process) costs exactly the test that was running -- it is reported as failed
  with the tail of the child's output -- and a fresh child picks up the remaining tests of the file.
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "aaai2023-pvd_amd")
for p in (REPO, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

# kernel-level oracle parity, then step / workload parity, then behaviour; graph recorders last (and isolated)
ORDER = [
    "test_hip_reference_kernels.py", "test_hip_parity.py", "test_hip_edges.py", "test_hip_golden.py", "test_hip_vm.py", "test_hip_plenoxel.py", "test_hip_occupancy.py",
    "test_hip_head.py", "test_hip_mlp_frozen.py", "test_hip_infer_rounds.py",
    "test_hip_golden_step.py", "test_hip_fullsize.py", "test_hip_render_parity.py", "test_hip_workloads.py",
    "test_hip_fused_misc.py", "test_hip_bench_line.py",
]
ISOLATED = ["test_hip_amp_parity.py", "test_hip_budget.py", "test_hip_graph.py", "test_hip_dp_graph.py"]
ISOLATED += [n for n in os.environ.get("PVD_TEST_ISOLATE_EXTRA", "").split(",") if n]  # (tests/test_isolation_plumbing.py)
_CHILD = os.environ.get("PVD_TEST_CHILD") == "1"
_CHILD_TIMEOUT_S = 900


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        # the binding settles GPU_MAX_HW_QUEUES when it is imported, which has to happen before anything touches the device
        # (pvd_hip/__init__.py); test modules import torch first and the binding lazily, so do it here, once
        import pvd_hip  # noqa: F401
    except Exception:  # no library built yet (CPU-only checkout): the tests that need it say so themselves
        pass
    config._pvd_isolated = {}  # file basename -> {"todo": [nodeids], "done": {nodeid: [records]}}


def _rank(item):
    name = os.path.basename(str(item.fspath))
    if name in ORDER:
        return (0, ORDER.index(name))
    if name in ISOLATED:
        return (2, ISOLATED.index(name))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_rank)  # stable: the order inside a file is kept
    if _CHILD:
        return
    for it in items:
        name = os.path.basename(str(it.fspath))
        if name in ISOLATED and it.get_closest_marker("gpu") is not None:
            config._pvd_isolated.setdefault(name, {"todo": [], "done": {}})["todo"].append(it.nodeid)


# ---------------------------------------------------------------- child side: one JSON record per test phase
def pytest_runtest_logreport(report):
    path = os.environ.get("PVD_TEST_CHILD_REPORT")
    if not (_CHILD and path):
        return
    rec = {"nodeid": report.nodeid, "when": report.when, "outcome": report.outcome, "duration": float(getattr(report, "duration", 0.0)),
           "longrepr": str(report.longrepr) if report.longrepr is not None and report.outcome != "passed" else ""}
    with open(path, "a") as f:
        f.write(json.dumps(rec) + "\n")
        f.flush()
        os.fsync(f.fileno())


def pytest_runtest_logstart(nodeid, location):
    path = os.environ.get("PVD_TEST_CHILD_REPORT")
    if _CHILD and path:
        with open(path, "a") as f:
            f.write(json.dumps({"nodeid": nodeid, "when": "start"}) + "\n")
            f.flush()
            os.fsync(f.fileno())


# ---------------------------------------------------------------- parent side
def _run_children(config, state):
    """Run state["todo"] in child interpreters until every node id has a record (a crash consumes one test)."""
    while state["todo"]:
        fd, rep_path = tempfile.mkstemp(prefix="pvd_child_", suffix=".jsonl")
        os.close(fd)
        env = dict(os.environ, PVD_TEST_CHILD="1", PVD_TEST_CHILD_REPORT=rep_path)
        env.setdefault("AMD_LOG_LEVEL", "1")  # errors only: if the HIP runtime takes the child down, its last words are in the report
        cmd = [sys.executable, "-X", "faulthandler", "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--rootdir", str(config.rootpath)]
        cmd += state["todo"]
        t0 = time.time()
        try:
            proc = subprocess.run(cmd, env=env, cwd=str(config.rootpath), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=_CHILD_TIMEOUT_S)
            rc, out = proc.returncode, proc.stdout.decode(errors="replace")
        except subprocess.TimeoutExpired as e:
            rc, out = -999, (e.stdout or b"").decode(errors="replace") + "\n[child killed after %d s]" % _CHILD_TIMEOUT_S
        records, started = {}, None
        with open(rep_path) as f:
            for line in f:
                try:
                    r = json.loads(line)
                except ValueError:
                    continue
                if r["when"] == "start":
                    started = r["nodeid"]
                else:
                    records.setdefault(r["nodeid"], []).append(r)
        os.unlink(rep_path)
        finished = {n for n, rs in records.items() if any(r["when"] == "teardown" for r in rs)}
        for n in finished:
            state["done"][n] = records[n]
        remaining = [n for n in state["todo"] if n not in finished]
        if remaining and (rc not in (0, 1) or len(remaining) == len(state["todo"])):
            # the child died (or ran nothing): charge the test that was running -- or, failing that, the first one left
            victim = started if started in remaining else remaining[0]
            cut = out.find("Fatal Python error")
            tail = out[-6000:] if cut < 0 else out[max(0, cut - 2500):cut + 3500]  # what was printed BEFORE the interpreter's own dump matters most
            state["done"][victim] = records.get(victim, []) + [{
                "nodeid": victim, "when": "call", "outcome": "failed", "duration": time.time() - t0,
                "longrepr": "child interpreter ended with rc=%s while this test was running; tail of its output:\n%s" % (rc, tail)}]
            remaining = [n for n in remaining if n != victim]
        elif remaining:  # (a child that returned normally but skipped node ids: -x is not passed, so this is deselection)
            for n in remaining:
                state["done"][n] = [{"nodeid": n, "when": "setup", "outcome": "skipped", "duration": 0.0,
                                     "longrepr": "('%s', 0, 'not run by the child interpreter')" % n}]
            remaining = []
        state["todo"] = remaining


def pytest_runtest_protocol(item, nextitem):
    if _CHILD:
        return None
    name = os.path.basename(str(item.fspath))
    state = item.config._pvd_isolated.get(name)
    if state is None or item.get_closest_marker("gpu") is None:
        return None
    if item.nodeid not in state["done"]:
        _run_children(item.config, state)
    recs = state["done"].get(item.nodeid) or [{"when": "call", "outcome": "failed", "duration": 0.0, "longrepr": "no record from the child interpreter"}]
    from _pytest.reports import TestReport
    item.ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
    whens = {r["when"]: r for r in recs if r["when"] in ("setup", "call", "teardown")}
    if "setup" not in whens:
        whens["setup"] = {"when": "setup", "outcome": "passed", "duration": 0.0, "longrepr": ""}
    if "teardown" not in whens:
        whens["teardown"] = {"when": "teardown", "outcome": "passed", "duration": 0.0, "longrepr": ""}
    for when in ("setup", "call", "teardown"):
        r = whens.get(when)
        if r is None:
            continue
        longrepr = None
        if r["outcome"] == "skipped":
            longrepr = (str(item.fspath), 0, "Skipped: " + (r.get("longrepr") or ""))
        elif r["outcome"] != "passed":
            longrepr = r.get("longrepr") or "failed in the child interpreter"
        rep = TestReport(nodeid=item.nodeid, location=item.location, keywords={k: 1 for k in item.keywords}, outcome=r["outcome"],
                         longrepr=longrepr, when=when, duration=r.get("duration", 0.0))
        item.ihook.pytest_runtest_logreport(report=rep)
    item.ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
    return True


@pytest.fixture(scope="session")
def hip_lib_built():
    """The HIP library must exist in-tree (it cross-compiles without a GPU)."""
    so = os.path.join(PKG, "libpvd_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g

        g.build()
    assert os.path.exists(so)
    return so
