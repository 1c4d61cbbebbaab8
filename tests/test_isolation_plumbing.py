"""tests/conftest.py's ordering and child-interpreter isolation, exercised on the CPU with a throw-away test file: a test
that ABORTS the interpreter (what a fault inside the HIP runtime does) must cost exactly itself -- the tests before and after
it in the same file, and the files that sort before it, are all reported under their own node ids."""
import os
import re
import shutil
import subprocess
import sys
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, extra_args=()):
    shutil.copy(os.path.join(HERE, "conftest.py"), tmp_path / "conftest.py")
    (tmp_path / "test_iso_sample.py").write_text(textwrap.dedent('''
        import os
        import pytest
        pytestmark = pytest.mark.gpu

        def test_a_passes():
            assert os.environ.get("PVD_TEST_CHILD") == "1"  # runs in the child interpreter

        def test_b_aborts():
            os.abort()

        @pytest.mark.parametrize("v", [1, 2])
        def test_c_after_the_abort(v):
            assert v in (1, 2)

        def test_d_fails():
            assert 1 == 2, "plain failure"

        def test_e_skips():
            pytest.skip("nothing to do")
    '''))
    (tmp_path / "test_hip_parity.py").write_text(textwrap.dedent('''
        import os
        import pytest
        pytestmark = pytest.mark.gpu

        def test_parity_runs_first_and_in_process():
            assert os.environ.get("PVD_TEST_CHILD") != "1"
    '''))
    (tmp_path / "test_aaa_sorts_first_by_name.py").write_text(textwrap.dedent('''
        import pytest
        pytestmark = pytest.mark.gpu

        def test_unlisted_file():
            pass
    '''))
    env = dict(os.environ, PVD_TEST_ISOLATE_EXTRA="test_iso_sample.py")
    env.pop("PVD_TEST_CHILD", None)
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-p", "no:cacheprovider", "-v", "--rootdir", str(tmp_path), *extra_args, str(tmp_path)]
    p = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    return p.returncode, p.stdout.decode()


def test_an_abort_costs_one_test_and_the_order_puts_parity_first(tmp_path):
    rc, out = _run(tmp_path)
    assert rc == 1, out
    verdicts = re.findall(r"^(\S+::\S+) (PASSED|FAILED|SKIPPED)", out, flags=re.M)
    names = [n.split("::")[1] for n, _ in verdicts]
    assert names == ["test_parity_runs_first_and_in_process", "test_unlisted_file", "test_a_passes", "test_b_aborts", "test_c_after_the_abort[1]",
                     "test_c_after_the_abort[2]", "test_d_fails", "test_e_skips"], out
    assert [v for _, v in verdicts] == ["PASSED", "PASSED", "PASSED", "FAILED", "PASSED", "PASSED", "FAILED", "SKIPPED"], out
    assert "child interpreter ended with rc=-6" in out and "plain failure" in out
    assert re.search(r"2 failed, 5 passed, 1 skipped", out), out


def test_dash_x_stops_at_the_abort_with_the_parity_results_already_reported(tmp_path):
    rc, out = _run(tmp_path, ("-x",))
    assert rc == 1
    verdicts = re.findall(r"^(\S+::\S+) (PASSED|FAILED|SKIPPED)", out, flags=re.M)
    assert [v for _, v in verdicts] == ["PASSED", "PASSED", "PASSED", "FAILED"], out
