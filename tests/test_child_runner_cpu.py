"""The child-process test runner (tests/_child.py) that isolates RCCL / hipGraph test bodies on the GPU box, exercised on the CPU:
a passing body, a failing assertion, a body that dies from a signal in the interpreter (the case round 4's driver run could not
attribute), a body that outlives its timeout; and the rank spawner's early detection of a rank that dies without reporting."""
import os
import signal
import time

import pytest

from tests._child import run_child, spawn_ranks


def child_ok(a, b):
    assert a + b == 5
    print("computed", a + b)


def child_assert():
    assert 1 + 1 == 3, "arithmetic"


def child_abort():
    os.kill(os.getpid(), signal.SIGABRT)          # what std::terminate in a non-Python thread looks like from outside


def child_sleep():
    time.sleep(60)


def test_child_passes_and_relays_its_output(capsys):
    run_child(__file__, "child_ok", 2, 3)
    assert "computed 5" in capsys.readouterr().out


def test_child_assertion_failure_is_reported_with_its_traceback():
    with pytest.raises(pytest.fail.Exception) as e:
        run_child(__file__, "child_assert")
    assert "exit code 1" in str(e.value) and "arithmetic" in str(e.value)


def test_child_killed_by_a_signal_is_named():
    with pytest.raises(pytest.fail.Exception) as e:
        run_child(__file__, "child_abort")
    assert "killed by SIGABRT" in str(e.value) and "child_abort" in str(e.value)


def test_child_timeout_is_reported():
    with pytest.raises(pytest.fail.Exception) as e:
        run_child(__file__, "child_sleep", timeout=3)
    assert "still running after 3 s" in str(e.value)


def _rank_worker(rank, world, port, mode, q):
    if mode == "ok":
        q.put((rank, "ok", rank * 10))
    elif mode == "one_dies" and rank == 1:
        os.kill(os.getpid(), signal.SIGKILL)      # dies without reporting
    elif mode == "one_fails" and rank == 0:
        q.put((rank, "fail", "Traceback: boom"))
    elif mode == "dies_in_teardown":
        q.put((rank, "ok"))
        if rank == 1:                             # reports first, THEN aborts (a fault in destroy_process_group / interpreter exit)
            time.sleep(0.5)                       # (lets the queue's feeder thread hand the report over)
            os.kill(os.getpid(), signal.SIGABRT)
    else:
        time.sleep(2.0)
        q.put((rank, "ok"))


def test_spawn_ranks_collects_reports_in_rank_order():
    res = spawn_ranks(_rank_worker, 2, "ok", timeout=60)
    assert [r[0] for r in res] == [0, 1] and res[1][2] == 10


def test_spawn_ranks_names_a_rank_that_died_silently():
    t0 = time.time()
    with pytest.raises(AssertionError) as e:
        spawn_ranks(_rank_worker, 2, "one_dies", timeout=120)
    assert "rank 1 ended without reporting" in str(e.value) and "SIGKILL" in str(e.value)
    assert time.time() - t0 < 60                   # noticed at once, not after the queue's timeout


def test_spawn_ranks_relays_a_reported_failure():
    with pytest.raises(AssertionError) as e:
        spawn_ranks(_rank_worker, 2, "one_fails", timeout=60)
    assert "rank 0" in str(e.value) and "boom" in str(e.value)


def test_spawn_ranks_names_a_rank_that_died_after_reporting_ok():
    """advisor, round 5: a rank that reports ok and then dies in its teardown used to be joined without a look at its exit status"""
    with pytest.raises(AssertionError) as e:
        spawn_ranks(_rank_worker, 2, "dies_in_teardown", timeout=60)
    assert "rank 1 reported ok but exited with killed by SIGABRT" in str(e.value)
