import faulthandler
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "orbit-2_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Attribution of a crash (round 4's driver run died with SIGABRT and named no test): every test's node id goes, flushed, to
# file descriptor 1 BEFORE the test starts -- so the last such line of the run's output is the test that was running -- and to a
# side file under gpurun_out/ with a wall-clock stamp.  The interpreter's fault dump (all threads) goes to a second side file, not
# to the terminal: with its list of extension modules it is ~5 KB, enough to push the test's own frames (and the node id) out of
# the tail a harness keeps.  pytest's own faulthandler plugin is switched off in pytest.ini for the same reason.
_PROGRESS_DIR = os.environ.get("ORBIT2_TEST_LOG_DIR", os.path.join(ROOT, "gpurun_out"))
_progress = None
_fault = None


def _open_side_files():
    global _progress, _fault
    if _progress is not None:
        return
    try:
        os.makedirs(_PROGRESS_DIR, exist_ok=True)
        _progress = open(os.path.join(_PROGRESS_DIR, "pytest_nodeids.log"), "a", buffering=1)
        _fault = open(os.path.join(_PROGRESS_DIR, "pytest_fault.log"), "a", buffering=1)
        faulthandler.enable(file=_fault, all_threads=True)
    except OSError:                                  # read-only tree: stdout still names the test, and the fault dump goes to
        _progress = False                            # stderr (pytest's own faulthandler plugin is off: without this a crash of the
        faulthandler.enable(all_threads=True)        # parent would leave no Python traceback anywhere)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _open_side_files()
    if _progress:
        head = "==== session %s pid %d args %s\n" % (time.strftime("%Y-%m-%d %H:%M:%S"), os.getpid(), " ".join(sys.argv[1:]))
        _progress.write(head)
        if _fault:
            _fault.write(head)                       # (append mode: a dump is then attributable to its session)


def pytest_runtest_logstart(nodeid, location):
    line = "\n[test] %s\n" % nodeid
    try:
        os.write(1, line.encode())                   # the process's real stdout, whatever pytest's capture holds
    except OSError:
        pass
    if _progress:
        _progress.write("%s start %s\n" % (time.strftime("%H:%M:%S"), nodeid))


def pytest_runtest_logreport(report):
    if _progress and (report.when == "call" or (report.when == "setup" and report.outcome != "passed")):
        _progress.write("%s %-7s %s (%.1f s)\n" % (time.strftime("%H:%M:%S"), report.outcome, report.nodeid, report.duration))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
