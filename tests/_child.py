"""One test body per fresh interpreter.

Tests whose body creates an RCCL process group or captures a hipGraph run that body in a CHILD process: a watchdog-thread abort,
a runtime crash in hipStreamEndCapture or a teardown fault then costs that one test (the parent reports the child's exit status
and the tail of its output) instead of the interpreter that holds every other result -- round 4's driver run ended in SIGABRT with no
test named.  The child is a plain `python tests/_child.py <test file> <function> <json args>`: it imports the test module and
calls the function; it is started with subprocess (a new program in a child process, never an exec of the process that holds the
GPU), one at a time, so the card sees the parent plus one child.

In a test module:

    def child_sharded_optimizer(golden_dir): ...          # the body; plain asserts; os.environ instead of monkeypatch
    def test_sharded_optimizer(golden_dir):
        run_child(__file__, "child_sharded_optimizer", golden_dir)
"""
import json
import os
import signal
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def free_port():
    """a TCP port nobody listens on right now (rendezvous of a single test; no hard-coded ports shared between tests)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tail(text, n=6000):
    return text if len(text) <= n else "...[cut]...\n" + text[-n:]


def run_child(test_file, func, *args, timeout=900, env=None):
    """runs `func(*args)` of `test_file` in a fresh interpreter; fails the calling test with the child's status and output"""
    import pytest
    e = dict(os.environ)
    e.update(env or {})
    e["ORBIT2_TEST_CHILD"] = "1"
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.setdefault("PYTHONFAULTHANDLER", "1")
    cmd = [sys.executable, os.path.join(HERE, "_child.py"), os.path.abspath(test_file), func, json.dumps(list(args))]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as t:
        out = t.stdout.decode(errors="replace") if isinstance(t.stdout, bytes) else (t.stdout or "")
        err = t.stderr.decode(errors="replace") if isinstance(t.stderr, bytes) else (t.stderr or "")
        pytest.fail("child %s::%s still running after %d s (killed)\n--- stdout\n%s\n--- stderr\n%s"
                    % (os.path.basename(test_file), func, timeout, _tail(out), _tail(err)), pytrace=False)
    if r.returncode != 0:
        how = "exit code %d" % r.returncode
        if r.returncode < 0:
            try:
                how = "killed by %s" % signal.Signals(-r.returncode).name
            except ValueError:
                how = "killed by signal %d" % -r.returncode
        pytest.fail("child %s::%s: %s\n--- stdout\n%s\n--- stderr\n%s"
                    % (os.path.basename(test_file), func, how, _tail(r.stdout), _tail(r.stderr)), pytrace=False)
    sys.stdout.write(_tail(r.stdout, 2000))
    return r


def _how(code):
    """exit status in words: 'exit code 3', 'killed by SIGSEGV'"""
    if isinstance(code, int) and code < 0:
        try:
            return "killed by %s" % signal.Signals(-code).name
        except ValueError:
            return "killed by signal %d" % -code
    return "exit code %s" % code if isinstance(code, int) else str(code)


def spawn_ranks(worker, world, *args, timeout=900):
    """`worker(rank, world, port, *args, q)` in `world` spawned processes; every rank reports (rank, "ok" | text, ...) on q.  A
    rank that dies without reporting (signal, abort in a non-Python thread) is named with its exit status at once instead of
    leaving the parent waiting for the queue's timeout; the other ranks are then terminated (they would wait in a collective)."""
    import queue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res, t0, dead = [], time.time(), None
    while len(res) < world and dead is None:
        try:
            res.append(q.get(timeout=1.0))
            continue
        except queue.Empty:
            pass
        reported = {r[0] for r in res}
        for r, p in enumerate(procs):
            if r not in reported and p.exitcode not in (None, 0):
                try:                                             # its report may still be in the pipe
                    res.append(q.get(timeout=2.0))
                except queue.Empty:
                    dead = (r, p.exitcode)
                break
        if time.time() - t0 > timeout:
            dead = (-1, "no report after %d s" % timeout)
    hung = []
    for r, p in enumerate(procs):
        p.join(60 if dead is None else 5)
        if p.is_alive():
            hung.append(r)
            p.terminate()
            p.join(10)
    if dead is not None:
        r, code = dead
        raise AssertionError("rank %s ended without reporting: %s; reports so far: %r" % (r, _how(code), [x[:2] for x in res]))
    for r in res:
        assert r[1] == "ok", "rank %s:\n%s" % (r[0], "\n".join(str(x) for x in r[1:]))
    # a rank that reported "ok" and THEN died (destroy_process_group in its `finally`, interpreter or RCCL teardown) is the crash
    # class this runner exists to attribute: every exit status is examined after the joins (advisor, round 5)
    late = [(r, p.exitcode) for r, p in enumerate(procs) if r not in hung and p.exitcode != 0]
    assert not hung, "rank(s) %s reported ok but were still running 60 s later (terminated)" % hung
    assert not late, "; ".join("rank %d reported ok but exited with %s" % (r, _how(c)) for r, c in late)
    return sorted(res, key=lambda r: r[0])


def _main(argv):
    import faulthandler
    import importlib.util
    faulthandler.enable(all_threads=True)
    test_file, func, args = argv[1], argv[2], json.loads(argv[3])
    for p in (ROOT, os.path.join(ROOT, "orbit-2_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    name = os.path.splitext(os.path.basename(test_file))[0]
    spec = importlib.util.spec_from_file_location(name, test_file)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    print("[child] %s::%s%s pid %d" % (name, func, tuple(args), os.getpid()), flush=True)
    getattr(mod, func)(*args)
    print("[child] %s::%s ok" % (name, func), flush=True)
    sys.stdout.flush()
    sys.stderr.flush()


if __name__ == "__main__":
    _main(sys.argv)
