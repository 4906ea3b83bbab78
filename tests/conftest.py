import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "aberration-aware-depth-from-focus_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Watchdog outside the per-test timeout (pytest.ini): a run of this suite takes 1-5 minutes; one that is still alive
    # after 30 dumps every thread's stack and exits non-zero instead of hanging the caller.
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(1800, exit=True)
    # children that exist before the first test (plugin helpers) are not this suite's to reap
    _FOREIGN.update(_own_children())
    # The oracle is torch on the CPU.  The GPU box shows 256 logical CPUs under a 16-CPU quota and this container 8: an
    # OpenMP pool sized by the logical count stalls small ops (and has been seen to wedge a run), so size it by what is usable.
    try:
        import torch
        usable = len(os.sched_getaffinity(0))
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                usable = min(usable, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
        torch.set_num_threads(max(1, min(usable, 16)))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def repo_root():
    return REPO


# Parity margins: tests record (name, measured, tolerance); the terminal summary prints them so every run (also `-q`)
# shows how much of each tolerance is used, not only pass/fail.
_MARGINS = []


@pytest.fixture(scope="session")
def margin():
    def record(name, value, tol):
        _MARGINS.append((name, float(value), float(tol)))
        assert value <= tol, f"{name}: {value:.3e} exceeds {tol:.1e}"
    return record


_FOREIGN = set()


def _own_children():
    """PIDs whose parent is this process and that were started after the session began (exact PIDs from /proc: nothing is
    matched by name; helper processes a plugin started before the first test are left alone)."""
    me, out = os.getpid(), []
    for p in os.listdir("/proc"):
        if p.isdigit():
            try:
                if int(open(f"/proc/{p}/stat").read().rsplit(")", 1)[1].split()[1]) == me and int(p) not in _FOREIGN:
                    out.append(int(p))
            except (OSError, ValueError, IndexError):
                pass
    return out


def _reap_children(grace=5.0):
    """Leave no child behind.  What was found when 'pytest does not end after its last test' was chased (tools/exit_hang_probe.py,
    8 clean exits of 8 here; the reader of a pipe is a different matter): the multi-rank tests start multiprocessing's resource
    tracker, a helper process that INHERITS this session's stderr and lives until every process holding its pipe has gone - any
    straggling rank keeps it, and with it the caller's `2>&1 |` pipe, open after pytest itself has exited, which reads as a hang;
    and a tracker orphaned by an abrupt exit stays behind as a <defunct> child of PID 1.  So: stop the tracker the way
    multiprocessing does at interpreter shutdown (close its pipe, wait for it), then terminate and reap whatever else is still
    a child of this process - with a deadline, by exact PID."""
    import signal
    import time
    try:
        from multiprocessing import resource_tracker as rt
        tr = rt._resource_tracker
        if getattr(tr, "_fd", None) is not None:
            os.close(tr._fd)                      # the tracker exits on EOF once nobody else holds the pipe
            tr._fd = None
    except Exception:
        pass
    deadline = time.monotonic() + grace
    kids = _own_children()
    while kids and time.monotonic() < deadline:
        for pid in kids:
            try:
                os.waitpid(pid, os.WNOHANG)
            except ChildProcessError:
                pass
        time.sleep(0.05)
        kids = _own_children()
    for sig in (signal.SIGTERM, signal.SIGKILL):
        for pid in _own_children():
            try:
                os.kill(pid, sig)
            except ProcessLookupError:
                pass
        t1 = time.monotonic() + 2.0
        while _own_children() and time.monotonic() < t1:
            for pid in _own_children():
                try:
                    os.waitpid(pid, os.WNOHANG)
                except ChildProcessError:
                    pass
            time.sleep(0.05)
    try:
        tr._pid = None                            # multiprocessing's own shutdown hook must not wait for it again
    except Exception:
        pass


@pytest.hookimpl(trylast=True)
def pytest_unconfigure(config):
    # Everything is reported.  The interpreter now exits the NORMAL way (exit handlers, module teardown, native
    # destructors); the only thing added is that no child process outlives the session.  A watchdog covers the
    # finalisation: if it is still running after two minutes every thread's stack is dumped and the process ends with a
    # NON-ZERO status - a wedged exit is a failure to look at, not a pass (it does not restart or re-exec anything).
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    # The reaping touches multiprocessing's private tracker handle and ends leftover children of this process: not under
    # pytest-xdist (controller or worker: their children are xdist's), and switchable off (AADFF_TEST_NO_REAP=1).
    xdist = os.environ.get("PYTEST_XDIST_WORKER") or getattr(getattr(config, "option", None), "numprocesses", None)
    if not xdist and os.environ.get("AADFF_TEST_NO_REAP", "0") != "1":
        _reap_children()
    sys.stdout.flush()
    sys.stderr.flush()
    faulthandler.dump_traceback_later(120, exit=True)


def pytest_terminal_summary(terminalreporter):
    if _MARGINS:
        terminalreporter.section("parity margins (measured / tolerance)")
        for name, v, tol in _MARGINS:
            terminalreporter.write_line(f"{name:<62s} {v:.3e} / {tol:.1e}  ({100 * v / tol:5.1f} % of budget)")
