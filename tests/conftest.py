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
    # Watchdogs outside the per-test timeout (pytest.ini): a run of this suite takes 1-5 minutes; one that is still alive
    # after 30 dumps every thread's stack and exits instead of hanging the caller (seen three times in this container
    # with the output piped: no test was running, the process simply never ended).
    import faulthandler
    faulthandler.enable()
    faulthandler.dump_traceback_later(1800, exit=True)
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


_STATUS = {"exit": 0}


def pytest_sessionfinish(session, exitstatus):
    _STATUS["exit"] = int(exitstatus)


@pytest.hookimpl(trylast=True)
def pytest_unconfigure(config):
    # Everything is reported and pytest's own clean-up (tmp_path, capture) has run.  Python-level exit handlers still run
    # (whatever the caller registered with atexit included), then the process leaves WITHOUT the native finalisation
    # (destructors of the OpenMP / HIP runtimes, thread pools): that is where a run whose tests had all passed has wedged.
    # A C-level watchdog covers the exit handlers themselves.
    import atexit
    import faulthandler
    faulthandler.cancel_dump_traceback_later()
    if os.environ.get("AADFF_TEST_NORMAL_EXIT", "0") == "1":
        return
    faulthandler.dump_traceback_later(120, exit=True)
    sys.stdout.flush()
    sys.stderr.flush()
    atexit._run_exitfuncs()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(_STATUS["exit"])


def pytest_terminal_summary(terminalreporter):
    if _MARGINS:
        terminalreporter.section("parity margins (measured / tolerance)")
        for name, v, tol in _MARGINS:
            terminalreporter.write_line(f"{name:<62s} {v:.3e} / {tol:.1e}  ({100 * v / tol:5.1f} % of budget)")
