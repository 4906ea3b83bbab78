#!/usr/bin/env python3
"""Reproduce and diagnose the 'pytest does not exit after the last test' hang.

Runs the CPU suite N times (the conftest no longer short-cuts the interpreter exit; round 2 did, with os._exit).  A run that is still alive
`--grace` seconds after pytest printed its summary line is a reproduction: every thread of the process and of its
children is listed with its kernel wait channel, state and kernel stack (/proc), then the exact PIDs are killed.
Usage: python tools/exit_hang_probe.py [--runs 10] [--grace 60] [--out build/hang]"""
import argparse
import os
import re
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def children(pid):
    out = []
    for p in os.listdir("/proc"):
        if p.isdigit():
            try:
                st = open(f"/proc/{p}/stat").read()
                ppid = int(st.rsplit(")", 1)[1].split()[1])
                if ppid == pid:
                    out.append(int(p))
                    out += children(int(p))
            except OSError:
                pass
    return out


def describe(pid):
    lines = []
    try:
        cmd = open(f"/proc/{pid}/cmdline").read().replace("\0", " ")[:200]
        state = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0]
        lines.append(f"PID {pid} state {state} cmd {cmd}")
        for t in sorted(os.listdir(f"/proc/{pid}/task"), key=int):
            base = f"/proc/{pid}/task/{t}"
            rd = lambda n: (open(f"{base}/{n}").read().strip() if os.path.exists(f"{base}/{n}") else "?")
            try:
                st = rd("stat").rsplit(")", 1)[1].split()[0]
                lines.append(f"  tid {t} comm {rd('comm'):<18s} state {st} wchan {rd('wchan'):<28s} syscall {rd('syscall')[:60]}")
                ks = rd("stack")
                if ks and ks != "?":
                    lines += ["      " + l for l in ks.splitlines()[:8]]
            except OSError as e:
                lines.append(f"  tid {t}: {e}")
        fds = []
        for fd in os.listdir(f"/proc/{pid}/fd"):
            try:
                fds.append(f"{fd}->{os.readlink(f'/proc/{pid}/fd/{fd}')}")
            except OSError:
                pass
        lines.append("  fds: " + " ".join(fds)[:1500])
    except OSError as e:
        lines.append(f"PID {pid}: {e}")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--grace", type=float, default=60.0)
    ap.add_argument("--out", default=os.path.join(REPO, "build", "hang"))
    ap.add_argument("pytest_args", nargs="*", default=["tests", "-x", "-q", "-m", "not gpu"])
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ, AADFF_TEST_NORMAL_EXIT="1", PYTHONFAULTHANDLER="1")
    hung = 0
    for r in range(a.runs):
        log = os.path.join(a.out, f"run{r}.log")
        t0 = time.monotonic()
        with open(log, "w") as f:
            p = subprocess.Popen([sys.executable, "-X", "faulthandler", "-m", "pytest"] + a.pytest_args, cwd=REPO, env=env, stdout=f, stderr=subprocess.STDOUT)
            done_at = None
            while p.poll() is None:
                time.sleep(1.0)
                if done_at is None and re.search(r"\d+ (passed|failed)", open(log).read()):
                    done_at = time.monotonic()
                if done_at is not None and time.monotonic() - done_at > a.grace:
                    hung += 1
                    rep = [f"run {r}: still alive {a.grace:.0f} s after the summary line", describe(p.pid)] + [describe(c) for c in children(p.pid)]
                    open(os.path.join(a.out, f"run{r}.hang.txt"), "w").write("\n".join(rep))
                    print("\n".join(rep), flush=True)
                    for c in children(p.pid) + [p.pid]:
                        try:
                            os.kill(c, 9)
                        except OSError:
                            pass
                    break
                if time.monotonic() - t0 > 1500:
                    p.kill()
                    break
            p.wait()
        kids = [c for c in children(os.getpid())]
        print(f"run {r}: rc {p.returncode} in {time.monotonic() - t0:.0f} s; exit took "
              f"{(time.monotonic() - done_at) if done_at else float('nan'):.1f} s after the summary; leftover children of the probe: {kids}", flush=True)
    print(f"{hung} of {a.runs} runs hung")


if __name__ == "__main__":
    main()
