"""Runs LAST (file name): a rank whose rendezvous cannot complete.  Kept out of tests/test_dist_gloo.py on purpose - in this build
container the CPU-heavy oracle tests that followed it in collection order ran 20-50x slower about two times in three after a child
process had failed a torch.distributed rendezvous (any such child, also a bare `dist.init_process_group` without this package; a
child that just sleeps does not do it; cause not found) - with nothing behind it the suite's run time is what it was."""
import os

from aadff import dist as adist  # noqa: F401


def test_init_timeout_ends_a_rank_with_a_message_and_a_status(repo_root):
    """A rank whose peers never arrive must not hang the launcher: init_from_env gives up after AADFF_INIT_TIMEOUT_S with a message
    naming the rendezvous address and exits with aadff.dist.INIT_EXIT_CODE (the process exits; it never re-executes itself)."""
    import subprocess
    import sys
    import time
    from aadff import dist as adist
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from aadff import dist as d\n"
            "d.init_from_env(backend='gloo')\n"
            "print('unexpectedly up')\n") % (repo_root, os.path.join(repo_root, "aberration-aware-depth-from-focus_amd"))
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(adist.free_port()),
               AADFF_INIT_TIMEOUT_S="3")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == adist.INIT_EXIT_CODE, (p.returncode, p.stderr[-400:])
    assert "MASTER_ADDR=127.0.0.1" in p.stderr and "unexpectedly up" not in p.stdout
    assert time.monotonic() - t0 < 60
