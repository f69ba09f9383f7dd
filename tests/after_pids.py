"""Run a command once the given processes have exited (test plumbing, no GPU use of its own).

    python tests/after_pids.py <pid> [<pid> ...] -- <command ...>

tests/conftest.py starts the multi-rank rehearsals as fresh processes before pytest touches the GPU; a GPU box allows at
most six processes on its card at once, so the four-rank rehearsal waits here until the two-rank children are gone.
A pid counts as gone when /proc/<pid> has disappeared or the process is a zombie (exited, not yet reaped by pytest)."""
import os
import sys
import time


def gone(pid):
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] == "Z"
    except OSError:
        return True


def main():
    sep = sys.argv.index("--")
    pids = [int(p) for p in sys.argv[1:sep]]
    t0 = time.time()
    while not all(gone(p) for p in pids):
        if time.time() - t0 > 900:
            print("after_pids: gave up waiting for %s" % pids, file=sys.stderr)
            return 3
        time.sleep(0.5)
    time.sleep(2.0)                      # (device contexts of the exited ranks are torn down asynchronously)
    # exec, not spawn: this process has never touched the GPU, and the command inherits the pid the test session holds
    # (terminate() then reaches bench.py's launcher, whose SIGTERM handler takes its ranks down)
    os.execv(sys.argv[sep + 1], sys.argv[sep + 1:])


if __name__ == "__main__":
    sys.exit(main())
