"""One process per GPU without an external launcher.

The reference runs replicas as separate processes and sums their outputs afterwards
(scripts/do-parallel.sh:23-29, vis/seisplot/combine.m:26-33).  Here the replicas are the
ranks of one torch.distributed job; `spawn_ranks` starts them when the program was called
directly (``python bench.py --gpus 8``) rather than through ``torch.distributed.run``.

The parent never touches the GPU (it does not even import torch): the ranks are fresh child
processes started with the launcher's environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
MASTER_PORT), so nothing is re-exec'ed after a HIP call.  Rank 0 inherits stdout (its JSON line
is the program's output); every rank inherits stderr.
"""
import os
import socket
import subprocess
import sys
import time


def under_launcher(env=None):
    """True when this process is one rank of an already launched job."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env and "MASTER_PORT" in env


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_ADDR_IN_USE = ("EADDRINUSE", "Address already in use", "address already in use")


class _Tee:
    """Rank 0's stderr, passed on line by line as it comes and its tail kept: the one thing the
    parent reads of a rank's output -- whether rank 0 said it could not bind the rendezvous port."""

    def __init__(self, pipe, keep=1 << 16):
        import threading
        self.pipe, self.keep, self.tail = pipe, keep, b""
        self.thread = threading.Thread(target=self._pump, daemon=True)
        self.thread.start()

    def _pump(self):
        out = getattr(sys.stderr, "buffer", None)
        for line in iter(self.pipe.readline, b""):
            self.tail = (self.tail + line)[-self.keep:]
            if out is not None:
                out.write(line), out.flush()
            else:   # (a replaced sys.stderr without a byte stream, e.g. under a test runner)
                sys.stderr.write(line.decode(errors="replace")), sys.stderr.flush()
        self.pipe.close()

    def text(self):
        self.thread.join(timeout=5)
        return self.tail.decode(errors="replace")


def _run_once(n, argv, port, extra_env, poll_s):
    """One attempt: start the ranks, wait.  Returns (largest exit code, the rank that failed first or
    None, seconds until that failure or None, the tail of rank 0's stderr).  Whatever happens -- a
    failing rank, Ctrl-C in this process, an exception -- no started rank is left behind: the ones
    still running are terminated by pid and reaped."""
    procs = []
    tee = None
    t0 = time.monotonic()
    first_failed, first_failure = None, None
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if r == 0 else None))
            if r == 0:
                tee = _Tee(procs[0].stderr)
        worst = 0
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                worst = max(worst, abs(rc))
                if rc != 0:
                    if first_failed is None:
                        first_failed, first_failure = r, time.monotonic() - t0
                    print(f"[launch] rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                    for o in pending:
                        procs[o].terminate()
            if pending:
                time.sleep(poll_s)
        return worst, first_failed, first_failure, tee.text() if tee else ""
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()


def spawn_ranks(n, argv, extra_env=None, poll_s=0.2, port=None, attempts=3, early_s=20.0):
    """Run `argv` (a full command line) as ranks 0..n-1 of one job on this node and wait.

    Returns the largest exit code.  If a rank fails, the ranks still running are terminated
    (by pid -- the processes started here, nothing else) so that a dead peer cannot leave the
    others waiting in a collective.  The rendezvous port is probed and released before rank 0
    binds it, so another process can take it in between.  The job is started again on a fresh
    port (up to `attempts` times; never when the caller named the port) only in exactly that
    case: RANK 0 was the first to fail, within `early_s` seconds, its stderr says the address
    was in use, AND a listener still holds the port.  Any other failure -- another rank first, a
    crash after the rendezvous (engine error, out of memory, a GPU fault, an assertion) -- is
    returned as it is: a rerun could hide a flaky crash behind a passing second attempt."""
    if n < 1:
        raise ValueError("need at least one rank")
    fixed = port is not None
    worst = 1
    for attempt in range(max(1, attempts)):
        use = port if fixed else free_port()
        worst, first_failed, first_failure, err0 = _run_once(n, argv, use, extra_env, poll_s)
        bind_failed = (first_failed == 0 and first_failure is not None and first_failure <= early_s
                       and any(m in err0 for m in _ADDR_IN_USE) and _port_taken(use))
        if worst == 0 or fixed or not bind_failed:
            return worst
        print(f"[launch] port {use} was taken by another process; starting the ranks again", file=sys.stderr, flush=True)
    return worst


def _port_taken(port):
    """True when some socket is LISTENING on the port.  (SO_REUSEADDR: sockets of a finished rank 0 in
    TIME_WAIT on that port do not count -- without it every crash after the rendezvous looked like a
    taken port.)"""
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            return True
    return False
