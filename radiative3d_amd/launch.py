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


def _run_once(n, argv, port, extra_env, poll_s):
    """One attempt: start the ranks, wait.  Returns (largest exit code, seconds until the first
    failure or None).  Whatever happens -- a failing rank, Ctrl-C in this process, an exception --
    no started rank is left behind: the ones still running are terminated by pid and reaped."""
    procs = []
    t0 = time.monotonic()
    first_failure = None
    try:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
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
                    if first_failure is None:
                        first_failure = time.monotonic() - t0
                    print(f"[launch] rank {r} exited with {rc}; stopping the other ranks", file=sys.stderr, flush=True)
                    for o in pending:
                        procs[o].terminate()
            if pending:
                time.sleep(poll_s)
        return worst, first_failure
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
    binds it, so another process can take it in between: when rank 0 itself is the first to fail
    within `early_s` seconds and the port is then found taken, the job is started again on a
    fresh port (up to `attempts` times; never when the caller named the port)."""
    if n < 1:
        raise ValueError("need at least one rank")
    fixed = port is not None
    worst = 1
    for attempt in range(max(1, attempts)):
        use = port if fixed else free_port()
        worst, first_failure = _run_once(n, argv, use, extra_env, poll_s)
        if worst == 0 or fixed or first_failure is None or first_failure > early_s or not _port_taken(use):
            return worst
        print(f"[launch] port {use} was taken by another process; starting the ranks again", file=sys.stderr, flush=True)
    return worst


def _port_taken(port):
    with socket.socket() as s:
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            return True
    return False
