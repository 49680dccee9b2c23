"""bench.py --gpus N as the driver calls it (no launcher environment): the program must start N
ranks itself and form one process group of that size (the reference's replicas:
scripts/do-parallel.sh:23-29).  --rendezvous-only runs exactly that part on CPU (gloo)."""
import json
import os
import subprocess
import sys

from radiative3d_amd import launch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_gpus_2_spawns_two_ranks_and_reports_the_group_size():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2


def test_world_size_must_match_gpus_under_a_launcher():
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(launch.free_port()))
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "--gpus 2" in out.stderr


def test_a_failing_rank_stops_the_job():
    prog = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
            "if r == 1: sys.exit(7)\n"
            "time.sleep(60)\n")
    rc = launch.spawn_ranks(3, [sys.executable, "-c", prog])
    assert rc != 0


def test_a_rendezvous_port_taken_in_between_is_retried_on_a_fresh_one(monkeypatch):
    """free_port() releases its probe socket before rank 0 binds MASTER_PORT; if another process takes
    the port in between, rank 0 fails at once -- spawn_ranks then starts the job again on a fresh port
    (never when the caller fixed the port)."""
    import socket
    taken = socket.socket()
    taken.bind(("127.0.0.1", 0))
    taken.listen(1)
    busy = taken.getsockname()[1]
    handed_out = []
    real_free_port = launch.free_port

    def fake_free_port():
        handed_out.append(busy if not handed_out else real_free_port())
        return handed_out[-1]

    monkeypatch.setattr(launch, "free_port", fake_free_port)
    prog = ("import os, socket, sys\n"
            "if os.environ['RANK'] == '0':\n"
            "    s = socket.socket()\n"
            "    try:\n"
            "        s.bind(('127.0.0.1', int(os.environ['MASTER_PORT'])))\n"
            "    except OSError as exc:\n"
            "        print(exc, file=sys.stderr)\n"   # 'Address already in use', as torch's TCPStore reports it
            "        sys.exit(1)\n")
    try:
        assert launch.spawn_ranks(2, [sys.executable, "-c", prog]) == 0
        assert len(handed_out) == 2 and handed_out[0] == busy and handed_out[1] != busy
        assert launch.spawn_ranks(2, [sys.executable, "-c", prog], port=busy) != 0     # a named port is not replaced
    finally:
        taken.close()


def test_an_early_crash_of_rank_0_is_not_taken_for_a_port_collision(monkeypatch):
    """Only rank 0 failing to BIND is retried.  A crash of rank 0 after it bound the port (its sockets
    then sit in TIME_WAIT), a crash that says nothing about the address, and a failure of another rank
    first are returned as they are, after ONE attempt."""
    ports = []
    real_free_port = launch.free_port

    def counting_free_port():
        ports.append(real_free_port())
        return ports[-1]

    monkeypatch.setattr(launch, "free_port", counting_free_port)
    crash_after_bind = ("import os, socket, sys\n"
                        "if os.environ['RANK'] == '0':\n"
                        "    s = socket.socket(); s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)\n"
                        "    s.bind(('127.0.0.1', int(os.environ['MASTER_PORT']))); s.listen(1)\n"
                        "    c = socket.create_connection(('127.0.0.1', int(os.environ['MASTER_PORT'])))\n"
                        "    a, _ = s.accept(); a.close(); c.close(); s.close()\n"   # leaves TIME_WAIT behind
                        "    print('engine create failed: Address already in use (not really)', file=sys.stderr)\n"
                        "    sys.exit(3)\n")
    assert launch.spawn_ranks(2, [sys.executable, "-c", crash_after_bind]) == 3
    assert len(ports) == 1                       # the port is free again (nobody listens): no second start
    silent = "import os, sys\nsys.exit(5 if os.environ['RANK'] == '0' else 0)\n"
    assert launch.spawn_ranks(2, [sys.executable, "-c", silent]) == 5 and len(ports) == 2
    other_rank = ("import os, sys, time\n"
                  "if os.environ['RANK'] == '1':\n"
                  "    print('Address already in use', file=sys.stderr); sys.exit(9)\n"
                  "time.sleep(30)\n")
    assert launch.spawn_ranks(2, [sys.executable, "-c", other_rank]) != 0 and len(ports) == 3


def test_under_launcher_detection():
    assert not launch.under_launcher({})
    assert launch.under_launcher({"RANK": "0", "WORLD_SIZE": "1", "MASTER_PORT": "1"})


def test_counter_key_follows_the_code_not_the_comments(tmp_path):
    """The recorded PMC counters are used only while bench.kernel_source_hash() matches: the key must
    move with the kernel's code and build flags and stay put when a comment or the spacing changes."""
    import shutil
    sys.path.insert(0, REPO)
    import bench
    src = os.path.join(REPO, "radiative3d_amd", "csrc")
    base = bench.kernel_source_hash(src)
    assert base == bench.kernel_source_hash()
    work = tmp_path / "csrc"
    shutil.copytree(src, work)
    pool = work / "r3d_pool.h"
    text = pool.read_text()
    pool.write_text("// a new remark\n" + text.replace("\n", "\n   \n", 3) + "\n/* and a block\n   comment */\n")
    assert bench.kernel_source_hash(str(work)) == base
    pool.write_text(text.replace("constexpr int kPoolMovesThin = 32;", "constexpr int kPoolMovesThin = 33;"))
    assert "kPoolMovesThin = 33" in pool.read_text()
    assert bench.kernel_source_hash(str(work)) != base
