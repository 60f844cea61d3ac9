"""`bench.py --gpus N` must not be able to lie or to hang silently (VERDICT r04 "next" item 2): CPU tests of
the pieces that protect the first multi-GPU run -- no GPU, two `gloo` ranks.

* the communicator setup cannot desynchronise: rank 0's id creation forced to fail -> `Context.comm_init` broadcasts
  (status, id) unconditionally, BOTH ranks raise, `bench.setup_native_comm` returns False on both, and the next collective
  (the one the r04 code would have met with mismatched peers) completes;
* a rank that cannot create ITS communicator after a good id: the agreement makes every rank drop to the fallback;
* a stalled phase: the watchdog prints the one line and every rank exits with a NON-ZERO code; armed before the first
  collective (a stall with no headline yet still produces a line and code 3);
* the line is printed once even when the watchdog and the main thread race;
* a bare `python3 bench.py --gpus N` (no torch.distributed.run, WORLD_SIZE unset: the driver's command shape) is its own
  launcher (VERDICT r05 "next" item 1): N child ranks with the rendezvous environment, rank 0's line relayed once, the
  largest child code returned, a silent child ended at the launcher's deadline with a non-zero code and a null line;
* a rank other than 0 fires its watchdog later than rank 0, and a teardown that hangs after the complete line
  does not turn the run into a failure.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeLib:
    """Stands in for the C ABI where a communicator would need a GPU: records calls, fails where the test says."""

    def __init__(self, fail_id, fail_init_on):
        self.fail_id, self.fail_init_on, self.calls = fail_id, fail_init_on, []

    def pm_comm_unique_id(self, ident):
        self.calls.append("id")
        if self.fail_id:
            return -9
        for i in range(len(ident)):
            ident[i] = (7 * i + 1) & 0xff
        return 0

    def pm_comm_init(self, h, ident, rank, world):
        self.calls.append(("init", bytes(ident)[:4], rank, world))
        return -9 if rank in self.fail_init_on else 0

    def pm_comm_destroy(self, h):
        self.calls.append("destroy")
        return 0

    def pm_last_error(self, h):
        return b"forced by the test"


def _comm_worker(rank, world, port, fail_id, fail_init_on, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import bench
    import plonk_prototype_amd as pa
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = object.__new__(pa.Context)                 # no pm_init: there is no device here
        ctx._lib, ctx._h, ctx.comm_world = _FakeLib(fail_id, fail_init_on), None, 1
        cpu = torch.device("cpu")
        native = bench.setup_native_comm(ctx, rank, world, dist, cpu)
        # the collective that follows in bench.py: with r04's code rank 0 was here while rank 1 sat in the broadcast
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        q.put((rank, native, int(t.item()), ctx._lib.calls, bench.agree(dist, rank != 1, cpu)))
    finally:
        dist.destroy_process_group()


def _run_comm(fail_id, fail_init_on):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_comm_worker, args=(r, 2, port, fail_id, fail_init_on, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_failing_id_creation_cannot_desynchronise_the_ranks():
    res = _run_comm(fail_id=True, fail_init_on=())
    for rank, native, total, calls, agreed in res:
        assert native is False and total == 3 and agreed is False, (rank, native, total)
        assert not any(c[0] == "init" for c in calls if isinstance(c, tuple)), calls   # nobody tried to join a half-made group
    assert res[0][3] == ["id"] and res[1][3] == []


def test_one_rank_without_a_communicator_means_none_uses_it():
    res = _run_comm(fail_id=False, fail_init_on=(1,))
    for rank, native, total, calls, _ in res:
        assert native is False and total == 3
        assert ("init", bytes([1, 8, 15, 22]), rank, 2) in calls             # the id reached both ranks intact
    assert "destroy" in res[0][3] and "destroy" not in res[1][3]             # the rank that had one dropped it


def test_good_setup_is_native_on_both_ranks():
    for rank, native, total, calls, _ in _run_comm(fail_id=False, fail_init_on=()):
        assert native is True and total == 3 and "destroy" not in calls


_STALL = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import bench
fd = os.dup(1)
os.dup2(2, 1)
em = bench.LineEmitter(fd)
state = {{"emit": None}}
def on_timeout(phase):
    if state["emit"]:
        state["emit"]("watchdog: " + phase)
    else:
        em.emit({{"metric": "m", "value": None, "note": "watchdog: " + phase}})
dog = bench.Watchdog({first}, on_timeout, "startup")
if {headline}:
    state["emit"] = lambda note=None: em.emit({{"metric": "m", "value": 1.0, "note": note}})
    dog.extend({second}, "legs")
if {race}:
    # the main thread prints at the moment the watchdog fires: one line, whoever wins
    time.sleep({second} - 0.05)
    state["emit"]()
    dog.finish()
    time.sleep(0.6)
    sys.exit(0)
time.sleep(30)          # the stalled collective
"""


def _run_stall(first, second, headline, race=False):
    code = _STALL.format(root=ROOT, first=first, second=second, headline=headline, race=race)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines


def test_stalled_legs_print_the_line_and_exit_non_zero():
    rc, lines = _run_stall(first=20, second=1, headline=True)
    assert rc == 3 and len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.0 and "watchdog: legs" in line["note"]


def test_a_stall_before_the_headline_is_caught_too():
    rc, lines = _run_stall(first=1, second=1, headline=False)
    assert rc == 3 and len(lines) == 1
    assert json.loads(lines[0])["value"] is None


def test_the_line_is_printed_once_when_watchdog_and_main_thread_race():
    for _ in range(3):
        rc, lines = _run_stall(first=20, second=0.5, headline=True, race=True)
        assert len(lines) == 1 and rc in (0, 3), (rc, lines)


# ---------------------------------------------------------------------------------------------- the bare launcher
_STUB = r"""
import json, os, signal, sys, time
mode, piddir = sys.argv[1], sys.argv[2]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
open(os.path.join(piddir, f"pid{rank}"), "w").write(str(os.getpid()))
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert int(os.environ["LOCAL_RANK"]) == rank
if mode == "silent":
    time.sleep(120)                       # the rank that never prints
if mode == "peerdies":
    # rank 1 dies at start-up; rank 0 sits "in a collective" under the real watchdog, which the launcher's abort file cuts short
    if rank == 1:
        sys.exit(7)
    sys.path.insert(0, sys.argv[3])
    import bench
    fd = os.dup(1)
    em = bench.LineEmitter(fd)
    dog = bench.Watchdog(600, lambda phase: em.emit({"metric": "m", "value": None, "note": "watchdog: " + phase}), "startup",
                         abort_file=os.environ.get("PM_BENCH_ABORT_FILE"))
    time.sleep(120)
if rank == 0:
    print("RCCL banner on stdout", flush=True)
    print(json.dumps({"metric": "m", "value": 1.0, "world": world, "rank": rank}), flush=True)
    if mode == "twice":
        print(json.dumps({"metric": "m", "value": 2.0}), flush=True)
else:
    print(json.dumps({"metric": "m", "value": -1.0, "rank": rank}), flush=True)     # not rank 0: must not be relayed
    if mode == "exit3":
        sys.exit(3)
    if mode == "sigkill":
        os.kill(os.getpid(), signal.SIGKILL)
"""


def _run_launcher(tmp_path, mode, n=2, deadline=30):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launcher-timeout", str(deadline),
                        "--child-cmd", f"{sys.executable} {stub} {mode} {tmp_path} {ROOT}"],
                       capture_output=True, text=True, timeout=120, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr


def test_bare_gpus_n_starts_its_own_ranks_and_relays_one_line(tmp_path):
    rc, lines, err = _run_launcher(tmp_path, "twice", n=3)
    assert rc == 0 and len(lines) == 1, (rc, lines, err)
    line = json.loads(lines[0])
    assert line == {"metric": "m", "value": 1.0, "world": 3, "rank": 0}
    assert "has imported torch: False" in err                     # the launcher makes no GPU call: it never loads torch
    assert "RCCL banner on stdout" in err and "-1.0" in err       # every other stdout byte of the ranks lands on stderr
    assert sorted(os.listdir(tmp_path))[:3] == ["pid0", "pid1", "pid2"]


def test_launcher_returns_the_largest_child_code(tmp_path):
    rc, lines, _ = _run_launcher(tmp_path, "exit3")
    assert rc == 3 and len(lines) == 1 and json.loads(lines[0])["value"] == 1.0
    rc, lines, _ = _run_launcher(tmp_path, "sigkill")
    assert rc == 128 + 9 and len(lines) == 1


def test_launcher_deadline_ends_silent_ranks_with_a_null_line(tmp_path):
    rc, lines, err = _run_launcher(tmp_path, "silent", deadline=2)
    assert rc != 0 and len(lines) == 1, (rc, lines, err)
    line = json.loads(lines[0])
    assert line["value"] is None and "deadline" in line["note"]
    for r in range(2):                                            # the exact PIDs it started are gone
        pid = int((tmp_path / f"pid{r}").read_text())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_under_a_launcher_the_world_size_must_match():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


_GRACE = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import bench
t0 = time.monotonic()
fired = []
dog = bench.Watchdog(0.5, lambda phase: fired.append((phase, time.monotonic() - t0)), "startup",
                     exit_fn=lambda code: fired.append(code), grace={grace})
time.sleep(0.1)
if {teardown}:
    dog.extend(0.5, "teardown", exit_code=0)
time.sleep({grace} + 1.5)
print(repr(fired))
"""


def _run_grace(grace, teardown):
    p = subprocess.run([sys.executable, "-c", _GRACE.format(root=ROOT, grace=grace, teardown=teardown)],
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    return eval(p.stdout.strip())                                 # noqa: S307 -- our own repr


def test_other_ranks_fire_after_rank_zero_and_teardown_hang_exits_zero():
    r0 = _run_grace(0.0, False)
    r1 = _run_grace(1.0, False)
    assert r0[0][0] == "startup" and r0[1] == 3 and r1[1] == 3
    assert 0.5 <= r0[0][1] < 1.2 and r1[0][1] >= 1.5              # the grace separates them
    td = _run_grace(0.0, True)
    assert td[0][0] == "teardown" and td[1] == 0


def test_a_rank_that_dies_at_start_up_costs_its_peers_seconds_not_the_deadline(tmp_path):
    """Rank 1 exits with code 7 before any collective; rank 0 is blocked under its 600 s start-up watchdog.  The launcher
    creates the abort file, rank 0's watchdog prints the line (value null, the reason in the note) and leaves with code 3; the
    launcher returns the largest code -- all within seconds."""
    import time
    t0 = time.monotonic()
    rc, lines, err = _run_launcher(tmp_path, "peerdies", deadline=300)
    assert time.monotonic() - t0 < 30, err
    assert rc == 7 and len(lines) == 1, (rc, lines, err)
    line = json.loads(lines[0])
    assert line["value"] is None and "a peer rank exited" in line["note"]
    assert "signalling the others" in err and "[3, 7]" in err


def test_a_terminated_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to the launcher (the caller's own timeout): the ranks are terminated, a null line says why, the exit code is 128 + 15."""
    import signal
    import time
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-timeout", "300",
                          "--child-cmd", f"{sys.executable} {stub} silent {tmp_path} {ROOT}"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    t0 = time.monotonic()
    while not all((tmp_path / f"pid{r}").exists() for r in range(2)) and time.monotonic() - t0 < 60:
        time.sleep(0.1)
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert p.returncode == 128 + 15 and len(lines) == 1 and "signal 15" in json.loads(lines[0])["note"], (p.returncode, out, err)
    for r in range(2):
        pid = int((tmp_path / f"pid{r}").read_text())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
