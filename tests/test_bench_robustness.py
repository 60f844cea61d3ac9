"""`bench.py --gpus N` must not be able to lie or to hang silently (VERDICT r04 "next" item 2): CPU tests of
the pieces that protect the first multi-GPU run -- no GPU, two `gloo` ranks.

* the communicator setup cannot desynchronise: rank 0's id creation forced to fail -> `Context.comm_init` broadcasts
  (status, id) unconditionally, BOTH ranks raise, `bench.setup_native_comm` returns False on both, and the next collective
  (the one the r04 code would have met with mismatched peers) completes;
* a rank that cannot create ITS communicator after a good id: the agreement makes every rank drop to the fallback;
* a stalled phase: the watchdog prints the one line and every rank exits with a NON-ZERO code; armed before the first
  collective (a stall with no headline yet still produces a line and code 3);
* the line is printed once even when the watchdog and the main thread race.
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
