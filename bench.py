"""Headline benchmark: BLS12-381 Fr NTT butterflies/s (+ G1 MSM scalar-muls/s) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs either under torch.distributed.run (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment) or bare: with WORLD_SIZE unset this process starts the N ranks itself as child processes (launch_ranks),
relays rank 0's one line and exits with the largest child code.

A step = one pass of the hot path over one batch of synthetic input resident in HBM:
a forward and an inverse 2^20-point NTT (BASELINE.json configs[1]).  `value` is whole-job
butterflies/s over all ranks (weak scaling: every rank transforms its own polynomials, no
collective on the data path).  The same JSON line carries the MSM leg (configs[2]: 2^20-point
G1 MSM; with N > 1 the points are sharded over the ranks and the 144-byte partials are
all-gathered over RCCL and folded), the full-prove leg (configs[3]: a 2^20-gate synthetic circuit
through the native prover: `prover`, with `prover.all_gate_kinds` = the same size with every widget selector
present, `prover.two_contexts_ms_per_proof` = two proofs in flight, and `prover.large` = BASELINE configs[4]: one
2^24-gate proof through pm_plonk_prove and through pm_plonk_prove_dist -- one rank at N = 1, the N ranks otherwise),
`msm.pcie_inclusive` (pm_g1_msm with host scalars: the CommitKey::commit drop-in signature), the roofline of the dominant kernel from
live hipEvent timings, and the CPU baseline (the oracle's C restatement, timed on this box's host cores).
`ntt_extra` holds the 2^24 transform (+ the rank-split transform's kernels on one rank), the PCIe-inclusive
host-pointer call and the prover's coset shape; `ntt_fourstep` (N > 1, opt-in) one transform over all ranks.

Role of oracle/ here: it is the CPU baseline, it synthesises seeded inputs with known answers
(scalars, an SRS stand-in of known discrete logs) and it checks every leg's result after the timed
region (asserts below).  Nothing that is timed on the GPU calls it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md


PMC_ROUND = "r06"          # the round whose committed PMC summaries describe THIS tree's kernels


def pmc_traffic(kernel, grid=None, also=(), kinds=("pmc_summary", "pmc_prover_summary", "pmc_big_summary")):
    """HBM bytes per launch of `kernel` from the PMC counters (2 x FETCH_SIZE + WRITE_SIZE: the gfx950 correction of
    MI355X_MICROARCH.md) -> (bytes, source) or (None, None).  The counters cannot be collected inside this process: the PMC
    passes are separate `rocprofv3 --pmc` runs (tools/collect_pmc.sh -> tools/pmc_summary.py) whose summaries are committed under
    profiles/.  Only the CURRENT round's summaries are quoted as they stand (a kernel whose tiling changed must not
    report an older kernel's bytes as its own); when this round has none yet, the previous round's figure is returned with its
    source string marked "STALE rNN" -- the caller passes the string through, so the line says what it is.  Several entries
    at different grids and no grid asked for: the largest grid (the full-size launch) is taken."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for rnd in (PMC_ROUND, "r05"):
        for kind in kinds:
            path = os.path.join(root, f"{rnd}_{kind}.json")
            try:
                with open(path) as f_:
                    ents = json.load(f_)
            except (OSError, ValueError):
                continue
            hits = [(nm, e) for nm, e in ents.items()
                    if kernel in nm and all(a in nm for a in also) and "hbm_bytes_per_launch_corrected" in e
                    and (grid is None or f"grid={grid} " in nm + " ")]
            if not hits:
                continue

            def grid_of(nm_):
                try:
                    return int(nm_.split("grid=")[1].split()[0])
                except (IndexError, ValueError):
                    return 0
            nm, e = max(hits, key=lambda h_: grid_of(h_[0]))
            src = f"profiles/{rnd}_{kind}.json [{nm}]"
            if rnd != PMC_ROUND:
                src = f"STALE {rnd} (no {PMC_ROUND} summary has this kernel; the kernel may have changed since): " + src
            return int(e["hbm_bytes_per_launch_corrected"]), src
    return None, None



class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when a communicator is created; this script's stdout carries exactly
    one JSON line, so file descriptor 1 points at stderr while a communicator is being made."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class LineEmitter:
    """The ONE JSON line.  `emit(obj)` writes it to the saved stdout descriptor exactly once per process, whoever calls
    first (the main thread at the end, or the watchdog): a lock and a printed flag, so a watchdog that fires while the
    main thread is printing cannot produce a second line."""

    def __init__(self, fd):
        import threading
        self.fd, self.lock, self.printed = fd, threading.Lock(), False

    def emit(self, obj) -> bool:
        with self.lock:
            if self.printed:
                return False
            self.printed = True
            sys.stdout.flush()
            os.write(self.fd, (json.dumps(obj) + "\n").encode())
            return True


class Watchdog:
    """N > 1: a collective that never returns has no other way out.  Armed BEFORE the first collective (r04 armed it after
    the headline: a desynchronised communicator setup hung with no line and no exit).  `extend(seconds, phase)` moves the
    deadline when a phase completes; when it passes, `on_timeout(phase)` runs (rank 0 prints the line with what there is)
    and EVERY rank leaves with a NON-ZERO code -- a hang must not read as success (r04 exited 0).

    `abort_file`: a path the bare launcher (launch_ranks) creates as soon as one rank has exited with a non-zero code; the
    watchdog polls it and treats its appearance like a passed deadline, so a rank that died at start-up costs its peers seconds,
    not the ten minutes of the start-up deadline.
    `grace`: seconds added to every deadline of this rank.  Rank 0 runs with 0 and the other ranks with a few
    seconds, so that rank 0 -- the one that prints -- always fires first: under torch.distributed.run the first non-zero
    exit makes the agent SIGTERM the remaining ranks, and a rank other than 0 firing first could get rank 0 killed before
    its line is out.  `extend(..., exit_code=c)` sets the code a timeout of THAT phase ends with (the teardown phase, after
    the complete line is printed, ends with 0: the measurement is whole, only close / destroy hung)."""
    EXIT_CODE = 3

    def __init__(self, seconds, on_timeout, phase="startup", exit_fn=os._exit, grace=0.0, abort_file=None):
        import threading
        self._grace = grace
        self._abort_file = abort_file          # the bare launcher creates it when a peer rank has exited with an error
        self._deadline, self._phase, self._code = time.monotonic() + seconds + grace, phase, self.EXIT_CODE
        self._done, self._lock = threading.Event(), threading.Lock()
        self._on_timeout, self._exit = on_timeout, exit_fn
        threading.Thread(target=self._run, daemon=True).start()

    def extend(self, seconds, phase, exit_code=None):
        with self._lock:
            self._deadline, self._phase = time.monotonic() + seconds + self._grace, phase
            self._code = self.EXIT_CODE if exit_code is None else exit_code

    def finish(self):
        self._done.set()

    def _run(self):
        while not self._done.wait(0.2):
            with self._lock:
                late, phase, code = time.monotonic() > self._deadline, self._phase, self._code
            if not late and self._abort_file and os.path.exists(self._abort_file):
                # a peer is gone: whatever collective this rank sits in will never complete -- do now what the deadline would
                # do later (the watchdog thread runs while the main thread is blocked in native code; a signal handler would not)
                late, phase = True, f"{phase} -- cut short: a peer rank exited with an error (the launcher's abort file)"
                if code == 0:
                    code = self.EXIT_CODE
                time.sleep(self._grace / 5.0)       # rank 0 first, as for a deadline
            if late:
                try:
                    self._on_timeout(phase)
                finally:
                    self._exit(code)
                return


def launch_ranks(n, child_cmd, json_fd, deadline_s, env=None, log=lambda s: print(s, file=sys.stderr, flush=True)):
    """`python3 bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process becomes the launcher.  It has
    made no GPU call (only `import numpy` so far) and makes none: it starts N CHILD processes of `child_cmd` -- one rank per
    GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, exactly what
    torch.distributed.run would set -- relays rank 0's ONE JSON line to `json_fd` (once; anything else rank 0 writes on its
    stdout goes to stderr), waits for all of them and returns the LARGEST child exit code (a child killed by signal s counts
    as 128 + s).  It never replaces itself (no exec) and it kills only the exact PIDs it started.  When `deadline_s` passes
    with children still running they are terminated and the return code is non-zero; if rank 0 had not printed by then a
    line with value null and the reason is printed, so that the caller always gets exactly one line."""
    import socket
    import subprocess
    import threading
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import tempfile
    abort_dir = tempfile.mkdtemp(prefix="pm_bench_")
    abort_file = os.path.join(abort_dir, "abort")
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                PM_BENCH_ABORT_FILE=abort_file)
    emitter = LineEmitter(json_fd)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        # rank 0's stdout is a pipe this process reads; the other ranks' stdout joins stderr (they print no line)
        procs.append(subprocess.Popen(child_cmd, env=e, stdin=subprocess.DEVNULL,
                                      stdout=subprocess.PIPE if r == 0 else 2))
    log(f"[bench] launcher: started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}, "
        f"deadline {deadline_s:.0f} s; this process has imported torch: {'torch' in sys.modules}")

    def relay():
        for raw in procs[0].stdout:
            txt = raw.decode(errors="replace").strip()
            obj = None
            if txt.startswith("{"):
                try:
                    obj = json.loads(txt)
                except ValueError:
                    obj = None
            if isinstance(obj, dict) and "metric" in obj and emitter.emit(obj):
                continue
            if txt:
                log(f"[bench] rank 0 stdout: {txt}")

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    # the launcher itself may be told to stop (the caller's own timeout: SIGTERM; Ctrl-C): the ranks must not outlive it
    import signal
    stop = {"sig": None}

    def on_signal(signum, _frame):
        stop["sig"] = signum
    old_handlers = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    t_end = time.monotonic() + deadline_s
    timed_out = False
    peer_failed = False
    while any(p.poll() is None for p in procs):
        if time.monotonic() > t_end or stop["sig"] is not None:
            timed_out = True
            break
        if not peer_failed and any(p.poll() not in (None, 0) for p in procs):
            # a rank is gone: its peers' collectives cannot complete.  Tell their watchdogs (they print the line with what had
            # finished and leave with a non-zero code) and give them a bounded time to do so
            peer_failed = True
            bad = [(i, p.returncode) for i, p in enumerate(procs) if p.poll() not in (None, 0)]
            log(f"[bench] launcher: rank(s) {bad} exited with an error; signalling the others through {abort_file}")
            with open(abort_file, "w") as f_:
                f_.write(repr(bad))
            t_end = min(t_end, time.monotonic() + 90)
        time.sleep(0.1)
    if timed_out:
        alive = [p for p in procs if p.poll() is None]
        why = f"signal {stop['sig']} received" if stop["sig"] is not None else f"deadline of {deadline_s:.0f} s passed"
        log(f"[bench] launcher: {why} with ranks {[procs.index(p) for p in alive]} still running; terminating them")
        for p in alive:
            p.terminate()
        t_kill = time.monotonic() + 10
        for p in alive:
            try:
                p.wait(max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    th.join(5)
    for sg, h_ in old_handlers.items():
        signal.signal(sg, h_)
    codes = [p.returncode if p.returncode >= 0 else 128 - p.returncode for p in procs]
    log(f"[bench] launcher: rank exit codes {codes}")
    try:
        if os.path.exists(abort_file):
            os.unlink(abort_file)
        os.rmdir(abort_dir)
    except OSError:
        pass
    rc = max(codes)
    if timed_out:
        rc = max(rc, Watchdog.EXIT_CODE + 1 if stop["sig"] is None else 128 + stop["sig"])
    if not emitter.printed:
        emitter.emit({"metric": "bls12_381_fr_ntt_butterflies_per_s", "value": None, "unit": "butterflies/s", "n_gpus": n,
                      "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                      "note": f"launcher: rank 0 printed no line (rank exit codes {codes}"
                              + (f", signal {stop['sig']}" if stop["sig"] is not None else
                                 f", deadline of {deadline_s:.0f} s passed" if timed_out else "") + ")"})
        rc = max(rc, 1)
    return rc


def group_kernels(prof) -> dict:
    """pm_profile_read of one proof -> milliseconds by phase."""
    grp = {"msm": 0.0, "ntt": 0.0, "quotient": 0.0, "permutation": 0.0, "openings": 0.0}
    for name, (_, ms) in prof.items():
        key = ("msm" if name.startswith("msm_") else "ntt" if name.startswith("ntt_") else
               "quotient" if name == "plonk_quotient" else
               "permutation" if name in ("plonk_perm_terms", "fr_batch_inverse", "fr_vec_mul", "fr_prefix_product")
               else "openings")
        grp[key] += ms
    return grp


def agree(dist, ok: bool, device) -> bool:
    """True iff `ok` on EVERY rank: one all_reduce(MIN) that every rank reaches whatever happened to it locally, so no
    rank skips a collective its peers are waiting in."""
    import torch
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def setup_native_comm(ctx, rank, world, dist, coll_dev, log=lambda s: None) -> bool:
    """Create the library's RCCL communicator on every rank, or on none.  ctx.comm_init broadcasts rank 0's (status, id)
    unconditionally, so a failing id creation cannot leave the ranks in different collectives; the all_reduce(MIN) then
    settles whether EVERY rank has a communicator -- if not, the ones that have one drop it and all use torch.distributed."""
    try:
        with stdout_to_stderr():
            ctx.comm_init(rank, world, coll_dev)
        ok_local = True
    except Exception as e:                                   # noqa: BLE001
        log(f"[bench] rank {rank}: pm_comm_init failed ({e}); using torch.distributed")
        ok_local = False
    native = agree(dist, ok_local, coll_dev)
    if not native and ok_local:
        ctx.comm_destroy()
    return native


def main():
    # stdout carries exactly ONE line, the result: file descriptor 1 points at stderr for the whole run (RCCL, gloo and the
    # HIP runtime print banners on stdout from native code, on every rank), and the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--msm-log-n", type=int, default=20)
    ap.add_argument("--msm-steps", type=int, default=5)
    ap.add_argument("--msm-large-log-n", type=int, default=24,
                    help="second MSM leg (BASELINE configs[4]); 0 disables it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-msm", action="store_true")
    ap.add_argument("--no-poly", action="store_true")
    ap.add_argument("--no-prover", action="store_true")
    ap.add_argument("--no-dist-prover", action="store_true", help="skip the pm_plonk_prove_dist leg of the prover")
    ap.add_argument("--no-msm-extra", action="store_true",
                    help="skip the witness-like and batched MSM measurements (PMC passes: one MSM shape per mode)")
    ap.add_argument("--no-ntt-extra", action="store_true",
                    help="skip the PCIe-inclusive and coset-4n NTT measurements (PMC passes: one NTT size only)")
    ap.add_argument("--prover-log-n", type=int, default=20,
                    help="gates of the synthetic circuit for the full-prove entry (BASELINE configs[3])")
    ap.add_argument("--prover-large-log-n", type=int, default=24,
                    help="gates of the second full-prove entry (BASELINE configs[4]: pm_plonk_prove and one-rank / N-rank "
                         "pm_plonk_prove_dist); 0 disables it")
    ap.add_argument("--cpu-prover-log-n", type=int, default=20,
                    help="gates of the CPU-baseline proof (2^18: ~8 s on 16 threads; 2^20, the GPU leg's size: ~35 s)")
    ap.add_argument("--leg-timeout", type=int, default=420,
                    help="N > 1: seconds the sharded MSM / prover legs may take before the watchdog prints the line without them")
    ap.add_argument("--startup-timeout", type=int, default=600,
                    help="N > 1: seconds the process group, the communicator and the headline may take before the watchdog "
                         "ends every rank with a non-zero code (armed before the first collective)")
    ap.add_argument("--comm-timeout-ms", type=int, default=180000,
                    help="N > 1 with the library's RCCL communicator: option comm_timeout_ms of the context (0 = off); well above "
                         "any legitimate exchange, below --leg-timeout")
    ap.add_argument("--fourstep-log-n", type=int, default=0,
                    help="N > 1 only, off by default: also time ONE 2^K transform split over the ranks "
                         "(pm_fr_ntt_fourstep_dev, SURVEY 8f N5) through the library's RCCL communicator")
    ap.add_argument("--launcher-timeout", type=int, default=0,
                    help="bare `--gpus N` (no torch.distributed.run around it): seconds the N child ranks may take in all; "
                         "0 = --startup-timeout + 2 x --leg-timeout + 300 (the ranks' own watchdogs end them before that)")
    ap.add_argument("--child-cmd", default=None,
                    help="bare `--gpus N` only: the command each rank runs instead of this script (tests drive the launcher "
                         "with a stub child)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher.  Nothing above has touched the GPU (imports: numpy) and nothing in
        # launch_ranks does; the ranks are CHILD processes of this script, never an exec
        import shlex
        cmd = shlex.split(args.child_cmd) if args.child_cmd else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(launch_ranks(args.gpus, cmd, json_fd,
                              args.launcher_timeout or args.startup_timeout + 2 * args.leg_timeout + 300))

    import torch
    import torch.distributed as dist
    import plonk_prototype_amd as pa
    from plonk_prototype_amd.dist import allgather_fold, shard_range
    from oracle.cpu_oracle import INVERSE, CpuOracle, ints_to_limbs  # cpu_baseline + input synthesis + checks
    from oracle.bigint_oracle import R_MOD

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: run `python3 bench.py --gpus N` bare (it starts its own "
                         f"N ranks) or under torch.distributed.run with --nproc-per-node N")
    # Rehearsal knobs (not used by the driver): PM_BENCH_DEVICE pins every rank to one GPU and
    # PM_BENCH_BACKEND=gloo moves the 144-byte collectives to the CPU, so the multi-rank code path
    # can be exercised on a one-GPU box.
    backend = os.environ.get("PM_BENCH_BACKEND", "nccl")
    if "PM_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PM_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    emitter = LineEmitter(json_fd)
    state = {"emit": None}             # set once the headline exists: the full line with the legs that have finished
    dog = None
    if world > 1:
        def on_timeout(phase):
            note = (f"watchdog: phase '{phase}' did not finish in time (a collective that never returned?); line printed by the "
                    f"watchdog with what had finished; every rank exits with code {Watchdog.EXIT_CODE}")
            print(f"[bench] rank {rank}: {note}", file=sys.stderr, flush=True)
            if rank == 0:
                if state["emit"]:
                    state["emit"](note)
                else:
                    emitter.emit({"metric": "bls12_381_fr_ntt_butterflies_per_s", "value": None, "unit": "butterflies/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
                                  "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic",
                                  "note": note})
        dog = Watchdog(args.startup_timeout, on_timeout, "process group + communicator + headline",
                       grace=0.0 if rank == 0 else 10.0, abort_file=os.environ.get("PM_BENCH_ABORT_FILE"))
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        ctx.sync()                       # the library's own stream
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def local_sync():                    # for rank-0-only sections: no collective
        ctx.sync()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    oracle = CpuOracle()
    ctx = pa.Context(local_rank)
    # N > 1: the exchange step runs through the library's own RCCL communicator (pm_comm_init); if creating it
    # fails on any rank, every rank falls back to torch.distributed for the 2.3 KB all-gathers (same fold)
    native_comm = False
    if world > 1 and backend == "nccl":
        native_comm = setup_native_comm(ctx, rank, world, dist, coll_dev, lambda m_: print(m_, file=sys.stderr, flush=True))
        if native_comm and args.comm_timeout_ms > 0:
            # a peer that dies inside an exchange: the library ends the blocked call with ncclCommAbort after this long and the
            # leg records an error (every later leg on this communicator fails at once) instead of the watchdog ending the run
            ctx.set_option("comm_timeout_ms", args.comm_timeout_ms)
        # self-diagnosing first multi-GPU run: every rank says what communicator it ended up with
        import ctypes as _C
        _r, _w = _C.c_int(-1), _C.c_int(-1)
        ctx._lib.pm_comm_info(ctx._h, _C.byref(_r), _C.byref(_w))
        print(f"[bench] rank {rank}/{world} device {local_rank}: library communicator "
              f"{'rank %d of %d' % (_r.value, _w.value) if native_comm else 'NOT in use (torch.distributed ' + backend + ')'}",
              file=sys.stderr, flush=True)
    # The launches go to the context's OWN stream (stream = 0 in the calls below: the null handle means that; a foreign
    # stream costs one ordering event per call, +2 % on the headline step).  The timing events must be recorded on that
    # same stream -- torch's default stream would bracket nothing -- so it is wrapped for torch (pm_ctx_stream).
    tstream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
    stream = 0
    # host threads we may use: the box's CPU share, not every core the kernel lists
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))

    # ------------------------------------------------------------------ NTT leg (the step)
    k = args.log_n
    n = 1 << k
    host_in = oracle.fr_sample(0x504C4F4E4B + rank, n)          # uniform Fr, Montgomery limbs
    d_a = torch.from_numpy(host_in.view(np.int64)).to(dev)
    d_b = torch.empty_like(d_a)
    d_c = torch.empty_like(d_a)

    def step():
        ctx.fr_ntt_dev(d_a.data_ptr(), n, d_b.data_ptr(), k, 0, stream=stream)
        ctx.fr_ntt_dev(d_b.data_ptr(), n, d_c.data_ptr(), k, INVERSE, stream=stream)

    # the timed region runs with every timer OFF (an event pair is ~5 us of stream time: r02's headline carried the
    # dominant kernel's pair inside the loop and read 4-9 % low); the kernels' durations -- the roofline needs the
    # dominant one's -- come from a separate, fully profiled loop of the same step right after it
    step()                                                          # first launches: tables, code load
    barrier()
    for _ in range(max(args.warmup, 1)):
        step()
    # The timed region: EXACTLY args.steps steps between two barrier + synchronize brackets -- measured `repeats` times,
    # each with the host clock around the brackets AND with a hipEvent pair recorded on the launch stream right after the
    # opening bracket / right before the closing one.  `value` is from the MEDIAN region by the device clock: the host
    # clock also counts the closing bracket itself (ctx.sync + synchronize, ~0.3 ms here), which a 5 ms region of 20
    # steps sees as 6 % and a 50 ms region of 200 steps as 0.6 % (the driver's 20-step run read 8.4e10
    # where the kernels do 9.6e10).  Both clocks are in the line (`timing`).
    repeats = 9
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(repeats)]
    host_dts = []
    for r_ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        ev[r_][0].record(tstream)
        for _ in range(args.steps):
            step()
        ev[r_][1].record(tstream)
        barrier()
        host_dts.append(time.perf_counter() - t0)
    dev_dts = [a_.elapsed_time(b_) * 1e-3 for a_, b_ in ev]
    dt = max_over_ranks(float(np.median(dev_dts)))
    dt_host = max_over_ranks(float(np.median(host_dts)))
    assert torch.equal(d_c, d_a), "iNTT(NTT(a)) != a"            # round trip inside the bench
    butterflies_per_step = 2 * (n // 2) * k
    value = world * butterflies_per_step * args.steps / dt
    timing = {"clock": "hipEvent pair on the launch stream around exactly `steps` steps, inside the barrier + synchronize brackets",
              "repeats": repeats, "region_ms_device_median": round(dt * 1e3, 4),
              "region_ms_device_min_max": [round(min(dev_dts) * 1e3, 4), round(max(dev_dts) * 1e3, 4)],
              "region_ms_host_clock_median": round(dt_host * 1e3, 4),
              "value_by_host_clock": world * butterflies_per_step * args.steps / dt_host}
    prof_steps = 200                                              # fixed: the kernels' mean durations must not depend on --steps
    ctx.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(prof_steps):
        step()
    barrier()
    dt_prof = (time.perf_counter() - t0) / prof_steps
    pre = {name: v for name, v in ctx.profile_read().items() if name.startswith("ntt_pass")}
    ctx.profile(False)
    dom = max(pre, key=lambda s_: pre[s_][1])
    prof = pre

    # roofline of the dominant kernel: algorithmic bytes of one launch / its mean duration
    passes = pa.ntt_plan(k)
    kern = dict(pre)                                               # live: the profiled loop of this run
    dom_ms = kern[dom][1] / kern[dom][0]
    s_dom = passes[0] if dom.endswith("first") or dom.endswith("single") else passes[-1]
    algo_bytes = 64 * n * s_dom / k                                # 64 N bytes per transform, S/k of it per pass
    achieved = algo_bytes / (dom_ms * 1e-3)
    # `traffic` (HBM bytes per launch from the PMC counters, FETCH_SIZE x 2 + WRITE_SIZE: the gfx950 correction of
    # MI355X_MICROARCH.md) cannot be collected inside this process: the PMC passes are separate rocprofv3 runs of THIS
    # command (tools/collect_pmc.sh -> tools/pmc_summary.py).  When their summary for this round is in profiles/ and has
    # the dominant kernel at this size, its figure is quoted (with its source); otherwise null.
    # ntt_pass4_kernel<S, LT, OUT_UFAST, IN_WIDE, OUT_WIDE>: the first pass reads canonical and writes wide, the last the reverse
    role_args = {"first": "false, true>", "last": "true, false>", "single": "false, false>", "middle": "true, true>"}
    traffic, traffic_src = pmc_traffic("ntt_pass", n // 4, also=(role_args[dom.rsplit("_", 1)[1]],))
    # the committed rocprofv3 --kernel-trace --stats summary of the headline leg alone (tools/final_gpu_run.sh): the same kernel's
    # average there, quoted beside the live figure (an event pair adds ~2 us of stream time to what it brackets)
    prof_avg, prof_src = None, None
    try:
        import csv
        for rnd_ in (PMC_ROUND, "r05"):
            path_ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"{rnd_}_headline_kernel_stats.csv")
            if not os.path.exists(path_):
                continue
            with open(path_) as f_:
                for row_ in csv.DictReader(f_):
                    if "ntt_pass" in row_["Name"] and role_args[dom.rsplit("_", 1)[1]] in row_["Name"]:
                        prof_avg, prof_src = round(float(row_["AverageNs"]) / 1e3, 2), f"profiles/{rnd_}_headline_kernel_stats.csv"
            if prof_avg is not None:
                break
    except Exception:                                            # noqa: BLE001 -- evidence only
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(dom_ms * 1e3, 2), "rocprof_avg_launch_us": prof_avg, "rocprof_source": prof_src,
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "all_kernels_us": {s: round(v[1] / v[0] * 1e3, 2) for s, v in kern.items()},
                "measured_in": f"a separate profiled loop of {prof_steps} steps ({dt_prof * 1e3:.4f} ms per step with an event "
                               f"pair around every kernel); `value` is from the loop with the timers off",
                "pmc_evidence": "profiles/r0N_pmc_summary.json (offline rocprofv3 --pmc passes of this command, tools/collect_pmc.sh)",
                "note": "integer-ALU bound (Fr products: DESIGN.md section 2 for the issue ceiling by opcode class).  A 2^20 pass is ONE "
                        "round of workgroups (r05: 512 of two columns, two per CU; r01 - r04: 256 of four columns, one per CU): load, "
                        "barrier-separated exchange steps and store of a workgroup overlap only with its one neighbour's, so about a "
                        "quarter of the launch is not arithmetic and an issue-ceiling fraction quoted for the pass is not a "
                        "whole-kernel efficiency"}

    # ------------------------------------------------------------------ the one JSON line, and a safety net for N > 1
    # The headline above needs no data-path collective; the legs below do when N > 1 (the sharded MSM and prover through
    # RCCL).  `legs` collects what has finished; with N > 1 a watchdog prints the line with what there is and ends the
    # process if the remaining legs do not finish in time (a collective that never returns has no other way out), so the
    # driver always gets its line.
    legs = {}

    def emit_line(note=None):
        out = {"metric": "bls12_381_fr_ntt_butterflies_per_s", "value": value, "unit": "butterflies/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
               "timing": timing,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, Fr)",
               "data": "synthetic",
               "config": {"workload": f"forward + inverse BLS12-381 Fr NTT, 2^{k} points, device resident, "
                                      f"natural order in/out, bit-exact vs oracle", "log_n": k, "passes": passes,
                          "parallelism": f"{world} independent polynomial(s), one per GPU"},
               "exchange": (None if world == 1 else "library RCCL communicator (pm_comm_init / pm_g1_allgather_fold)"
                            if native_comm else f"torch.distributed ({backend})"),
               "roofline": roofline, "cpu_baseline": legs.get("cpu"), "ntt_extra": legs.get("ntt_extra"),
               "ntt_fourstep": legs.get("fourstep"), "msm": legs.get("msm"), "msm_large": legs.get("msm_large"),
               "next_rows": legs.get("poly"), "prover": legs.get("prover")}
        if note:
            out["note"] = note
        return emitter.emit(out)

    state["emit"] = emit_line
    if dog:
        dog.extend(args.leg_timeout, "multi-GPU legs (sharded MSM / prover)")

    # ------------------------------------------------------------------ NTT extras (rank 0): SURVEY 8d
    ntt_extra = None
    if rank == 0 and not args.no_ntt_extra:
        # (i) the host-pointer ABI call (what a patched dusk-plonk `fft` makes): H2D + NTT + D2H
        host_out = np.empty_like(host_in)             # a touched buffer: first-touch page faults are the caller's
        ctx.fr_ntt(host_in, k, 0, out=host_out)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.fr_ntt(host_in, k, 0, out=host_out)
        e2e = (time.perf_counter() - t0) / reps
        assert np.array_equal(host_out, d_b.cpu().numpy().view(np.uint64)), "host-pointer NTT != device-resident NTT"
        # (i') a round's worth of host-resident polynomials: uploads, transforms and downloads overlap on
        # three streams (pm_fr_ntt_batch); the same call with the overlap switched off for comparison
        hb_in = np.ascontiguousarray(np.stack([host_in] * 4))
        hb_out = np.zeros_like(hb_in)                 # touched once: no first-touch faults in the timing
        pipe = {}
        for mode in (1, 0):
            ctx.set_option("ntt_pipeline", mode)
            ctx.fr_ntt_batch(hb_in, k, 0, out=hb_out)
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.fr_ntt_batch(hb_in, k, 0, out=hb_out)
            pipe[mode] = (time.perf_counter() - t0) / 3 / 4
            assert np.array_equal(hb_out[3], host_out), "batched host-pointer NTT differs"
        ctx.set_option("ntt_pipeline", 1)
        del hb_in, hb_out
        # (ii) the prover's shape: coset NTT of a 2^k-coefficient polynomial on the 4x domain
        d4 = torch.empty((4 * n, 4), dtype=torch.int64, device=dev)
        for _ in range(3):
            ctx.fr_ntt_dev(d_a.data_ptr(), n, d4.data_ptr(), k + 2, pa.NTT_COSET, stream=stream)
        local_sync()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.fr_ntt_dev(d_a.data_ptr(), n, d4.data_ptr(), k + 2, pa.NTT_COSET, stream=stream)
        local_sync()
        c4 = (time.perf_counter() - t0) / 20
        # (iii) the metric's second size: forward + inverse at 2^24, device resident
        big = None
        if k < 24:
            kb, nb = 24, 1 << 24
            xb = torch.from_numpy(oracle.fr_sample(0x504C4F4E4C, nb).view(np.int64)).to(dev)
            yb, zb = torch.empty_like(xb), torch.empty_like(xb)
            for _ in range(2):
                ctx.fr_ntt_dev(xb.data_ptr(), nb, yb.data_ptr(), kb, 0, stream=stream)
                ctx.fr_ntt_dev(yb.data_ptr(), nb, zb.data_ptr(), kb, INVERSE, stream=stream)
            local_sync()
            assert torch.equal(xb, zb), "iNTT(NTT(a)) != a at 2^24"
            ctx.profile(True)
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.fr_ntt_dev(xb.data_ptr(), nb, yb.data_ptr(), kb, 0, stream=stream)
                ctx.fr_ntt_dev(yb.data_ptr(), nb, zb.data_ptr(), kb, INVERSE, stream=stream)
            local_sync()
            tb = (time.perf_counter() - t0) / 10
            bprof = {nm: v for nm, v in ctx.profile_read().items() if nm.startswith("ntt_pass")}
            ctx.profile(False)
            bdom = max(bprof, key=lambda s_: bprof[s_][1] / bprof[s_][0])      # longest launch
            bdom_ms = bprof[bdom][1] / bprof[bdom][0]
            bpasses = pa.ntt_plan(kb)
            bs = bpasses[0] if bdom.endswith("first") else (bpasses[-1] if bdom.endswith("last") else bpasses[1])
            balgo = 64 * nb * bs / kb
            big_role = {"first": "false, true>", "last": "true, false>", "middle": "true, true>"}[bdom.rsplit("_", 1)[1]]
            big_traffic = pmc_traffic("ntt_pass", nb // 4, also=(big_role,), kinds=("pmc_big_summary",))
            big = {"log_n": kb, "ms_per_step": round(tb * 1e3, 3), "butterflies_per_s": nb * kb / tb,
                   "passes": bpasses,
                   "roofline": {"bound": "hbm", "kernel": bdom, "achieved": round(balgo / (bdom_ms * 1e-3) / 1e9, 2),
                                "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(balgo / (bdom_ms * 1e-3) / HBM_PEAK, 4),
                                "traffic": big_traffic[0], "traffic_source": big_traffic[1],
                                "avg_launch_us": round(bdom_ms * 1e3, 2),
                                "algorithmic_bytes_per_launch": int(balgo),
                                "all_kernels_us": {s_: round(v[1] / v[0] * 1e3, 2) for s_, v in bprof.items()},
                                "whole_transform_frac": round(2 * 64 * nb / tb / HBM_PEAK, 4),
                                "note": "every kernel timed (an event pair costs ~5 us per launch: <1 % at this size)"}}
            # (iv) SURVEY 8f N5 on one rank: the rank-split transform's own kernels (pack / unpack transposes + the
            # batched sub-transforms; the all-to-all degenerates to nothing), result equal to the plan's bit for bit
            ctx.fr_ntt_dev(xb.data_ptr(), nb, yb.data_ptr(), kb, 0, stream=stream)
            zb.copy_(xb)
            local_sync()
            ctx.fr_ntt_fourstep_dev(zb.data_ptr(), yb.data_ptr(), kb, 1, 0, 0)      # yb doubles as the staging buffer
            ctx.sync()
            ctx.fr_ntt_dev(xb.data_ptr(), nb, yb.data_ptr(), kb, 0, stream=stream)
            local_sync()
            four_ok = bool(torch.equal(zb, yb))
            stage = torch.empty_like(xb)
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.fr_ntt_fourstep_dev(zb.data_ptr(), stage.data_ptr(), kb, 1, 0, 0)
            ctx.sync()
            four_ms = (time.perf_counter() - t0) / 5 * 1e3
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.fr_ntt_fourstep_dev(zb.data_ptr(), stage.data_ptr(), kb, 1, 0, pa.NTT_TRANSPOSED)
            ctx.sync()
            four_t_ms = (time.perf_counter() - t0) / 5 * 1e3
            big["fourstep_world1"] = {"ms_per_transform": round(four_ms, 3),
                                      "ms_per_transform_transposed_order": round(four_t_ms, 3),
                                      "equals_single_gpu_plan": four_ok,
                                      "note": "pm_fr_ntt_fourstep_dev with world = 1: three transposes (two with PM_NTT_TRANSPOSED: the result stays in the block-transposed order a pointwise consumer accepts) and four sub-transform passes; over several GPUs each transpose adds one all-to-all (not measurable on one GPU)"}
            del xb, yb, zb, stage
        ntt_extra = {"fwd_inv_2^24": big, "pcie_inclusive": {"ms_per_transform": round(e2e * 1e3, 3), "butterflies_per_s": (n // 2) * k / e2e,
                                        "note": "pm_fr_ntt with pageable host buffers: H2D + transform + D2H; never `value`",
                                        "batch4_ms_per_transform": round(pipe[1] * 1e3, 3),
                                        "batch4_no_overlap_ms_per_transform": round(pipe[0] * 1e3, 3)},
                     "coset_4n": {"log_n": k + 2, "in_len": n, "us_per_transform": round(c4 * 1e6, 1),
                                  "butterflies_per_s": (2 * n) * (k + 2) / c4}}
        del d4

    legs.update(ntt_extra=ntt_extra)
    # ------------------------------------------------------------------ MSM legs
    k0, dd = 0x1234567, 0xabcdef123456789abcdef

    def run_msm(mk, steps, table):
        """KZG-commit shaped MSM of 2^mk points, sharded by points over the ranks."""
        mn = 1 << mk
        lo, hi = shard_range(mn, rank, world)
        k0_shard = ints_to_limbs([(k0 + lo * dd) % R_MOD], 4)[0]          # P_i = (k0 + i d) G, i in shard
        pts = oracle.g1_bases_arith(k0_shard, ints_to_limbs([dd], 4)[0], hi - lo, cores)
        full_sc = oracle.fr_sample(0x5343414C, mn)
        sc = full_sc[lo:hi]
        bases = pa.host.Bases(ctx, pts)
        t_pre = time.perf_counter()
        if table:                       # resident SRS (CommitKey): window multiples precomputed once
            bases.precompute()
        t_pre = time.perf_counter() - t_pre
        d_sc = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).to(dev)

        def msm_step():
            part = bases.msm_dev(d_sc.data_ptr(), hi - lo, stream=stream)
            if native_comm:                                      # the library's own RCCL communicator
                return ctx.g1_allgather_fold(part)[0]
            return allgather_fold(part, coll_dev if world > 1 else None)

        res = msm_step()
        barrier()
        # as for the headline: the timed loop runs with every timer off (an event pair is ~5 us of stream time, a dozen
        # kernels per MSM), the kernels' durations come from a profiled loop of the same step right after it
        t0 = time.perf_counter()
        for _ in range(steps):
            res = msm_step()
        barrier()
        mdt = max_over_ranks(time.perf_counter() - t0)
        ctx.profile(True)
        for _ in range(steps):
            msm_step()
        barrier()
        mprof = ctx.profile_read()
        ctx.profile(False)
        # parity inside the bench: discrete-log identity (bases are known multiples of G)
        dl = oracle.expected_dlog(full_sc, 0, ints_to_limbs([k0], 4)[0], ints_to_limbs([dd], 4)[0])
        ok = bool(np.array_equal(pa.g1_to_affine(res)[0], oracle.g1_mul(oracle.g1_generator(), dl)))
        acc_ms = mprof["msm_accumulate_l1"][1] / mprof["msm_accumulate_l1"][0]
        # the summary of the pass that ran THIS shape: 2^20 with the table in pmc_summary, 2^24 in pmc_big_summary
        msm_traffic = (None, None)
        if table and world == 1 and mk in (20, 24):
            msm_traffic = pmc_traffic("msm_accumulate_l1_kernel", kinds=("pmc_summary",) if mk == 20 else ("pmc_big_summary",))
        out = {"metric": "bls12_381_g1_msm_scalar_muls_per_s", "value": mn * steps / mdt,
               "unit": "scalar-muls/s", "points": mn, "ms_per_msm": mdt / steps * 1e3,
               "scaling": "strong" if world > 1 else None, "bit_exact_vs_oracle": ok,
               "srs_window_table": bool(table), "table_build_ms": round(t_pre * 1e3, 1) if table else None,
               "kernels_us": {s: round(v[1] / v[0] * 1e3, 1) for s, v in mprof.items()},
               "roofline": {"bound": "hbm", "kernel": "msm_accumulate_l1",
                            "achieved": round(128 * (hi - lo) / (acc_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK / 1e9,
                            "unit": "GB/s", "frac": round(128 * (hi - lo) / (acc_ms * 1e-3) / HBM_PEAK, 5),
                            "traffic": msm_traffic[0], "traffic_source": msm_traffic[1],
                            "traffic_note": "random 96-byte gathers from the window table: the x2 FETCH_SIZE correction is calibrated "
                                            "for coalesced streams, the truth lies between half this figure and it (DESIGN.md section 4.2)",
                            "avg_launch_us": round(acc_ms * 1e3, 1),
                            "algorithmic_bytes_per_launch": 128 * (hi - lo)}}
        if world > 1:
            # the exchange alone (2312-byte all-gather + fold), timed on every rank: max over ranks
            xt = None
            try:
                barrier()
                t0 = time.perf_counter()
                for _ in range(20):
                    if native_comm:
                        ctx.g1_allgather_fold(res.reshape(1, 18))
                    else:
                        allgather_fold(res, coll_dev)
                xt = max_over_ranks((time.perf_counter() - t0) / 20)
            except Exception as e:                                   # noqa: BLE001
                print(f"[bench] rank {rank}: exchange timing failed ({e})", file=sys.stderr)
            out["exchange_us"] = round(xt * 1e6, 1) if xt is not None else None
            print(f"[bench] rank {rank}: 2^{mk} MSM shard [{lo}, {hi}) {mdt / steps * 1e3:.3f} ms per MSM, exchange "
                  f"{out['exchange_us']} us", file=sys.stderr, flush=True)
        assert ok, "MSM result differs from the discrete-log identity"
        if table and world == 1 and not args.no_msm_extra:
            # what an 8-way point shard of this MSM costs on one GPU (its own window table, sized for the
            # shard): the measured basis of the 8-GPU projection -- time(full) / (time(shard) + exchange)
            sh_n = mn // 8
            sh_bases = pa.host.Bases(ctx, pts[:sh_n])
            sh_bases.precompute()
            sh_bases.msm_dev(d_sc.data_ptr(), sh_n, stream=stream)
            barrier()
            ctx.profile(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                sh_bases.msm_dev(d_sc.data_ptr(), sh_n, stream=stream)
            barrier()
            sdt = (time.perf_counter() - t0) / steps
            sprof = ctx.profile_read()
            ctx.profile(False)
            sh_bases.free()
            tail = sum(v[1] / v[0] for s_, v in sprof.items() if s_ in ("msm_bucket_chunk", "msm_window_sum", "msm_accumulate_ln"))
            out["shard_1_of_8"] = {"points": sh_n, "ms_per_msm": sdt * 1e3,
                                   "kernels_us": {s_: round(v[1] / v[0] * 1e3, 1) for s_, v in sprof.items()},
                                   "tail_frac": round(tail * 1e-3 / sdt, 3),
                                   "note": "tail = partial-list levels + bucket reduction + window sums"}
            # NOT a measurement of 8 GPUs: the one-GPU time over (the measured time of a 1/8 shard + the measured
            # time of the library's exchange on a ONE-rank communicator: staging copies + ncclAllGather + fold)
            exch = None
            try:
                with stdout_to_stderr():
                    ctx.comm_init(0, 1)
                    ctx.g1_allgather_fold(res.reshape(1, 18))
                t0 = time.perf_counter()
                for _ in range(20):
                    ctx.g1_allgather_fold(res.reshape(1, 18))
                exch = (time.perf_counter() - t0) / 20
            except Exception as e:                                   # noqa: BLE001
                print(f"[bench] one-rank communicator unavailable ({e})", file=sys.stderr)
            finally:
                try:
                    ctx.comm_destroy()
                except Exception:                                    # noqa: BLE001
                    pass
            if exch is not None:
                out["projection"] = {"scaling_8_gpus": round((mdt / steps) / (sdt + exch), 2),
                                     "is_a_measurement": False,
                                     "assumptions": "8 ranks each run a 1/8 point shard with its own window table (shard_1_of_8, "
                                                    "measured on this GPU), then one pm_g1_allgather_fold; the exchange is priced "
                                                    "at its measured ONE-rank cost (no xGMI hop, one message instead of eight)",
                                     "exchange_us_world1": round(exch * 1e6, 1)}
        if table and world == 1 and mk == args.msm_log_n:
            # the drop-in signature itself (CommitKey::commit / msm_variable_base over a resident SRS): pm_g1_msm with the
            # scalars in pageable HOST memory -- H2D of 32 N bytes + MSM + the 144-byte result back ("scalars
            # H2D counted in the end-to-end figure"); never `value`
            h_sc = np.ascontiguousarray(sc)
            r_h = bases.msm(h_sc)
            assert np.array_equal(pa.g1_to_affine(r_h)[0], pa.g1_to_affine(res)[0]), "host-scalar MSM != device-resident MSM"
            t0 = time.perf_counter()
            for _ in range(steps):
                bases.msm(h_sc)
            hdt = (time.perf_counter() - t0) / steps
            out["pcie_inclusive"] = {"ms_per_msm": round(hdt * 1e3, 3), "value": mn / hdt,
                                     "note": "pm_g1_msm: scalars in pageable host memory (32 N bytes H2D) + MSM + result D2H, "
                                             "SRS and its window table resident; never `value`"}
        if table and world == 1 and mk <= 20 and not args.no_msm_extra:
            # "witness-like" scalars: 90 % below 2^16, 5 % zero, 1 % one -- bucket skew and shortcuts
            rs = np.random.default_rng(0x5343414C)
            wl = full_sc.copy()
            u_ = rs.random(mn)
            small = oracle.fr_to_mont(np.concatenate([rs.integers(0, 1 << 16, size=(mn, 1), dtype=np.uint64),
                                                      np.zeros((mn, 3), np.uint64)], axis=1))
            wl[u_ < 0.90] = small[u_ < 0.90]
            wl[(u_ >= 0.90) & (u_ < 0.95)] = 0
            wl[(u_ >= 0.95) & (u_ < 0.96)] = oracle.fr_to_mont(ints_to_limbs([1], 4))[0]
            d_wl = torch.from_numpy(np.ascontiguousarray(wl).view(np.int64)).to(dev)
            rw = bases.msm_dev(d_wl.data_ptr(), mn, stream=stream)
            dlw = oracle.expected_dlog(wl, 0, ints_to_limbs([k0], 4)[0], ints_to_limbs([dd], 4)[0])
            okw = bool(np.array_equal(pa.g1_to_affine(rw)[0], oracle.g1_mul(oracle.g1_generator(), dlw)))
            assert okw, "witness-like MSM differs from the discrete-log identity"
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                bases.msm_dev(d_wl.data_ptr(), mn, stream=stream)
            barrier()
            wdt = (time.perf_counter() - t0) / steps
            out["witness_like"] = {"ms_per_msm": wdt * 1e3, "value": mn / wdt, "bit_exact_vs_oracle": okw,
                                   "note": "90 % of scalars < 2^16, 5 % zero, 1 % one, rest uniform"}
            # a prover round: 4 wire polynomials committed in one pass over the same SRS
            kb = 4
            sc4 = np.concatenate([sc] + [oracle.fr_sample(0x5343414D + j, mn) for j in range(1, kb)])
            d_sc4 = torch.from_numpy(np.ascontiguousarray(sc4).view(np.int64)).to(dev)
            r4 = bases.msm_batch_dev(d_sc4.data_ptr(), mn, kb)
            assert np.array_equal(r4[0], res), "batched commit differs from the single one"
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                bases.msm_batch_dev(d_sc4.data_ptr(), mn, kb)
            barrier()
            bdt = time.perf_counter() - t0
            out["batch4"] = {"ms_per_msm": bdt / steps / kb * 1e3, "value": mn * kb * steps / bdt,
                             "note": "4 scalar vectors per call over the resident SRS (pm_g1_msm_batch_dev)"}
        bases.free()
        return out, pts, sc

    msm = msm_large = None
    if not args.no_msm:
        msm_plain, pts, sc = run_msm(args.msm_log_n, args.msm_steps, False)   # msm_variable_base: bases as given
        msm, _, _ = run_msm(args.msm_log_n, args.msm_steps, True)             # CommitKey::commit: resident SRS
        msm["without_table"] = {k: msm_plain[k] for k in ("value", "ms_per_msm", "kernels_us")}
        if args.msm_large_log_n > args.msm_log_n:
            msm_large, _, _ = run_msm(args.msm_large_log_n, 2, True)

    legs.update(msm=msm, msm_large=msm_large)
    # ------------------------------------------------------------------ optional: one transform over all ranks (N5)
    fourstep = None
    if world > 1 and args.fourstep_log_n and native_comm:
        fk = args.fourstep_log_n
        fblk = (1 << fk) // world
        fx = torch.from_numpy(oracle.fr_sample(0x4E35 + rank, fblk).view(np.int64)).to(dev)
        fstage = torch.empty((2 * fblk, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize(dev)
        for _ in range(2):
            ctx.fr_ntt_fourstep_dev(fx.data_ptr(), fstage.data_ptr(), fk, world, rank, 0)
        ctx.sync()
        barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            ctx.fr_ntt_fourstep_dev(fx.data_ptr(), fstage.data_ptr(), fk, world, rank, 0)
        ctx.sync()
        barrier()
        fdt = max_over_ranks(time.perf_counter() - t0) / 10
        fourstep = {"log_n": fk, "ms_per_transform": round(fdt * 1e3, 3), "butterflies_per_s": (1 << (fk - 1)) * fk / fdt,
                    "scaling": "strong", "exchange": "grouped ncclSend / ncclRecv inside the library, three all-to-alls per transform"}
        del fx, fstage

    # ------------------------------------------------------------------ next rows (N1/N2 helpers), rank 0
    poly = None
    if rank == 0 and not args.no_poly:
        pk = 22
        pn = 1 << pk
        va = pa.DeviceVector.from_host(ctx, oracle.fr_sample(11, pn))
        vb = pa.DeviceVector.from_host(ctx, oracle.fr_sample(12, pn))
        vo = pa.DeviceVector(ctx, pn)
        pt = oracle.fr_sample(13, 1)[0]
        P, lib, h = pa.Polynomial, ctx._lib, ctx._h
        import ctypes as C
        pp = pt.ctypes.data_as(C.POINTER(C.c_uint64))
        ev = np.zeros(4, np.uint64)

        def timed(fn, reps=5):
            fn()
            ctx.sync()
            ctx.profile(True)
            for _ in range(reps):
                fn()
            ctx.sync()
            pr = ctx.profile_read()
            ctx.profile(False)
            (name, (cnt, ms)), = pr.items()
            return ms / reps * 1e-3

        rows = {
            "vec_add": (lambda: lib.pm_fr_vec_op_dev(h, 0, va._p, vb._p, pn, vo._p, pn, None), 96 * pn),
            "vec_mul": (lambda: lib.pm_fr_vec_op_dev(h, 2, va._p, vb._p, pn, vo._p, pn, None), 96 * pn),
            "poly_evaluate": (lambda: lib.pm_fr_poly_evaluate_dev(h, va._p, pn, pp, ev.ctypes.data_as(C.POINTER(C.c_uint64)), None), 32 * pn),
            "poly_ruffini": (lambda: lib.pm_fr_poly_ruffini_dev(h, va._p, pn, pp, vo._p, None), 64 * pn),
            "prefix_product": (lambda: lib.pm_fr_prefix_product_dev(h, va._p, pn, vo._p, None), 64 * pn),
            "batch_inverse": (lambda: lib.pm_fr_batch_inverse_dev(h, vo._p, pn, None), 64 * pn),
        }
        poly = {"n": pn, "note": "SURVEY 8f rows N1/N2 helpers; algorithmic bytes = operands read once + result written once"}
        for name, (fn, nbytes) in rows.items():
            sec = timed(fn)
            poly[name] = {"us": round(sec * 1e6, 1), "GB/s": round(nbytes / sec / 1e9, 1),
                          "hbm_frac": round(nbytes / sec / HBM_PEAK, 4)}
        # the two scans at the prover's own size (2^20: the one-pass look-back applies there, the 2^22 rows above use the
        # three-stage scan)
        qn = 1 << 20
        at20 = {}
        for name, fn in (("poly_ruffini", lambda: lib.pm_fr_poly_ruffini_dev(h, va._p, qn, pp, vo._p, None)),
                         ("prefix_product", lambda: lib.pm_fr_prefix_product_dev(h, va._p, qn, vo._p, None))):
            sec = timed(fn)
            at20[name] = {"us": round(sec * 1e6, 1), "hbm_frac": round(64 * qn / sec / HBM_PEAK, 4)}
        poly["at_2^20"] = at20
        for v in (va, vb, vo):
            v.free()

    # ------------------------------------------------------------------ full prove (N1 + N2)
    # N = 1: one GPU proves.  N > 1 (BASELINE configs[4] shape): every rank runs the rounds, the 11 MSMs
    # are split by coefficient range over the ranks' SRS shards, partial points all-gathered and folded.
    legs.update(fourstep=fourstep, poly=poly)
    prover = None
    if not args.no_prover and msm is not None and args.prover_log_n == args.msm_log_n:
        from plonk_prototype_amd.dist import ShardedCommitKey
        gk = args.prover_log_n
        gn = 1 << gk
        circuit, wit, pub = pa.synthetic.chain_circuit(gn, 1)
        tau = 0x1F2E3D4C5B6A79788796A5B4C3D2E1F00112233445566778899AABBCCDDEEFF % R_MOD
        srs_ms = None
        if world == 1:
            # a real powers-of-tau key, generated on the GPU (PublicParameters::setup), so that the opening
            # equations can be checked at full size below
            t0 = time.perf_counter()
            ck = pa.CommitKey.setup(gn - 1, pa.field.fr_to_limbs(tau), ctx)
            srs_ms = (time.perf_counter() - t0) * 1e3
            ck._bases.precompute()
        else:
            # the SRS split over the ranks, partial commitments exchanged by the library's own RCCL
            # communicator (pm_comm_init / pm_g1_allgather_fold); PM_BENCH_BACKEND=gloo rehearsals keep torch's
            native = native_comm
            ck = ShardedCommitKey(pts, shard_range(gn, rank, world)[0], gn, ctx, device=coll_dev, precompute=True,
                                  native=native)
        t0 = time.perf_counter()
        pkey = pa.preprocess(circuit, ctx, ck)               # key polynomials + the verifier key's 15 commitments
        ctx.sync()
        t_pre = time.perf_counter() - t0
        d_wit = pa.DeviceVector.from_host(ctx, wit.reshape(-1, 4))
        pub_sparse = pa.prover.sparse_public_inputs(pub)
        proof = pa.prove(pkey, ck, d_wit, pub_sparse)            # warm-up + the checked proof
        pub_z = pa.field.fr_from_limbs(oracle.fr_poly_evaluate(oracle.fr_ntt(pub, gk, INVERSE, cores),
                                                               pa.field.fr_to_limbs(proof.challenges["z"])))
        ident_ok = bool(pa.prover.check_identity(proof, gn, pub_z))
        fl, fi = pa.field.fr_to_limbs, pa.field.fr_from_limbs
        G1 = oracle.g1_generator()
        kzg_ok = None
        coeffs = pa.DeviceVector(ctx, gn)
        if world == 1:
            # commit(a) = [a(tau)] G, and the KZG equation of the shifted opening witness in the exponent:
            # [W_zw] (tau - z w) + F_s(z w) G - sum_i aw'^i [w_i(tau)] G = [z]   (F_s = z + aw' a + aw'^2 b + aw'^3 d),
            # with a(tau), b(tau), d(tau) evaluated on the device from the witness
            w_tau = []
            for j in (0, 1, 3):
                ctx.fr_ntt_dev(d_wit.ptr + 32 * j * gn, gn, coeffs.ptr, gk, pa.NTT_INVERSE)
                w_tau.append(fi(ctx.fr_evaluate(coeffs.ptr, gn, fl(tau))))
            comm_ok = bool(np.array_equal(proof.commitments["a"], oracle.g1_mul(G1, ints_to_limbs([w_tau[0]], 4)[0])))
            ev = {k_: fi(v_) for k_, v_ in proof.evaluations.items()}
            zw = proof.challenges["z"] * fi(pa.domain_info(gk)[0]) % R_MOD
            aws = proof.challenges["aw_shifted"]
            sh_eval = (ev["z_next"] + aws * ev["a_next"] + aws ** 2 * ev["b_next"] + aws ** 3 * ev["d_next"]) % R_MOD
            rest = (sh_eval - sum(pow(aws, i + 1, R_MOD) * w for i, w in enumerate(w_tau))) % R_MOD
            lhs = oracle.g1_add(oracle.g1_mul(proof.commitments["w_zw"], ints_to_limbs([(tau - zw) % R_MOD], 4)[0]),
                                oracle.g1_mul(G1, ints_to_limbs([rest], 4)[0]))
            kzg_ok = bool(np.array_equal(lhs, proof.commitments["z"]))
            assert kzg_ok, "opening witness fails the KZG equation"
        else:
            k0l, ddl = ints_to_limbs([k0], 4)[0], ints_to_limbs([dd], 4)[0]
            dl = oracle.expected_dlog(oracle.fr_ntt(wit[0], gk, INVERSE, cores), 0, k0l, ddl)
            comm_ok = bool(np.array_equal(proof.commitments["a"], oracle.g1_mul(G1, dl)))
        coeffs.free()
        assert ident_ok and comm_ok, "prover output fails the verifier identity / commitment check"
        # prove() returns when its last commitment is on the host, so each proof is timed on its own and
        # the median is reported: the ROCm runtime reclaims the previous legs' multi-GB frees in
        # the background and that one-off ~35 ms stall would otherwise be averaged into the proofs
        reps = 9
        barrier()
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            pa.prove(pkey, ck, d_wit, pub_sparse)
            times.append(time.perf_counter() - t0)
        barrier()
        pdt = max_over_ranks(float(np.median(times)))
        pmean = max_over_ranks(float(np.mean(times)))
        ctx.profile(True)
        pa.prove(pkey, ck, d_wit, pub_sparse)
        pprof = ctx.profile_read()
        ctx.profile(False)
        grp = group_kernels(pprof)
        q_ms = pprof["plonk_quotient"][1] / pprof["plonk_quotient"][0]
        q_bytes = 19 * 32 * 4 * gn                      # 18 operands read + 1 result written per coset point
        q_traffic = pmc_traffic("quotient_kernel", kinds=("pmc_prover_summary",)) if gk == 20 else (None, None)
        # two proofs in flight: a second context (own stream, own key and workspace) proving from a second host
        # thread over the same resident SRS -- one proof's NTT / quotient / opening phases and low-occupancy MSM
        # tails run under the other's accumulate kernels
        # the same proof with EVERY vector split over the ranks by coefficient range (pm_plonk_*_dist): N = 1
        # prices the decomposition itself (four size-n sub-coset transforms per coset form, each through the four-step
        # transform with one rank); N > 1 is the real thing -- the library's RCCL all-gather and all-to-all
        dist_leg = None
        # what has finished so far goes into the line's legs NOW: if the distributed leg below hangs in its first real all-to-all
        # (N > 1: never executed before the first multi-GPU run), the watchdog's line still carries the sharded prover's figures
        legs["prover"] = {"gates": gn, "ms_per_proof": round(pdt * 1e3, 2), "gates_per_s": gn / pdt, "n_gpus": world,
                          "entry_point": "pm_plonk_prove" if world == 1 else "pm_plonk_prove_sharded",
                          "kernel_ms": {k_: round(v_, 3) for k_, v_ in grp.items()},
                          "verifier_identity_holds": ident_ok, "commitment_matches_dlog": comm_ok,
                          "note": "partial entry: the legs after this one (pm_plonk_prove_dist, two contexts, ...) had not finished"}
        if not args.no_dist_prover and gn % (world * world) == 0:
            # Every phase that holds collectives runs under try / except on every rank and is followed by an agreement
            # (all_reduce MIN, which every rank reaches): a rank whose phase failed never leaves its peers waiting in a
            # barrier it skipped -- either all ranks go on, or all record the error.
            def all_ok(ok_):
                return agree(dist, ok_, coll_dev) if world > 1 else ok_
            dkey = d_wsl = dproof = None
            err, d_times, t_dpre = None, [], 0.0
            try:
                from plonk_prototype_amd.dist import DistGroup
                dgrp = DistGroup(native=(world > 1 and native_comm), device=dev)
                m_ = gn // world
                t0 = time.perf_counter()
                dkey = pa.prover.DistProverKey(circuit, ctx, dgrp)
                dkey.commit(ck._bases)
                ctx.sync()
                t_dpre = time.perf_counter() - t0
                d_wsl = pa.DeviceVector.from_host(ctx, np.ascontiguousarray(wit[:, dgrp.rank * m_:(dgrp.rank + 1) * m_]).reshape(-1, 4))
                dproof = dkey.prove(ck._bases, d_wsl, pub_sparse)
            except Exception as e:                                   # noqa: BLE001
                err = f"setup: {e}"
            if all_ok(err is None):
                barrier()
                try:
                    ctx.comm_stats(reset=True)
                    for _ in range(5):
                        t0 = time.perf_counter()
                        dkey.prove(ck._bases, d_wsl, pub_sparse)
                        d_times.append(time.perf_counter() - t0)
                    dstats = {k_: v_ // 5 for k_, v_ in ctx.comm_stats().items()}
                except Exception as e:                               # noqa: BLE001
                    err = f"timed proofs: {e}"
                if all_ok(err is None):
                    barrier()
                    dist_leg = {"ms_per_proof": round(max_over_ranks(float(np.median(d_times))) * 1e3, 2), "world": world,
                                "equals_the_replicated_prover_byte_for_byte": dproof.to_bytes() == proof.to_bytes(),
                                "device_bytes_per_rank": dkey.device_bytes, "preprocess_ms": round(t_dpre * 1e3, 1),
                                "exchanges_per_proof": dict(dstats, note="pm_comm_stats: transpose_steps = all-to-all CALLS a proof makes "
                                                            "as soon as world > 1 (r04: 102), allgather_calls = 2312-byte message all-gathers; "
                                                            "time over xGMI is NOT measured on one GPU"),
                                "exchange": ("library RCCL communicator" if world > 1 and native_comm else
                                             f"torch.distributed ({backend})" if world > 1 else "none (one rank)"),
                                "note": "pm_plonk_prove_dist: rows / coefficients [rank n / world, (rank + 1) n / world) of every vector per "
                                        "rank, transforms as four-step NTTs over the ranks, nothing replicated"}
            if dist_leg is None:
                dist_leg = {"error": err or "a peer rank failed"}
            try:
                if d_wsl is not None:
                    d_wsl.free()
                if dkey is not None:
                    dkey.free()
            except Exception:                                        # noqa: BLE001
                pass
        two_ms = None
        if world == 1:
            import threading
            ctx2 = pa.Context(local_rank)
            ck2 = pa.CommitKey.__new__(pa.CommitKey)
            ck2.__dict__.update(ck.__dict__)
            pk2 = pa.preprocess(circuit, ctx2, ck2)
            d_wit2 = pa.DeviceVector.from_host(ctx2, wit.reshape(-1, 4))
            two_ok = pa.prove(pk2, ck2, d_wit2, pub_sparse).to_bytes() == proof.to_bytes()
            assert two_ok, "the second context's proof differs"
            per = 6

            def worker(pk_, wit_):
                for _ in range(per):
                    pa.prove(pk_, ck, wit_, pub_sparse)
            th = [threading.Thread(target=worker, args=a_) for a_ in ((pkey, d_wit), (pk2, d_wit2))]
            barrier()
            ctx2.sync()
            t0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            ctx2.sync()
            barrier()
            two_ms = (time.perf_counter() - t0) / (2 * per) * 1e3
            d_wit2.free()
            pk2.free()
            ctx2.close()
        prover = {"workload": f"full PLONK prove, 2^{gk}-gate synthetic arithmetic circuit (4 wires, copy permutation, "
                              f"1 public input): 5 rounds, 11 commitments, 16 openings, Merlin transcript seeded with the "
                              f"verifier key",
                  "gates": gn, "ms_per_proof": round(pdt * 1e3, 2), "gates_per_s": gn / pdt,
                  "timing": f"median of {reps} proofs (mean {pmean * 1e3:.2f} ms, max {max(times) * 1e3:.2f} ms)",
                  "entry_point": "pm_plonk_prove (one C-ABI call)" if world == 1 else "pm_plonk_prove_sharded (one C-ABI call per rank)",
                  "two_contexts_ms_per_proof": round(two_ms, 2) if two_ms else None,
                  "distributed": dist_leg,
                  "n_gpus": world, "scaling": "strong" if world > 1 else None,
                  "parallelism": ("one GPU" if world == 1 else
                                  f"rounds replicated on {world} ranks, every MSM split by coefficient range, "
                                  f"144-byte partial points all-gathered ("
                                  + ("RCCL inside the library: pm_g1_allgather_fold" if native_comm else "torch.distributed")
                                  + ") and folded"),
                  "kernel_ms": {k_: round(v_, 3) for k_, v_ in grp.items()},
                  "kernel_ms_total": round(sum(grp.values()), 2), "preprocess_ms": round(t_pre * 1e3, 1),
                  "quotient_roofline": {"bound": "hbm", "kernel": "plonk_quotient", "unit": "GB/s",
                                        "achieved": round(q_bytes / (q_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9,
                                        "frac": round(q_bytes / (q_ms * 1e-3) / HBM_PEAK, 4),
                                        "algorithmic_bytes_per_launch": q_bytes, "traffic": q_traffic[0],
                                        "traffic_source": q_traffic[1]},
                  "verifier_identity_holds": ident_ok, "commitment_matches_dlog": comm_ok,
                  "kzg_opening_equation_holds": kzg_ok, "srs_setup_ms": round(srs_ms, 1) if srs_ms else None,
                  "inputs": "witness resident in HBM; proving key, verifier key and SRS table resident"}
        d_wit.free()
        if world == 1:
            # every gate kind: the same size with all 11 selector polynomials present (wide_mixed_circuit: half the
            # rows arithmetic, blocks of rows under each widget selector), i.e. the quotient and linearisation of a
            # real dusk-plonk circuit -- the arithmetic-only circuit above skips the widget arithmetic
            pkey.free()
            m_circ, m_wit, _ = pa.synthetic.wide_mixed_circuit(gn, ctx, 3)
            m_key = pa.preprocess(m_circ, ctx, ck)
            del m_circ
            m_proof = pa.prove(m_key, ck, m_wit, None)
            m_times = []
            for _ in range(5):
                ctx.sync()
                t0 = time.perf_counter()
                pa.prove(m_key, ck, m_wit, None)
                m_times.append(time.perf_counter() - t0)
            ctx.profile(True)
            pa.prove(m_key, ck, m_wit, None)
            mprof = ctx.profile_read()
            ctx.profile(False)
            prover["all_gate_kinds"] = {
                "workload": f"2^{gk} gates, all 11 selector polynomials present (arithmetic, range, logic, fixed-base and variable-base rows)",
                "ms_per_proof": round(float(np.median(m_times)) * 1e3, 2),
                "quotient_ms": round(mprof["plonk_quotient"][1] / mprof["plonk_quotient"][0], 3),
                "verifier_identity_holds": bool(pa.prover.check_identity(m_proof, gn, 0))}
            m_wit.free()
            m_key.free()
            # small circuits: latency of one proof (a chain of four batched MSM calls of ~0.5 ms each, not a throughput
            # number); 2^12 is the domain of BASELINE configs[0], the reference's own CPU-runnable circuit
            small = {}
            for sk_ in (12, 16):
                if sk_ >= gk:
                    continue
                sn_ = 1 << sk_
                c_s, w_s, p_s = pa.synthetic.chain_circuit(sn_, 5)
                ck_s = pa.CommitKey(pts[:sn_], ctx, precompute=True)
                key_s = pa.preprocess(c_s, ctx, ck_s)
                dw_s = pa.DeviceVector.from_host(ctx, w_s.reshape(-1, 4))
                pub_s = pa.prover.sparse_public_inputs(p_s)
                pr_s = pa.prove(key_s, ck_s, dw_s, pub_s)
                pz_s = pa.field.fr_from_limbs(oracle.fr_poly_evaluate(oracle.fr_ntt(p_s, sk_, INVERSE, cores),
                                                                      pa.field.fr_to_limbs(pr_s.challenges["z"])))
                assert pa.prover.check_identity(pr_s, sn_, pz_s), "small proof fails the verifier identity"
                for _ in range(8):                      # the clock settles over the first few proofs after an idle spell
                    pa.prove(key_s, ck_s, dw_s, pub_s)
                ts_ = []
                for _ in range(15):
                    t0 = time.perf_counter()
                    pa.prove(key_s, ck_s, dw_s, pub_s)
                    ts_.append(time.perf_counter() - t0)
                small[f"2^{sk_}"] = round(float(np.median(ts_)) * 1e3, 3)
                dw_s.free()
                key_s.free()
            prover["latency_ms_by_gates"] = dict(small, note="median of 15 proofs each after 8 untimed ones; 2^12 = the domain of BASELINE configs[0]")
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # the same rounds on the host cores: the C restatement composed by oracle/cpu_prover.py, on a
            # bounded sample (a 2^16-gate circuit), outputs compared with a GPU proof of that circuit
            from oracle import cpu_prover as CP
            ck_ = min(gk, args.cpu_prover_log_n)
            cn = 1 << ck_
            c_circ, c_wit, c_pub = pa.synthetic.chain_circuit(cn, 2)
            c_srs = pts[:cn]
            g_ck = pa.CommitKey(c_srs, ctx)
            g_proof = pa.prove(pa.preprocess(c_circ, ctx, g_ck), g_ck, c_wit, c_pub)
            cpk = CP.preprocess(oracle, {k_: getattr(c_circ, k_) for k_ in CP.SELECTORS}, c_circ.sigma_index, cores)
            t0 = time.perf_counter()
            c_out = CP.prove(oracle, cpk, c_srs, c_wit, c_pub, g_proof.challenges, cores)
            ct = time.perf_counter() - t0
            same = all(np.array_equal(g_proof.commitments[k_], v_) for k_, v_ in c_out["commitments"].items()) and \
                all(np.array_equal(g_proof.evaluations[k_], v_) for k_, v_ in c_out["evaluations"].items())
            assert same, "GPU proof differs from the CPU restatement's proof"
            # the reference's own default is ONE thread (default-features = false: no rayon, ref:Cargo.toml:19): the same
            # prover on one thread, on a smaller sample (2^14 gates) so that the leg stays within its time budget
            s1k = min(gk, 14)
            s_circ, s_wit, s_pub = pa.synthetic.chain_circuit(1 << s1k, 3)
            s_srs = pts[:1 << s1k]
            s_ck = pa.CommitKey(s_srs, ctx)
            s_proof = pa.prove(pa.preprocess(s_circ, ctx, s_ck), s_ck, s_wit, s_pub)
            spk = CP.preprocess(oracle, {k_: getattr(s_circ, k_) for k_ in CP.SELECTORS}, s_circ.sigma_index, 1)
            t0 = time.perf_counter()
            CP.prove(oracle, spk, s_srs, s_wit, s_pub, s_proof.challenges, 1)
            st1 = time.perf_counter() - t0
            prover["cpu_baseline"] = {"value": cn / ct, "unit": "gates/s", "cores": cores, "kind": "port",
                                      "ms_per_proof": round(ct * 1e3, 1), "bit_exact_vs_gpu": bool(same),
                                      "sample": f"one 2^{ck_}-gate proof (5 rounds, 11 Pippenger MSMs, radix-2 NTTs) with "
                                                f"the C restatement on {cores} threads",
                                      "single_thread_value": (1 << s1k) / st1,
                                      "single_thread_sample": f"one 2^{s1k}-gate proof on 1 thread ({st1:.1f} s) -- the reference's "
                                                              f"default build has no rayon",
                                      "why_not_full_size": None if ck_ == gk else f"the 2^{gk}-gate proof takes ~{ct * (1 << (gk - ck_)):.0f} s on "
                                                           f"{cores} threads: outside the bounded-sample budget; pass --cpu-prover-log-n {gk}"}

    legs.update(prover=prover)
    # ------------------------------------------------------------------ BASELINE configs[4]: one 2^24-gate proof
    # N = 1 (the anchor of any scaling curve of the prover): pm_plonk_prove, and the same proof through pm_plonk_prove_dist
    # with one rank (coefficient-range ownership, four-step transforms: what N > 1 runs), byte-compared; SRS = real powers
    # of tau from pm_g1_fixed_base_mul_dev, so the commitment is checked in the exponent.  N > 1: pm_plonk_prove_dist over
    # the ranks (every vector split by coefficient range, the SRS slice generated on each rank's own GPU).  No CPU leg at
    # this size (a 2^24-gate proof is ~8 minutes on 16 threads).
    lk = args.prover_large_log_n
    if prover is not None and lk > args.prover_log_n and (1 << lk) % (world * world) == 0:
        from plonk_prototype_amd.dist import DistGroup, ShardedCommitKey
        if dog:
            dog.extend(args.leg_timeout, f"2^{lk}-gate proof leg")
        ln = 1 << lk
        fl, fi = pa.field.fr_to_limbs, pa.field.fr_from_limbs
        tau = 0x1F2E3D4C5B6A79788796A5B4C3D2E1F00112233445566778899AABBCCDDEEFF % R_MOD

        def all_ok(ok_):
            return agree(dist, ok_, coll_dev) if world > 1 else ok_

        def timed_proofs(fn, reps_):
            ts_ = []
            for _ in range(reps_):
                t0_ = time.perf_counter()
                fn()
                ts_.append(time.perf_counter() - t0_)
            return ts_

        try:
            ck._bases.free()                                   # the 2^20 leg's key and table: not needed any more
        except Exception:                                      # noqa: BLE001
            pass
        ctx.trim()
        hbm_free0 = torch.cuda.mem_get_info(dev)[0]
        t0 = time.perf_counter()
        l_circ, l_wit, _ = pa.synthetic.wide_circuit(ln, ctx, 7)     # every rank builds the same circuit (seeded)
        t_circ = time.perf_counter() - t0
        large = {"workload": f"full PLONK prove, 2^{lk}-gate synthetic arithmetic circuit (synthetic.wide_circuit: 4 wires, copy "
                             f"cycles across three columns), SRS = powers of tau generated on the GPU",
                 "gates": ln, "n_gpus": world, "circuit_synthesis_host_s": round(t_circ, 1)}
        lproof = None
        a_tau_ok = None
        if world == 1:
            t0 = time.perf_counter()
            lck = pa.CommitKey.setup(ln - 1, fl(tau), ctx)
            ctx.sync()
            l_srs_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            lck._bases.precompute()
            ctx.sync()
            l_tab_ms = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            lkey = pa.preprocess(l_circ, ctx, lck)
            ctx.sync()
            l_pre = time.perf_counter() - t0
            lproof = pa.prove(lkey, lck, l_wit, None)
            l_bytes = hbm_free0 - torch.cuda.mem_get_info(dev)[0]
            l_ident = bool(pa.prover.check_identity(lproof, ln, 0))
            # commit(a) = [a(tau)] G: a(tau) from the device (iNTT of the a column, Horner at tau), one scalar-mul on the host
            co = pa.DeviceVector(ctx, ln)
            ctx.fr_ntt_dev(l_wit.ptr, ln, co.ptr, lk, pa.NTT_INVERSE)
            a_tau = ctx.fr_evaluate(co.ptr, ln, fl(tau))
            co.free()
            a_tau_ok = bool(np.array_equal(lproof.commitments["a"], oracle.g1_mul(oracle.g1_generator(), ints_to_limbs([fi(a_tau)], 4)[0])))
            assert l_ident and a_tau_ok, "2^24-gate proof fails the verifier identity / the commitment check"
            l_times = timed_proofs(lambda: pa.prove(lkey, lck, l_wit, None), 3)
            ctx.profile(True)
            pa.prove(lkey, lck, l_wit, None)
            lprof = ctx.profile_read()
            ctx.profile(False)
            lkey.free()
            large["replicated"] = {"entry_point": "pm_plonk_prove (one C-ABI call)",
                                   "ms_per_proof": round(float(np.median(l_times)) * 1e3, 1),
                                   "gates_per_s": ln / float(np.median(l_times)),
                                   "timing": f"median of 3 proofs (min {min(l_times) * 1e3:.1f}, max {max(l_times) * 1e3:.1f} ms)",
                                   "kernel_ms": {k_: round(v_, 2) for k_, v_ in group_kernels(lprof).items()},
                                   "kernels_ms": {k_: round(v_[1], 2) for k_, v_ in sorted(lprof.items(), key=lambda kv: -kv[1][1])[:8]},
                                   "device_bytes": int(l_bytes),
                                   "device_bytes_note": "HBM in use by this process after the first proof minus before the circuit: "
                                                        "SRS + window table + prover key + workspace + witness (hipMemGetInfo)",
                                   "preprocess_ms": round(l_pre * 1e3, 1), "srs_setup_ms": round(l_srs_ms, 1),
                                   "srs_window_table_ms": round(l_tab_ms, 1),
                                   "verifier_identity_holds": l_ident, "commitment_matches_a_of_tau": a_tau_ok}
            l_bases = lck._bases
        else:
            m_l = ln // world
            lck = ShardedCommitKey.setup(ln, tau, rank * m_l, (rank + 1) * m_l, ctx, precompute=True, native=native_comm,
                                         device=coll_dev)
            l_bases = lck._bases
        prover["large"] = large                               # (the watchdog's line carries what has finished: legs["prover"] is this dict)
        dl_leg, err, dkey, d_wsl, dproof, d_times, t_dpre, dstats = None, None, None, None, None, [], 0.0, {}
        if not args.no_dist_prover:
            try:
                dgrp = DistGroup(native=(world > 1 and native_comm), device=dev)
                m_l = ln // world
                t0 = time.perf_counter()
                dkey = pa.prover.DistProverKey(l_circ, ctx, dgrp)
                dkey.commit(l_bases)
                ctx.sync()
                t_dpre = time.perf_counter() - t0
                if world == 1:
                    d_wsl = l_wit
                else:
                    w_all = l_wit.to_host().reshape(4, ln, 4)
                    d_wsl = pa.DeviceVector.from_host(ctx, np.ascontiguousarray(w_all[:, rank * m_l:(rank + 1) * m_l]).reshape(-1, 4))
                    del w_all
                dproof = dkey.prove(l_bases, d_wsl, None)
            except Exception as e:                                   # noqa: BLE001
                err = f"setup: {e}"
            if all_ok(err is None):
                barrier()
                try:
                    ctx.comm_stats(reset=True)
                    d_times = timed_proofs(lambda: dkey.prove(l_bases, d_wsl, None), 3)
                    dstats = {k_: v_ // 3 for k_, v_ in ctx.comm_stats().items()}
                except Exception as e:                               # noqa: BLE001
                    err = f"timed proofs: {e}"
                if all_ok(err is None):
                    barrier()
                    d_med = max_over_ranks(float(np.median(d_times)))
                    dl_leg = {"entry_point": "pm_plonk_prove_dist (one C-ABI call per rank)", "world": world,
                              "ms_per_proof": round(d_med * 1e3, 1), "gates_per_s": ln / d_med,
                              "device_bytes_per_rank": dkey.device_bytes, "preprocess_ms": round(t_dpre * 1e3, 1),
                              "exchanges_per_proof": dstats,
                              "exchange": ("library RCCL communicator" if world > 1 and native_comm else
                                           f"torch.distributed ({backend})" if world > 1 else "none (one rank)"),
                              "verifier_identity_holds": bool(pa.prover.check_identity(dproof, ln, 0)),
                              "equals_the_replicated_prover_byte_for_byte":
                                  (dproof.to_bytes() == lproof.to_bytes()) if lproof is not None else None}
                    if world > 1:
                        # the replicated proof is not run at N > 1; the commitment is checked in the exponent instead
                        co = pa.DeviceVector(ctx, ln)
                        ctx.fr_ntt_dev(l_wit.ptr, ln, co.ptr, lk, pa.NTT_INVERSE)
                        a_tau = ctx.fr_evaluate(co.ptr, ln, fl(tau))
                        co.free()
                        dl_leg["commitment_matches_a_of_tau"] = bool(np.array_equal(
                            dproof.commitments["a"], oracle.g1_mul(oracle.g1_generator(), ints_to_limbs([fi(a_tau)], 4)[0])))
            if dl_leg is None:
                dl_leg = {"error": err or "a peer rank failed"}
            try:
                if d_wsl is not None and d_wsl is not l_wit:
                    d_wsl.free()
                if dkey is not None:
                    dkey.free()
            except Exception:                                        # noqa: BLE001
                pass
        large["distributed"] = dl_leg
        prover["large"] = large
        l_wit.free()
        try:
            l_bases.free()
        except Exception:                                            # noqa: BLE001
            pass
        del l_circ
        ctx.trim()

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1)
    legs.update(prover=prover)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu_oracle = CpuOracle(native=True)       # rebuild with -march=native for this host
        except Exception:
            cpu_oracle = oracle
        x = host_in.copy()
        t0 = time.perf_counter()
        f1 = cpu_oracle.fr_ntt(x, k, 0, 1)
        cpu_oracle.fr_ntt(f1, k, INVERSE, 1)
        t1 = time.perf_counter() - t0
        reps, tm = 0, 0.0
        while tm < 6.0 and reps < 50:
            t0 = time.perf_counter()
            f = cpu_oracle.fr_ntt(x, k, 0, cores)
            cpu_oracle.fr_ntt(f, k, INVERSE, cores)
            tm += time.perf_counter() - t0
            reps += 1
        assert np.array_equal(f, d_b.cpu().numpy().view(np.uint64)), "GPU NTT != CPU restatement"
        cpu = {"value": butterflies_per_step * reps / tm, "unit": "butterflies/s", "cores": cores,
               "kind": "port",
               "sample": f"{reps} x (forward + inverse 2^{k} NTT), C restatement of dusk-plonk serial_fft/"
                         f"parallel_fft (oracle/c, gcc -O3), {cores} threads; 1 thread: "
                         f"{butterflies_per_step / t1:.3e} butterflies/s",
               "single_thread_value": butterflies_per_step / t1}
        if msm is not None:
            sk = args.msm_log_n
            sn = 1 << sk
            t0 = time.perf_counter()
            cpu_oracle.g1_msm(pts[:sn], sc[:sn], 0, cores)
            tmsm = time.perf_counter() - t0
            cpu["msm_value"] = sn / tmsm
            cpu["msm_unit"] = "scalar-muls/s"
            cpu["msm_sample"] = f"one 2^{sk}-point Pippenger MSM (c rule of the reference), {cores} threads"
            s1n = min(sn, 1 << 17)                         # one thread = the reference's default build; bounded sample
            t0 = time.perf_counter()
            cpu_oracle.g1_msm(pts[:s1n], sc[:s1n], 0, 1)
            t1msm = time.perf_counter() - t0
            cpu["msm_single_thread_value"] = s1n / t1msm
            cpu["msm_single_thread_sample"] = f"one 2^{s1n.bit_length() - 1}-point Pippenger MSM on 1 thread ({t1msm:.1f} s)"

    legs.update(cpu=cpu, ntt_extra=ntt_extra, fourstep=fourstep, msm=msm, msm_large=msm_large, poly=poly, prover=prover)
    if dog:
        # the line goes out with the watchdog still armed; close / destroy then run under a short deadline of their own, and a
        # teardown that hangs AFTER the complete line is out ends the rank with code 0
        dog.extend(90, "teardown (context close + process group destroy)", exit_code=0)
    if rank == 0:
        emit_line()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if dog:
        dog.finish()


if __name__ == "__main__":
    main()
