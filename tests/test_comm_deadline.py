"""A dead peer inside an exchange must end in an error code, not in a hang (VERDICT r05 "next" item 5; SURVEY.md section 5
"failure detection").  CPU tests of the deadline logic of the library's guarded exchanges (csrc/comm.hip: ExchangeWatch +
guarded_exchange, the code path of pm_g1_allgather_fold and of the all-to-all) against a STUB RCCL table -- an
ncclAllGather that blocks until ncclCommAbort is called on its communicator, i.e. a peer that never arrives.  What cannot be
tested here: the real ncclCommAbort over xGMI (one GPU, no second rank)."""
import ctypes as C
import time

import plonk_prototype_amd as pa

PM_OK, PM_ERR_EXCHANGE = 0, -7


def _run(timeout_ms, peer_answers):
    lib = pa.load()
    first, second, aborts, elapsed = C.c_int(99), C.c_int(99), C.c_int(-1), C.c_long(-1)
    err = C.create_string_buffer(512)
    t0 = time.monotonic()
    rc = lib.pm_test_comm_deadline(timeout_ms, peer_answers, C.byref(first), C.byref(elapsed), C.byref(second), C.byref(aborts),
                                   err, len(err))
    assert rc == PM_OK
    return first.value, elapsed.value, second.value, aborts.value, err.value.decode(), time.monotonic() - t0


def test_a_peer_that_never_arrives_ends_in_pm_err_exchange_within_the_deadline():
    first, elapsed, second, aborts, err, wall = _run(300, 0)
    assert first == PM_ERR_EXCHANGE and aborts == 1
    assert 250 <= elapsed < 3000 and wall < 5              # the deadline, not for ever
    assert "comm_timeout_ms" in err and "ncclCommAbort" in err
    assert second == PM_ERR_EXCHANGE                       # the communicator is dead: the second exchange fails at once,
    #                                                        without entering the stub again (the hook returns -100 if it did)


def test_exchanges_that_complete_are_left_alone():
    first, elapsed, second, aborts, err, _ = _run(2000, 1)
    assert (first, second, aborts, err) == (PM_OK, PM_OK, 0, "") and elapsed < 1000


def test_without_the_option_nothing_is_armed():
    first, _, second, aborts, _, _ = _run(0, 1)
    assert (first, second, aborts) == (PM_OK, PM_OK, 0)


def test_the_deadline_can_be_armed_again_and_again():
    for _ in range(3):
        first, elapsed, _, aborts, _, _ = _run(100, 0)
        assert first == PM_ERR_EXCHANGE and aborts == 1 and 80 <= elapsed < 2000
