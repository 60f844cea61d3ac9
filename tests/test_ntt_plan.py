"""Host logic of the transform plan (csrc/ntt.hip, make_plan), through the library's pure-host test hook -- no GPU: for
every domain size EvaluationDomain::new accepts and every value the tunables take, the passes multiply up to the domain,
each has a kernel, and a workgroup fits a CU."""
import ctypes as C

import pytest

LDS_BYTES, MAX_THREADS = 160 * 1024, 1024


@pytest.fixture(scope="module")
def lib():
    from plonk_prototype_amd import _lib
    return _lib.load()


def _plan(lib, log_n, batch=1, tile_log=0, max_radix=0, radix=0, num_cus=0):
    out = (C.c_uint32 * 20)()
    rc = lib.pm_test_ntt_plan(log_n, batch, tile_log, max_radix, radix, num_cus, out)
    if rc:
        return rc, None
    v = list(out)
    return 0, dict(npass=v[0], S=[v[1 + 4 * i] for i in range(v[0])], LT=[v[2 + 4 * i] for i in range(v[0])],
                   threads=[v[3 + 4 * i] for i in range(v[0])], lds=[v[4 + 4 * i] for i in range(v[0])], have=v[17],
                   glog=v[18])


def _check(p, log_n, tag):
    if log_n < 3:
        assert p["npass"] == 0, tag                               # direct evaluation, one thread per output
        return
    assert 1 <= p["npass"] <= 4 and sum(p["S"]) == log_n, tag
    assert p["have"] == (1 << p["npass"]) - 1, tag                 # every pass has an instantiated kernel
    for S, LT, threads, lds in zip(p["S"], p["LT"], p["threads"], p["lds"]):
        assert 3 <= S <= 10 and 0 <= LT and S + LT <= 12 and S + LT <= log_n + 0, tag
        assert 64 <= threads <= MAX_THREADS and lds <= LDS_BYTES, tag
        assert p["npass"] == 1 or lds >= (36 << (S + LT)) or lds == 0, tag      # the tile (9 limbs per element) sits in LDS
    if p["npass"] > 1:
        assert min(p["S"]) >= 5 and max(p["S"]) - min(p["S"]) <= 1, tag          # balanced radices: 2^20 = 2^10 x 2^10
        assert p["glog"] in (2, 3) and all(LT >= 0 for LT in p["LT"]), tag


def test_default_plan_for_every_domain_size(lib):
    for log_n in range(0, 32):
        for batch in (1, 4, 15):
            rc, p = _plan(lib, log_n, batch)
            assert rc == 0
            _check(p, log_n, (log_n, batch, p))
    assert _plan(lib, 32)[0] == -2 and _plan(lib, 40)[0] == -2       # PM_ERR_DOMAIN_TOO_LARGE, as EvaluationDomain::new
    rc, p = _plan(lib, 20)
    # BASELINE configs[1]: two radix-2^10 passes, two columns per workgroup = two 512-thread workgroups per CU (r05; r01 - r04: four columns, one)
    assert p["S"] == [10, 10] and p["LT"] == [1, 1] and p["threads"] == [512, 512]
    assert _plan(lib, 24)[1]["S"] == [8, 8, 8] and _plan(lib, 30)[1]["S"] == [10, 10, 10]
    # fewer tiles than CUs: narrower tiles (2^18: 2^9 x 2^9 with two columns)
    p18 = _plan(lib, 18)[1]
    assert (1 << 18) >> (p18["S"][0] + p18["LT"][0]) >= 256


def test_every_tunable_value_gives_a_runnable_plan(lib):
    n = 0
    for log_n in range(3, 32):
        for radix in (4, 8):
            for max_radix in range(6, 11):
                for tile_log in (0, 10, 11, 12):
                    for num_cus in (256, 64):
                        rc, p = _plan(lib, log_n, 1, tile_log, max_radix, radix, num_cus)
                        assert rc == 0
                        _check(p, log_n, (log_n, radix, max_radix, tile_log, num_cus, p))
                        # ntt_max_radix bounds the radix unless that would need passes below 2^5 (no such kernels) or more
                        # than four passes (ntt_max_radix 6 above 2^24 once overran the plan's arrays: found by this test)
                        assert max(p["S"]) <= max_radix or p["npass"] == min(-(-log_n // max_radix), log_n // 5, 4) or log_n <= 10, \
                            (log_n, max_radix, p)
                        n += 1
    assert n == 29 * 2 * 5 * 4 * 2
