"""Host logic of the MSM bucket fill (csrc/msm_sort.hip.h, make_geom / part_of / part_range), through the library's
pure-host test hook -- no GPU: for every shape the library accepts, the layout fits the kernels' fixed resources and the
bucket -> partition map is a monotone cover whose inverse (first bucket, width) agrees with it."""
import ctypes as C

import numpy as np
import pytest

MAX_BINS, MAX_RBITS, LDS_BYTES, TILE2_PAIRS = 4096, 12, 160 * 1024, 16384


@pytest.fixture(scope="module")
def lib():
    from plonk_prototype_amd import _lib
    return _lib.load()


def _geom(lib, n, c, table_c, batch, maps=False):
    out = (C.c_uint32 * 16)()
    rc = lib.pm_test_msm_geometry(n, c, table_c, batch, out, None, None, None)
    assert rc == 0
    keys = ("c", "nwin", "nsets", "bbits", "pbits", "rbits", "pps", "bins", "np", "ts", "tiles", "lds_scatter", "lds_local",
            "na", "sa")
    g = dict(zip(keys, list(out)))
    if maps:
        part = np.zeros(1 << g["bbits"], np.uint32)
        first, width = np.zeros(g["pps"], np.uint32), np.zeros(g["pps"], np.uint32)
        u32p = C.POINTER(C.c_uint32)
        assert lib.pm_test_msm_geometry(n, c, table_c, batch, out, part.ctypes.data_as(u32p), first.ctypes.data_as(u32p),
                                        width.ctypes.data_as(u32p)) == 0
        g.update(part=part, first=first, width=width)
    return g


def _shapes():
    for log_n in (0, 3, 8, 10, 12, 15, 17, 18, 20, 22, 24, 26):
        for n in {1 << log_n, (1 << log_n) + 1, max(1, (1 << log_n) - 3)}:
            yield n, 0, 0, 1                                   # no table, the library's width
            for c in range(8, 25):
                yield n, 0, c, 1                               # window table of every width
            for c in (4, 5, 8, 13, 16):
                yield n, c, 0, 1                               # explicit width without a table
            for batch in (2, 4, 15):
                yield n, 0, 0, batch
                yield n, 0, (13 if log_n <= 15 else 16 if log_n <= 18 else 20 if log_n <= 22 else 22), batch


def test_layout_fits_the_kernels_and_the_partition_map_is_a_cover(lib):
    checked = 0
    for n, c, table_c, batch in _shapes():
        g = _geom(lib, n, c, table_c, batch)
        tag = (n, c, table_c, batch, g)
        assert g["c"] == (table_c or c or g["c"]) and 2 <= g["c"] <= 24, tag
        assert g["nwin"] == -(-256 // g["c"]) and g["nsets"] == (1 if table_c else g["nwin"]), tag
        assert g["bbits"] == g["c"] - 1 and g["pbits"] + g["rbits"] == g["bbits"], tag
        assert g["bins"] == g["nsets"] * g["pps"] and g["np"] == batch * g["bins"], tag
        # what msm_piece refuses (PM_ERR_BAD_ARG) instead of launching: more bins or local bits than the kernels hold
        if g["bins"] > MAX_BINS or g["rbits"] > MAX_RBITS or g["ts"] == 0:
            continue
        assert g["lds_scatter"] <= LDS_BYTES and g["lds_local"] <= LDS_BYTES, tag
        assert 1 <= g["ts"] <= 1024 and g["ts"] * g["nwin"] <= 14 * 1024, tag    # a scalar's digits fit the staged tile
        assert g["tiles"] == -(-n // g["ts"]), tag
        assert g["pps"] >= 1 << g["pbits"] and (g["na"] == 0) == (g["sa"] == 0), tag
        if g["bbits"] > 16:
            continue                                            # the maps of the wide windows are checked below on a sample
        m = _geom(lib, n, c, table_c, batch, maps=True)
        part, first, width = m["part"], m["first"], m["width"]
        assert part[0] == 0 and part[-1] == g["pps"] - 1 and np.all(np.diff(part.astype(np.int64)) >= 0), tag
        assert np.all(np.diff(part.astype(np.int64)) <= 1), tag                   # no partition without a bucket
        sizes = np.bincount(part, minlength=g["pps"])
        assert np.array_equal(sizes, 1 << width.astype(np.int64)), tag            # part_range's width is the bucket count
        assert np.array_equal(first, np.concatenate([[0], np.cumsum(sizes)[:-1]])), tag
        assert np.all(width <= g["rbits"]) and np.all(width <= MAX_RBITS), tag    # local bins fit the LDS histogram
        checked += 1
    assert checked > 300


@pytest.mark.parametrize("n,table_c", [(1 << 20, 20), (1 << 20, 22), (1 << 24, 22), (1 << 24, 24), (1 << 17, 16)])
def test_low_buckets_of_a_short_top_window_get_finer_partitions(lib, n, table_c):
    """One bucket set for all windows (table mode): the short top window loads only the low 2^(top bits - 1) buckets, so
    those partitions are cut finer by 2^sa -- every partition then expects about the same number of pairs."""
    g = _geom(lib, n, 0, table_c, 1, maps=True)
    nwin, c = g["nwin"], g["c"]
    top_bits = 256 - c * (nwin - 1)
    part, width = g["part"], g["width"]
    # expected pairs per bucket: (nwin - 1) uniform windows + the top window over its 2^(top_bits - 1) low buckets
    load = np.full(1 << g["bbits"], (nwin - 1) / (1 << g["bbits"]))
    if top_bits < c:
        load[: 1 << (top_bits - 1)] += 1.0 / (1 << (top_bits - 1))
    per_part = np.bincount(part, weights=load, minlength=g["pps"])
    assert per_part.max() / per_part.mean() < 1.35, (g["na"], g["sa"], per_part.max() / per_part.mean())
    assert per_part.mean() * n <= TILE2_PAIRS or n >= 1 << 22                     # single-tile partitions up to 2^20
    if g["sa"]:
        assert np.all(width[: g["na"]] == g["rbits"] - g["sa"]) and np.all(width[g["na"]:] == g["rbits"])


def _sizing(lib, n, batch=1, c=0, table_c=0, cus=0):
    out = (C.c_uint64 * 4)()
    assert lib.pm_test_msm_sizing(n, batch, c, table_c, cus, out) == 0
    return C.c_int64(out[0]).value, int(out[1]), int(out[2]), int(out[3])


def test_sizing_pass_accepts_every_shape_and_stays_proportionate(lib):
    """The library's own sizing pass (msm_piece without a workspace), on a context that never touches a device: every
    shape up to 2^27 points is accepted, the workspace is what the layout says -- pairs three times over (partitioned,
    sorted keys, sorted values) plus 32 B of integer scalar per point, 256 B per bucket, the partial lists -- and grows
    with n (without a table the width follows n)."""
    n_checked = 0
    for table_c in [0] + list(range(8, 25)):
        for batch in (1, 2, 4, 15):
            prev = 0
            for n in sorted({(1 << k) + d for k in (0, 1, 5, 10, 13, 16, 17, 20, 22, 24, 26, 27) for d in (0, 7)}):
                if True:
                    g = _geom(lib, n, 0, table_c, batch)
                    if g["bins"] > MAX_BINS or g["rbits"] > MAX_RBITS:
                        assert _sizing(lib, n, batch, 0, table_c)[0] == -1, (n, batch, table_c)   # refused, not launched
                        continue
                    rc, ws, pinned, pairs = _sizing(lib, n, batch, 0, table_c)
                    tag = (n, batch, table_c, rc, ws, pinned, pairs)
                    assert rc == 0 and pairs == n * batch * g["nwin"], tag
                    buckets = batch * g["nsets"] << g["bbits"]
                    assert 16 * pairs + 256 * buckets <= ws <= 21 * pairs + 32 * n * batch + 300 * buckets + (64 << 20), tag
                    # 7 + 5 levels sequences of <= 4 entries per set, or one record per set when the device applies the weights itself
                    assert 256 * batch * g["nsets"] <= pinned <= 22 * 4 * 256 * batch * g["nsets"], tag
                    # (the chunk length steps with the grid's rounds, and -- r05 -- small grids take the chunk a latency model picks:
                    # the per-thread head slots follow the thread count)
                    assert ws >= 0.6 * prev or not table_c, tag
                    prev = ws
                    n_checked += 1
    assert n_checked > 1500
    # fewer CUs (a partitioned device): still accepted, same pair count
    assert _sizing(lib, 1 << 20, 4, 0, 20, 64)[0] == 0
    assert _sizing(lib, 0)[0:2] == (0, 0)
