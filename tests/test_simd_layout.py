"""QEX's SIMD field memory <-> the library's host format (qex_amd/csrc/simd_layout.cpp, CPU only).

The only code between a QEX `Field` and the C ABI is the layout conversion of the shim (qex_amd/nim/qexhip.nim toHost / fromHost /
toHostG, after src/quda/qudaWrapperImpl.nim:198-260: l.coord -> lo1.rankIndex per site).  It is restated in C
(layoutSetupQ / layoutIndexQ / layoutCoordQ, src/layout/qlayout.nim:10-66,110-185; default inner geometry
src/layout/layoutX.nim:19-42,98-111) and held here against
  * its own inverse on every site (the check layoutCoordQ itself ends with, qlayout.nim:177-185),
  * the independent map of oracle/cpu_simd (bench.py's CPU baseline lays QEX's V = 8 AoSoA fields out by it),
  * the closed form of the default V = 8 geometry {1,2,2,2},
  * coordinate-coded fields through the data movers, both directions, vector and gauge forms."""
import ctypes as C

import numpy as np
import pytest


def _lib():
    import qex_amd

    return qex_amd.lib()


def _i4(v):
    return (C.c_int * 4)(*[int(x) for x in v])


def default_inner(lat, V):
    out = (C.c_int * 4)()
    rc = _lib().qexhip_layout_default_inner(_i4(lat), V, out)
    return rc, list(out)


def simd_map(lat, inner):
    m = np.zeros(int(np.prod(lat)), dtype=np.int32)
    rc = _lib().qexhip_layout_simd_map(_i4(lat), _i4(inner), m.ctypes.data_as(C.POINTER(C.c_int)))
    return rc, m


def v1_coords(lat):
    """coordinates of every site in V=1 even-odd order (x fastest inside a parity block; layoutIndexQ at innerGeom 1)"""
    vol = int(np.prod(lat))
    lex = np.arange(vol)
    c = np.zeros((vol, 4), dtype=np.int64)
    r = lex.copy()
    for i in range(4):
        c[:, i] = r % lat[i]
        r //= lat[i]
    idx = lex // 2 + (c.sum(1) % 2) * (vol // 2)
    out = np.zeros_like(c)
    out[idx] = c
    return out


def test_default_inner_geometry_is_newLayoutX():
    assert default_inner([8, 8, 8, 8], 8) == (0, [1, 2, 2, 2])            # layoutX.nim:19-42: t, z, y are halved in turn
    assert default_inner([32, 32, 32, 32], 8) == (0, [1, 2, 2, 2])
    assert default_inner([48, 48, 48, 12], 8) == (0, [2, 2, 2, 1])        # dist = 1 splits the longest unsplit extents first
    assert default_inner([16, 8, 8, 4], 8) == (0, [2, 2, 2, 1])
    assert default_inner([8, 8, 8, 8], 1) == (0, [1, 1, 1, 1])
    assert default_inner([8, 8, 8, 8], 4) == (0, [1, 1, 2, 2])
    assert default_inner([8, 8, 8, 8], 16) == (0, [2, 2, 2, 2])
    # the fix-up of layoutX.nim:98-111: an odd outer extent moves its split to an unsplit direction with a factor 4
    assert default_inner([8, 6, 6, 6], 8)[0] != 0                           # ... and gives up where there is none
    assert default_inner([12, 12, 12, 4], 8) == (0, [2, 2, 2, 1])
    assert default_inner([4, 6, 10, 6], 8)[0] != 0                         # QEX: "can't lay out inner geom"
    assert default_inner([6, 6, 6, 6], 32)[0] != 0                         # "not enough 2's"
    assert default_inner([8, 8, 8, 8], 3)[0] != 0


def test_odd_local_extents_are_refused():
    """With an odd local extent the parity that splits a QEX field into halves depends on the rank's origin (qlayout.nim:133-185 counts
    GLOBAL coordinates): a rank at an odd coordinate would be permuted wrongly by a map built from local coordinates.  The entry
    points are given no origin, so they refuse instead of guessing (sharded handles have even local extents anyway)."""
    for lat in ([3, 4, 4, 4], [4, 4, 4, 5], [6, 3, 4, 4]):
        rc, _ = simd_map(lat, [1, 1, 1, 1])
        assert rc < 0, lat
    assert simd_map([6, 4, 4, 2], [1, 1, 1, 1])[0] == 0


@pytest.mark.parametrize("lat,inner", [([8, 8, 8, 8], [1, 2, 2, 2]), ([4, 8, 12, 8], [1, 2, 2, 2]), ([16, 8, 8, 4], [2, 2, 2, 1]),
                                       ([8, 4, 6, 4], [2, 2, 1, 1]), ([4, 6, 10, 6], [1, 1, 1, 1]), ([8, 8, 8, 8], [2, 2, 2, 2]),
                                       ([16, 16, 16, 32], [1, 2, 2, 2]), ([12, 4, 4, 4], [1, 2, 2, 2]),
                                       ([6, 4, 4, 2], [2, 1, 1, 1])])     # the last one: odd outer extent, inner checkerboard shift (innerCb = 1 along t)
def test_map_is_a_permutation_and_matches_the_closed_form(lat, inner):
    rc, m = simd_map(lat, inner)
    assert rc == 0
    vol = int(np.prod(lat))
    assert np.array_equal(np.sort(m), np.arange(vol))
    V = int(np.prod(inner))
    if V == 1:
        assert np.array_equal(m, np.arange(vol))                           # the host format IS the V = 1 layout
    outer = [lat[i] // inner[i] for i in range(4)]
    if all(o % 2 == 0 for o in outer):
        # closed form (no inner checkerboard shift): lane = lex of (c // outer) over innerGeom, outer index = lex(c % outer) // 2
        # (+ half for odd sites); qlayout.nim:110-131
        c = v1_coords(lat)                                                  # coordinates by V=1 index
        k = c // outer
        o = c % outer
        lane = k[:, 0] + inner[0] * (k[:, 1] + inner[1] * (k[:, 2] + inner[2] * k[:, 3]))
        olex = o[:, 0] + outer[0] * (o[:, 1] + outer[1] * (o[:, 2] + outer[2] * o[:, 3]))
        nouter = int(np.prod(outer))
        oidx = olex // 2 + (c.sum(1) % 2) * (nouter // 2)
        simd_of_v1 = oidx * V + lane
        assert np.array_equal(m[simd_of_v1], np.arange(vol))


@pytest.mark.parametrize("lat", [[8, 8, 8, 8], [4, 8, 12, 8], [16, 16, 16, 32]])
def test_map_against_the_independent_cpu_simd_layout(lat):
    """oracle/cpu_simd builds its V = 8 even-odd AoSoA map on its own (lay_init / build_map) and is itself checked against the
    oracle's operator (tests/test_cpu_simd.py): the product's restatement must name the same (outer, lane) for every site"""
    from oracle import cpu_simd as cs

    L = cs.Lattice(lat, np.zeros((int(np.prod(lat)), 4, 3, 3, 2)))
    simd_of_v1 = L.site_map()
    rc, m = simd_map(lat, [1, 2, 2, 2])
    assert rc == 0 and np.array_equal(m[simd_of_v1], np.arange(len(m)))


@pytest.mark.parametrize("lat,inner", [([8, 8, 8, 8], [1, 2, 2, 2]), ([16, 8, 8, 4], [2, 2, 2, 1]), ([8, 4, 6, 4], [2, 2, 1, 1])])
def test_data_movers_on_coordinate_coded_fields(lat, inner):
    """tests/base/tshift.nim's device of a field whose value IS its coordinate: every number must land on the site it names"""
    L = _lib()
    vol, V = int(np.prod(lat)), int(np.prod(inner))
    c = v1_coords(lat)
    code = c[:, 0] + 100 * c[:, 1] + 10000 * c[:, 2] + 1000000 * c[:, 3]
    rc, m = simd_map(lat, inner)
    assert rc == 0
    # vector: v1[site][colour][re|im] = code + colour/10 + im/100
    v1 = code[:, None, None] + np.arange(3)[None, :, None] / 10.0 + np.arange(2)[None, None, :] / 100.0
    simd = np.zeros((vol // V, 3, 2, V))
    assert L.qexhip_layout_vec_v1_to_simd(_i4(lat), _i4(inner), v1.ctypes.data, simd.ctypes.data) == 0
    for i in (0, 1, V, vol // 2 + 3, vol - 1):
        assert np.array_equal(simd[i // V, :, :, i % V], v1[m[i]])
    back = np.zeros_like(v1)
    assert L.qexhip_layout_vec_simd_to_v1(_i4(lat), _i4(inner), simd.ctypes.data, back.ctypes.data) == 0
    assert np.array_equal(back, v1)
    sites = np.arange(vol)
    assert np.array_equal(simd[sites // V, 1, 1, sites % V], v1[m, 1, 1])      # every site, one component
    # gauge: four fields [outer][3][3][2][V] <-> [site][4][3][3][2]
    g1 = code[:, None, None, None, None] + np.arange(4)[None, :, None, None, None] / 8.0 + np.arange(3)[None, None, :, None, None] / 64.0 \
        + np.arange(3)[None, None, None, :, None] / 512.0 + np.arange(2)[None, None, None, None, :] / 4096.0
    gs = [np.zeros((vol // V, 3, 3, 2, V)) for _ in range(4)]
    ptrs = (C.c_void_p * 4)(*[a.ctypes.data for a in gs])
    assert L.qexhip_layout_gauge_v1_to_simd(_i4(lat), _i4(inner), g1.ctypes.data, ptrs) == 0
    for mu in range(4):
        assert np.array_equal(gs[mu][sites // V, 2, 1, 0, sites % V], g1[m, mu, 2, 1, 0])
    gb = np.zeros_like(g1)
    assert L.qexhip_layout_gauge_simd_to_v1(_i4(lat), _i4(inner), ptrs, gb.ctypes.data) == 0
    assert np.array_equal(gb, g1)


def test_errors():
    L = _lib()
    m = np.zeros(8 ** 4, dtype=np.int32)
    assert L.qexhip_layout_simd_map(_i4([8, 8, 8, 8]), _i4([1, 3, 2, 2]), m.ctypes.data_as(C.POINTER(C.c_int))) != 0
    assert b"multiple" in L.qexhip_last_error()
    # an inner split with an odd outer extent and no even unsplit direction to carry the checkerboard shift (qlayout.nim:31)
    assert L.qexhip_layout_simd_map(_i4([6, 6, 6, 6]), _i4([2, 2, 2, 2]), m.ctypes.data_as(C.POINTER(C.c_int))) != 0
    assert L.qexhip_layout_vec_simd_to_v1(_i4([8, 8, 8, 8]), _i4([1, 2, 2, 2]), None, None) != 0
