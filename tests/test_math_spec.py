"""The product's arithmetic spec header (scvod_math.h, compiled for the CPU) against glibc -- what
the reference calls -- and against the oracle.  Not gpu."""
import ctypes as C

import numpy as np


def _cmp32(spec, y, x):
    fb = C.c_long(-1)
    return spec.spec_cmp_atan2f(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_long(len(y)), C.byref(fb))


def test_atan2f_bit_identical_to_glibc(spec):
    rng = np.random.default_rng(1)
    n = 2_000_000
    sets = [
        (rng.uniform(-80, 80, n), rng.uniform(-80, 80, n)),
        (rng.uniform(-5, 5, n), rng.uniform(0, 80, n)),                       # getAzimuth(z, dis)
        (rng.choice([-1, 1], n) * np.exp(rng.uniform(-40, 40, n)), rng.choice([-1, 1], n) * np.exp(rng.uniform(-40, 40, n))),
    ]
    for y, x in sets:
        assert _cmp32(spec, y.astype(np.float32), x.astype(np.float32)) == 0
    bits = lambda: rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    assert _cmp32(spec, bits(), bits()) == 0
    assert _cmp32(spec, bits(), np.ones(n, np.float32)) == 0                   # atanf path
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, 0.4375, 0.6875, 1.1875, 2.4375,
                     2.0**25, 2.0**-29, 1.5, 30.0, 80.0], np.float32)
    yy, xx = np.meshgrid(edge, edge)
    assert _cmp32(spec, yy.ravel().copy(), xx.ravel().copy()) == 0


def test_atan2_f64_within_one_ulp_of_glibc(spec):
    rng = np.random.default_rng(2)
    n = 2_000_000
    y = rng.uniform(-80, 80, n).astype(np.float32).astype(np.float64)
    x = rng.uniform(-80, 80, n).astype(np.float32).astype(np.float64)
    worst = spec.spec_cmp_atan2(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_long(n))
    assert worst <= 1.0
    # exact directions land exactly on sector boundaries: both must agree bit for bit there
    for yy, xx in [(0.0, 1.0), (1.0, 0.0), (1.0, 1.0), (0.0, -1.0), (-1.0, 0.0), (-1.0, -1.0), (2.5, 2.5), (-3.0, 3.0)]:
        assert spec.spec_atan2(yy, xx) == np.arctan2(yy, xx)


def test_svd3_spec_equals_oracle(spec, oracle):
    rng = np.random.default_rng(3)
    for k in range(2000):
        if k % 4 == 0:   # nearly planar cloud (the Patchwork case)
            p = rng.normal(0, [3.0, 2.0, 0.02], (50, 3))
        elif k % 4 == 1:
            p = rng.normal(0, 1, (20, 3)) * rng.uniform(0.01, 30)
        elif k % 4 == 2:  # rank deficient
            p = np.outer(rng.normal(0, 1, 30), rng.normal(0, 1, 3))
        else:
            p = rng.normal(0, 1, (10, 3)) + rng.uniform(-60, 60, 3)
        cov = np.cov(p.T, bias=True).astype(np.float32)
        sv = np.zeros(3, np.float32)
        U = np.zeros(9, np.float32)
        c = np.ascontiguousarray(cov.reshape(9))
        spec.spec_svd3(c.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p), U.ctypes.data_as(C.c_void_p))
        osv, oU = oracle.svd3(cov)
        assert np.array_equal(sv.view(np.uint32), osv.view(np.uint32))
        assert np.array_equal(U.view(np.uint32), oU.reshape(9).view(np.uint32))
        # and it is an SVD: singular values descending, U orthonormal, cov ~= U S U^T up to sign
        assert sv[0] >= sv[1] >= sv[2] >= 0
        Um = U.reshape(3, 3).astype(np.float64)
        assert np.allclose(Um.T @ Um, np.eye(3), atol=1e-5)
        assert np.allclose(np.sort(np.linalg.svd(cov.astype(np.float64), compute_uv=False))[::-1], sv, rtol=2e-4,
                           atol=2e-6 * max(1.0, float(sv[0])))


def test_apri_spec_equals_oracle_binning(spec, oracle, scvod):
    rng = np.random.default_rng(4)
    for preset in ("semantickitti", "parkinglot", "os128_fine"):
        P = scvod.make_params(preset)
        dims = oracle.grid_dims(P)
        x = rng.uniform(-45, 45, (20000, 4)).astype(np.float32)
        x[:, 2] = rng.uniform(-4, 8, 20000)
        x[:50, 1] = 0.0
        x[50:60, :2] = 0.0
        x[60:70, 0], x[60:70, 1] = P.min_dis, 0.0
        b = oracle.bin(P, x, False)
        g = (C.c_float * 9)(P.min_dis, P.max_dis, P.min_angle, P.max_angle, P.min_azimuth, P.max_azimuth, P.range_res,
                            P.sector_res, P.azimuth_res)
        d = (C.c_int * 4)(*dims)
        of, oi = (C.c_float * 7)(), (C.c_int * 4)()
        keep_o = oracle.bin(P, x, True)["src"]
        keep_set = set(keep_o.tolist())
        for i in range(0, 20000, 7):
            p = (C.c_float * 4)(*x[i])
            keep = spec.spec_apri(g, d, p, of, oi)
            a = b["apri"][i]
            got = np.array(list(of), np.float32)
            exp = np.array([a["x"], a["y"], a["z"], a["range"], a["angle"], a["azimuth"], a["intensity"]], np.float32)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (preset, i)
            assert list(oi) == [a["range_idx"], a["sector_idx"], a["azimuth_idx"], a["voxel_idx"]]
            assert bool(keep) == (i in keep_set)


def test_angle_estimate_stays_inside_its_error_bound(spec):
    """czm_patch_of trusts atan2_abs_rad_fast wherever theta / sector_size is farther than 4e-6 rad from an integer: the estimate has
    to stay within the 6.1e-7 rad its header claims.  The polynomial over every 64th float of [0, 1] plus ALL floats of the stretch
    around its worst point (the exhaustive run over all 2^30 floats found 1.05e-7 rad at t = 0.98361367: a minute of CPU, done once),
    the whole estimate on points of every octant (host arithmetic: an exact quotient where the device takes v_rcp_f32, whose 1 ulp is
    1.2e-7 of the bound)."""
    spec.spec_atan01_worst.restype = C.c_double
    spec.spec_atan2_abs_worst.restype = C.c_double
    one = int(np.float32(1.0).view(np.uint32))
    assert spec.spec_atan01_worst(C.c_uint(0), C.c_uint(one), C.c_uint(64)) <= 1.06e-7
    lo, hi = int(np.float32(0.97).view(np.uint32)), int(np.float32(0.995).view(np.uint32))
    w = spec.spec_atan01_worst(C.c_uint(lo), C.c_uint(hi), C.c_uint(1))
    assert 1.0e-7 < w <= 1.06e-7
    rng = np.random.default_rng(11)
    n = 4_000_000
    ay = np.abs(rng.choice([1e-3, 0.1, 1.0, 30.0, 80.0], n) * rng.uniform(0.01, 1.0, n)).astype(np.float32)
    x = (rng.choice([-1, 1], n) * rng.choice([1e-3, 0.1, 1.0, 30.0, 80.0], n) * rng.uniform(0.0, 1.0, n)).astype(np.float32)
    ay[ay == 0] = 1e-3
    x[:1000] = 0.0
    x[1000:2000] = ay[1000:2000]
    x[2000:3000] = -ay[2000:3000]
    assert spec.spec_atan2_abs_worst(ay.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_long(n)) <= 4.9e-7


def test_patch_ids_spec_equals_libm_oracle(spec, oracle, scvod):
    """pc2czm zone / ring / sector of the spec (fp32 atan2 fast path + fp64 fdlibm fall-back) == the oracle's
    glibc-double evaluation, including points a few 1e-7 rad away from sector boundaries and on the axes."""
    rng = np.random.default_rng(6)
    P = scvod.make_params("semantickitti")
    n = 400_000
    r = rng.uniform(2.0, 85.0, n)
    t = rng.uniform(0, 2 * np.pi, n)
    for S in (16, 32, 54):                       # cluster a share of the angles around the sector boundaries
        k = rng.integers(0, S, n // 8)
        sl = slice((S // 16 - 1) * (n // 8), (S // 16 - 1) * (n // 8) + n // 8) if S != 54 else slice(3 * (n // 8), 4 * (n // 8))
        # (3e-7 rad: inside the fdlibm path's margin; 4e-6 / 3e-5: around the margin of the branch-free estimate in front of it)
        t[sl] = k * (2 * np.pi / S) + rng.normal(0, 1, n // 8) * rng.choice([3e-7, 4e-6, 3e-5], n // 8)
    # a share of the radii within a few fp32 ulps (and within the fast path's margin) of every zone / ring boundary
    mn, mx = 2.7, 80.0
    zb = [mn, (7 * mn + mx) / 8, (3 * mn + mx) / 4, (mn + mx) / 2, mx]
    bounds = []
    for k, nr in enumerate((2, 4, 4, 4)):
        bounds += [zb[k] + j * (zb[k + 1] - zb[k]) / nr for j in range(nr + 1)]
    bsel = rng.choice(np.asarray(bounds), n // 8)
    r[4 * (n // 8):5 * (n // 8)] = bsel + rng.choice([0.0, 1e-7, 1e-6, 1e-5, 1e-4, 5e-4, 2e-3], n // 8) * rng.choice([-1, 1], n // 8) * bsel / 10
    x = np.stack([r * np.cos(t), r * np.sin(t), rng.uniform(-4, 3, n), np.zeros(n)], 1).astype(np.float32)
    x[:200, 1] = 0.0                             # exact axis directions
    x[200:400, 0] = 0.0
    x[400:600, 1] = x[400:600, 0]
    x[600:700, 1] = -0.0
    x[600:700, 0] = np.abs(x[600:700, 0])
    got = np.zeros(n, np.int32)
    spec.spec_patch_ids(C.c_float(P.sensor_height), x.ctypes.data_as(C.c_void_p), C.c_long(n), got.ctypes.data_as(C.c_void_p))
    ref = np.zeros(n, np.int32)
    oracle.lib.oracle_patch_ids(C.byref(P), x.ctypes.data_as(C.c_void_p), n, ref.ctypes.data_as(C.c_void_p))
    assert np.array_equal(got, ref)
    assert (ref >= 0).sum() > n // 2 and ref.max() == 503


def test_fast_keep_verdict_equals_reference_arithmetic(spec, scvod):
    """keep_of_point (range test + tan() window instead of two atan2f) == apri_of_point's verdict on random clouds and on
    points packed around every boundary (min/max range, azimuth limits within 1e-7..1e-3 rad, the +x axis from below where
    the polar angle wraps to 360, the z axis), for both YAML presets and a preset that disables the shortcut."""
    rng = np.random.default_rng(31)
    for name, extra in (("semantickitti", {}), ("parkinglot", {}), ("semantickitti", dict(max_angle=300.0))):
        P = scvod.make_params(name, **extra)
        g = np.array([P.min_dis, P.max_dis, P.min_angle, P.max_angle, P.min_azimuth, P.max_azimuth, P.range_res, P.sector_res,
                      P.azimuth_res], np.float32)
        n = 3_000_000
        r = rng.uniform(0.0, 1.2 * P.max_dis, n)
        th = rng.uniform(0, 2 * np.pi, n)
        az = np.deg2rad(rng.uniform(-89, 89, n))
        # azimuth limits: a third of the points within 1e-7 .. 1e-3 rad of one of them
        k = n // 3
        lim = np.deg2rad(rng.choice([P.min_azimuth, P.max_azimuth], k))
        az[:k] = lim + rng.choice([-1, 1], k) * 10.0 ** rng.uniform(-7.5, -3, k)
        # range limits
        r[k:k + k // 2] = rng.choice([P.min_dis, P.max_dis], k // 2) * (1 + rng.choice([-1, 1], k // 2) * 10.0 ** rng.uniform(-8, -4, k // 2))
        x = np.stack([r * np.cos(th), r * np.sin(th), r * np.tan(az)], 1).astype(np.float32)
        x[-3000:-2000, 1] = -np.abs(x[-3000:-2000, 1]) * 1e-6       # just below the +x axis: polar angle -> 360
        x[-3000:-2000, 0] = np.abs(x[-3000:-2000, 0])
        x[-2000:-1000, :2] = 0.0                                     # on the z axis (dis == 0)
        x[-1000:-500, 2] = 0.0
        x[-500:] = rng.integers(0, 2**32, (500, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)   # arbitrary bit patterns
        n_fast = C.c_long(0)
        spec.spec_keep_compare.restype = C.c_long
        bad = spec.spec_keep_compare(g.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.c_long(n), C.byref(n_fast))
        assert bad == 0, (name, extra, bad)
        if extra:
            assert n_fast.value == 0                                  # shortcut disabled: max_angle < 360
        else:
            assert n_fast.value > 0.6 * n                             # and it really decides most points


def test_voxel_index_estimate_never_disagrees_with_the_reference_arithmetic(spec):
    """scvod_math.h::voxel_idx_fast (polynomial arctangent + distance to the bin edges; used where SSC::tracking re-bins
    transformed points, ssc.cpp:1280-1286) against apri_of_point: every point it DECIDES carries the reference's voxel
    index -- uniform points, points scattered 2e-3 degrees around sector / azimuth bin edges, both reference grids and a
    fine one -- and its angle estimates stay within 2.5e-4 degrees of the reference's (the guard is 1.5e-3)."""
    import ctypes as C
    spec.spec_voxel_fast_compare.restype = C.c_long
    rng = np.random.default_rng(5)

    def run(g, dims, xyz):
        xyz = np.ascontiguousarray(xyz, np.float32)
        nf, w = C.c_long(), C.c_double()
        gg, dd = np.asarray(g, np.float32), np.asarray(dims, np.int32)
        bad = spec.spec_voxel_fast_compare(gg.ctypes.data_as(C.c_void_p), dd.ctypes.data_as(C.c_void_p), xyz.ctypes.data_as(C.c_void_p),
                                           C.c_long(len(xyz)), C.byref(nf), C.byref(w))
        return bad, nf.value / len(xyz), w.value
    n = 3_000_000
    grids = [([1.5, 30, 0, 360, -40, 80, 0.4, 1.2, 2.0], [72, 300, 60]), ([0.8, 40, 0, 360, -30, 60, 0.4, 1.2, 2.0], [98, 300, 45]),
             ([1.5, 30, 0, 360, -40, 80, 0.2, 0.6, 1.0], [143, 600, 120])]
    for g, dims in grids:
        xyz = np.stack([rng.uniform(-60, 60, n), rng.uniform(-60, 60, n), rng.uniform(-6, 12, n)], 1)
        bad, frac, worst = run(g, dims, xyz)
        assert bad == 0 and frac > 0.98 and worst < 2.5e-4, (bad, frac, worst)
        r = rng.uniform(1, 60, n)
        th = np.deg2rad(rng.integers(0, int(360 / g[7]), n) * g[7] + rng.normal(0, 2e-3, n))
        bad, frac, worst = run(g, dims, np.stack([r * np.cos(th), r * np.sin(th), rng.uniform(-6, 12, n)], 1))
        assert bad == 0 and worst < 2.5e-4, (bad, worst)
        az = np.deg2rad(rng.integers(-20, 40, n) * g[8] + rng.normal(0, 2e-3, n))
        th = rng.uniform(0, 2 * np.pi, n)
        bad, frac, worst = run(g, dims, np.stack([r * np.cos(th), r * np.sin(th), r * np.tan(az)], 1))
        assert bad == 0 and worst < 2.5e-4, (bad, worst)
    # the corner cases that must never be decided by the estimate: y == +-0, the origin, non-finite input
    edge = np.array([[3.0, 0.0, 0.5], [-3.0, 0.0, 0.5], [3.0, -0.0, 0.5], [-3.0, -0.0, 0.5], [0.0, 0.0, 1.0], [np.nan, 1.0, 1.0],
                     [1.0, np.inf, 1.0], [1e30, 1e30, 1.0]], np.float32)
    bad, frac, _ = run(grids[0][0], grids[0][1], edge)
    assert bad == 0


def test_atan2_overload_decision_moves_a_handful_of_indices(oracle, scvod):
    """the unqualified `atan2` of utility.h:382-391 is atan2f when <cmath>'s float overload is in scope (the
    reading this library and its oracle follow) and float(atan2(double, double)) otherwise.  The two round differently in the
    last place on a few arguments; an index only flips when such an argument sits on a bin edge.  Counted on 4 M uniform
    points per grid (same seed as DESIGN.md section 2 quotes): a few per million, never the range / FOV verdict."""
    rng = np.random.default_rng(20241026)
    for preset, want in (("semantickitti", (2571252, 3, 0, 0)), ("parkinglot", (3441127, 5, 0, 0)), ("os128_fine", (2571808, 5, 0, 0))):
        P = scvod.make_params(preset)
        n = 4_000_000
        r = rng.uniform(1.0, 45.0, n)
        th = rng.uniform(0, 2 * np.pi, n)
        z = rng.uniform(-3, 6, n)
        x = np.stack([r * np.cos(th), r * np.sin(th), z, np.zeros(n)], 1).astype(np.float32)
        got = oracle.atan2_overload_flips(P, x)
        assert got[0] > 2_000_000 and got[1] + got[2] < 1e-5 * got[0] and got[3] == 0, (preset, got)
        assert tuple(int(v) for v in got) == want, (preset, got)  # (glibc 2.35 of this image; pins the number DESIGN.md quotes)
