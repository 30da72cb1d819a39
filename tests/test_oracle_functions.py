"""Unit checks of the oracle's per-function restatements (parity unpinned: these check the
restatement against the mathematics and against the behaviours the reference source states,
not against outputs of the reference)."""
import math

import numpy as np

from raytracer_amd.capi import Config


def test_rot_matrix_vectors_quirk_and_axis_aligned_cases(oracle):
    # The axis b x a is not normalised (Matrix3.cpp:15-18): for a 45-degree normal the
    # "rotation" shrinks z to (0.5, 0, 0.7071) -- the value observed from the reference.
    m = oracle.rot_matrix_vectors((0.70710678, 0, 0.70710678), (0, 0, 1))
    np.testing.assert_allclose(oracle.mat_vec(m, (0, 0, 1)), (0.5, 0.0, 0.70710678), atol=1e-6)
    # |b x a| in {0, 1}: a proper rotation taking b to a.
    for a in [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1)]:
        m = oracle.rot_matrix_vectors(a, (0, 0, 1))
        np.testing.assert_allclose(m @ m.T, np.eye(3), atol=1e-6)
        np.testing.assert_allclose(oracle.mat_vec(m, (0, 0, 1)), a, atol=1e-6)
    # a = -b: c = -1, axis = 0 -> diag(-1,-1,-1): the hemisphere is mirrored, not rotated.
    m = oracle.rot_matrix_vectors((0, 0, -1), (0, 0, 1))
    np.testing.assert_allclose(m, -np.eye(3), atol=1e-6)


def test_triangle_intersect_against_float64(oracle):
    rng = np.random.default_rng(1)
    hits = 0
    for _ in range(2000):
        a, b, c = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
        o = rng.uniform(-2, 2, 3).astype(np.float32)
        tgt = (a + b + c) / 3 + rng.normal(0, 0.4, 3)
        d = (tgt - o).astype(np.float32)
        t = oracle.triangle_intersect(a, b, c, o, d)
        A, B, Cc, O, D = (x.astype(np.float64) for x in (a, b, c, o, d))
        n = np.cross(B - A, Cc - A)
        den = D @ n
        q = np.cross(O - A, D)
        u, v = -(q @ (Cc - A)) / den, (q @ (B - A)) / den
        inside = (u >= 0) and (u <= 1) and (v >= 0) and (u + v <= 1)
        margin = min(u, v, 1 - u - v, 1 - u)
        if abs(margin) < 1e-4:
            continue
        if inside:
            hits += 1
            assert math.isfinite(t) and abs(t - (-(n @ (O - A)) / den)) < 1e-3 * max(1, abs(t))
        else:
            assert t == math.inf
    assert hits > 200


def test_triangle_intersect_edges_and_parallel_rays(oracle):
    a, b, c = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    # Edges and vertices are inclusive (u<0 || u>1 || v<0 || u+v>1 rejects; Obj.h:119).
    assert oracle.triangle_intersect(a, b, c, (0.5, 0, 1), (0, 0, -1)) == 1.0
    assert oracle.triangle_intersect(a, b, c, (0, 0, 1), (0, 0, -1)) == 1.0
    assert oracle.triangle_intersect(a, b, c, (0.5, 0.5, 1), (0, 0, -1)) == 1.0
    assert oracle.triangle_intersect(a, b, c, (0.6, 0.6, 1), (0, 0, -1)) == math.inf
    # No backface culling.
    assert oracle.triangle_intersect(a, b, c, (0.2, 0.2, -1), (0, 0, 1)) == 1.0
    # A ray in the triangle's plane: 1/0 -> inf/NaN; NaN fails every rejection compare and the
    # result is NaN or +-inf, which the caller's `EPS < t && t < min_t` discards (Obj.h:116-120,189).
    t = oracle.triangle_intersect(a, b, c, (-1, 0.2, 0), (1, 0, 0))
    assert not (1e-3 < t < math.inf)


def test_sphere_intersect(oracle):
    c, r = (0, 0, 0), 1.0
    assert abs(oracle.sphere_intersect(c, r, (0, 0, 3), (0, 0, -1)) - 2.0) < 1e-6
    assert abs(oracle.sphere_intersect(c, r, (0, 0, 0), (0, 0, -1)) - 1.0) < 1e-6   # from inside: far root
    assert oracle.sphere_intersect(c, r, (0, 2, 3), (0, 0, -1)) == math.inf
    # Origin on the surface: the near root is ~0 <= EPS, so the far root is returned (Obj.h:266-267).
    assert abs(oracle.sphere_intersect(c, r, (0, 0, 1), (0, 0, -1)) - 2.0) < 1e-6
    # Behind the ray: both roots negative; the caller's t > EPS test rejects it.
    assert oracle.sphere_intersect(c, r, (0, 0, 3), (0, 0, 1)) < 0


def test_camera_rays(oracle):
    cfg = Config(width=800, height=800, fov=50.0)
    o, d, _ = oracle.camera_ray(cfg, 400.0, 400.0)
    np.testing.assert_array_equal(o, (0, 1, 3))
    np.testing.assert_allclose(d, (0, 0, -1), atol=1e-7)
    o, d, _ = oracle.camera_ray(cfg, 800.0, 0.0)   # right edge, top edge: +x, +y
    tx = math.tan(math.radians(25.0))
    want = np.array([tx, tx, -1.0]) / math.sqrt(2 * tx * tx + 1)
    np.testing.assert_allclose(d, want, atol=1e-6)
    # Non-square frame: the vertical half-angle is H/W * fov/2 (Camera.h:20).
    cfg2 = Config(width=800, height=400, fov=50.0)
    _, d, _ = oracle.camera_ray(cfg2, 400.0, 0.0)
    ty = math.tan(math.radians(0.5 * 50.0 / 2))
    np.testing.assert_allclose(d, np.array([0, ty, -1]) / math.sqrt(ty * ty + 1), atol=1e-6)
    # Orthogonal: parallel rays, origin moves (Camera.h:73-79).
    cfg3 = Config(width=800, height=800, camera=2, stretch_x=1.2, stretch_y=1.2)
    o, d, _ = oracle.camera_ray(cfg3, 800.0, 800.0)
    np.testing.assert_allclose(o, (1.2, 1 - 1.2, 3), atol=1e-6)
    np.testing.assert_array_equal(d, (0, 0, -1))
    # Thin lens: two more draws, origin within the lens radius, ray through the focal point.
    cfg4 = Config(width=800, height=800, camera=1, fov=60.0)
    o, d, st = oracle.camera_ray(cfg4, 123.0, 456.0, state=99)
    assert st != 99 and np.linalg.norm(o - np.array([0, 1, 3])) <= 0.09 + 1e-6
    assert abs(np.linalg.norm(d) - 1) < 1e-6


def test_cosine_sample_hemisphere(oracle):
    state, zs = 12345, []
    for _ in range(4000):
        v, state = oracle.cosine_sample_hemisphere(state)
        if np.isnan(v[2]):
            continue  # sqrt of a tiny negative: the reference has the same NaN (Integrator.h:24)
        assert v[2] >= 0 and abs(np.linalg.norm(v) - 1) < 1e-3
        zs.append(v[2])
    assert abs(np.mean(zs) - 2 / 3) < 0.02  # E[cos] for a cosine-weighted hemisphere
    v, _ = oracle.cosine_sample_hemisphere(0)   # first Uniform() of seed 0 is exactly 0 -> r = 0
    assert v[0] == 0 and v[1] == 0 and v[2] == 1


def test_schlick_and_sky(oracle):
    L = oracle.lib()
    assert abs(L.orc_schlick(0.5, 1.0) - 1 / 9) < 1e-7 and abs(L.orc_schlick(2.0, 1.0) - 1 / 9) < 1e-7
    assert abs(L.orc_schlick(0.5, 0.0) - 1.0) < 1e-6
    sun = np.array([0.2, -0.8, -0.4]) / np.linalg.norm([0.2, -0.8, -0.4])
    np.testing.assert_allclose(oracle.sky_emission(sun), (5.5, 5.7, 5.8), atol=1e-6)
    np.testing.assert_allclose(oracle.sky_emission(-sun), (0.5, 0.7, 0.8), atol=1e-7)
    # Inside the ramp 0.96 < a <= 0.999: quadratic from 0 to 5.
    t = math.acos(0.98)
    perp = np.cross(sun, [1, 0, 0])
    perp /= np.linalg.norm(perp)
    dvec = math.cos(t) * sun + math.sin(t) * perp
    ramp = 5 * (0.98 - 0.96) ** 2 / (0.999 - 0.96) ** 2
    np.testing.assert_allclose(oracle.sky_emission(dvec), np.array([0.5, 0.7, 0.8]) + ramp, atol=2e-3)


def test_triangle_sample_point_and_light_terms(oracle):
    a, b, c = np.array([0, 0, 0.0]), np.array([2, 0, 0.0]), np.array([0, 3, 0.0])
    state = 7
    for _ in range(500):
        p, n, rcp, state = oracle.triangle_sample_point(a, b, c, state)
        assert p[0] >= -1e-6 and p[1] >= -1e-6 and p[0] / 2 + p[1] / 3 <= 1 + 1e-6 and p[2] == 0
        assert abs(rcp - 1 / 3.0) < 1e-7          # rcp_area = 2/|cross| = 1/area
        np.testing.assert_allclose(n, (0, 0, 1), atol=1e-7)
    # u0 = 0 returns vertex a (first Uniform() of seed 0 is 0): b0 = 1, b1 = 0.
    p, _, _, _ = oracle.triangle_sample_point(a + 1, b, c, 0)
    np.testing.assert_array_equal(p, a + 1)


def test_img_update_running_mean_vs_sum(oracle):
    rng = np.random.default_rng(3)
    samples = rng.uniform(0, 4, (200, 3)).astype(np.float32)
    px = np.zeros(3, np.float32)
    for i, s in enumerate(samples):
        px = oracle.img_update(px, i, s)     # Raytracer.cpp:78-81
    total = np.zeros(3, np.float32)
    for s in samples:
        total += s
    np.testing.assert_allclose(px, total / np.float32(len(samples)), rtol=2e-5)


def test_box_and_plane(oracle):
    """Box follows the reference's commented-out class (Obj.h:43-87); Plane is build-defined."""
    mn, mx = (-1, -1, -1), (1, 1, 1)
    assert oracle.box_intersect(mn, mx, (0, 0, 3), (0, 0, -1)) == 2.0
    assert oracle.box_intersect(mn, mx, (0, 0, 0), (0, 0, -1)) == 1.0       # inside: tmin <= EPS -> tmax
    assert oracle.box_intersect(mn, mx, (0, 3, 3), (0, 0, -1)) == math.inf  # misses the y slab
    assert oracle.box_intersect(mn, mx, (0, 0, 3), (0, 0, 1)) < 0           # behind: negative, caller rejects
    # axis-parallel ray inside the other slabs: divisions by zero give +-inf, folded by min/max
    assert oracle.box_intersect(mn, mx, (0.5, 0.5, 3), (0, 0, -1)) == 2.0
    np.testing.assert_array_equal(oracle.box_normal(mn, mx, (0.2, 0.3, 1.0)), (0, 0, 1))
    np.testing.assert_array_equal(oracle.box_normal(mn, mx, (-1.0, 0.3, 0.5)), (-1, 0, 0))
    # a thin box: the face is chosen relative to the extent on each axis
    np.testing.assert_array_equal(oracle.box_normal((-2, -0.1, -2), (2, 0.1, 2), (1.5, 0.1, 0.0)), (0, 1, 0))
    assert oracle.plane_intersect((0, 0, 0), (0, 1, 0), (0, 2, 0), (0, -1, 0)) == 2.0
    assert oracle.plane_intersect((0, 0, 0), (0, 1, 0), (0, 2, 0), (0, 1, 0)) == -2.0
    t = oracle.plane_intersect((0, 0, 0), (0, 1, 0), (0, 2, 0), (1, 0, 0))   # parallel: -inf (or NaN), never a hit
    assert not (1e-3 < t < math.inf)
