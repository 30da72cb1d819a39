"""GPU parity, fixed ray sets: the HIP closest-hit / any-hit kernels against the
oracle's first-party linear scan (Mesh::intersect order, reference Obj.h:184-195,
combined by the castRay rules of Scene.h:144-189).  Bar: bit-exact kind, primitive
id, t and normal for every ray."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

pytestmark = pytest.mark.gpu


def camera_rays(oracle, cfg, n_side=96):
    rays = []
    for j in range(n_side):
        for i in range(n_side):
            px = (i + 0.5) * cfg.width / n_side
            py = (j + 0.5) * cfg.height / n_side
            o, d, _ = oracle.camera_ray(cfg, px, py)
            rays.append(np.concatenate([o, d]))
    return np.array(rays, np.float32)


def bounce_rays(oracle, osc, primary, seed=3):
    """Secondary rays leaving the primary hit points in random directions (origins lie
    exactly on surfaces: the eps and tie rules matter here)."""
    kind, prim, t, nrm = osc.trace_closest(primary)
    hit = kind != 0
    o = primary[hit, :3] + primary[hit, 3:] * t[hit, None]
    rng = np.random.default_rng(seed)
    d = rng.normal(size=o.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)


SCENES = {
    "cornell": lambda: scenes.cornell(),
    "cornell_4lights": lambda: scenes.cornell(lights=4),
    "sphere_sky": lambda: scenes.sphere_sky(),
    "soup10k": lambda: scenes.soup(10000, 3),
    # the wide tree's leaves of a fan-tessellated quad as one record of four vertices, beside leaves that do not qualify
    "quad_soup": lambda: scenes.quad_soup(2500, seed=4),
    "terrain40": lambda: scenes.terrain(40),
    "one_triangle": lambda: scenes.SceneData(scenes.quad((-1, 0, 0), (1, 0, 0), (1, 2, 0), (-1, 2, 0))[:1],
                                             np.zeros(1, np.uint32),
                                             np.array([scenes.material(scenes.WHITE)], scenes.MATERIAL_DTYPE)),
    "ref_scene0": lambda: scenes.reference_scene(0),
    "ref_scene1": lambda: scenes.reference_scene(1),
    "ref_scene3": lambda: scenes.reference_scene(3),
    "spheres_only": lambda: scenes.SceneData(np.zeros((0, 9), np.float32), np.zeros(0, np.uint32),
                                             np.array([scenes.material(scenes.WHITE)], scenes.MATERIAL_DTYPE),
                                             scenes.soup(1, 4).spheres),
}


def assert_same_hits(got, want):
    gk, gp, gt, gn = got
    wk, wp, wt, wn = want
    bad = np.nonzero((gk != wk) | (gp != wp) | (gt.view(np.uint32) != wt.view(np.uint32)))[0]
    assert bad.size == 0, (f"{bad.size} of {gk.size} rays differ; first: ray {bad[:5]}, "
                           f"gpu (kind,prim,t)={list(zip(gk[bad[:5]], gp[bad[:5]], gt[bad[:5]]))} "
                           f"oracle={list(zip(wk[bad[:5]], wp[bad[:5]], wt[bad[:5]]))}")
    hit = wk != 0
    assert np.array_equal(gn[hit].view(np.uint32), wn[hit].view(np.uint32)), "normals differ"


@pytest.mark.parametrize("name", list(SCENES))
def test_closest_hit_bit_exact(name, oracle, rtx_lib):
    cfg = Config(width=800, height=800, pool_paths=1 << 16)
    sc = SCENES[name]()
    osc = oracle.OracleScene(sc, cfg)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        prim = camera_rays(oracle, cfg)
        sets = [prim, bounce_rays(oracle, osc, prim), scenes.random_rays(20000, seed=11),
                scenes.random_rays(5000, seed=12, unit=False)]
        if name.startswith("ref_scene"):   # rays inside the +-800 rooms too
            sets.append(scenes.random_rays(20000, seed=13, origin_box=((-790, -790, -790), (790, 790, 790))))
        for rays in sets:
            if rays.shape[0] == 0:
                continue
            want = osc.trace_closest(rays, use_bvh=False)
            got = r.trace_closest(rays)
            assert_same_hits(got, want)


@pytest.mark.parametrize("name", ["cornell", "terrain", "soup", "huge_quad"])
def test_hits_at_distances_around_eps(name, oracle, rtx_lib):
    """The walk skips boxes the ray has left before eps / 2 (a hit needs eps < t): rays that start a
    fraction or a small multiple of eps in front of a surface, head-on and grazing, must still find
    exactly the hits the linear scan finds -- the ones whose computed t just passes eps included --
    and the same occlusion answers, on thin axis-aligned quads, a fine terrain, a soup and a quad
    thousands of units wide (whose computed t is noisy at the scale of eps)."""
    def huge_quad():   # the 2000-unit ground quad of the sphere scenes with enough small triangles on it for a tree
        g, small = scenes.sphere_sky(), scenes.soup(40, 0, seed=8, size=0.3)
        return scenes.SceneData(np.concatenate([g.triangles, small.triangles]), np.zeros(42, np.uint32), g.materials,
                                g.spheres, name="huge_quad")
    sc = {"cornell": scenes.cornell, "terrain": lambda: scenes.terrain(40), "soup": lambda: scenes.soup(3000, 0, seed=6),
          "huge_quad": huge_quad}[name]()
    cfg = Config(width=256, height=256, sky=1, pool_paths=1 << 14)
    osc = oracle.OracleScene(sc, cfg)
    primary = camera_rays(oracle, cfg, 40)
    kind, prim, t, nrm = osc.trace_closest(primary)
    hit = kind == 1
    p = (primary[hit, :3] + primary[hit, 3:] * t[hit, None]).astype(np.float32)
    n = nrm[hit]
    rng = np.random.default_rng(17)
    rays = []
    for scale in (0.3, 0.45, 0.5, 0.55, 0.9, 0.999, 1.0, 1.001, 1.1, 2.0, 5.0):
        d = rng.normal(size=p.shape)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        graze = rng.random(len(p)) < 0.5                   # half of them nearly in the surface's plane
        along = np.sum(d * n, 1, keepdims=True)
        d = np.where(graze[:, None], d - 0.98 * along * n, d)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = p - d * np.float32(scale * cfg.eps)             # the surface lies scale * eps ahead (or behind, for rays leaving it)
        rays.append(np.concatenate([o, d], 1))
        rays.append(np.concatenate([p + d * np.float32(scale * cfg.eps), -d], 1))
    rays = np.concatenate(rays).astype(np.float32)
    dist = rng.uniform(0.0, 6.0 * cfg.eps, rays.shape[0]).astype(np.float32)
    dist[::3] = rng.uniform(0.0, 3.0, len(dist[::3])).astype(np.float32)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        assert_same_hits(r.trace_closest(rays), osc.trace_closest(rays, use_bvh=False))
        assert np.array_equal(r.trace_occluded(rays, dist), osc.trace_occluded(rays, dist, use_bvh=False))


def test_two_child_tree_in_global_memory(oracle, rtx_lib, monkeypatch):
    """A small scene whose tree is too deep for the LDS budget keeps its two-child tree in global
    memory.  RTX_NO_LDS_SCENE forces that mode for an ordinary small scene: same hits, same
    occlusion answers and the same image bits as the LDS mode."""
    cfg = Config(width=120, height=80, pool_paths=1 << 12)
    for sc in (scenes.cornell(), scenes.soup(150, 3, seed=2)):
        osc = oracle.OracleScene(sc, cfg)
        rays = np.concatenate([scenes.random_rays(8000, seed=11), bounce_rays(oracle, osc, camera_rays(oracle, cfg, 48))])
        dist = np.random.default_rng(5).uniform(0.0, 3.0, rays.shape[0]).astype(np.float32)
        out = {}
        for mode in ("lds", "global"):
            if mode == "global":
                monkeypatch.setenv("RTX_NO_LDS_SCENE", "1")
            else:
                monkeypatch.delenv("RTX_NO_LDS_SCENE", raising=False)
            with Renderer(cfg) as r:
                info = r.set_scene(sc)
                assert info.bvh_width == 2
                hits = r.trace_closest(rays)
                occ = r.trace_occluded(rays, dist)
                acc = r.accum_alloc()
                r.render(acc, 3)
                out[mode] = (hits, occ, r.resolve(acc, 1), r.stats())
        monkeypatch.delenv("RTX_NO_LDS_SCENE", raising=False)
        assert_same_hits(out["global"][0], osc.trace_closest(rays, use_bvh=False))
        assert np.array_equal(out["global"][1], osc.trace_occluded(rays, dist, use_bvh=False))
        assert np.array_equal(out["global"][2].view(np.uint32), out["lds"][2].view(np.uint32))
        assert out["global"][3].bounces == out["lds"][3].bounces
        assert out["global"][3].launches > out["lds"][3].launches     # no one-kernel bounce without the LDS tree


def test_meshes_of_a_few_triangles_are_tested_without_a_tree(oracle, rtx_lib, monkeypatch):
    """Meshes of up to eight triangles (the ground quad of the sphere scenes) are tested triangle by
    triangle by every ray, with no tree.  RTX_NO_FEW_TRIANGLES switches that off: same hits
    (lowest-id tie-break included: the soup's triangles overlap), same occlusion answers, same
    image bits in one- and two-kernel bounces, and the same as the linear scan of the oracle."""
    cfg = Config(width=120, height=80, sky=1, pool_paths=1 << 12)
    for sc in (scenes.sphere_sky(glass=True), scenes.soup(8, 2, seed=4, size=0.9), scenes.soup(1, 0, seed=5, size=0.9)):
        osc = oracle.OracleScene(sc, cfg)
        rays = np.concatenate([scenes.random_rays(8000, seed=11), bounce_rays(oracle, osc, camera_rays(oracle, cfg, 48))])
        dist = np.random.default_rng(5).uniform(0.0, 3.0, rays.shape[0]).astype(np.float32)
        out = {}
        for mode in ("few", "tree"):
            if mode == "tree":
                monkeypatch.setenv("RTX_NO_FEW_TRIANGLES", "1")
            else:
                monkeypatch.delenv("RTX_NO_FEW_TRIANGLES", raising=False)
            images = []
            for fused in (0, 1):
                with Renderer(Config(width=120, height=80, sky=1, pool_paths=1 << 12, fused_trace=fused)) as r:
                    r.set_scene(sc)
                    hits = r.trace_closest(rays)
                    occ = r.trace_occluded(rays, dist)
                    acc = r.accum_alloc()
                    r.render(acc, 3)
                    images.append(r.resolve(acc, 1))
                    st = r.stats()
            assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
            out[mode] = (hits, occ, images[0], st)
        monkeypatch.delenv("RTX_NO_FEW_TRIANGLES", raising=False)
        assert_same_hits(out["few"][0], osc.trace_closest(rays, use_bvh=False))
        assert np.array_equal(out["few"][1], osc.trace_occluded(rays, dist, use_bvh=False))
        assert_same_hits(out["few"][0], out["tree"][0])
        assert np.array_equal(out["few"][1], out["tree"][1])
        assert np.array_equal(out["few"][2].view(np.uint32), out["tree"][2].view(np.uint32))
        assert out["few"][3].bounces == out["tree"][3].bounces


def test_triangles_that_share_a_vertex_or_a_plane(oracle, rtx_lib):
    """Consecutive triangles of a leaf reuse what they share: o-a and (o-a)xd when the first vertex is the same,
    1/(d.n) and t as well when the normals are equal (as floats: the halves of an axis-aligned rectangle give
    (4, 0, 0) and (4, -0, 0)).  Fans around one vertex whose triangles lie in one plane, in two, or in planes that
    differ in a last bit, as a mesh of up to eight triangles (one run of eight records) and with a ninth triangle
    (leaves of two in a tree): rays in the planes, starting on them, along the axes and at random must give the
    linear scan's hits and occlusion answers bit for bit."""
    rng = np.random.default_rng(23)

    def fan(center, rim):
        return [np.concatenate([center, rim[i], rim[i + 1]]) for i in range(len(rim) - 1)]

    c = np.array([0.1, 0.5, -0.2])
    flat = [c + np.array([np.cos(a), 0.0, np.sin(a)]) * r for a, r in zip(np.linspace(0, 4.5, 6), (1.0, 0.7, 1.2, 0.9, 1.1, 0.8))]
    bent = [p + np.array([0.0, 0.3 * (i % 2), 0.0]) for i, p in enumerate(flat)]
    angle = np.radians(17.0)   # a rotated rectangle: the normals of its halves differ in the last bits, or not at all
    u, v = np.array([np.cos(angle), 0.0, np.sin(angle)]) * 0.8, np.array([0.0, 1.3, 0.0])
    meshes = {
        "rectangles": np.concatenate([scenes.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)),
                                      scenes.quad((-1, 0, -1), (-1, 2, -1), (-1, 2, 1), (-1, 0, 1)),
                                      scenes.quad((1, 2, 1), (-1, 2, 1), (-1, 0, 1), (1, 0, 1)),
                                      scenes.quad(c, c + u, c + u + v, c + v)]),
        "flat_fan": np.array(fan(c, flat), np.float32),
        "bent_fan": np.array(fan(c, bent), np.float32),
        "mixed_fan": np.array(fan(c, flat[:3] + bent[3:]) + fan(c + 0.5, flat[:3]), np.float32),
    }
    extra = np.array([[3.0, 0.0, 3.0, 3.5, 0.0, 3.0, 3.0, 0.5, 3.5]], np.float32)   # a ninth triangle: the tree takes over
    for name, tris in meshes.items():
        for with_tree in (False, True):
            t = np.concatenate([tris, extra]) if with_tree and len(tris) <= 8 else tris
            sc = scenes.SceneData(t.astype(np.float32), np.zeros(len(t), np.uint32),
                                  np.array([scenes.material(scenes.WHITE)], scenes.MATERIAL_DTYPE))
            cfg = Config(width=64, height=64, pool_paths=1 << 12)
            osc = oracle.OracleScene(sc, cfg)
            tri = t.reshape(-1, 3, 3).astype(np.float64)
            rays = [scenes.random_rays(6000, seed=31, origin_box=((-2, -1, -2), (2, 3, 2)))]
            # from points on the triangles (t = 0 for the triangle itself), towards other points on them (rays in the plane
            # of a flat fan) and at random
            w = rng.dirichlet((1, 1, 1), (2, 3000))
            k = rng.integers(0, len(tri), (2, 3000))
            on = [(w[i][:, :, None] * tri[k[i]]).sum(1) for i in range(2)]
            d = on[1] - on[0]
            d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9)
            rays.append(np.concatenate([on[0], d], 1))
            r2 = rng.normal(size=(3000, 3))
            rays.append(np.concatenate([on[1], r2 / np.linalg.norm(r2, axis=1, keepdims=True)], 1))
            start = on[1] + r2   # aimed at the triangles from outside
            aim = on[0] - start
            rays.append(np.concatenate([start, aim / np.linalg.norm(aim, axis=1, keepdims=True)], 1))
            for axis in range(3):   # axis-parallel: d.n = 0 for the rectangles beside the ray, products with signed zeros
                for sign in (1.0, -1.0):
                    o = rng.uniform(-1.5, 2.5, (400, 3))
                    o[:200] = on[0][:200]
                    dd = np.zeros((400, 3))
                    dd[:, axis] = sign
                    rays.append(np.concatenate([o, dd], 1))
            rays = np.concatenate(rays).astype(np.float32)
            dist = rng.uniform(0.0, 3.0, len(rays)).astype(np.float32)
            with Renderer(cfg) as r:
                r.set_scene(sc)
                want = osc.trace_closest(rays, use_bvh=False)
                assert_same_hits(r.trace_closest(rays), want)
                assert np.array_equal(r.trace_occluded(rays, dist), osc.trace_occluded(rays, dist, use_bvh=False)), name
            assert (want[0] != 0).sum() > 2000, name


@pytest.mark.parametrize("name", ["cornell", "soup10k", "quad_soup", "terrain40", "sphere_sky", "spheres_only", "ref_scene0",
                                  "ref_scene1"])
def test_occlusion_bit_exact(name, oracle, rtx_lib):
    cfg = Config(width=800, height=800, pool_paths=1 << 16)
    sc = SCENES[name]()
    osc = oracle.OracleScene(sc, cfg)
    rng = np.random.default_rng(5)
    rays = np.concatenate([scenes.random_rays(20000, seed=21),
                           bounce_rays(oracle, osc, camera_rays(oracle, cfg, 64))])
    dist = rng.uniform(0.0, 3.0, rays.shape[0]).astype(np.float32)
    if name == "ref_scene1":
        dist *= 400
    dist[:50] = rng.uniform(0.0, 0.003, 50)  # exercises distance < 2*eps (Scene.h:192-194)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        got = r.trace_occluded(rays, dist)
    want = osc.trace_occluded(rays, dist, use_bvh=False)
    assert np.array_equal(got, want), f"{np.count_nonzero(got != want)} of {got.size} segments differ"
    assert 0 < want.sum() < want.size or name in ("spheres_only", "ref_scene0")


def test_empty_inputs(rtx_lib):
    cfg = Config(width=16, height=16, pool_paths=1 << 12)
    with Renderer(cfg) as r:
        r.set_scene(scenes.cornell())
        k, p, t, n = r.trace_closest(np.zeros((0, 6), np.float32))
        assert k.size == 0 and t.size == 0
        assert r.trace_occluded(np.zeros((0, 6), np.float32), np.zeros(0, np.float32)).size == 0


def test_degenerate_rays(oracle, rtx_lib):
    """Axis-parallel rays, rays in a triangle's plane, zero direction components: the NaN/Inf
    fall-through of Triangle::intersect (Obj.h:116-120) must match."""
    cfg = Config(width=800, height=800, pool_paths=1 << 12)
    sc = scenes.cornell()
    osc = oracle.OracleScene(sc, cfg)
    rays = []
    for o in [(0, 1, 3), (0, 0, 0), (0.5, 0, 0.5), (-1, 1, 0), (0, 2, 0), (0.3, 1.98, 0.1), (0, 1, 0)]:
        for d in [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1),
                  (1, 1, 0), (0, 1, 1), (1, 0, 1), (0.70710678, 0, -0.70710678), (1e-30, 1, 0)]:
            rays.append(o + d)
    rays = np.array(rays, np.float32)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        assert_same_hits(r.trace_closest(rays), osc.trace_closest(rays, use_bvh=False))


@pytest.mark.parametrize("name", ["soup_lds", "cornell", "soup10k"])
def test_far_origins_and_axis_parallel_rays(name, oracle, rtx_lib):
    """Box tests must never cull a hit of the float triangle test, whatever the origin: rays
    aimed at the scene from 10 to 1e4 scene sizes away (the box test works on origin*inverse
    products there) and axis-parallel rays through a grid of origins (zero direction
    components).  soup_lds is small enough for the LDS-resident two-child tree.
    (From about 1e6 scene sizes away the float triangle test itself returns t with errors
    as large as the scene; a traversal that culls by the closest t so far then differs from
    a linear scan whatever the boxes are.  That range is outside the guarantee.)"""
    sc = {"soup_lds": lambda: scenes.soup(140, 2, seed=8), "cornell": scenes.cornell,
          "soup10k": lambda: scenes.soup(10000, 3)}[name]()
    cfg = Config(width=64, height=64, pool_paths=1 << 14)
    osc = oracle.OracleScene(sc, cfg)
    rng = np.random.default_rng(17)
    tri = sc.triangles.reshape(-1, 3, 3)
    lo, hi = tri.min((0, 1)), tri.max((0, 1))
    rays = []
    for dist in (10.0, 1e2, 1e3, 1e4):
        target = rng.uniform(lo, hi, (3000, 3))
        dirs = rng.normal(size=(3000, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        o = (target - dirs * dist).astype(np.float32)
        d = target - o.astype(np.float64)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays.append(np.concatenate([o, d.astype(np.float32)], 1))
    for axis in range(3):
        for sign in (1.0, -1.0):
            g = rng.uniform(lo, hi, (1500, 3))
            g[:, axis] = (lo[axis] - 5.0) if sign > 0 else (hi[axis] + 5.0)
            d = np.zeros((1500, 3))
            d[:, axis] = sign
            rays.append(np.concatenate([g, d], 1))
    rays = np.concatenate(rays).astype(np.float32)
    with Renderer(cfg) as r:
        info = r.set_scene(sc)
        assert (info.bvh_width == 2) == (name != "soup10k")
        want = osc.trace_closest(rays, use_bvh=False)
        assert_same_hits(r.trace_closest(rays), want)
        assert (want[0] != 0).sum() > 1000
        dist = np.where(want[0] != 0, want[2] * 1.5, 10.0).astype(np.float32)
        dist[~np.isfinite(dist)] = 10.0
        assert np.array_equal(r.trace_occluded(rays, dist), osc.trace_occluded(rays, dist, use_bvh=False))


@pytest.mark.parametrize("name,make", [("cornell", scenes.cornell), ("soup2k", lambda: scenes.soup(2000, 3, seed=4))])
def test_matches_committed_vectors(name, make, rtx_lib):
    """The committed fixed ray sets (tests/golden/make_golden.py): bit-exact kind, id, t and occlusion."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"closest_{name}.npz"))
    with Renderer(Config(width=800, height=800, pool_paths=1 << 12)) as r:
        r.set_scene(make())
        kind, prim, t, _ = r.trace_closest(g["rays"])
        occ = r.trace_occluded(g["rays"], g["dist"])
    assert np.array_equal(kind, g["kind"]) and np.array_equal(prim, g["prim"])
    assert np.array_equal(t.view(np.uint32), g["t_bits"])
    assert np.array_equal(occ, g["occluded"])


def test_million_triangle_mesh_bit_exact(oracle, rtx_lib):
    """BASELINE's ~1M-triangle configuration: closest hits against the oracle.  The linear
    scan is too slow at this size, so the oracle's BVH (itself checked against the linear
    scan on smaller scenes) is the comparator; a sample of the rays is also checked against
    the linear scan."""
    sc = scenes.terrain(707)
    cfg = Config(width=1920, height=1080, pool_paths=1 << 16)
    osc = oracle.OracleScene(sc, cfg)
    rays = np.concatenate([camera_rays(oracle, cfg, 64), scenes.random_rays(20000, seed=5)])
    rays = np.concatenate([rays, bounce_rays(oracle, osc, rays[:4096])])
    with Renderer(cfg) as r:
        info = r.set_scene(sc)
        got = r.trace_closest(rays)
    assert info.num_triangles == 999710 and info.max_depth < info.stack_entries
    assert_same_hits(got, osc.trace_closest(rays, use_bvh=True))
    sub = rays[::97]
    want = osc.trace_closest(sub, use_bvh=False)
    assert np.array_equal(got[1][::97], want[1]) and np.array_equal(got[2][::97].view(np.uint32), want[2].view(np.uint32))
