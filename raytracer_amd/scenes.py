"""Synthetic scene builders (numpy only).

The reference ships no scene asset: its live scene loads a git-ignored OBJ
(Raytracer.cpp:162, .gitignore:3) and the other scenes are commented out
(Raytracer.cpp:180-226).  Everything here is therefore defined by this build,
following the roles and materials of the reference's scenes:

  cornell()      the live scene's roles (Raytracer.cpp:152-176): red/green/white
                 walls, two boxes, a 2-triangle quad light of emission 10, and the
                 glass sphere c=(-0.5,0.302,0.55) r=0.3
  sphere_sky()   "sample1"/"sample2" of the README: a sphere over a ground plane
                 lit by skyEmission (Integrator.h:73-85), diffuse or glass
  oven()         the oven test of Raytracer.cpp:203-214: a closed room whose
                 surfaces all have emission 0.5 and albedo 0.5 -> radiance 1
  terrain()      ~1M-triangle height field inside the Cornell room
  soup()         random triangles + spheres (parity tests)

Triangles are wound counter-clockwise seen from the side their normal
(b-a)x(c-a) points to; the integrator never flips normals, so walls face inwards
and boxes outwards.
"""
from __future__ import annotations

import numpy as np

from .capi import BOX_DTYPE, MATERIAL_DTYPE, PLANE_DTYPE, SPHERE_DTYPE, SceneData

DIFFUSE, SPECULAR, REFLECTIVE, VARNISH = 0, 1, 2, 3


def material(albedo, emission=0.0, surface=DIFFUSE):
    m = np.zeros((), MATERIAL_DTYPE)
    m["albedo"] = np.broadcast_to(np.asarray(albedo, np.float32), (3,))
    m["emission"] = emission
    m["surface"] = surface
    return m


def sphere(c, r, mat):
    s = np.zeros((), SPHERE_DTYPE)
    s["c"] = c
    s["r"] = r
    s["material"] = mat
    return s


def box_obj(mn, mx, mat):
    b = np.zeros((), BOX_DTYPE)
    b["min"] = mn
    b["max"] = mx
    b["material"] = mat
    return b


def plane_obj(point, normal, mat):
    p = np.zeros((), PLANE_DTYPE)
    p["point"] = point
    p["normal"] = normal
    p["material"] = mat
    return p


def quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3)."""
    p0, p1, p2, p3 = (np.asarray(p, np.float32) for p in (p0, p1, p2, p3))
    return np.stack([np.concatenate([p0, p1, p2]), np.concatenate([p0, p2, p3])])


def room(x0, x1, y0, y1, z0, z1, front=False):
    """Inward-facing walls; returns list of (name, (2,9) triangles)."""
    out = [
        ("left", quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1))),
        ("right", quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0))),
        ("floor", quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0))),
        ("ceiling", quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1))),
        ("back", quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0))),
    ]
    if front:
        out.append(("front", quad((x1, y0, z1), (x0, y0, z1), (x0, y1, z1), (x1, y1, z1))))
    return out


def box(center, size, angle_deg=0.0, bottom=False):
    """Outward-facing box rotated about Y through its centre. center=(cx,y0,cz), size=(sx,sy,sz)."""
    cx, y0, cz = center
    sx, sy, sz = size
    x0, x1, z0, z1, y1 = -sx / 2, sx / 2, -sz / 2, sz / 2, y0 + sy
    faces = [
        quad((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),  # +y
        quad((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # +z
        quad((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # -z
        quad((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # +x
        quad((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # -x
    ]
    if bottom:
        faces.append(quad((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)))
    t = np.concatenate(faces).reshape(-1, 3, 3).astype(np.float64)
    a = np.deg2rad(angle_deg)
    ca, sa = np.cos(a), np.sin(a)
    x, z = t[..., 0].copy(), t[..., 2].copy()
    t[..., 0] = ca * x + sa * z + cx
    t[..., 2] = -sa * x + ca * z + cz
    return t.reshape(-1, 9).astype(np.float32)


WHITE = (0.93, 0.93, 0.93)
RED = (0.55, 0.09, 0.09)
GREEN = (0.16, 0.55, 0.15)


def _cornell_materials(light_emission=10.0):
    # Slots follow the reference's overrides (Raytracer.cpp:163-171).
    return np.array([
        material(RED), material(GREEN), material(WHITE), material(WHITE), material(WHITE),
        material(WHITE), material(WHITE), material((0.78, 0.78, 0.78), light_emission),
    ], MATERIAL_DTYPE)


def cornell(glass_sphere=True, boxes=True, lights=1, light_emission=10.0) -> SceneData:
    """Procedural Cornell box, interior [-1,1]x[0,2]x[-1,1], open towards the camera (+z).

    lights=1: one quad light (2 triangles, the only case in which the reference's
    rand()%2 pick is correct).  lights=4: four quad lights (8 light triangles).
    """
    tris, mats = [], []
    slot = {"left": 0, "right": 1, "floor": 2, "ceiling": 3, "back": 4}
    for name, q in room(-1, 1, 0, 2, -1, 1):
        tris.append(q)
        mats += [slot[name]] * 2
    if boxes:
        b = box((0.35, 0.0, 0.25), (0.6, 0.6, 0.6), -17.0)
        tris.append(b)
        mats += [5] * len(b)
        b = box((-0.35, 0.0, -0.35), (0.6, 1.2, 0.6), 17.0)
        tris.append(b)
        mats += [6] * len(b)
    y = 1.98
    if lights == 1:
        rects = [(-0.25, 0.25, -0.25, 0.25)]
    else:
        rects = [(-0.7, -0.4, -0.7, -0.4), (0.4, 0.7, -0.7, -0.4), (-0.7, -0.4, 0.4, 0.7),
                 (0.4, 0.7, 0.4, 0.7)][:lights]
    for (x0, x1, z0, z1) in rects:
        tris.append(quad((x0, y, z0), (x1, y, z0), (x1, y, z1), (x0, y, z1)))  # faces down
        mats += [7, 7]
    sph = []
    if glass_sphere:
        sph.append(sphere((-0.5, 0.302, 0.55), 0.3, material((1.0, 1.0, 1.0), 0.0, SPECULAR)))
    return SceneData(np.concatenate(tris), np.array(mats, np.uint32), _cornell_materials(light_emission),
                     np.array(sph, SPHERE_DTYPE) if sph else np.zeros(0, SPHERE_DTYPE),
                     name=f"cornell_l{lights}")


def sphere_sky(glass=False) -> SceneData:
    """README sample1 (diffuse) / sample2 (glass): a sphere over a ground plane under the sky.
    The ground is two large triangles (the reference has no live Plane class).  Needs sky=1."""
    g = 1000.0
    tris = quad((-g, 0, g), (g, 0, g), (g, 0, -g), (-g, 0, -g))
    mats = np.array([material((0.73, 0.73, 0.73))], MATERIAL_DTYPE)
    sm = material((1.0, 1.0, 1.0), 0.0, SPECULAR) if glass else material(WHITE)
    sph = np.array([sphere((0.0, 1.0, -1.0), 0.63, sm)], SPHERE_DTYPE)
    return SceneData(tris, np.zeros(2, np.uint32), mats, sph, name="sample2_glass" if glass else "sample1_diffuse")


def oven(emission=0.5, albedo=0.5) -> SceneData:
    """Closed room around the camera; every surface emits `emission` with albedo `albedo`."""
    tris = np.concatenate([q for _, q in room(-2, 2, -1, 3, -2, 4, front=True)])
    mats = np.array([material((albedo,) * 3, emission)], MATERIAL_DTYPE)
    return SceneData(tris, np.zeros(len(tris), np.uint32), mats, name="oven")


def oven_objects() -> SceneData:
    """The oven with a glass and a mirror sphere inside.  Both have albedo 1 and no emission, and
    the integrator follows them with weight 1 (Integrator.h:198-205,248-278), so the radiance is
    still emission / (1 - albedo) = 1.0 everywhere: an analytic answer that exercises the
    specular branches, Russian roulette at the 0.99 cap and long chains inside the glass."""
    base = oven()
    sph = [sphere((-0.5, 0.6, 0.5), 0.5, material((1, 1, 1), 0.0, SPECULAR)),
           sphere((0.7, 0.8, 0.0), 0.6, material((1, 1, 1), 0.0, REFLECTIVE))]
    return SceneData(base.triangles, base.material_index, base.materials, np.array(sph, SPHERE_DTYPE),
                     name="oven_objects")


def reference_scene(i: int) -> SceneData:
    """The reference's commented-out analytic scenes (Raytracer.cpp:180-226): planes, spheres and
    axis-aligned boxes only, no mesh (so no NEE: the light list holds mesh triangles only).
    0: a plane, a glass sphere and a glass box (lit by the sky: use sky=1)
    1: Cornell box of six planes at +-800 with a glass sphere, a glass box and a box light
    2: the oven test: six planes of emission 0.5 / albedo 0.5
    3: Cornell planes with a diffuse sphere and a box light
    The +-800 rooms are centred on the origin like the reference's; the camera of Camera.h:23 at
    (0,1,3) is inside them."""
    mirror = material((1, 1, 1), 0.0, REFLECTIVE)  # noqa: F841  (mirrorMat, unused by the scenes as in the reference)
    dif = material(WHITE)
    red, green = material(RED), material(GREEN)
    glass = material((1, 1, 1), 0.0, SPECULAR)
    light = material((1, 1, 1), 16.0, SPECULAR)   # lightMat, Raytracer.cpp:159
    oven_m = material((0.5, 0.5, 0.5), 0.5)
    no_mesh = dict(triangles=np.zeros((0, 9), np.float32), material_index=np.zeros(0, np.uint32),
                   materials=np.array([dif], MATERIAL_DTYPE))

    def walls(m_back, m_front, m_top, m_bottom, m_right, m_left):
        return [plane_obj((0, 0, -800), (0, 0, 1), m_back), plane_obj((0, 0, 800), (0, 0, -1), m_front),
                plane_obj((0, 800, 0), (0, -1, 0), m_top), plane_obj((0, -800, 0), (0, 1, 0), m_bottom),
                plane_obj((800, 0, 0), (-1, 0, 0), m_right), plane_obj((-800, 0, 0), (1, 0, 0), m_left)]

    if i == 0:
        return SceneData(**no_mesh, name="ref_scene0",
                         planes=np.array([plane_obj((0, 2.05, 0), (0, -1, 0), dif)], PLANE_DTYPE),
                         spheres=np.array([sphere((1.7, 0.5, -4.0), 1.3, glass)], SPHERE_DTYPE),
                         boxes=np.array([box_obj((-1.4, 0.5, -3.0), (-0.4, 1.5, -2.0), glass)], BOX_DTYPE))
    if i == 1:
        return SceneData(**no_mesh, name="ref_scene1",
                         planes=np.array(walls(dif, dif, dif, dif, green, red), PLANE_DTYPE),
                         spheres=np.array([sphere((400, 440, -600), 200.3, glass)], SPHERE_DTYPE),
                         boxes=np.array([box_obj((-200, -800, -500), (200, -750, -100), glass),
                                         box_obj((-200, -800, -500), (200, -799.9, -100), light)], BOX_DTYPE))
    if i == 2:
        return SceneData(**no_mesh, name="ref_scene2_oven",
                         planes=np.array(walls(*([oven_m] * 6)), PLANE_DTYPE))
    if i == 3:
        return SceneData(**no_mesh, name="ref_scene3",
                         planes=np.array(walls(dif, dif, dif, dif, green, red), PLANE_DTYPE),
                         spheres=np.array([sphere((0, -40, -500), 200.3, dif)], SPHERE_DTYPE),
                         boxes=np.array([box_obj((-50, -800, -650), (50, -799.9, -550), light)], BOX_DTYPE))
    raise ValueError("scene index 0..3")


def _hash32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    st = (x * 747796405 + 2891336453) & 0xFFFFFFFF
    w = (((st >> ((st >> 28) + 4)) ^ st) * 277803737) & 0xFFFFFFFF
    return (((w >> 22) ^ w) & 0xFFFFFFFF).astype(np.uint32)


def terrain(cells=707, light_emission=10.0) -> SceneData:
    """Height field of 2*cells^2 triangles (cells=707 -> 999,698) on the floor of the Cornell room."""
    n = cells + 1
    xs = np.linspace(-0.98, 0.98, n, dtype=np.float64)
    X, Z = np.meshgrid(xs, xs, indexing="xy")  # X[j,i]=xs[i], Z[j,i]=xs[j]
    idx = (np.arange(n)[:, None] * n + np.arange(n)[None, :]).astype(np.uint32)
    jitter = _hash32(idx).astype(np.float64) / 2.0**32
    Y = 0.22 + 0.10 * np.sin(5.1 * X + 0.3) * np.cos(4.3 * Z - 0.7) + 0.05 * np.sin(17.0 * X * Z) \
        + 0.012 * jitter
    P = np.stack([X, Y, Z], -1).astype(np.float32)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]  # [z index, x index]
    # +y winding: (x0,z1),(x1,z1),(x1,z0) and (x0,z1),(x1,z0),(x0,z0)
    t1 = np.concatenate([p01, p11, p10], -1).reshape(-1, 9)
    t2 = np.concatenate([p01, p10, p00], -1).reshape(-1, 9)
    field_tris = np.stack([t1, t2], 1).reshape(-1, 9)
    tris, mats = [], []
    slot = {"left": 0, "right": 1, "floor": 2, "ceiling": 3, "back": 4}
    for name, q in room(-1, 1, 0, 2, -1, 1):
        tris.append(q)
        mats += [slot[name]] * 2
    y = 1.98
    tris.append(quad((-0.25, y, -0.25), (0.25, y, -0.25), (0.25, y, 0.25), (-0.25, y, 0.25)))
    mats += [7, 7]
    tris.append(field_tris)
    mats = np.concatenate([np.array(mats, np.uint32), np.full(len(field_tris), 5, np.uint32)])
    return SceneData(np.concatenate(tris), mats, _cornell_materials(light_emission), name=f"terrain_{cells}")


def soup(ntri=10000, nsph=3, seed=1, extent=1.0, size=0.15) -> SceneData:
    """Random triangle soup with mixed materials, for traversal parity tests."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (ntri, 1, 3)) + np.array([0.0, 1.0, 0.0])
    t = (c + rng.uniform(-size, size, (ntri, 3, 3))).astype(np.float32).reshape(-1, 9)
    mats = np.array([material(WHITE), material(RED), material(GREEN, 0.0, REFLECTIVE),
                     material((1, 1, 1), 0.0, SPECULAR), material((0.8, 0.8, 0.8), 4.0),
                     material((0.73,) * 3, 0.0, VARNISH)], MATERIAL_DTYPE)
    mi = rng.integers(0, len(mats), ntri).astype(np.uint32)
    sph = [sphere(rng.uniform(-0.6, 0.6, 3) + np.array([0, 1, 0]), float(rng.uniform(0.1, 0.3)),
                  mats[int(rng.integers(0, 4))]) for _ in range(nsph)]
    return SceneData(t, mi, mats, np.array(sph, SPHERE_DTYPE) if sph else np.zeros(0, SPHERE_DTYPE),
                     name=f"soup_{ntri}")


def quad_soup(nquad=2500, seed=1, extent=1.0, size=0.2) -> SceneData:
    """Random quadrilaterals as the reference's fan tessellation leaves them (Scene.h:55-115): consecutive
    triangles (a, b, c), (a, c, d) -- planar, folded and self-crossing ones, every tenth quad twice (hits at equal t),
    every seventh with its second half's vertices rotated to (c, d, a) (the same triangle, not the fan's order) and
    every eleventh with the halves swapped in the triangle order."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (nquad, 1, 3)) + np.array([0.0, 1.0, 0.0])
    v = (c + rng.uniform(-size, size, (nquad, 4, 3))).astype(np.float32)
    planar = rng.random(nquad) < 0.5     # d = a + (c - b): a parallelogram
    v[planar, 3] = (v[planar, 0] + (v[planar, 2] - v[planar, 1])).astype(np.float32)
    tris = []
    for k in range(nquad):
        a, b, cc, d = v[k]
        t0, t1 = np.concatenate([a, b, cc]), np.concatenate([a, cc, d])
        if k % 7 == 3:
            t1 = np.concatenate([cc, d, a])
        pair = [t1, t0] if k % 11 == 5 else [t0, t1]
        tris += pair
        if k % 10 == 0:
            tris += pair
    t = np.array(tris, np.float32)
    mats = np.array([material(WHITE), material(RED), material(GREEN, 0.0, REFLECTIVE),
                     material((1, 1, 1), 0.0, SPECULAR), material((0.8, 0.8, 0.8), 4.0)], MATERIAL_DTYPE)
    mi = rng.integers(0, len(mats), len(t)).astype(np.uint32)
    return SceneData(t, mi, mats, name=f"quad_soup_{nquad}")


def random_rays(n, seed=7, origin_box=((-1.2, -0.2, -1.2), (1.2, 2.2, 3.2)), unit=True) -> np.ndarray:
    """n rays (o, d) with origins uniform in a box and directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(origin_box[0]), np.array(origin_box[1])
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if not unit:
        d *= rng.uniform(0.5, 2.0, (n, 1))
    return np.concatenate([o, d], 1).astype(np.float32)


def write_obj(scene: SceneData, obj_path: str, num_slots: int = 8) -> None:
    """Writes the scene's triangles as a Wavefront OBJ with an MTL of `num_slots` materials
    (m0, m1, ...: one slot per material index, the numbering load_mesh derives from the MTL,
    Scene.h:55-115).  Vertices are shared where their coordinates are identical; coordinates
    are printed with nine significant digits, which a float32 survives unchanged.  Faces keep
    the triangle order and each triangle's vertex order."""
    import os
    tris = scene.triangles.reshape(-1, 3, 3)
    verts, inverse = np.unique(tris.reshape(-1, 3), axis=0, return_inverse=True)
    faces = inverse.reshape(-1, 3) + 1
    mtl_name = os.path.splitext(os.path.basename(obj_path))[0] + ".mtl"
    with open(os.path.join(os.path.dirname(obj_path), mtl_name), "w") as f:
        for i in range(max(num_slots, int(scene.material_index.max()) + 1 if len(scene.material_index) else 0)):
            f.write(f"newmtl m{i}\nKd 0.5 0.5 0.5\n")
    with open(obj_path, "w") as f:
        f.write(f"mtllib {mtl_name}\n")
        np.savetxt(f, verts, fmt="v %.9g %.9g %.9g")
        mi = scene.material_index
        # runs of equal material: one usemtl line per run
        starts = np.concatenate([[0], np.nonzero(np.diff(mi.astype(np.int64)))[0] + 1, [len(mi)]])
        for a, b in zip(starts[:-1], starts[1:]):
            f.write(f"usemtl m{int(mi[a])}\n")
            np.savetxt(f, faces[a:b], fmt="f %d %d %d")


def read_mesh_dump(path: str):
    """(triangles (n, 9) float32, material slots (n,) uint32, number of materials) from the
    binary file rt_host --dump-mesh writes."""
    with open(path, "rb") as f:
        ntri, nmat = (int(x) for x in np.frombuffer(f.read(16), "<u8"))
        tris = np.frombuffer(f.read(36 * ntri), "<f4").reshape(ntri, 9)
        mats = np.frombuffer(f.read(4 * ntri), "<u4")
    assert len(mats) == ntri
    return tris.copy(), mats.copy(), nmat


def through_obj(scene: SceneData, workdir=None) -> SceneData:
    """The same scene with its mesh taken through the OBJ path: written as OBJ + MTL, parsed by
    the C++ host layer's load_obj (raytracer_amd/host, the replacement of Scene::load_mesh,
    Scene.h:55-115) and read back.  Raises if the file round trip changed a bit."""
    import os
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "rt_host")
    if not os.path.exists(exe):
        raise FileNotFoundError(f"{exe} not built (make -C raytracer_amd/csrc host)")
    with tempfile.TemporaryDirectory(dir=workdir) as d:
        obj, dump = os.path.join(d, "mesh.obj"), os.path.join(d, "mesh.bin")
        write_obj(scene, obj)
        subprocess.check_call([exe, "--obj", obj, "--dump-mesh", dump], stdout=subprocess.DEVNULL)
        tris, mats, _ = read_mesh_dump(dump)
    if not (np.array_equal(tris.view(np.uint32), scene.triangles.view(np.uint32)) and
            np.array_equal(mats, scene.material_index)):
        raise ValueError("the OBJ round trip changed the mesh")
    return SceneData(tris, mats, scene.materials, scene.spheres, name=scene.name + "_obj", boxes=scene.boxes,
                     planes=scene.planes)
