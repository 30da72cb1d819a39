"""Generates the committed golden vectors.  Run from the repo root, in the container that
has /root/reference (the Pcg vectors come from the reference's own Sampler.h, compiled
unchanged into oracle/_ref/libref_pcg.so by oracle/Makefile):

    python tests/golden/make_golden.py

pcg_kat.json        rand_pcg / Uniform / pcg_hash sequences from the REFERENCE build
                    (this pins the oracle's generator to the reference).
functions.npz       camera rays on a 64x64 grid (800^2 and 1024^2, three cameras) and per-function
                    vectors (rotMatrixVectors, cosineSampleHemisphere, Triangle::samplePoint,
                    schlickApprox, skyEmission, box/plane/sphere/triangle intersect) from the
                    oracle.  Regression vectors of the restatement: parity unpinned.
image_*.npz         128x128 images of the oracle at IMAGE_SPP samples per pixel (sums / spp),
                    seed 0x5EED, with the bounce count.  Regression vectors as well.
formats/shapes.npz  what the REFERENCE's vendored OBJ reader (tiny_obj_loader.h behind Scene::load_mesh's
                    loop, compiled unchanged into oracle/_ref/libref_io.so) makes of formats/shapes.obj
                    + shapes.mtl (hand-written for this test): triangles, material ids, material count.
formats/image.npz   a small float image with ordinary and special values, and formats/image.exr, the
                    file the REFERENCE's writer (tinyexr.h through the call sequence of Img::saveEXR)
                    makes of it.  Both pin the host layer's file formats to the reference.
closest_*.npz       fixed ray sets with (kind, prim, t) from the oracle's first-party
                    linear scan.  These are regression vectors of the restatement, not
                    outputs of the reference (which cannot be built here): parity unpinned.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as orc  # noqa: E402
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.capi import Config  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pcg():
    ref = orc.ref_pcg()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_pcg.so missing: run `make -C oracle` where /root/reference exists")
    out = {"source": "reference Sampler.h:11-42 compiled unchanged (oracle/_ref/libref_pcg.so)", "seeds": {}}
    for seed in [0, 1, 12345, 0xDEADBEEF, 0xFFFFFFFF]:
        r, st = orc.pcg_rand(seed, 64, ref)
        u, st2 = orc.pcg_uniform(seed, 64, ref)
        out["seeds"][str(seed)] = {"rand": [int(x) for x in r], "state_after": int(st),
                                   "uniform_bits": [int(x) for x in u.view(np.uint32)]}
    out["constants"] = {"EPS_bits": int(np.float32(ref.ref_eps()).view(np.uint32)),
                        "NEE_EMISSION_THRESHOLD_bits": int(np.float32(ref.ref_nee_emission_threshold()).view(np.uint32)),
                        "WIDTH": int(ref.ref_width()), "HEIGHT": int(ref.ref_height()),
                        "source": "reference globals.h:3-8 compiled unchanged"}
    out["hash"] = {str(x): int(ref.ref_pcg_hash(x)) for x in [0, 1, 2, 12345, 0xDEADBEEF, 0xFFFFFFFF]}
    with open(os.path.join(HERE, "pcg_kat.json"), "w") as f:
        json.dump(out, f, indent=1)


def closest():
    cfg = Config(width=800, height=800)
    for name, sc in [("cornell", scenes.cornell()), ("soup2k", scenes.soup(2000, 3, seed=4))]:
        rays = np.concatenate([scenes.random_rays(3000, seed=31),
                               np.array([np.concatenate(orc.camera_ray(cfg, x, y)[:2])
                                         for x in np.linspace(0.5, 799.5, 32) for y in np.linspace(0.5, 799.5, 32)],
                                        np.float32)])
        osc = orc.OracleScene(sc, cfg)
        kind, prim, t, nrm = osc.trace_closest(rays, use_bvh=False)
        dist = np.random.default_rng(9).uniform(0, 3, rays.shape[0]).astype(np.float32)
        occ = osc.trace_occluded(rays, dist, use_bvh=False)
        np.savez_compressed(os.path.join(HERE, f"closest_{name}.npz"), rays=rays, kind=kind.astype(np.uint8),
                            prim=prim, t_bits=t.view(np.uint32), dist=dist, occluded=occ)


def functions():
    rng = np.random.default_rng(77)
    out = {}
    for res in (800, 1024):
        for cam, fov in ((0, 50.0), (1, 60.0), (2, 50.0)):
            cfg = Config(width=res, height=res, camera=cam, fov=fov)
            grid = [(x, y) for y in np.linspace(0.5, res - 0.5, 64) for x in np.linspace(0.5, res - 0.5, 64)]
            rays = np.array([np.concatenate(orc.camera_ray(cfg, x, y, state=int(i) * 2654435761 % 2**32)[:2])
                             for i, (x, y) in enumerate(grid)], np.float32)
            out[f"camera{cam}_{res}"] = rays
    n = rng.normal(size=(1000, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    out["rot_normals"] = n
    out["rot_matrices"] = np.array([orc.rot_matrix_vectors(v, (0, 0, 1)) for v in n], np.float32)
    states = rng.integers(0, 2**32, 512, dtype=np.uint64).astype(np.uint32)
    out["rng_states"] = states
    out["cosine_dirs"] = np.array([orc.cosine_sample_hemisphere(int(s))[0] for s in states], np.float32)
    tri = rng.uniform(-2, 2, (512, 9)).astype(np.float32)
    out["sample_tris"] = tri
    sp = [orc.triangle_sample_point(t[0:3], t[3:6], t[6:9], int(s)) for t, s in zip(tri, states)]
    out["sample_points"] = np.array([np.concatenate([x[0], x[1], [x[2]]]) for x in sp], np.float32)  # point, n, rcp_area
    cos = np.linspace(0, 1, 257, dtype=np.float32)
    out["schlick_cos"] = cos
    out["schlick"] = np.array([[orc.lib().orc_schlick(r, float(c)) for c in cos] for r in (0.5, 2.0)], np.float32)
    d = rng.normal(size=(512, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    out["sky_dirs"] = d
    out["sky"] = np.array([orc.sky_emission(v) for v in d], np.float32)
    o = rng.uniform(-3, 3, (512, 3)).astype(np.float32)
    out["isect_o"], out["isect_d"] = o, d
    out["isect_tri"] = np.array([orc.triangle_intersect(t[0:3], t[3:6], t[6:9], a, b)
                                 for t, a, b in zip(tri, o, d)], np.float32).view(np.uint32)
    out["isect_sphere"] = np.array([orc.sphere_intersect((0.1, -0.2, 0.3), 1.5, a, b)
                                    for a, b in zip(o, d)], np.float32).view(np.uint32)
    out["isect_box"] = np.array([orc.box_intersect((-1, -0.5, -1.5), (1, 0.75, 0.5), a, b)
                                 for a, b in zip(o, d)], np.float32).view(np.uint32)
    out["isect_plane"] = np.array([orc.plane_intersect((0, -1, 0), (0, 1, 0), a, b)
                                   for a, b in zip(o, d)], np.float32).view(np.uint32)
    np.savez_compressed(os.path.join(HERE, "functions.npz"), **out)


IMAGE_SPP = 1024
IMAGES = {
    "cornell": (lambda: scenes.cornell(), 0),
    "cornell_4lights": (lambda: scenes.cornell(lights=4), 0),
    "sample1": (lambda: scenes.sphere_sky(), 1),
    "sample2": (lambda: scenes.sphere_sky(glass=True), 1),
    "ref_scene0": (lambda: scenes.reference_scene(0), 1),
    # scenes past the LDS limit: the four-child tree, k_generate, the refilling closest-hit kernel
    "soup3k": (lambda: scenes.soup(3000, 3), 1),
    "terrain40": (lambda: scenes.terrain(40), 0),
}


def images():
    for name, (make, sky) in IMAGES.items():
        cfg = Config(width=128, height=128, sky=sky)
        img, st = orc.OracleScene(make(), cfg).render(IMAGE_SPP, seed=0x5EED)
        np.savez_compressed(os.path.join(HERE, f"image_{name}.npz"), mean=(img / IMAGE_SPP).astype(np.float32),
                            spp=IMAGE_SPP, sky=sky, bounces=st.bounces, nee_samples=st.nee_samples)


def exr_test_image():
    """13x7 image: smooth ramp, random magnitudes across the half range and beyond, and special values."""
    rng = np.random.default_rng(77)
    h, w = 7, 13
    img = (rng.standard_normal((h, w, 3)) * np.exp(rng.uniform(-22, 13, (h, w, 3)))).astype(np.float32)
    img[0] = np.linspace(0, 4, w * 3, dtype=np.float32).reshape(w, 3)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 65504.0, 65520.0, 65519.996, 6.1e-5, 5.96e-8, 2.98e-8,
                        2.9e-8, 1e-45, 1.0009766, 1.0004883, 1.0014648, 0.33333334, 2049.0, 2051.0], np.float32)
    img[1].reshape(-1)[:special.size] = special
    return img


def formats():
    if orc.ref_io() is None:
        raise SystemExit("oracle/_ref/libref_io.so missing: run `make -C oracle` where /root/reference exists")
    d = os.path.join(HERE, "formats")
    tris, mats, nmat = orc.ref_obj_triangles(os.path.join(d, "shapes.obj"))
    np.savez_compressed(os.path.join(d, "shapes.npz"), triangles=tris, material_ids=mats, num_materials=nmat)
    img = exr_test_image()
    np.savez_compressed(os.path.join(d, "image.npz"), rgb=img)
    orc.ref_exr_save(img, os.path.join(d, "image.exr"))


if __name__ == "__main__":
    formats()
    pcg()
    closest()
    functions()
    images()
    print("golden vectors written to", HERE)
