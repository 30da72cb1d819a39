"""ctypes binding of the CPU oracle (oracle/librt_oracle.so) and of the reference's
own Pcg build (oracle/_ref/libref_pcg.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (raytracer_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_oracle.so")
REF_PCG_PATH = os.path.join(_HERE, "_ref", "libref_pcg.so")
REF_IO_PATH = os.path.join(_HERE, "_ref", "libref_io.so")


class OMaterial(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("emission", C.c_float), ("surface", C.c_int32)]


class OSphere(C.Structure):
    _fields_ = [("c", C.c_float * 3), ("r", C.c_float), ("material", OMaterial)]


class OBox(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3), ("material", OMaterial)]


class OPlane(C.Structure):
    _fields_ = [("point", C.c_float * 3), ("normal", C.c_float * 3), ("material", OMaterial)]


class OConfig(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("fov", C.c_float),
        ("cam_origin", C.c_float * 3), ("eps", C.c_float), ("nee_threshold", C.c_float),
        ("rr_cap", C.c_float), ("sky", C.c_int32), ("max_bounces", C.c_uint32),
        ("camera", C.c_int32), ("integrator", C.c_int32),
        ("focal_length", C.c_float), ("lens_radius", C.c_float),
        ("stretch_x", C.c_float), ("stretch_y", C.c_float),
        ("rng_streams", C.c_int32),
    ]


class OStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("bounces", C.c_uint64), ("nee_samples", C.c_uint64),
                ("nan_paths", C.c_uint64)]


_lib = None


def build(force: bool = False):
    """Compiles the oracle with its Makefile (g++); the reference Pcg too when the
    reference checkout exists."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, sz, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_int
        L.orc_pcg_hash.restype = u32
        L.orc_pcg_hash.argtypes = [u32]
        L.orc_pcg_rand.argtypes = [C.POINTER(u32), vp, sz]
        L.orc_pcg_uniform.argtypes = [C.POINTER(u32), vp, sz]
        L.orc_path_seed.restype = u32
        L.orc_path_seed.argtypes = [u32, u32, u64]
        L.orc_rot_matrix_vectors.argtypes = [vp, vp, vp]
        L.orc_mat_vec.argtypes = [vp, vp, vp]
        L.orc_cosine_sample_hemisphere.argtypes = [C.POINTER(u32), vp]
        L.orc_schlick.restype = C.c_float
        L.orc_schlick.argtypes = [C.c_float, C.c_float]
        L.orc_sky_emission.argtypes = [vp, vp]
        L.orc_triangle_intersect.restype = C.c_float
        L.orc_triangle_intersect.argtypes = [vp] * 5
        L.orc_sphere_intersect.restype = C.c_float
        L.orc_sphere_intersect.argtypes = [vp, C.c_float, vp, vp]
        L.orc_triangle_sample_point.restype = C.c_float
        L.orc_triangle_sample_point.argtypes = [vp, vp, vp, C.POINTER(u32), vp, vp]
        L.orc_camera_ray.argtypes = [C.POINTER(OConfig), C.c_float, C.c_float, C.POINTER(u32), vp, vp]
        L.orc_img_update.argtypes = [vp, u32, vp]
        L.orc_scene_create.restype = vp
        L.orc_scene_create.argtypes = [vp, sz, vp, vp, sz, vp, sz, vp, sz, vp, sz, C.POINTER(OConfig)]
        L.orc_box_intersect.restype = C.c_float
        L.orc_box_intersect.argtypes = [vp] * 4
        L.orc_box_normal.argtypes = [vp] * 4
        L.orc_plane_intersect.restype = C.c_float
        L.orc_plane_intersect.argtypes = [vp] * 4
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_destroy.restype = None
        L.orc_scene_num_lights.restype = sz
        L.orc_scene_num_lights.argtypes = [vp]
        L.orc_trace_closest.argtypes = [vp, vp, sz, i, vp, vp, vp, vp]
        L.orc_trace_occluded.argtypes = [vp, vp, vp, sz, i, vp]
        L.orc_path_trace.argtypes = [vp, u32, u32, u64, vp, i, vp, C.POINTER(u32)]
        L.orc_render.argtypes = [vp, u32, u32, u64, vp, i, i, u32, u32, vp, C.POINTER(OStats)]
        _lib = L
    return _lib


def ref_pcg() -> Optional[C.CDLL]:
    """The reference's own Sampler.h compiled unchanged (None if it was never built)."""
    if not os.path.exists(REF_PCG_PATH):
        return None
    L = C.CDLL(REF_PCG_PATH)
    L.ref_pcg_hash.restype = C.c_uint32
    L.ref_pcg_hash.argtypes = [C.c_uint32]
    L.ref_pcg_rand.argtypes = [C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
    L.ref_pcg_uniform.argtypes = [C.POINTER(C.c_uint32), C.c_void_p, C.c_size_t]
    if hasattr(L, "ref_eps"):
        L.ref_eps.restype = C.c_float
        L.ref_nee_emission_threshold.restype = C.c_float
    return L


def ref_io() -> Optional[C.CDLL]:
    """The reference's vendored OBJ reader and EXR writer compiled unchanged (None if never built)."""
    if not os.path.exists(REF_IO_PATH):
        return None
    L = C.CDLL(REF_IO_PATH)
    L.ref_obj_triangles.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_int)),
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_exr_save.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    L.ref_exr_load.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return L


def ref_obj_triangles(path: str):
    """(triangles [n,3,3] float32, material id per triangle int32 (-1 = none), number of materials)
    as the reference's loader sees the file: tinyobj::ObjReader with its default configuration and
    the fan loop of Scene::load_mesh (reference Scene.h:55-115)."""
    L = ref_io()
    tris, mats = C.POINTER(C.c_float)(), C.POINTER(C.c_int)()
    n, nm = C.c_size_t(), C.c_size_t()
    rc = L.ref_obj_triangles(path.encode(), C.byref(tris), C.byref(mats), C.byref(n), C.byref(nm))
    if rc != 0:
        raise RuntimeError(f"reference OBJ reader failed ({rc}) on {path}")
    t = np.ctypeslib.as_array(tris, (n.value, 3, 3)).copy() if n.value else np.zeros((0, 3, 3), np.float32)
    m = np.ctypeslib.as_array(mats, (n.value,)).copy() if n.value else np.zeros((0,), np.int32)
    L.ref_free(tris)
    L.ref_free(mats)
    return t, m, nm.value


def ref_exr_save(rgb: np.ndarray, path: str) -> None:
    """rgb [h,w,3] float32 through the call sequence of Img::saveEXR (reference Raytracer.cpp:83-143)."""
    a = np.ascontiguousarray(rgb, np.float32)
    rc = ref_io().ref_exr_save(_p(a), a.shape[1], a.shape[0], path.encode())
    if rc != 0:
        raise RuntimeError(f"reference EXR writer failed ({rc})")


def ref_exr_load(path: str) -> np.ndarray:
    """[h,w,4] float32 RGBA as tinyexr's LoadEXR reads the file."""
    L = ref_io()
    w, h = C.c_int(), C.c_int()
    rc = L.ref_exr_load(path.encode(), None, 0, C.byref(w), C.byref(h))
    out = np.zeros((h.value, w.value, 4), np.float32)
    rc = L.ref_exr_load(path.encode(), _p(out), out.size, C.byref(w), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"reference EXR reader failed ({rc})")
    return out


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return np.ascontiguousarray(v, np.float32).reshape(3)


def config_from(cfg) -> OConfig:
    """cfg: any object with the fields of raytracer_amd.capi.Config."""
    o = OConfig()
    o.width, o.height, o.fov = cfg.width, cfg.height, cfg.fov
    o.cam_origin = (C.c_float * 3)(*cfg.cam_origin)
    o.eps, o.nee_threshold, o.rr_cap = cfg.eps, cfg.nee_threshold, cfg.rr_cap
    o.sky, o.max_bounces, o.camera, o.integrator = cfg.sky, cfg.max_bounces, cfg.camera, cfg.integrator
    o.focal_length, o.lens_radius = cfg.focal_length, cfg.lens_radius
    o.stretch_x, o.stretch_y = cfg.stretch_x, cfg.stretch_y
    o.rng_streams = getattr(cfg, "rng_streams", 0)
    return o


def path_seed(pixel_index: int, sample_index: int, seed: int = 0x5EED) -> int:
    """The starting state of a sample's generator (the build's counter-based seeding)."""
    return int(lib().orc_path_seed(pixel_index, sample_index, seed))


def pcg_rand(seed: int, n: int, which=None):
    L = which or lib()
    st = C.c_uint32(seed)
    out = np.empty(n, np.uint32)
    (L.ref_pcg_rand if which is not None else L.orc_pcg_rand)(C.byref(st), _p(out), n)
    return out, st.value


def pcg_uniform(seed: int, n: int, which=None):
    L = which or lib()
    st = C.c_uint32(seed)
    out = np.empty(n, np.float32)
    (L.ref_pcg_uniform if which is not None else L.orc_pcg_uniform)(C.byref(st), _p(out), n)
    return out, st.value


def rot_matrix_vectors(a, b) -> np.ndarray:
    m = np.empty(9, np.float32)
    lib().orc_rot_matrix_vectors(_p(_f3(a)), _p(_f3(b)), _p(m))
    return m.reshape(3, 3)


def mat_vec(m, v) -> np.ndarray:
    out = np.empty(3, np.float32)
    lib().orc_mat_vec(_p(np.ascontiguousarray(m, np.float32).reshape(9)), _p(_f3(v)), _p(out))
    return out


def cosine_sample_hemisphere(state: int):
    st = C.c_uint32(state)
    out = np.empty(3, np.float32)
    lib().orc_cosine_sample_hemisphere(C.byref(st), _p(out))
    return out, st.value


def sky_emission(d) -> np.ndarray:
    out = np.empty(3, np.float32)
    lib().orc_sky_emission(_p(_f3(d)), _p(out))
    return out


def triangle_intersect(a, b, c, o, d) -> float:
    return float(lib().orc_triangle_intersect(_p(_f3(a)), _p(_f3(b)), _p(_f3(c)), _p(_f3(o)), _p(_f3(d))))


def sphere_intersect(c, r, o, d) -> float:
    return float(lib().orc_sphere_intersect(_p(_f3(c)), r, _p(_f3(o)), _p(_f3(d))))


def box_intersect(mn, mx, o, d) -> float:
    return float(lib().orc_box_intersect(_p(_f3(mn)), _p(_f3(mx)), _p(_f3(o)), _p(_f3(d))))


def box_normal(mn, mx, p) -> np.ndarray:
    out = np.empty(3, np.float32)
    lib().orc_box_normal(_p(_f3(mn)), _p(_f3(mx)), _p(_f3(p)), _p(out))
    return out


def plane_intersect(point, normal, o, d) -> float:
    return float(lib().orc_plane_intersect(_p(_f3(point)), _p(_f3(normal)), _p(_f3(o)), _p(_f3(d))))


def triangle_sample_point(a, b, c, state: int):
    st = C.c_uint32(state)
    p = np.empty(3, np.float32)
    n = np.empty(3, np.float32)
    rcp = lib().orc_triangle_sample_point(_p(_f3(a)), _p(_f3(b)), _p(_f3(c)), C.byref(st), _p(p), _p(n))
    return p, n, float(rcp), st.value


def camera_ray(cfg, px: float, py: float, state: int = 0):
    oc = config_from(cfg)
    st = C.c_uint32(state)
    o = np.empty(3, np.float32)
    d = np.empty(3, np.float32)
    lib().orc_camera_ray(C.byref(oc), px, py, C.byref(st), _p(o), _p(d))
    return o, d, st.value


def img_update(pixel, sample_count: int, color) -> np.ndarray:
    px = _f3(pixel).copy()
    lib().orc_img_update(_p(px), sample_count, _p(_f3(color)))
    return px


class OracleScene:
    """Scene + config for the oracle; `scene` is a raytracer_amd.capi.SceneData-like
    object (triangles (n,9) f32, material_index u32, materials/spheres structured arrays
    with the 20/36-byte layouts)."""

    def __init__(self, scene, cfg):
        self.cfg = cfg
        self._oc = config_from(cfg)
        boxes = getattr(scene, "boxes", np.zeros(0))
        planes = getattr(scene, "planes", np.zeros(0))
        self._keep = (scene.triangles, scene.material_index, scene.materials, scene.spheres, boxes, planes)
        L = lib()
        self._h = L.orc_scene_create(_p(scene.triangles), scene.triangles.shape[0],
                                     _p(scene.material_index), _p(scene.materials),
                                     scene.materials.shape[0],
                                     _p(scene.spheres) if len(scene.spheres) else None, len(scene.spheres),
                                     _p(boxes) if len(boxes) else None, len(boxes),
                                     _p(planes) if len(planes) else None, len(planes), C.byref(self._oc))
        if not self._h:
            raise ValueError("orc_scene_create failed (material index out of range?)")

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_lights(self) -> int:
        return int(lib().orc_scene_num_lights(self._h))

    def trace_closest(self, rays, use_bvh: bool = False, want_normal: bool = True):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        kind = np.empty(n, np.uint32)
        prim = np.empty(n, np.uint32)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32) if want_normal else None
        lib().orc_trace_closest(self._h, _p(rays), n, int(use_bvh), _p(kind), _p(prim), _p(t), _p(nrm))
        return kind, prim, t, nrm

    def trace_occluded(self, rays, dist, use_bvh: bool = False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        dist = np.ascontiguousarray(dist, np.float32).reshape(-1)
        out = np.empty(rays.shape[0], np.uint8)
        lib().orc_trace_occluded(self._h, _p(rays), _p(dist), rays.shape[0], int(use_bvh), _p(out))
        return out

    def path_trace(self, pixel_index: int, sample_index: int, seed: int = 0x5EED,
                   cam_offset: Optional[Sequence[float]] = None, use_bvh: bool = True):
        col = np.empty(3, np.float32)
        b = C.c_uint32()
        off = _f3(cam_offset) if cam_offset is not None else None
        lib().orc_path_trace(self._h, pixel_index, sample_index, seed, _p(off), int(use_bvh), _p(col),
                             C.byref(b))
        return col, b.value

    def render(self, spp: int, sample_base: int = 0, seed: int = 0x5EED,
               cam_offset: Optional[Sequence[float]] = None, use_bvh: bool = True, nthreads: int = 0,
               row0: int = 0, row1: int = 0, accum: Optional[np.ndarray] = None):
        """Returns (sum image (H,W,3) float32, OStats)."""
        if accum is None:
            accum = np.zeros((self.cfg.height, self.cfg.width, 3), np.float32)
        st = OStats()
        off = _f3(cam_offset) if cam_offset is not None else None
        lib().orc_render(self._h, spp, sample_base, seed, _p(off), int(use_bvh), nthreads, row0, row1,
                         _p(accum), C.byref(st))
        return accum, st
