"""ctypes binding of the C ABI in include/rtx_capi.h.

This is the Python-side host layer used by the tests and bench.py; the C++ host
layer for a Raytracer.cpp-shaped driver is raytracer_amd/host/.  Nothing here
computes: every method forwards to librtx_hip.so and raises RtxError with the
library's message on a non-zero status.  There is no fallback of any kind: a
missing library or GPU is an error.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtx_hip.so")


class RtxError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rtx status {status}: {message}")
        self.status = status


# ---- struct mirrors (same layouts as include/rtx_capi.h) ---------------------
class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("emission", C.c_float), ("surface", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("c", C.c_float * 3), ("r", C.c_float), ("material", Material)]


class Box(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("max", C.c_float * 3), ("material", Material)]


class Plane(C.Structure):
    _fields_ = [("point", C.c_float * 3), ("normal", C.c_float * 3), ("material", Material)]


class CConfig(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("fov", C.c_float),
        ("cam_origin", C.c_float * 3), ("eps", C.c_float), ("nee_threshold", C.c_float),
        ("rr_cap", C.c_float), ("sky", C.c_int32), ("max_bounces", C.c_uint32),
        ("camera", C.c_int32), ("integrator", C.c_int32),
        ("focal_length", C.c_float), ("lens_radius", C.c_float),
        ("stretch_x", C.c_float), ("stretch_y", C.c_float),
        ("device", C.c_int32), ("pool_paths", C.c_uint32),
        ("finish_paths", C.c_uint32), ("result_buffer_mib", C.c_uint32), ("primary_rays", C.c_int32), ("fused_trace", C.c_int32),
        ("rng_streams", C.c_int32),
    ]


class CScene(C.Structure):
    _fields_ = [
        ("triangles", C.POINTER(C.c_float)), ("num_triangles", C.c_size_t),
        ("material_index", C.POINTER(C.c_uint32)),
        ("materials", C.POINTER(Material)), ("num_materials", C.c_size_t),
        ("spheres", C.POINTER(Sphere)), ("num_spheres", C.c_size_t),
        ("boxes", C.POINTER(Box)), ("num_boxes", C.c_size_t),
        ("planes", C.POINTER(Plane)), ("num_planes", C.c_size_t),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("paths", C.c_uint64), ("bounces", C.c_uint64), ("nee_samples", C.c_uint64),
        ("shadow_rays", C.c_uint64), ("iterations", C.c_uint64), ("launches", C.c_uint64),
        ("ms_generate", C.c_double), ("ms_intersect", C.c_double), ("ms_shade", C.c_double),
        ("ms_shadow", C.c_double), ("ms_finish", C.c_double), ("ms_reduce", C.c_double),
        ("ms_other", C.c_double),
        ("timed_launches_intersect", C.c_uint64), ("timed_launches_shade", C.c_uint64),
        ("timed_launches_shadow", C.c_uint64), ("timed_launches_generate", C.c_uint64),
        ("nan_paths", C.c_uint64), ("truncated_paths", C.c_uint64),
        ("drain_launches", C.c_uint64), ("overlapped_sums", C.c_uint64),
        ("stored_paths", C.c_uint64), ("device_allocations", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_uint32), ("num_triangles", C.c_uint32), ("num_lights", C.c_uint32),
        ("num_spheres", C.c_uint32), ("max_depth", C.c_uint32), ("stack_entries", C.c_uint32),
        ("build_ms", C.c_double), ("bvh_width", C.c_uint32), ("stack_bound", C.c_uint32),
    ]


# Every symbol include/rtx_capi.h declares (tests check the library exports them all).
SYMBOLS = [
    "rtx_last_error", "rtx_create", "rtx_destroy", "rtx_set_scene", "rtx_get_scene_info",
    "rtx_render", "rtx_resolve", "rtx_trace_closest", "rtx_trace_occluded", "rtx_get_stats",
    "rtx_reset_stats", "rtx_set_profiling", "rtx_accum_alloc", "rtx_accum_free", "rtx_accum_clear",
    "rtx_accum_upload", "rtx_debug_pcg", "rtx_trace_closest_noalloc",
    "rtx_comm_create_id", "rtx_comm_init", "rtx_comm_reduce", "rtx_comm_destroy", "rtx_stream_synchronize",
    "rtx_render_async", "rtx_wait", "rtx_debug_streams",
]

COMM_ID_BYTES = 128
_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Loads librtx_hip.so (built in-tree by raytracer_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RTX_LIB") or LIB_PATH  # RTX_LIB: another build of the same library (A/B runs)
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build it with `make -C raytracer_amd/csrc` "
            "(there is no CPU fallback for the renderer)")
    lib = C.CDLL(p)
    vp, u32, u64, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t
    fp = C.POINTER(C.c_float)
    lib.rtx_last_error.restype = C.c_char_p
    lib.rtx_last_error.argtypes = []
    lib.rtx_create.argtypes = [C.POINTER(CConfig), C.POINTER(vp)]
    lib.rtx_destroy.argtypes = [vp]
    lib.rtx_destroy.restype = None
    lib.rtx_set_scene.argtypes = [vp, C.POINTER(CScene)]
    lib.rtx_get_scene_info.argtypes = [vp, C.POINTER(SceneInfo)]
    lib.rtx_render.argtypes = [vp, vp, u32, u32, u64, fp, u32, u32, vp]
    lib.rtx_render_async.argtypes = [vp, vp, u32, u32, u64, fp, u32, u32, vp]
    lib.rtx_wait.argtypes = [vp, vp]
    lib.rtx_resolve.argtypes = [vp, vp, u32, vp, vp]
    lib.rtx_trace_closest.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    lib.rtx_trace_closest_noalloc.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    lib.rtx_trace_occluded.argtypes = [vp, vp, vp, sz, vp]
    lib.rtx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.rtx_reset_stats.argtypes = [vp]
    lib.rtx_set_profiling.argtypes = [vp, C.c_int]
    lib.rtx_accum_alloc.argtypes = [vp, C.POINTER(vp)]
    lib.rtx_accum_free.argtypes = [vp, vp]
    lib.rtx_accum_clear.argtypes = [vp, vp, vp]
    lib.rtx_accum_upload.argtypes = [vp, vp, vp, vp]
    lib.rtx_debug_pcg.argtypes = [vp, u32, u32, vp, vp, vp]
    lib.rtx_debug_streams.argtypes = [vp, u32, u32, u64, vp, vp]
    lib.rtx_comm_create_id.argtypes = [C.c_char_p]
    lib.rtx_comm_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.rtx_comm_reduce.argtypes = [vp, vp, C.c_int, vp]
    lib.rtx_comm_destroy.argtypes = [vp]
    lib.rtx_stream_synchronize.argtypes = [vp, vp]
    if path is None:
        _lib = lib
    return lib


@dataclass
class Config:
    """Run-time form of the reference's compile-time constants (see rtx_config)."""
    width: int = 800
    height: int = 800
    fov: float = 50.0
    cam_origin: Sequence[float] = (0.0, 1.0, 3.0)
    eps: float = 1.0e-3
    nee_threshold: float = 0.05
    rr_cap: float = 0.99
    sky: int = 0
    max_bounces: int = 0
    camera: int = 0
    integrator: int = 0
    focal_length: float = 2.4
    lens_radius: float = 0.09
    stretch_x: float = 1.2
    stretch_y: float = 1.2
    device: int = -1
    pool_paths: int = 0
    finish_paths: int = 0
    result_buffer_mib: int = 0
    primary_rays: int = 0
    fused_trace: int = 0
    rng_streams: int = 0

    def to_c(self) -> CConfig:
        c = CConfig()
        c.width, c.height, c.fov = self.width, self.height, self.fov
        c.cam_origin = (C.c_float * 3)(*self.cam_origin)
        c.eps, c.nee_threshold, c.rr_cap = self.eps, self.nee_threshold, self.rr_cap
        c.sky, c.max_bounces, c.camera, c.integrator = self.sky, self.max_bounces, self.camera, self.integrator
        c.focal_length, c.lens_radius = self.focal_length, self.lens_radius
        c.stretch_x, c.stretch_y = self.stretch_x, self.stretch_y
        c.device, c.pool_paths = self.device, self.pool_paths
        c.finish_paths, c.result_buffer_mib = self.finish_paths, self.result_buffer_mib
        c.primary_rays = self.primary_rays
        c.fused_trace = self.fused_trace
        c.rng_streams = self.rng_streams
        return c


MATERIAL_DTYPE = np.dtype([("albedo", "<f4", 3), ("emission", "<f4"), ("surface", "<i4")])
SPHERE_DTYPE = np.dtype([("c", "<f4", 3), ("r", "<f4"), ("material", MATERIAL_DTYPE)])
BOX_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("material", MATERIAL_DTYPE)])
PLANE_DTYPE = np.dtype([("point", "<f4", 3), ("normal", "<f4", 3), ("material", MATERIAL_DTYPE)])
assert BOX_DTYPE.itemsize == C.sizeof(Box) == 44 and PLANE_DTYPE.itemsize == C.sizeof(Plane) == 44
assert MATERIAL_DTYPE.itemsize == C.sizeof(Material) == 20
assert SPHERE_DTYPE.itemsize == C.sizeof(Sphere) == 36


@dataclass
class SceneData:
    """Flat scene arrays in the layout rtx_set_scene takes."""
    triangles: np.ndarray            # (n, 9) float32: a, b, c
    material_index: np.ndarray       # (n,) uint32
    materials: np.ndarray            # (m,) MATERIAL_DTYPE
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, SPHERE_DTYPE))
    name: str = ""
    boxes: np.ndarray = field(default_factory=lambda: np.zeros(0, BOX_DTYPE))
    planes: np.ndarray = field(default_factory=lambda: np.zeros(0, PLANE_DTYPE))

    def __post_init__(self):
        self.triangles = np.ascontiguousarray(self.triangles, dtype=np.float32).reshape(-1, 9)
        self.material_index = np.ascontiguousarray(self.material_index, dtype=np.uint32)
        self.materials = np.ascontiguousarray(self.materials, dtype=MATERIAL_DTYPE)
        self.spheres = np.ascontiguousarray(self.spheres, dtype=SPHERE_DTYPE)
        self.boxes = np.ascontiguousarray(self.boxes, dtype=BOX_DTYPE)
        self.planes = np.ascontiguousarray(self.planes, dtype=PLANE_DTYPE)
        assert self.triangles.shape[0] == self.material_index.shape[0]


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dev_ptr(buf) -> int:
    """Address of a device buffer: an int, or anything with data_ptr() (a torch tensor)."""
    if hasattr(buf, "data_ptr"):
        return int(buf.data_ptr())
    return int(buf)


class Renderer:
    """One rtx_ctx: the replacement for the reference's scene/camera/gens globals
    and its render(img, pathTrace) call (Raytracer.cpp:145-148,233,236-286)."""

    def __init__(self, cfg: Config, lib: Optional[C.CDLL] = None):
        self.lib = lib or load_library()
        self.cfg = cfg
        self._ctx = C.c_void_p()
        cc = cfg.to_c()
        self._check(self.lib.rtx_create(C.byref(cc), C.byref(self._ctx)))
        self._own_accum = []

    def _check(self, status: int):
        if status != 0:
            raise RtxError(status, self.lib.rtx_last_error().decode())

    def close(self):
        if self._ctx:
            for p in self._own_accum:
                self.lib.rtx_accum_free(self._ctx, p)
            self._own_accum = []
            self.lib.rtx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene ---------------------------------------------------------------
    def set_scene(self, s: SceneData) -> SceneInfo:
        cs = CScene()
        cs.triangles = s.triangles.ctypes.data_as(C.POINTER(C.c_float))
        cs.num_triangles = s.triangles.shape[0]
        cs.material_index = s.material_index.ctypes.data_as(C.POINTER(C.c_uint32))
        cs.materials = s.materials.ctypes.data_as(C.POINTER(Material))
        cs.num_materials = s.materials.shape[0]
        cs.spheres = s.spheres.ctypes.data_as(C.POINTER(Sphere)) if len(s.spheres) else None
        cs.num_spheres = len(s.spheres)
        cs.boxes = s.boxes.ctypes.data_as(C.POINTER(Box)) if len(s.boxes) else None
        cs.num_boxes = len(s.boxes)
        cs.planes = s.planes.ctypes.data_as(C.POINTER(Plane)) if len(s.planes) else None
        cs.num_planes = len(s.planes)
        self._check(self.lib.rtx_set_scene(self._ctx, C.byref(cs)))
        return self.scene_info()

    def scene_info(self) -> SceneInfo:
        info = SceneInfo()
        self._check(self.lib.rtx_get_scene_info(self._ctx, C.byref(info)))
        return info

    # -- accumulation buffers --------------------------------------------------
    def accum_alloc(self) -> int:
        p = C.c_void_p()
        self._check(self.lib.rtx_accum_alloc(self._ctx, C.byref(p)))
        self._own_accum.append(p.value)
        return p.value

    def accum_clear(self, accum, stream: int = 0):
        self._check(self.lib.rtx_accum_clear(self._ctx, _dev_ptr(accum), stream))

    def accum_upload(self, accum, sums: np.ndarray, stream: int = 0):
        """Per-pixel sums (H, W, 3) from a checkpoint into the device buffer."""
        a = np.ascontiguousarray(sums, np.float32)
        assert a.size == self.cfg.width * self.cfg.height * 3
        self._check(self.lib.rtx_accum_upload(self._ctx, _dev_ptr(accum), _ptr(a), stream))

    # -- rendering ---------------------------------------------------------------
    def render(self, accum, spp: int, sample_base: int = 0, seed: int = 0x5EED,
               cam_offset: Optional[Sequence[float]] = None, row0: int = 0, row1: int = 0,
               stream: int = 0):
        off = (C.c_float * 3)(*cam_offset) if cam_offset is not None else None
        self._check(self.lib.rtx_render(self._ctx, _dev_ptr(accum), spp, sample_base, seed, off,
                                        row0, row1, stream))

    def render_async(self, accum, spp: int, sample_base: int = 0, seed: int = 0x5EED,
                     cam_offset: Optional[Sequence[float]] = None, row0: int = 0, row1: int = 0,
                     stream: int = 0):
        """rtx_render_async: the frame is left in flight (at most two are); same sums as render()."""
        off = (C.c_float * 3)(*cam_offset) if cam_offset is not None else None
        self._check(self.lib.rtx_render_async(self._ctx, _dev_ptr(accum), spp, sample_base, seed, off,
                                              row0, row1, stream))

    def wait(self, accum=None):
        """Waits for the frames of render_async that add to `accum` (None: for all)."""
        self._check(self.lib.rtx_wait(self._ctx, _dev_ptr(accum) if accum is not None else None))

    def resolve(self, accum, total_spp: int, stream: int = 0) -> np.ndarray:
        out = np.empty((self.cfg.height, self.cfg.width, 3), np.float32)
        self._check(self.lib.rtx_resolve(self._ctx, _dev_ptr(accum), total_spp, _ptr(out), stream))
        return out

    # -- fixed ray sets ------------------------------------------------------------
    def trace_closest(self, rays: np.ndarray, want_normal: bool = True, noalloc: bool = False):
        """noalloc: through rtx_trace_closest_noalloc (no device allocation on the call path)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        kind = np.empty(n, np.uint32)
        prim = np.empty(n, np.uint32)
        t = np.empty(n, np.float32)
        nrm = np.empty((n, 3), np.float32) if want_normal else None
        fn = self.lib.rtx_trace_closest_noalloc if noalloc else self.lib.rtx_trace_closest
        self._check(fn(self._ctx, _ptr(rays), n, _ptr(kind), _ptr(prim), _ptr(t), _ptr(nrm)))
        return kind, prim, t, nrm

    def trace_occluded(self, rays: np.ndarray, dist: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        dist = np.ascontiguousarray(dist, np.float32).reshape(-1)
        assert dist.shape[0] == rays.shape[0]
        out = np.empty(rays.shape[0], np.uint8)
        self._check(self.lib.rtx_trace_occluded(self._ctx, _ptr(rays), _ptr(dist), rays.shape[0],
                                                _ptr(out)))
        return out

    # -- multi-GPU exchange step -----------------------------------------------------
    @staticmethod
    def comm_create_id(lib: Optional[C.CDLL] = None) -> bytes:
        """A fresh communicator id (made on one rank, carried to the others by the caller)."""
        lib = lib or load_library()
        buf = C.create_string_buffer(COMM_ID_BYTES)
        st = lib.rtx_comm_create_id(buf)
        if st != 0:
            raise RtxError(st, lib.rtx_last_error().decode())
        return buf.raw

    def comm_init(self, comm_id: bytes, rank: int, world_size: int):
        assert len(comm_id) == COMM_ID_BYTES
        self._check(self.lib.rtx_comm_init(self._ctx, comm_id, rank, world_size))

    def comm_reduce(self, accum, root: int = 0, stream: int = 0):
        """Enqueues the sum of every rank's buffer into the root's (RCCL reduce, in place)."""
        self._check(self.lib.rtx_comm_reduce(self._ctx, _dev_ptr(accum), root, stream))

    def comm_destroy(self):
        self._check(self.lib.rtx_comm_destroy(self._ctx))

    def stream_synchronize(self, stream: int = 0):
        self._check(self.lib.rtx_stream_synchronize(self._ctx, stream))

    def debug_pcg(self, seed: int, n: int):
        """(rand_pcg draws, Uniform draws, pcg_hash(seed + i)) computed on the device."""
        rnd, uni, hsh = np.empty(n, np.uint32), np.empty(n, np.float32), np.empty(n, np.uint32)
        self._check(self.lib.rtx_debug_pcg(self._ctx, seed, n, _ptr(rnd), _ptr(uni), _ptr(hsh)))
        return rnd, uni, hsh

    def debug_streams(self, out_dev, spp: int, sample_base: int = 0, seed: int = 0x5EED, stream: int = 0):
        """state | increment << 32 of every sample's generator into a device array of width*height*spp uint64."""
        self._check(self.lib.rtx_debug_streams(self._ctx, spp, sample_base, seed, _dev_ptr(out_dev), stream))

    # -- statistics ------------------------------------------------------------------
    def stats(self) -> Stats:
        s = Stats()
        self._check(self.lib.rtx_get_stats(self._ctx, C.byref(s)))
        return s

    def reset_stats(self):
        self._check(self.lib.rtx_reset_stats(self._ctx))

    def set_profiling(self, on: bool):
        self._check(self.lib.rtx_set_profiling(self._ctx, 1 if on else 0))
