"""The C++ host layer (raytracer_amd/host): the Raytracer.cpp-shaped driver above the C ABI.
CPU: it builds with plain g++ and its OBJ front end reproduces Scene::load_mesh's rules
(convex faces come out as a fan around the first vertex, one material slot per MTL material,
Scene.h:55-115; tests/test_formats.py holds the comparison with the reference's own loader).
GPU: a progressive render through the driver equals the same render through the Python binding."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "host", "rt_host")


@pytest.fixture(scope="module")
def rt_host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "raytracer_amd", "csrc"), "host"])
    return EXE


OBJ = """# test mesh
mtllib test.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 1.5 0
v 0 0 1
usemtl second
f 1 2 3 4
usemtl first
f 1/1/1 2/2/2 3/3/3
f -6//1 -5//1 -4//1 -3//1 -2//1
usemtl unknown
f 1 2 6
"""
MTL = "newmtl first\nKd 1 0 0\nnewmtl second\nKd 0 1 0\n"


def test_obj_front_end_matches_load_mesh_rules(rt_host, tmp_path):
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    out = subprocess.check_output([rt_host, "--obj", str(tmp_path / "test.obj"), "--parse-only"]).decode().split("\n")
    assert out[0] == "triangles 7 materials 3"   # 2 MTL materials + one slot for faces without a known material
    rows = np.array([[float(x) for x in l.split()] for l in out[1:8]])
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 1.5, 0], [0, 0, 1]], float)
    want = [
        (1, 0, 1, 2), (1, 0, 2, 3),              # quad, material "second" = slot 1: fan (a, b, c), (a, c, d)
        (0, 0, 1, 2),                            # v/vt/vn form, material "first" = slot 0
        (0, 0, 1, 2), (0, 0, 2, 3), (0, 0, 3, 4),  # pentagon with negative indices: fan around the first vertex
        (2, 0, 1, 5),                            # unknown material -> the extra slot
    ]
    for row, (m, a, b, c) in zip(rows, want):
        assert int(row[0]) == m
        np.testing.assert_array_equal(row[1:], np.concatenate([v[a], v[b], v[c]]))


def test_driver_reports_errors(rt_host, tmp_path):
    r = subprocess.run([rt_host, "--obj", str(tmp_path / "missing.obj"), "--parse-only"], capture_output=True)
    assert r.returncode == 1 and b"cannot open" in r.stderr
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    r = subprocess.run([rt_host, "--obj", str(tmp_path / "bad.obj"), "--parse-only"], capture_output=True)
    assert r.returncode == 1 and b"bad face index" in r.stderr


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        assert float(f.readline()) < 0
        data = np.frombuffer(f.read(), "<f4").reshape(h, w, 3)
    return data[::-1]  # stored bottom row first


@pytest.mark.gpu
def test_driver_render_equals_python_binding(rt_host, tmp_path):
    from raytracer_amd.capi import Config, Renderer
    out = str(tmp_path / "img")
    log = subprocess.check_output([rt_host, "--width", "64", "--height", "48", "--spp", "4", "--passes", "2",
                                   "--out", out]).decode()
    assert "Accumulated 8 samples per pixel" in log and "Saved" in log
    got = read_pfm(out + ".pfm")
    # The same scene as the driver's procedural_cornell(): slot 7 keeps the loader's placeholder
    # albedo and only gets emission 10 (Raytracer.cpp:171).
    sc = scenes.cornell(boxes=False)
    sc.materials[7]["albedo"] = (0.65, 0.05, 0.05)
    with Renderer(Config(width=64, height=48)) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 8)
        want = r.resolve(acc, 8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ppm = open(out + ".ppm", "rb").read()
    assert ppm.startswith(b"P6\n64 48\n255\n") and len(ppm) == 13 + 64 * 48 * 3


@pytest.mark.gpu
def test_driver_checkpoint_and_resume_equal_one_run(rt_host, tmp_path):
    """Two runs of two passes, the second resumed from the first's checkpoint, write the image of one
    run of four passes, bit for bit (sums, sample count and sample indices carry over)."""
    common = [rt_host, "--width", "72", "--height", "40", "--spp", "3"]
    subprocess.check_call(common + ["--passes", "4", "--out", str(tmp_path / "whole")], stdout=subprocess.DEVNULL)
    subprocess.check_call(common + ["--passes", "2", "--out", str(tmp_path / "half"),
                                    "--checkpoint", str(tmp_path / "half.ckpt")], stdout=subprocess.DEVNULL)
    log = subprocess.check_output(common + ["--passes", "2", "--out", str(tmp_path / "resumed"),
                                            "--resume", str(tmp_path / "half.ckpt")]).decode()
    assert "Resumed" in log and "at 6 samples per pixel" in log and "Accumulated 12 samples per pixel" in log
    whole, half, resumed = (read_pfm(str(tmp_path / n) + ".pfm") for n in ("whole", "half", "resumed"))
    assert np.array_equal(whole.view(np.uint32), resumed.view(np.uint32))
    assert not np.array_equal(whole, half)
    assert (tmp_path / "resumed.exr").read_bytes() == (tmp_path / "whole.exr").read_bytes()
    r = subprocess.run(common + ["--width", "64", "--resume", str(tmp_path / "half.ckpt")], capture_output=True)
    assert r.returncode == 1 and b"does not fit" in r.stderr
    # another configuration (sky on) or a damaged file is refused as well; no half-written file is left
    r = subprocess.run(common + ["--sky", "--resume", str(tmp_path / "half.ckpt")], capture_output=True)
    assert r.returncode == 1 and b"another scene or configuration" in r.stderr
    data = (tmp_path / "half.ckpt").read_bytes()
    assert data[:8] == b"RTXCKPT1" and len(data) == 8 + 8 + 24 + 72 * 40 * 12
    assert int.from_bytes(data[16:24], "little") == 6 and int.from_bytes(data[24:32], "little") == 0x5EED
    (tmp_path / "cut.ckpt").write_bytes(data[:-100])
    r = subprocess.run(common + ["--resume", str(tmp_path / "cut.ckpt")], capture_output=True)
    assert r.returncode == 1 and b"cut short" in r.stderr
    assert not list(tmp_path.glob("*.tmp"))


@pytest.mark.gpu
def test_driver_passes_in_flight_write_the_same_files(rt_host, tmp_path):
    """--in-flight (rtx_render_async, two passes at a time, image read back at checkpoints and at the
    end) writes the files of the pass-by-pass run, byte for byte, checkpoint included."""
    common = [rt_host, "--width", "72", "--height", "40", "--spp", "3", "--passes", "5", "--checkpoint-every", "2"]
    subprocess.check_call(common + ["--out", str(tmp_path / "a"), "--checkpoint", str(tmp_path / "a.ckpt")],
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(common + ["--in-flight", "--out", str(tmp_path / "b"), "--checkpoint", str(tmp_path / "b.ckpt")],
                          stdout=subprocess.DEVNULL)
    for ext in (".pfm", ".exr", ".ppm", ".ckpt"):
        assert (tmp_path / ("a" + ext)).read_bytes() == (tmp_path / ("b" + ext)).read_bytes(), ext


def test_write_obj_round_trip_through_load_obj(rt_host, tmp_path):
    """scenes.through_obj: a mesh written as OBJ + MTL and parsed by the host layer's load_obj comes
    back bit for bit (coordinates, triangle order, vertex order, material slots)."""
    sc = scenes.terrain(24)
    back = scenes.through_obj(sc, workdir=str(tmp_path))
    assert np.array_equal(back.triangles.view(np.uint32), sc.triangles.view(np.uint32))
    assert np.array_equal(back.material_index, sc.material_index)
    soup = scenes.soup(500, 0)
    back = scenes.through_obj(soup, workdir=str(tmp_path))
    assert np.array_equal(back.triangles.view(np.uint32), soup.triangles.view(np.uint32))


@pytest.mark.gpu
def test_million_triangle_obj_through_the_driver_equals_python_binding(rt_host, tmp_path):
    """BASELINE configs[3] is an OBJ mesh scene: the 1 M-triangle terrain written as an OBJ file,
    loaded by the driver's load_mesh path (Scene.h:55-115), with buildScene's material overrides
    and glass sphere (Raytracer.cpp:163-176), renders the bits the Python binding renders from the
    arrays it was written from."""
    from raytracer_amd.capi import Config, Renderer
    sc = scenes.terrain(707)
    obj = str(tmp_path / "terrain.obj")
    scenes.write_obj(sc, obj)
    out = str(tmp_path / "img")
    log = subprocess.check_output([rt_host, "--obj", obj, "--width", "320", "--height", "180", "--spp", "2",
                                   "--passes", "2", "--out", out]).decode()
    assert "999710 triangles, 8 materials" in log and "Accumulated 4 samples per pixel" in log
    got = read_pfm(out + ".pfm")
    # what the driver builds: slots 0/1 red/green, 2..6 white, 7 = the loader's placeholder albedo with
    # emission 10, plus the glass sphere
    mats = np.array([scenes.material((0.55, 0.09, 0.09)), scenes.material((0.16, 0.55, 0.15))] +
                    [scenes.material((0.93, 0.93, 0.93))] * 5 + [scenes.material((0.65, 0.05, 0.05), 10.0)],
                    scenes.MATERIAL_DTYPE)
    glass = np.array([scenes.sphere((-0.5, 0.302, 0.55), 0.3, scenes.material((1, 1, 1), 0.0, scenes.SPECULAR))],
                     scenes.SPHERE_DTYPE)
    want_scene = scenes.SceneData(sc.triangles, sc.material_index, mats, glass)
    with Renderer(Config(width=320, height=180)) as r:
        info = r.set_scene(want_scene)
        assert info.bvh_width == 4
        acc = r.accum_alloc()
        r.render(acc, 4)
        want = r.resolve(acc, 4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def gpu_count():
    import ctypes
    n = ctypes.c_int(0)
    try:
        ctypes.CDLL("libamdhip64.so.7").hipGetDeviceCount(ctypes.byref(n))
    except OSError:
        return 0
    return n.value


@pytest.mark.gpu
def test_driver_exchange_step_with_one_rank(rt_host, tmp_path):
    """rt_host --comm: the multi-GPU start-up (child process per rank, communicator id through a
    file, ncclCommInitRank, ncclReduce on the render stream, resolve on the root) with a single
    rank gives the bits of the plain single-process run.  (RCCL refuses two ranks on one GPU, so
    a one-GPU box can exercise everything but the wire.)"""
    common = [rt_host, "--width", "96", "--height", "64", "--spp", "3", "--passes", "2"]
    log = subprocess.check_output(common + ["--gpus", "1", "--comm", "--out", str(tmp_path / "comm")],
                                  timeout=300).decode()
    assert "1 ranks" in log and "Accumulated 6 samples per pixel" in log
    subprocess.check_call(common + ["--out", str(tmp_path / "plain")], stdout=subprocess.DEVNULL)
    a, b = read_pfm(str(tmp_path / "comm") + ".pfm"), read_pfm(str(tmp_path / "plain") + ".pfm")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not list(tmp_path.glob("*.commid*"))


@pytest.mark.gpu
@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_driver_on_two_gpus_adds_up_to_one_gpu(rt_host, tmp_path):
    """rt_host --gpus 2: every rank renders its share of the sample indices, the library's RCCL
    reduce adds the sums on rank 0.  One pass of 4 spp is exactly rank 0's two samples plus rank
    1's two (a + b, whatever the order); over two passes the image agrees with the one-process
    render up to the order of the float additions."""
    from raytracer_amd.capi import Config, Renderer
    common = [rt_host, "--width", "96", "--height", "64", "--spp", "4"]
    log = subprocess.check_output(common + ["--gpus", "2", "--passes", "1", "--out", str(tmp_path / "two")],
                                  timeout=300).decode()
    assert "2 ranks" in log and "Accumulated 4 samples per pixel" in log and "rank 1: paths %d" % (96 * 64 * 2) in log
    got = read_pfm(str(tmp_path / "two") + ".pfm")
    sc = scenes.cornell(boxes=False)
    sc.materials[7]["albedo"] = (0.65, 0.05, 0.05)
    with Renderer(Config(width=96, height=64)) as r:
        r.set_scene(sc)
        a, b = r.accum_alloc(), r.accum_alloc()
        r.render(a, 2, sample_base=0)
        r.render(b, 2, sample_base=2)
        want = (r.resolve(a, 1) + r.resolve(b, 1)) / np.float32(4)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    subprocess.check_call(common + ["--gpus", "2", "--passes", "2", "--out", str(tmp_path / "two2")],
                          stdout=subprocess.DEVNULL, timeout=300)
    subprocess.check_call(common + ["--passes", "2", "--out", str(tmp_path / "one2")], stdout=subprocess.DEVNULL)
    two, one = read_pfm(str(tmp_path / "two2") + ".pfm"), read_pfm(str(tmp_path / "one2") + ".pfm")
    np.testing.assert_allclose(two, one, rtol=2e-6, atol=1e-7)


def test_rank_shares_cover_a_pass_exactly(tmp_path):
    """rt::split_range, the share of a pass's sample indices a rank takes: compiled from the host
    header, it cuts n samples into contiguous shares that differ by at most one and add up to n."""
    src = tmp_path / "split.cpp"
    src.write_text('''#include "rt_host.hpp"
#include <cstdio>
int main() {
  for (unsigned n : {0u, 1u, 4u, 7u, 256u, 1000u})
    for (int parts : {1, 2, 3, 8}) {
      unsigned next = 0, lo = ~0u, hi = 0;
      for (int i = 0; i < parts; i++) {
        uint32_t first, count;
        rt::split_range(n, parts, i, first, count);
        if (first != next) return 1;
        next = first + count;
        lo = count < lo ? count : lo;
        hi = count > hi ? count : hi;
      }
      if (next != n || hi - lo > 1) return 2;
    }
  std::puts("ok");
  return 0;
}
''')
    exe = tmp_path / "split"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I",
                           os.path.join(ROOT, "raytracer_amd", "host"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "raytracer_amd"), "-lrtx_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "raytracer_amd")])
    assert subprocess.check_output([str(exe)]).decode().strip() == "ok"
