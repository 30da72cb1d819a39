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
