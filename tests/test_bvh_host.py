"""The product's host-side BVH builder (raytracer_amd/csrc/bvh_build.cpp), checked on the
CPU through a test-only probe: every triangle sits in exactly one leaf, child boxes
contain their triangles, the reported depth bounds every root-to-leaf path, and a CPU walk
of the flat tree returns what a linear scan returns."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    out = tmp_path_factory.mktemp("bvh") / "libbvh_probe.so"
    src = os.path.join(ROOT, "raytracer_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-I", src,
                           os.path.join(ROOT, "tests", "native", "bvh_probe.cpp"),
                           os.path.join(src, "bvh_build.cpp"), "-o", str(out)])
    lib = C.CDLL(str(out))
    lib.bvh_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def build(probe, tris, max_leaf=4, width=2, want_stack=False):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    n = probe.bvh_probe(tris.ctypes.data, len(tris), max_leaf, width, None, None, None, None)
    nodes = np.zeros((n, 16), np.float32)
    order = np.zeros(len(tris), np.uint32)
    depth = C.c_uint32()
    stack = C.c_uint32()
    probe.bvh_probe(tris.ctypes.data, len(tris), max_leaf, width, nodes.ctypes.data, order.ctypes.data,
                    C.byref(depth), C.byref(stack))
    if want_stack:
        return nodes, order, depth.value, stack.value
    return nodes, order, depth.value


def walk(nodes, order, tris, max_leaf=4):
    """Returns (leaf ranges seen, max depth); asserts box containment."""
    seen = np.zeros(len(order), np.int32)
    maxd = 0
    stack = [(0, 1)]
    kids = nodes[:, 12:14].copy().view(np.int32)
    while stack:
        i, d = stack.pop()
        maxd = max(maxd, d)
        q = nodes[i]
        boxes = [((q[0], q[2], q[8]), (q[1], q[3], q[9])), ((q[4], q[6], q[10]), (q[5], q[7], q[11]))]
        for (lo, hi), c in zip(boxes, kids[i]):
            if c >= 0:
                # the child's own children boxes must lie inside this child's box
                cq = nodes[c]
                clo = np.minimum([cq[0], cq[2], cq[8]], [cq[4], cq[6], cq[10]])
                chi = np.maximum([cq[1], cq[3], cq[9]], [cq[5], cq[7], cq[11]])
                assert np.all(clo >= np.array(lo) - 1e-4 * (1 + np.abs(lo))) and \
                       np.all(chi <= np.array(hi) + 1e-4 * (1 + np.abs(hi)))
                stack.append((int(c), d + 1))
            else:
                v = ~int(c)
                first, cnt = v >> 3, (v & 7) + 1
                assert cnt <= max_leaf and first + cnt <= len(order)
                seen[first:first + cnt] += 1
                pts = tris[order[first:first + cnt]].reshape(-1, 3)
                assert np.all(pts >= np.array(lo)) and np.all(pts <= np.array(hi))
    return seen, maxd


@pytest.mark.parametrize("name,make", [
    ("cornell", lambda: scenes.cornell().triangles), ("soup", lambda: scenes.soup(3000, 0).triangles),
    ("terrain", lambda: scenes.terrain(24).triangles),
    ("one", lambda: scenes.cornell().triangles[:1]), ("three", lambda: scenes.cornell().triangles[:3]),
    ("degenerate", lambda: np.tile(scenes.cornell().triangles[:1], (37, 1))),
])
@pytest.mark.parametrize("max_leaf", [1, 2, 4])
def test_flat_bvh_structure(name, make, max_leaf, probe):
    tris = make()
    nodes, order, depth = build(probe, tris, max_leaf)
    assert sorted(order.tolist()) == list(range(len(tris)))
    seen, maxd = walk(nodes, order, tris, max_leaf)
    if len(tris) > 1:
        assert np.all(seen == 1), "every triangle is in exactly one leaf"
    else:
        assert np.all(seen == 2)  # a single triangle is referenced by both children of the root
    assert maxd <= depth and depth < 64


def test_empty_mesh_has_no_nodes(probe):
    assert probe.bvh_probe(None, 0, 4, 2, None, None, None, None) == 0
    assert probe.bvh_probe(None, 0, 4, 4, None, None, None, None) == 0


def test_flat_bvh_walk_equals_linear_scan(probe, oracle):
    """Traversing the product's tree on the CPU (plain slab test in float, with the kernels'
    visiting rule) with the oracle's triangle test gives the oracle's linear-scan answer."""
    from raytracer_amd.capi import Config
    sc = scenes.soup(1500, 0, seed=8)
    nodes, order, depth = build(probe, sc.triangles)
    kids = nodes[:, 12:14].copy().view(np.int32)
    rays = scenes.random_rays(300, seed=4)
    want = oracle.OracleScene(sc, Config()).trace_closest(rays, use_bvh=False)
    T = sc.triangles.reshape(-1, 3, 3)
    for r, wk, wp, wt in zip(rays, *want[:3]):
        o, d = r[:3], r[3:]
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = np.float32(1) / d
        best_t, best_p = np.float32(np.inf), -1
        stack = [0]
        while stack:
            cur = stack.pop()
            if cur >= 0:
                q = nodes[cur]
                for (lo, hi), c in zip([((q[0], q[2], q[8]), (q[1], q[3], q[9])),
                                        ((q[4], q[6], q[10]), (q[5], q[7], q[11]))], kids[cur]):
                    with np.errstate(invalid="ignore"):
                        a = (np.array(lo, np.float32) - o) * inv
                        b = (np.array(hi, np.float32) - o) * inv
                    tn = max(np.nanmax(np.fmin(a, b)), 0)
                    tf = min(np.nanmin(np.fmax(a, b)), best_t)
                    if tn <= tf * np.float32(1.0000004):
                        stack.append(int(c))
            else:
                v = ~cur
                for slot in range(v >> 3, (v >> 3) + (v & 7) + 1):
                    p = int(order[slot])
                    t = np.float32(oracle.triangle_intersect(T[p, 0], T[p, 1], T[p, 2], o, d))
                    if 1e-3 < t < np.inf and (t < best_t or (t == best_t and p < best_p)):
                        best_t, best_p = t, p
        if wk == 0:
            assert best_p == -1
        else:
            assert best_p == wp and best_t == wt


def decode4(node):
    """(lo[4,3], hi[4,3], refs[4]) of a four-child quantised node (format: bvh_build.h)."""
    u = node.view(np.uint32)
    origin = node[0:3].astype(np.float32)
    ew = int(u[3])
    scale = np.array([np.array([((ew >> (8 * k)) & 0xFF) << 23], np.uint32).view(np.float32)[0] for k in range(3)],
                     np.float32)
    lo_w, hi_w = u[4:7], u[7:10]
    lo = np.zeros((4, 3), np.float32)
    hi = np.zeros((4, 3), np.float32)
    for c in range(4):
        for k in range(3):
            lo[c, k] = origin[k] + np.float32((int(lo_w[k]) >> (8 * c)) & 0xFF) * scale[k]
            hi[c, k] = origin[k] + np.float32((int(hi_w[k]) >> (8 * c)) & 0xFF) * scale[k]
    return lo, hi, node[12:16].view(np.int32)


@pytest.mark.parametrize("name,make", [("soup", lambda: scenes.soup(3000, 0).triangles),
                                       ("terrain", lambda: scenes.terrain(24).triangles),
                                       ("cornell", lambda: scenes.cornell().triangles),
                                       ("five", lambda: scenes.cornell().triangles[:5])])
def test_wide_bvh_structure(name, make, probe):
    """Four-child quantised tree: every triangle in exactly one leaf, decoded child boxes contain
    their triangles (and their own children's decoded boxes), absent children are inverted boxes,
    the stack bound is 3 entries per level."""
    tris = make()
    nodes, order, depth, stack = build(probe, tris, 4, 4, want_stack=True)
    assert sorted(order.tolist()) == list(range(len(tris)))
    assert stack == 3 * depth
    seen = np.zeros(len(order), np.int32)
    info = {"maxd": 0}

    def visit(i, d):
        """Returns the bounds of the triangles under node i; checks every decoded child box."""
        info["maxd"] = max(info["maxd"], d)
        lo, hi, refs = decode4(nodes[i])
        tlo, thi = np.full(3, np.inf), np.full(3, -np.inf)
        nkids = 0
        for c in range(4):
            r = int(refs[c])
            if r == -2**31:
                assert np.all(lo[c] > hi[c])          # absent child: lo byte 255, hi byte 0
                continue
            nkids += 1
            if r >= 0:
                clo, chi = visit(r, d + 1)
            else:
                v = ~r
                first, cnt = v >> 3, (v & 7) + 1
                assert cnt <= 4
                seen[first:first + cnt] += 1
                pts = tris[order[first:first + cnt]].reshape(-1, 3)
                clo, chi = pts.min(0), pts.max(0)
            # the decoded (quantised, outward-rounded) box contains everything beneath the child
            assert np.all(clo >= lo[c]) and np.all(chi <= hi[c])
            tlo, thi = np.minimum(tlo, clo), np.maximum(thi, chi)
        assert nkids >= 2
        return tlo, thi

    visit(0, 1)
    assert np.all(seen == 1) and info["maxd"] == depth
