"""Generates the committed golden vectors.  Run from the repo root, in the container that
has /root/reference (the Pcg vectors come from the reference's own Sampler.h, compiled
unchanged into oracle/_ref/libref_pcg.so by oracle/Makefile):

    python tests/golden/make_golden.py

pcg_kat.json        rand_pcg / Uniform / pcg_hash sequences from the REFERENCE build
                    (this pins the oracle's generator to the reference).
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


if __name__ == "__main__":
    pcg()
    closest()
    print("golden vectors written to", HERE)
