#!/usr/bin/env python3
"""Benchmark of the wavefront path tracer: Mpath-bounces/s and time-to-image.

  python bench.py --gpus N --steps K --warmup W

A step is one full render of the workload through the C ABI (rtx_render): every
primary sample of the frame traced to termination, accumulated into a device
buffer that is already resident (no host transfers in the timed region).  The
default workload is BASELINE.json's configs[1], the one its metric is quoted on:
the glass-sphere scene ("sample2") at 1024x1024, 256 samples per pixel.

N > 1 (launched by torch.distributed.run, one process per GPU): every rank renders
the whole frame with its own, disjoint range of 256 sample indices (weak scaling:
the per-GPU work is the N=1 workload) and the per-GPU sum buffers are added onto
rank 0 with one RCCL reduce over xGMI, inside the timed region.

One JSON line is printed by rank 0.  Besides the contract's fields it carries
  roofline      the dominant kernel's algorithmic bytes per launch / its average
                launch duration (HIP events on the render stream), against HBM peak;
  cpu_baseline  the CPU oracle (a port of the reference's OpenMP loop) timed on
                this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# Algorithmic bytes per unit: the minimum queue traffic of a wavefront design.
# Path state = o,d (24) + attenuation (12) + pixel (4) + rng (4) + mis_brdf_pdf (4) + nee_pdf (4) = 52 B.
BYTES = {
    "intersect": 24 + 8,                  # ray read + hit record written
    "shade": 52 + 8 + 52 + 4 + 4 + 12,    # state read, hit read, state written, compaction index r/w, radiance RMW
    "trace_shade": 52 + 52 + 4 + 4 + 12,  # search and shading in one kernel: no ray or hit record in between
    "shadow": 44 + 44,                    # shadow record written by shade and read by shadow
    "generate": 52,                       # state written per primary sample
}


def workloads():
    from raytracer_amd import scenes
    return {
        "sample2_glass": dict(scene=lambda: scenes.sphere_sky(glass=True), sky=1, width=1024, height=1024, spp=256),
        "sample1_diffuse": dict(scene=lambda: scenes.sphere_sky(glass=False), sky=1, width=512, height=512, spp=64),
        "sample3_mis_cornell": dict(scene=lambda: scenes.cornell(), sky=0, width=1024, height=1024, spp=256),
        "sample3_mis_cornell_4lights": dict(scene=lambda: scenes.cornell(lights=4), sky=0, width=1024, height=1024, spp=256),
        "mesh_1m": dict(scene=lambda: scenes.terrain(707), sky=0, width=1920, height=1080, spp=256),
        "soup150": dict(scene=lambda: scenes.soup(150, 3, seed=2), sky=1, width=1024, height=1024, spp=64),
        "soup3k": dict(scene=lambda: scenes.soup(3000, 3, seed=2), sky=1, width=1024, height=1024, spp=64),
        "mis_4k": dict(scene=lambda: scenes.cornell(), sky=0, width=3840, height=2160, spp=4096),
    }


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE over this same command, reduced by tools/pmc_traffic.py).
    Counters cannot be read from inside this process, so the figure comes from the file."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    try:
        kernels = json.load(open(path))["kernels"]
        name = next(k for k in sorted(kernels) if k.startswith(kernel))  # k_shade -> k_shade_wave_append, ...
        return round(kernels[name]["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError, StopIteration):
        return None, None


def cpu_baseline(sc, cfg, spp_full, budget_s=12.0):
    """Times the oracle (OpenMP rows, dynamic schedule) on a bounded sample: the same
    frame at as many samples per pixel as fit the time budget."""
    from oracle import oracle_py
    threads = min(16, os.cpu_count() or 1)
    osc = oracle_py.OracleScene(sc, cfg)
    rows = max(8, cfg.height // 16)
    t0 = time.perf_counter()
    _, st = osc.render(1, nthreads=threads, row0=0, row1=rows)  # calibration: a band of rows
    dt = max(time.perf_counter() - t0, 1e-4)
    per_spp_s = dt * cfg.height / rows
    spp = int(max(1, min(spp_full, budget_s / per_spp_s)))
    t0 = time.perf_counter()
    _, st = osc.render(spp, nthreads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": round(st.bounces / dt / 1e6, 3), "unit": "Mpath-bounces/s", "cores": threads,
        "kind": "port",
        "sample": f"{cfg.width}x{cfg.height}, {spp} of {spp_full} spp, {st.paths} paths in {dt:.2f} s "
                  f"(oracle BVH in place of Embree; {st.paths / dt / 1e6:.3f} Mpaths/s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="sample2_glass")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--pool", type=int, default=0, help="in-flight path pool (0 = library default)")
    ap.add_argument("--finish-paths", type=int, default=0, help="drain-kernel threshold (0 = library default)")
    ap.add_argument("--fused-trace", type=int, default=0, choices=[0, 1, 2],
                    help="rtx_config.fused_trace: 0 library chooses, 1 search and shading in two kernels, 2 in one")
    ap.add_argument("--primary-rays", type=int, default=0, choices=[0, 1],
                    help="1 = primary rays always written by the generate kernel and read back (0 = library's choice)")
    ap.add_argument("--strong", action="store_true", help="split the workload's spp across ranks instead")
    ap.add_argument("--tiles", type=int, default=1,
                    help="row bands: ranks form TILES groups, group g renders band g of the frame and the "
                         "ranks of a group take disjoint sample ranges (for frames too large per GPU)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + RTX_BENCH_SAME_GPU=1 rehearses the N>1 logic on one GPU (buffers are summed "
                         "through host memory); the real run uses nccl = RCCL")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    if os.environ.get("RTX_BENCH_SAME_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from raytracer_amd.capi import Config, Renderer

    wl = workloads()[args.workload]
    W = args.width or wl["width"]
    H = args.height or wl["height"]
    spp_full = args.spp or wl["spp"]
    if world % args.tiles:
        raise SystemExit("--tiles must divide the number of ranks")
    band, replica, replicas = rank % args.tiles, rank // args.tiles, world // args.tiles
    row0, row1 = band * H // args.tiles, (band + 1) * H // args.tiles
    if args.strong:
        spp = spp_full // replicas + (1 if replica < spp_full % replicas else 0)
        base = replica * (spp_full // replicas) + min(replica, spp_full % replicas)
    else:
        spp, base = spp_full, replica * spp_full
    cfg = Config(width=W, height=H, sky=wl["sky"], device=local_rank, pool_paths=args.pool,
                 finish_paths=args.finish_paths, primary_rays=args.primary_rays, fused_trace=args.fused_trace)
    sc = wl["scene"]()

    r = Renderer(cfg)
    info = r.set_scene(sc)
    # Two sum buffers, used in turn: the RCCL reduce of one step runs on RCCL's stream while the
    # next step renders into the other buffer.
    accums = [torch.zeros(H * W * 3, dtype=torch.float32, device="cuda") for _ in range(2 if world > 1 else 1)]
    accum = accums[0]
    pending = [None] * len(accums)
    stream = torch.cuda.current_stream().cuda_stream
    step_no = [0]

    def step():
        b = step_no[0] % len(accums)
        step_no[0] += 1
        if pending[b] is not None:  # the reduce that last read this buffer (two steps ago)
            pending[b].wait()
            pending[b] = None
        buf = accums[b]
        buf.zero_()
        r.render(buf, spp, sample_base=base, seed=0x5EED, row0=row0, row1=row1, stream=stream)
        if world > 1:  # bands are disjoint and zero elsewhere: one sum serves tiles and sample ranges
            if args.backend == "nccl":
                pending[b] = dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, async_op=True)
            else:
                host = buf.cpu()
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    buf.copy_(host)

    def fence():
        for b, p in enumerate(pending):
            if p is not None:
                p.wait()
                pending[b] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:
        # Communicator set-up (RCCL builds its rings on the first collective) stays outside
        # the timed region whatever --warmup is.
        if args.backend == "nccl":
            dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        dist.barrier()
    for _ in range(args.warmup):
        step()
    fence()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st = r.stats()

    tvals = torch.tensor([elapsed, float(st.bounces), float(st.paths), float(st.shadow_rays),
                          float(st.nee_samples)], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = tvals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tvals, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    bounces, paths, shadows, nee = (float(tvals[i]) for i in range(1, 5))

    # Per-kernel pass: the same step once more with HIP events around every launch.
    roof = None
    stage = {}
    extra_ms = {}
    if rank == 0 and not args.no_profile:
        r.reset_stats()
        r.set_profiling(True)
        accum.zero_()
        r.render(accum, spp, sample_base=base, seed=0x5EED, row0=row0, row1=row1, stream=stream)
        torch.cuda.synchronize()
        r.set_profiling(False)
        p = r.stats()
        stage = {
            "intersect": (p.ms_intersect, p.timed_launches_intersect, p.bounces),
            "shade": (p.ms_shade, p.timed_launches_shade, p.bounces),
            "shadow": (p.ms_shadow, p.timed_launches_shadow, p.shadow_rays),
            "generate": (p.ms_generate, p.timed_launches_generate, p.paths),
        }
        extra_ms = {"finish": p.ms_finish, "reduce": p.ms_reduce}
        dom = max(stage, key=lambda k: stage[k][0])
        ms, launches, units = stage[dom]
        fused = p.timed_launches_intersect == 0      # k_trace_shade is timed as the shade stage
        if dom == "shade" and fused:
            dom = "trace_shade"
        if launches and ms > 0:
            bytes_per_launch = BYTES[dom] * units / launches
            avg_ms = ms / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(args.workload, "k_" + dom)
            roof = {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(bytes_per_launch),
                    "algorithmic_bytes_per_unit": BYTES[dom], "units_per_launch": round(units / launches, 1),
                    "avg_launch_ms": round(avg_ms, 5), "launches": int(launches)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        fused_run = bool(stage) and stage["intersect"][1] == 0
        pipeline_bytes = (124 if fused_run else 164) * bounces + 88 * shadows + 52 * paths
        out = {
            "metric": "Mpath-bounces/sec & time-to-image (1024², 256 spp) at 1/2/4/8 GPUs",
            "value": round(bounces / elapsed / 1e6, 2),
            "unit": "Mpath-bounces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} {W}x{H} {spp_full} spp" +
                                   (" per GPU (disjoint sample ranges, RCCL reduce)" if world > 1 and not args.strong else "") +
                                   (f", {args.tiles} row bands" if args.tiles > 1 else ""),
                       "triangles": int(info.num_triangles), "spheres": int(info.num_spheres),
                       "lights": int(info.num_lights), "bvh_nodes": int(info.num_nodes),
                       "pool_paths": int(args.pool) or (1 << 24)},
            "time_to_image_ms": round(elapsed / args.steps * 1e3, 3),
            "mpaths_per_s": round(paths / elapsed / 1e6, 2),
            "bounces_per_path": round(bounces / max(paths, 1), 3),
            "shadow_rays_per_path": round(shadows / max(paths, 1), 3),
            "pipeline_algorithmic_gbs": round(pipeline_bytes / elapsed / 1e9, 2),
            "pipeline_frac_of_hbm_peak": round(pipeline_bytes / elapsed / 1e9 / HBM_PEAK_GBS, 5),
            "stage_ms": {**{k: round(v[0], 3) for k, v in stage.items()},
                         **{k: round(v, 3) for k, v in (extra_ms if stage else {}).items()}},
            "roofline": roof,
        }
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(sc, cfg, spp_full)
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1) if cpu["value"] else None
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
