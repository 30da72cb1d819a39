#!/usr/bin/env python3
"""Benchmark of the wavefront path tracer: Mpath-bounces/s and time-to-image.

  python bench.py --gpus N --steps K --warmup W

A step is one full render of the workload through the C ABI (rtx_render): every
primary sample of the frame traced to termination, accumulated into a device
buffer that is already resident (no host transfers in the timed region).  The
default workload is BASELINE.json's configs[1], the one its metric is quoted on:
the glass-sphere scene ("sample2") at 1024x1024, 256 samples per pixel.

N > 1: one process per GPU.  Launched by torch.distributed.run the ranks are already
there; typed as a plain command, this process starts them itself (before it touches
the GPU) and passes rank 0's line through.  Every rank renders the whole frame with
its own, disjoint range of 256 sample indices (weak scaling: the per-GPU work is the
N=1 workload) and the per-GPU sum buffers are added onto rank 0 with one RCCL reduce
over xGMI, inside the timed region.

One JSON line is printed by rank 0.  Besides the contract's fields it carries
  roofline      for the dominant kernel (by HIP-event time on the render stream):
                `traffic` = HBM bytes per launch from the committed counter passes,
                `achieved` = traffic / the launch duration measured live here, `frac` =
                achieved / 8 TB/s; beside it the VALU issue fraction and the lanes
                active per VALU instruction from the committed SQ counter passes, and
                the algorithmic-bytes figure (minimum queue traffic of a wavefront
                design: 164 B per bounce, 88 B per shadow ray, 52 B per primary
                sample) with a note where that model exceeds what the kernel moves.
                `bound` is "hbm" or "valu", whichever fraction is larger, or
                "latency" when both are under one half;
  cpu_baseline  the CPU oracle (a port of the reference's OpenMP loop) timed on this
                host on a bounded sample of the same workload, at all of the host's
                cores and at the reference's own 8 threads (Raytracer.cpp:41);
  mis_cornell   a second, short leg on the MIS Cornell scene (1024x1024, 256 spp)
                with its own CPU baseline.
"""
import argparse
import csv
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
SIMDS = 256 * 4        # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # peak engine clock used to turn a launch duration into issue slots
VALU_CYCLES = 2        # issue cycles of one wave64 VALU instruction at full rate

# Algorithmic bytes per unit: the minimum queue traffic of a wavefront design with one kernel
# per stage.  Path state = o,d (24) + attenuation (12) + pixel (4) + rng (4) + mis_brdf_pdf (4)
# + nee_pdf (4) = 52 B.  Per bounce: 52 read + 52 written by shade, 24 ray re-read by intersect,
# 8 + 8 hit record, 4 + 4 compaction index, 12 radiance = 164 B.
BYTES = {
    "intersect": 24 + 8,
    "shade": 52 + 8 + 52 + 4 + 4 + 12,
    "trace_shade": 164,   # one kernel does the whole bounce: it is charged the whole bounce
    "shadow": 44 + 44,
    "generate": 52,
}
KERNEL_PREFIX = {"intersect": "k_intersect", "shade": "k_shade", "trace_shade": "k_trace_shade",
                 "shadow": "k_shadow", "generate": "k_generate"}


def workloads():
    from raytracer_amd import scenes
    return {
        "sample2_glass": dict(scene=lambda: scenes.sphere_sky(glass=True), sky=1, width=1024, height=1024, spp=256),
        "sample1_diffuse": dict(scene=lambda: scenes.sphere_sky(glass=False), sky=1, width=512, height=512, spp=64),
        "sample3_mis_cornell": dict(scene=lambda: scenes.cornell(), sky=0, width=1024, height=1024, spp=256),
        "sample3_mis_cornell_4lights": dict(scene=lambda: scenes.cornell(lights=4), sky=0, width=1024, height=1024, spp=256),
        # the mesh goes through the OBJ path: written as OBJ + MTL and parsed by the host layer's load_obj
        "mesh_1m": dict(scene=lambda: scenes.through_obj(scenes.terrain(707)), sky=0, width=1920, height=1080, spp=256),
        "soup150": dict(scene=lambda: scenes.soup(150, 3, seed=2), sky=1, width=1024, height=1024, spp=64),
        "soup3k": dict(scene=lambda: scenes.soup(3000, 3, seed=2), sky=1, width=1024, height=1024, spp=64),
        "terrain40": dict(scene=lambda: scenes.terrain(40), sky=0, width=1024, height=1024, spp=64),
        "mis_4k": dict(scene=lambda: scenes.cornell(), sky=0, width=3840, height=2160, spp=4096),
    }


# ---- committed profiler evidence (counters cannot be read from inside this process) ------------
def _load_json(path):
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


def _kernel_entry(table, prefix):
    """The entry of the first kernel (by name) that starts with `prefix` (k_shade -> k_shade_block_append)."""
    if not table:
        return None, None
    for name in sorted(table):
        if name.startswith(prefix):
            return name, table[name]
    return None, None


def pmc_traffic(workload, prefix):
    """HBM bytes per launch from the --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
    (tools/profile_round.sh, reduced by tools/pmc_traffic.py with the gfx950 corrections)."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
    d = _load_json(path)
    _, e = _kernel_entry(d and d.get("kernels"), prefix)
    if not e:
        return None, None
    return round(e["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)


def sq_counters(workload, prefix):
    """Per-launch means of the SQ counters of one kernel (tools/profile_round.sh, tools/pmc_sq.py)."""
    path = os.path.join(ROOT, "profiles", f"sq_counters_{workload}.json")
    d = _load_json(path)
    _, e = _kernel_entry(d and d.get("kernels"), prefix)
    return (e, os.path.relpath(path, ROOT)) if e else (None, None)


def rocprof_average_ms(workload, prefix):
    """Average launch duration of the kernel in the committed rocprofv3 --kernel-trace --stats summary."""
    path = os.path.join(ROOT, "profiles", f"kernel_stats_{workload}.csv")
    try:
        for row in csv.DictReader(open(path)):
            name = row["Name"].split("(")[0].split("::")[-1].split("<")[0]
            if name.startswith(prefix):
                return float(row["AverageNs"]) * 1e-6, os.path.relpath(path, ROOT)
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """The cores this process may use: the box's share where the launcher exported one
    (OMP_NUM_THREADS), else what the scheduler allows, else the machine's count."""
    env = os.environ.get("OMP_NUM_THREADS", "")
    if env.isdigit() and int(env) > 0:
        return int(env)
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def cpu_baseline(sc, cfg, spp_full, budget_s=9.0):
    """Times the oracle (OpenMP rows, dynamic schedule: the shape of Raytracer.cpp:242) on a
    bounded sample of the workload: the same frame at as many samples per pixel as fit the
    budget, once with every core this process may use and once with the reference's THREADS = 8."""
    from oracle import oracle_py
    osc = oracle_py.OracleScene(sc, cfg)
    rows = max(8, cfg.height // 16)

    def timed(threads):
        t0 = time.perf_counter()
        osc.render(1, nthreads=threads, row0=0, row1=rows)  # calibration: a band of rows
        dt = max(time.perf_counter() - t0, 1e-4)
        per_spp_s = dt * cfg.height / rows
        spp = int(max(1, min(spp_full, budget_s / per_spp_s)))
        t0 = time.perf_counter()
        _, st = osc.render(spp, nthreads=threads)
        dt = time.perf_counter() - t0
        return {"value": round(st.bounces / dt / 1e6, 3), "cores": threads,
                "sample": f"{cfg.width}x{cfg.height}, {spp} of {spp_full} spp, {st.paths} paths in {dt:.2f} s "
                          f"({st.paths / dt / 1e6:.3f} Mpaths/s)"}

    allc = timed(host_threads())
    eight = timed(min(8, host_threads()))
    return {
        "value": allc["value"], "unit": "Mpath-bounces/s", "cores": allc["cores"], "kind": "port",
        "sample": allc["sample"] + "; reference algorithm on the oracle's own BVH, not Embree",
        "cpu_model": cpu_model(),
        "threads8": {"value": eight["value"], "cores": eight["cores"], "sample": eight["sample"],
                     "note": "THREADS = 8 as in Raytracer.cpp:41"},
    }


def build_roofline(workload, stage, fused):
    """The roofline object for the stage with the largest HIP-event time."""
    dom = max(stage, key=lambda k: stage[k][0])
    ms, launches, units = stage[dom]
    if dom == "shade" and fused:
        dom = "trace_shade"
    if not launches or ms <= 0:
        return None
    prefix = KERNEL_PREFIX[dom]
    avg_ms = ms / launches
    alg_launch = BYTES[dom] * units / launches
    alg_gbs = alg_launch / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(workload, prefix)
    sq, sq_src = sq_counters(workload, prefix)
    prof_ms, prof_src = rocprof_average_ms(workload, prefix)
    hbm_gbs = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
    hbm_frac = hbm_gbs / HBM_PEAK_GBS if hbm_gbs else None
    valu_frac = lanes = None
    if sq and sq.get("SQ_INSTS_VALU"):
        # issue slots the launch's VALU instructions need / issue slots the chip has in its duration
        valu_frac = sq["SQ_INSTS_VALU"] * VALU_CYCLES / (avg_ms * 1e-3 * CLOCK_HZ * SIMDS)
        if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
            lanes = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"]
    fr = [f for f in (hbm_frac, valu_frac) if f is not None]
    if not fr:
        bound = "unmeasured"   # no counter files for this workload under profiles/: the byte model alone bounds nothing
    elif max(fr) < 0.5:
        bound = "latency"
    else:
        bound = "hbm" if (hbm_frac or 0) >= (valu_frac or 0) else "valu"
    roof = {
        "bound": bound, "kernel": prefix,
        "achieved": round(hbm_gbs, 2) if hbm_gbs else None,
        "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(hbm_gbs / HBM_PEAK_GBS, 5) if hbm_gbs else None,
        "achieved_is": "counter HBM bytes per launch / live launch duration" if hbm_gbs else
                       "not measured: no counter file for this workload (see `algorithmic` for the byte model)",
        "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_ms": round(avg_ms, 5), "launches": int(launches),
        "units_per_launch": round(units / launches, 1),
        "rocprof_avg_launch_ms": round(prof_ms, 5) if prof_ms else None, "rocprof_source": prof_src,
        "valu_issue_frac": round(valu_frac, 4) if valu_frac is not None else None,
        "valu_issue_formula": "SQ_INSTS_VALU per launch x 2 cycles / (avg_launch_s x 2.4 GHz x 1024 SIMDs)",
        "lanes_active": round(lanes, 2) if lanes is not None else None,
        "sq_source": sq_src,
        "algorithmic": {
            "bytes_per_unit": BYTES[dom], "bytes_per_launch": round(alg_launch),
            "gbs": round(alg_gbs, 2), "frac_of_peak": round(alg_gbs / HBM_PEAK_GBS, 5),
        },
    }
    if traffic and alg_launch > traffic:
        roof["algorithmic"]["note"] = (
            "the byte model counts queue traffic this kernel does not move (search and shading in one "
            "kernel: no ray re-read or hit record; primary rays recomputed; fresh and finished paths "
            "skip half their state)" + ("; it exceeds the HBM peak" if alg_gbs > HBM_PEAK_GBS else ""))
    return roof


def run_leg(args, workload, world, rank, local_rank, steps, warmup, profile, torch, dist):
    """Renders `workload` warmup + steps times; returns the measurements of this rank's view."""
    from raytracer_amd.capi import Config, Renderer
    if workload.startswith("soup") and workload not in workloads():  # soupN: N random triangles, any N
        from raytracer_amd import scenes
        n = int(workload[4:])
        wl = dict(scene=lambda: scenes.soup(n, 3, seed=2), sky=1, width=1024, height=1024, spp=64)
    else:
        wl = workloads()[workload]
    W = args.width or wl["width"]
    H = args.height or wl["height"]
    spp_full = args.spp or wl["spp"]
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
    native = world > 1 and args.reduce == "native"
    if native:  # the library's own communicator: rank 0 makes the id, torch's group carries it
        box = [Renderer.comm_create_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        r.comm_init(box[0], rank, world)
    # A ring of four sum buffers.  A step leaves its frame in flight (rtx_render_async): the next step's
    # rounds fill the GPU while the last long paths of this one run out, and the exchange step of a frame
    # (RCCL reduce onto rank 0) is issued once the next frame is under way.  A buffer is cleared two steps
    # before its turn, on a stream of its own: a clear issued between two frames runs beside the ordered sum
    # of the frame that has just ended, which has the memory system to itself, and a frame that had to wait
    # for it would start a third of a millisecond late.  --sync: every frame waited for (rtx_render), its
    # reduce issued at once and overlapping the next step's render.
    NB = 4
    accums = [torch.zeros(H * W * 3, dtype=torch.float32, device="cuda") for _ in range(NB)]
    pending = [None] * NB
    owed = [False] * NB            # frames rendered and not yet exchanged
    cleared = [None] * NB          # event behind the clear of a buffer that is ready for its next frame
    main = torch.cuda.current_stream()
    clear_stream = torch.cuda.Stream()
    stream = main.cuda_stream
    step_no = [0]

    def exchange(b):
        buf = accums[b]
        owed[b] = False
        if world == 1:
            return
        r.wait(buf)
        if native:
            r.comm_reduce(buf, root=0, stream=stream)  # ordered on the render stream
        elif args.backend == "nccl":  # bands are disjoint and zero elsewhere: one sum serves tiles and sample ranges
            pending[b] = dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, async_op=True)
        else:
            host = buf.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                buf.copy_(host)

    def clear(b):
        """Clears buffer b once its last frame and that frame's exchange are done."""
        if owed[b]:
            exchange(b)
        r.wait(accums[b])
        clear_stream.wait_stream(main)          # the native reduce and the gloo copy run on the main stream
        with torch.cuda.stream(clear_stream):
            if pending[b] is not None:          # (a stream-level wait for RCCL's stream)
                pending[b].wait()
                pending[b] = None
            accums[b].zero_()
            cleared[b] = clear_stream.record_event()

    def step():
        k = step_no[0]
        b = k % NB
        step_no[0] += 1
        if cleared[b] is None:
            clear(b)
        cleared[b].synchronize()   # (long done; a wait enqueued on the stream would be one more thing in its queue)
        cleared[b] = None
        buf = accums[b]
        if args.sync:
            r.render(buf, spp, sample_base=base, seed=0x5EED, row0=row0, row1=row1, stream=stream)
            exchange(b)
        else:
            r.render_async(buf, spp, sample_base=base, seed=0x5EED, row0=row0, row1=row1, stream=stream)
            owed[b] = True
            if owed[(b - 1) % NB]:
                exchange((b - 1) % NB)
        clear((b + 2) % NB)

    def fence():
        for b in range(NB):
            if owed[b]:
                exchange(b)
        r.wait()
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
        if native:
            r.comm_reduce(accums[0], root=0, stream=stream)
            torch.cuda.synchronize()
        elif args.backend == "nccl":
            dist.reduce(accums[0], dst=0, op=dist.ReduceOp.SUM)
        dist.barrier()
    for _ in range(warmup):
        step()
    fence()
    r.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st = r.stats()

    tvals = torch.tensor([elapsed, float(st.bounces), float(st.paths), float(st.shadow_rays),
                          float(st.nee_samples), float(st.nan_paths), float(st.truncated_paths)],
                         dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = tvals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tvals, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    bounces, paths, shadows, nee, nans, truncated = (float(tvals[i]) for i in range(1, 7))

    # Per-kernel pass: the same step once more with HIP events around every launch.
    stage, extra_ms, roof = {}, {}, None
    if rank == 0 and profile:
        r.reset_stats()
        r.set_profiling(True)
        accums[0].zero_()
        r.render(accums[0], spp, sample_base=base, seed=0x5EED, row0=row0, row1=row1, stream=stream)
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
        roof = build_roofline(workload, stage, fused=p.timed_launches_intersect == 0)
    if world > 1:
        dist.barrier()
    r.close()
    del accums
    return dict(W=W, H=H, spp_full=spp_full, cfg=cfg, scene=sc, info=info, elapsed=elapsed, steps=steps,
                bounces=bounces, paths=paths, shadows=shadows, nee=nee, nans=nans, truncated=truncated,
                stage=stage, extra_ms=extra_ms, roofline=roof)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
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
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event pass")
    ap.add_argument("--no-mis", action="store_true", help="skip the second leg on the MIS Cornell scene")
    ap.add_argument("--sync", action="store_true",
                    help="wait for every frame (rtx_render) instead of leaving it in flight while the next one starts")
    ap.add_argument("--reduce", default="torch", choices=["torch", "native"],
                    help="who issues the sum of the per-GPU buffers: torch.distributed (RCCL through torch), or the "
                         "library's own exchange step (rtx_comm_reduce: RCCL loaded by librtx_hip.so, enqueued on "
                         "the render stream)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + RTX_BENCH_SAME_GPU=1 rehearses the N>1 logic on one GPU (buffers are summed "
                         "through host memory); the real run uses nccl = RCCL")
    args = ap.parse_args()

    # Typed as a plain command with N > 1: start one child process per GPU now, before this
    # process has made any GPU call, and hand rank 0's line through.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        from raytracer_amd.dist import spawn_local_ranks
        sys.exit(spawn_local_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the renderer has no CPU path")
    if os.environ.get("RTX_BENCH_SAME_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world % args.tiles:
        raise SystemExit("--tiles must divide the number of ranks")
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    m = run_leg(args, args.workload, world, rank, local_rank, args.steps, args.warmup,
                profile=not args.no_profile, torch=torch, dist=dist)

    # Second leg (one GPU, default workload only): the closed Cornell box with next-event
    # estimation and MIS, the scene the >= 100x-over-CPU target is stated on.
    mis = None
    if world == 1 and not args.no_mis and args.workload == "sample2_glass" and not (args.spp or args.width or args.height):
        mis = run_leg(args, "sample3_mis_cornell", 1, 0, local_rank, steps=2, warmup=1,
                      profile=not args.no_profile, torch=torch, dist=dist)

    if rank == 0:
        W, H, spp_full, info = m["W"], m["H"], m["spp_full"], m["info"]
        elapsed, bounces, paths, shadows = m["elapsed"], m["bounces"], m["paths"], m["shadows"]
        stage = m["stage"]
        pipeline_bytes = 164 * bounces + 88 * shadows + 52 * paths
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
                                   (f" per GPU (disjoint sample ranges, {'RCCL' if args.backend == 'nccl' else 'gloo'} reduce)" if world > 1 and not args.strong else "") +
                                   (f", {args.tiles} row bands" if args.tiles > 1 else ""),
                       "triangles": int(info.num_triangles), "spheres": int(info.num_spheres),
                       "lights": int(info.num_lights), "bvh_nodes": int(info.num_nodes),
                       "bvh_width": int(info.bvh_width), "pool_paths": int(args.pool) or (1 << 24),
                       # 2: a step leaves its frame in flight and the next step starts beside its last long paths
                       # (rtx_render_async); all frames are complete when the timed region ends
                       "frames_in_flight": 1 if args.sync else 2},
            "time_to_image_ms": round(elapsed / args.steps * 1e3, 3) if args.sync else None,
            "mpaths_per_s": round(paths / elapsed / 1e6, 2),
            "mshadow_rays_per_s": round(shadows / elapsed / 1e6, 2),
            "bounces_per_path": round(bounces / max(paths, 1), 3),
            "shadow_rays_per_path": round(shadows / max(paths, 1), 3),
            "nan_paths": int(m["nans"]), "truncated_paths": int(m["truncated"]),
            "pipeline_algorithmic_gbs": round(pipeline_bytes / elapsed / 1e9, 2),
            "pipeline_algorithmic_frac_of_hbm_peak": round(pipeline_bytes / elapsed / 1e9 / HBM_PEAK_GBS, 5),
            "stage_ms": {**{k: round(v[0], 3) for k, v in stage.items()},
                         **{k: round(v, 3) for k, v in (m["extra_ms"] if stage else {}).items()}},
            "roofline": m["roofline"],
        }
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(m["scene"], m["cfg"], spp_full)
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1) if cpu["value"] else None
            out["gpu_over_cpu_8_threads"] = (round(out["value"] / cpu["threads8"]["value"], 1)
                                             if cpu["threads8"]["value"] else None)
        else:
            out["cpu_baseline"] = None
        if mis is not None:
            mv = mis["bounces"] / mis["elapsed"] / 1e6
            leg = {
                "workload": f"sample3_mis_cornell {mis['W']}x{mis['H']} {mis['spp_full']} spp",
                "steps": mis["steps"], "ms_per_step": round(mis["elapsed"] / mis["steps"] * 1e3, 3),
                "value": round(mv, 2), "unit": "Mpath-bounces/s",
                "mshadow_rays_per_s": round(mis["shadows"] / mis["elapsed"] / 1e6, 2),
                "bounces_per_path": round(mis["bounces"] / max(mis["paths"], 1), 3),
                "stage_ms": {**{k: round(v[0], 3) for k, v in mis["stage"].items()},
                             **{k: round(v, 3) for k, v in (mis["extra_ms"] if mis["stage"] else {}).items()}},
                "roofline": mis["roofline"],
            }
            if not args.no_cpu:
                cpu = cpu_baseline(mis["scene"], mis["cfg"], mis["spp_full"], budget_s=6.0)
                leg["cpu_baseline"] = cpu
                leg["gpu_over_cpu"] = round(mv / cpu["value"], 1) if cpu["value"] else None
                leg["gpu_over_cpu_8_threads"] = (round(mv / cpu["threads8"]["value"], 1)
                                                 if cpu["threads8"]["value"] else None)
            out["mis_cornell"] = leg
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
