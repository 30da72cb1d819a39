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
  roofline      for the dominant kernel (by HIP-event time on the render stream), against
                the roof that binds it.  Two roofs are priced from the committed counter
                passes and the launch duration measured live here: HBM (`traffic` = HBM bytes
                per launch; traffic / duration / 8 TB/s) and VALU issue (SQ_INSTS_VALU per
                launch x 2 cycles / (duration x 2.4 GHz x 1024 SIMDs)).  `bound` names the
                larger.  Where it is "valu", `achieved` is the lane-instructions per second
                the kernel's threads execute (VALU instructions x lanes active in each),
                `peak` what 1024 SIMDs x 32 lanes x 2.4 GHz can issue, and `frac` their
                ratio = issue fraction x lanes active / 64: the part of the chip's
                lane-issue capacity that does work.  Where it is "hbm", achieved = traffic /
                duration.  `design_bytes` is the byte budget of the pipeline as built (path
                state written and read back only by the paths that cross a launch, 12 B per
                finished sample, the ordered sum's 12 B read; hit records and queued light
                samples where the workload has them), from this run's own counts; the
                wavefront model of one kernel per stage (164 B per bounce, 88 B per shadow
                ray, 52 B per sample) is kept as a labelled side value, `stage_model`.
                `profiles_stale` is true when the kernel sources have changed since the
                counter files under profiles/ were collected;
  cpu_baseline  the CPU oracle (a port of the reference's OpenMP loop) timed on this
                host on a bounded sample of the same workload, at all of the host's
                cores and at the reference's own 8 threads (Raytracer.cpp:41);
  time_to_image_ms  the second half of the metric: the same frame rendered and waited for,
                one frame at a time (rtx_render), median of a few frames (what the
                reference prints per pass, Raytracer.cpp:281-285);
  legs          the other single-GPU configurations of BASELINE.json, a few steps each, with
                their own roofline and CPU baseline: the MIS Cornell scene (also under
                `mis_cornell`), the 1 M-triangle OBJ mesh, the diffuse sphere scene (whose
                CPU baseline is timed on the whole workload).
N > 1 adds `strong`: the same 256 spp split over the ranks, frame by frame, with its
time_to_image_ms.
"""
import argparse
import csv
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
SIMDS = 256 * 4        # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # peak engine clock used to turn a launch duration into issue slots
VALU_CYCLES = 2        # issue cycles of one wave64 VALU instruction at full rate
LANE_ISSUE_PEAK = SIMDS * 32 * CLOCK_HZ   # lane-instructions per second the chip can issue (32 lanes per SIMD and cycle)

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
def csrc_sha256():
    """Hash of the kernel sources and their build file: what a counter file under profiles/ was collected from."""
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def profiles_stale(workload):
    """True when profiles/MANIFEST.json has no entry for the workload or its sources' hash differs from the tree's."""
    m = _load_json(os.path.join(ROOT, "profiles", "MANIFEST.json")) or {}
    e = m.get(workload)
    return not e or e.get("csrc_sha256") != csrc_sha256()


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


def cpu_baseline(sc, cfg, spp_full, budget_s=9.0, whole=False):
    """Times the oracle (OpenMP rows, dynamic schedule: the shape of Raytracer.cpp:242) on a
    bounded sample of the workload: the same frame at as many samples per pixel as fit the
    budget, once with every core this process may use and once with the reference's THREADS = 8.
    whole: every sample of the workload, whatever it takes (the CPU-runnable configuration)."""
    from oracle import oracle_py
    osc = oracle_py.OracleScene(sc, cfg)
    rows = max(8, cfg.height // 16)

    def timed(threads):
        t0 = time.perf_counter()
        osc.render(1, nthreads=threads, row0=0, row1=rows)  # calibration: a band of rows
        dt = max(time.perf_counter() - t0, 1e-4)
        per_spp_s = dt * cfg.height / rows
        spp = spp_full if whole else int(max(1, min(spp_full, budget_s / per_spp_s)))
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


def state_bytes(carries_colour):
    """Bytes of one stored path: ray 16 + 8, attenuation and MIS pdf 16, sample ordinal and generator 8,
    and colour and light pdf 16 where anything in the scene emits."""
    return 48 + (16 if carries_colour else 0)


def design_bytes(m):
    """HBM byte budget of one frame of the pipeline as it is built, from this run's counts.
    Path state crosses memory only for the paths a launch stores for the next (written once, read once;
    fresh samples are recomputed from their number where the camera allows it, else 32 B written and read);
    a finished sample writes 12 B and the ordered sum reads them; where search and shading are separate
    kernels (the wide tree) a bounce re-reads its ray (24 B) and passes an 8 B hit record; a queued light
    sample is 48 B written, 32 B read by the any-hit kernel, and, unoccluded, 16 B of contribution plus the
    32 B read-modify-write of its path's colour (charged to all of them here: an upper bound)."""
    st = m["per_frame"]
    sb = state_bytes(m["carries_colour"])
    b = {"path_state": st["stored_paths"] * 2 * sb,
         "results": st["paths"] * 24,
         "fresh_samples": 0 if m["recompute_primaries"] else st["paths"] * 64,
         "hit_records": 0 if m["fused"] else st["bounces"] * (24 + 16),
         "light_samples": 0 if m["inline_light_samples"] else st["shadow_rays"] * (48 + 32 + 16 + 32)}
    b["total"] = sum(b.values())
    return b


def build_roofline(workload, m):
    """The roofline object for the stage with the largest HIP-event time."""
    stage = m["stage"]
    if not stage:
        return None
    fused = m["fused"]
    dom = max(stage, key=lambda k: stage[k][0])
    ms, launches, units = stage[dom]
    if dom == "shade" and fused:
        dom = "trace_shade"
    if not launches or ms <= 0:
        return None
    prefix = KERNEL_PREFIX[dom]
    avg_ms = ms / launches
    avg_s = avg_ms * 1e-3
    traffic, traffic_src = pmc_traffic(workload, prefix)
    sq, sq_src = sq_counters(workload, prefix)
    prof_ms, prof_src = rocprof_average_ms(workload, prefix)
    hbm_gbs = traffic / avg_s / 1e9 if traffic else None
    hbm_frac = hbm_gbs / HBM_PEAK_GBS if hbm_gbs else None
    valu_frac = lanes = lane_rate = None
    if sq and sq.get("SQ_INSTS_VALU"):
        # issue slots the launch's VALU instructions need / issue slots the chip has in its duration
        valu_frac = sq["SQ_INSTS_VALU"] * VALU_CYCLES / (avg_s * CLOCK_HZ * SIMDS)
        if sq.get("SQ_THREAD_CYCLES_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
            lanes = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"]
            lane_rate = sq["SQ_INSTS_VALU"] * lanes / avg_s
    if hbm_frac is None and valu_frac is None:
        bound = "unmeasured"   # no counter files for this workload under profiles/
    else:
        bound = "hbm" if (hbm_frac or 0) >= (valu_frac or 0) else "valu"
    roof = {"bound": bound, "kernel": prefix}
    if bound == "valu" and lane_rate is not None:
        roof.update({
            "achieved": round(lane_rate / 1e12, 3), "peak": round(LANE_ISSUE_PEAK / 1e12, 3), "unit": "Tlane-instr/s",
            "frac": round(lane_rate / LANE_ISSUE_PEAK, 4),
            "formula": "achieved = SQ_INSTS_VALU per launch x lanes_active / avg_launch_s; peak = 1024 SIMDs x 32 lanes "
                       "x 2.4 GHz; frac = valu_issue_frac x lanes_active / 64"})
    elif bound == "hbm":
        roof.update({
            "achieved": round(hbm_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4),
            "formula": "achieved = traffic / avg_launch_s; frac = achieved / 8000 GB/s"})
    else:
        roof.update({"achieved": None, "peak": None, "unit": None, "frac": None})
    roof.update({
        "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_ms": round(avg_ms, 5), "launches": int(launches),
        "units_per_launch": round(units / launches, 1),
        "rocprof_avg_launch_ms": round(prof_ms, 5) if prof_ms else None, "rocprof_source": prof_src,
        "hbm": {"achieved_gbs": round(hbm_gbs, 2) if hbm_gbs else None,
                "frac": round(hbm_frac, 4) if hbm_frac is not None else None,
                "formula": "traffic / avg_launch_s / 8000 GB/s"},
        "valu": {"issue_frac": round(valu_frac, 4) if valu_frac is not None else None,
                 "lanes_active": round(lanes, 2) if lanes is not None else None,
                 "insts_per_launch": round(sq["SQ_INSTS_VALU"]) if sq and sq.get("SQ_INSTS_VALU") else None,
                 "formula": "SQ_INSTS_VALU per launch x 2 cycles / (avg_launch_s x 2.4 GHz x 1024 SIMDs); "
                            "lanes_active = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU",
                 "source": sq_src},
        "profiles_stale": profiles_stale(workload),
    })
    # Side value: the minimum queue traffic of a wavefront design with one kernel per stage.  A kernel that does
    # search and shading in one launch, recomputes primary rays and makes two bounces per launch moves less.
    alg_launch = BYTES[dom] * units / launches
    roof["stage_model"] = {"bytes_per_unit": BYTES[dom], "bytes_per_launch": round(alg_launch),
                           "note": "byte model of a one-kernel-per-stage wavefront design, not of this kernel: "
                                   "what the kernel moves is `traffic`, what the pipeline must move is `design_bytes`"}
    return roof


def run_leg(args, workload, world, rank, local_rank, steps, warmup, profile, torch, dist, sync=None, strong=None,
            spp_slice=0):
    """Renders `workload` warmup + steps times; returns the measurements of this rank's view."""
    from raytracer_amd.capi import Config, Renderer
    sync = args.sync if sync is None else sync
    strong = args.strong if strong is None else strong
    if workload.startswith("soup") and workload not in workloads():  # soupN: N random triangles, any N
        from raytracer_amd import scenes
        n = int(workload[4:])
        wl = dict(scene=lambda: scenes.soup(n, 3, seed=2), sky=1, width=1024, height=1024, spp=64)
    else:
        wl = workloads()[workload]
    W = args.width or wl["width"]
    H = args.height or wl["height"]
    spp_full = spp_slice or args.spp or wl["spp"]   # spp_slice: a leg that renders part of a configuration's samples
    band, replica, replicas = rank % args.tiles, rank // args.tiles, world // args.tiles
    row0, row1 = band * H // args.tiles, (band + 1) * H // args.tiles
    if strong:
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
    # for it would start a third of a millisecond late.  sync: every frame waited for (rtx_render) and
    # exchanged before the next one starts: the time of one image.
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
        if sync:
            r.render(buf, spp, sample_base=base, seed=0x5EED, row0=row0, row1=row1, stream=stream)
            exchange(b)
            if world > 1:          # the image exists when the sums have arrived on rank 0
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
                torch.cuda.synchronize()
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
    # Host time at the end of every step.  With frames in flight a step returns when the frame two steps back has
    # finished (a set holds one frame), so the intervals are the period at which frames complete; with sync they are
    # the time of one image each.
    marks = [time.perf_counter()]
    t0 = marks[0]
    for _ in range(steps):
        step()
        marks.append(time.perf_counter())
    fence()
    elapsed = time.perf_counter() - t0
    st = r.stats()
    step_ms = [(b - a) * 1e3 for a, b in zip(marks, marks[1:])]

    tvals = torch.tensor([elapsed, float(st.bounces), float(st.paths), float(st.shadow_rays),
                          float(st.nee_samples), float(st.nan_paths), float(st.truncated_paths)],
                         dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = tvals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tvals, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    bounces, paths, shadows, nee, nans, truncated = (float(tvals[i]) for i in range(1, 7))
    per_frame = {k: getattr(st, k) / max(steps, 1) for k in ("paths", "bounces", "shadow_rays", "nee_samples", "stored_paths")}

    # Per-kernel pass: the same step once more with HIP events around every launch.
    stage, extra_ms = {}, {}
    fused = info.bvh_width == 2 and args.fused_trace != 1   # search and shading of a bounce in one kernel
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
    if world > 1:
        dist.barrier()
    r.close()
    del accums
    mats = list(sc.materials["emission"]) + [float(s["material"]["emission"]) for s in sc.spheres]
    m = dict(W=W, H=H, spp_full=spp_full, spp_rank=spp, cfg=cfg, scene=sc, info=info, elapsed=elapsed, steps=steps,
             step_ms=step_ms, sync=sync, bounces=bounces, paths=paths, shadows=shadows, nee=nee, nans=nans,
             truncated=truncated, stage=stage, extra_ms=extra_ms, fused=fused, per_frame=per_frame,
             carries_colour=any(e > 0 for e in mats),
             recompute_primaries=args.primary_rays != 1 and info.bvh_width != 4 and cfg.camera != 1,
             # (light samples traced inside the bounce kernel: seen from the launches of the per-kernel pass; a run
             # without that pass charges them as queued, an upper bound)
             inline_light_samples=bool(stage) and fused and stage["shadow"][1] == 0)
    m["roofline"] = build_roofline(workload, m) if stage else None
    return m


def spread(ms):
    """min / median / mean of per-step times, so that a reader can tell a hiccup from the level."""
    if not ms:
        return None
    return {"min": round(min(ms), 3), "median": round(statistics.median(ms), 3), "mean": round(sum(ms) / len(ms), 3),
            "max": round(max(ms), 3), "n": len(ms)}


def leg_report(workload, m, cpu_budget_s, no_cpu, whole_cpu=False):
    """One workload's measurements as a JSON object (the extra legs of the default run)."""
    v = m["bounces"] / m["elapsed"] / 1e6
    db = design_bytes(m)
    leg = {
        "workload": f"{workload} {m['W']}x{m['H']} {m['spp_full']} spp",
        "steps": m["steps"], "frames_in_flight": 1 if m["sync"] else 2,
        "ms_per_step": round(m["elapsed"] / m["steps"] * 1e3, 3), "step_ms": spread(m["step_ms"]),
        "value": round(v, 2), "unit": "Mpath-bounces/s",
        "mpaths_per_s": round(m["paths"] / m["elapsed"] / 1e6, 2),
        "mshadow_rays_per_s": round(m["shadows"] / m["elapsed"] / 1e6, 2),
        "bounces_per_path": round(m["bounces"] / max(m["paths"], 1), 3),
        "triangles": int(m["info"].num_triangles), "bvh_width": int(m["info"].bvh_width),
        "stage_ms": {**{k: round(x[0], 3) for k, x in m["stage"].items()},
                     **{k: round(x, 3) for k, x in (m["extra_ms"] if m["stage"] else {}).items()}},
        "design_bytes_per_frame": db,
        "design_frac_of_hbm_peak": round(db["total"] / (m["elapsed"] / m["steps"]) / 1e9 / HBM_PEAK_GBS, 4),
        "roofline": m["roofline"],
    }
    if not no_cpu:
        cpu = cpu_baseline(m["scene"], m["cfg"], m["spp_full"], budget_s=cpu_budget_s, whole=whole_cpu)
        leg["cpu_baseline"] = cpu
        leg["gpu_over_cpu"] = round(v / cpu["value"], 1) if cpu["value"] else None
        leg["gpu_over_cpu_8_threads"] = round(v / cpu["threads8"]["value"], 1) if cpu["threads8"]["value"] else None
    return leg


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
    ap.add_argument("--no-mis", "--no-legs", dest="no_legs", action="store_true",
                    help="skip the extra legs (frame-by-frame timing, MIS Cornell, 1 M-triangle mesh, diffuse sphere; "
                         "for N > 1 the strong-scaling leg)")
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

    default_run = args.workload == "sample2_glass" and not (args.spp or args.width or args.height) and not args.no_legs
    # The time of one image (the metric's second half): the same frame, rendered and waited for one at a time.
    # N > 1: the frame's 256 spp split over the ranks and summed onto rank 0 (strong scaling), which is what
    # "time-to-image at N GPUs" means; the main leg above keeps the per-GPU work fixed (weak scaling).
    tti = None
    if default_run and not args.sync:
        tti = run_leg(args, args.workload, world, rank, local_rank, steps=5, warmup=1, profile=False,
                      torch=torch, dist=dist, sync=True, strong=world > 1)
    # Further single-GPU legs: BASELINE.json's other configurations that fit one GPU.
    extra = {}
    if world == 1 and default_run:
        # (leg, workload, samples per pixel rendered (0: the workload's own), steps, warm-up steps)
        for leg, name, spp_slice, steps, warmup in (
                ("sample3_mis_cornell", "sample3_mis_cornell", 0, 2, 1),            # the scene of the >= 100x target
                ("mesh_1m", "mesh_1m", 0, 2, 1),                                    # configs[3]
                ("sample1_diffuse", "sample1_diffuse", 0, 20, 3),                   # configs[0]
                ("sample3_mis_cornell_1024spp", "sample3_mis_cornell", 1024, 1, 1),  # configs[2] on one GPU
                ("mis_4k_128_of_4096spp", "mis_4k", 128, 1, 1)):                    # configs[4]: one GPU's share of 32
            extra[leg] = run_leg(args, name, 1, 0, local_rank, steps=steps, warmup=warmup,
                                 profile=not args.no_profile, torch=torch, dist=dist, sync=False, strong=False,
                                 spp_slice=spp_slice)
            extra[leg]["workload_name"] = name

    if rank == 0:
        W, H, spp_full, info = m["W"], m["H"], m["spp_full"], m["info"]
        elapsed, bounces, paths, shadows = m["elapsed"], m["bounces"], m["paths"], m["shadows"]
        stage = m["stage"]
        db = design_bytes(m)
        stage_model_bytes = 164 * bounces + 88 * shadows + 52 * paths
        sync_ms = spread(m["step_ms"]) if m["sync"] else (spread(tti["step_ms"]) if tti else None)
        out = {
            "metric": "Mpath-bounces/sec & time-to-image (1024², 256 spp) at 1/2/4/8 GPUs",
            "value": round(bounces / elapsed / 1e6, 2),
            "unit": "Mpath-bounces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "ms_per_step_is": ("time of one image (every frame waited for)" if m["sync"] else
                               "throughput period: elapsed / steps with two frames in flight, not the latency of one image "
                               "(that is time_to_image_ms)"),
            "step_ms": spread(m["step_ms"]),
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
            # median of frames rendered and waited for one at a time (N > 1: 256 spp split over the ranks, reduce included)
            "time_to_image_ms": sync_ms["median"] if sync_ms else None,
            "time_to_image": ({"frames": sync_ms, "spp_per_gpu": (tti or m)["spp_rank"],
                               "scaling": "strong" if world > 1 else "one GPU",
                               "value_mbounces_per_s": round((tti or m)["bounces"] / (tti or m)["elapsed"] / 1e6, 2)}
                              if sync_ms else None),
            "mpaths_per_s": round(paths / elapsed / 1e6, 2),
            "mshadow_rays_per_s": round(shadows / elapsed / 1e6, 2),
            "bounces_per_path": round(bounces / max(paths, 1), 3),
            "shadow_rays_per_path": round(shadows / max(paths, 1), 3),
            "nan_paths": int(m["nans"]), "truncated_paths": int(m["truncated"]),
            # what this pipeline has to move per frame (design_bytes) over the frame's time, against the HBM peak
            "design_bytes_per_frame": db,
            "design_frac_of_hbm_peak": round(db["total"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else None,
            # side value: the one-kernel-per-stage byte model (164 B per bounce, 88 per shadow ray, 52 per sample);
            # above 1 means this pipeline does not move the bytes that model charges
            "stage_model_frac_of_hbm_peak": round(stage_model_bytes / elapsed / 1e9 / HBM_PEAK_GBS / world, 4),
            "stage_ms": {**{k: round(v[0], 3) for k, v in stage.items()},
                         **{k: round(v, 3) for k, v in (m["extra_ms"] if stage else {}).items()}},
            "roofline": m["roofline"],
        }
        if world > 1 and tti is not None:
            out["strong"] = {"workload": f"{args.workload} {W}x{H} {spp_full} spp split over {world} GPUs, one RCCL reduce per frame",
                             "spp_per_gpu": tti["spp_rank"], "time_to_image_ms": sync_ms["median"], "frames": sync_ms,
                             "value": round(tti["bounces"] / tti["elapsed"] / 1e6, 2), "unit": "Mpath-bounces/s"}
        if not args.no_cpu and world == 1:
            cpu = cpu_baseline(m["scene"], m["cfg"], spp_full)
            out["cpu_baseline"] = cpu
            out["gpu_over_cpu"] = round(out["value"] / cpu["value"], 1) if cpu["value"] else None
            out["gpu_over_cpu_8_threads"] = (round(out["value"] / cpu["threads8"]["value"], 1)
                                             if cpu["threads8"]["value"] else None)
        else:
            out["cpu_baseline"] = None
        if extra:
            legs = {}
            for name, mm in extra.items():
                legs[name] = leg_report(mm["workload_name"], mm, cpu_budget_s=3.0, no_cpu=args.no_cpu,
                                        whole_cpu=name == "sample1_diffuse")
            out["legs"] = legs
            out["mis_cornell"] = "legs.sample3_mis_cornell"
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
