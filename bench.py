#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: Mpixels/s (and frames/s) at 4K on the
1M-triangle synthetic soup (config 4: 3840x2160, z-test, depth-only).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame of the hot path (clear + vertex transform + setup + binning + tile raster
+ framebuffer write) over the resident scene; inputs and the framebuffer stay in HBM.

N GPUs: the framebuffer is split into N tile-row bands; every GPU holds the whole scene and
renders its band; no collective on the data path (bands are disjoint).  Two ways to get there:
  * under a launcher (WORLD_SIZE = N): one process per GPU, torch.distributed only for the timing barrier and the
    MAX over ranks (nccl = RCCL);
  * WITHOUT a launcher: ONE process, ONE multi-device context (swr_config.device_count = N): the library fans every
    call out to N per-device sub-contexts on their own host threads.  Needs N visible GPUs, otherwise the run
    exits non-zero (SWR_BENCH_ALLOW_SHARED=1 lets bands share GPUs as a labelled rehearsal — not a result).
The total work is fixed as N grows => "scaling": "strong".

Prints ONE JSON line on rank 0.
  value / ms_per_step   W untimed + exactly K timed frames, no HIP event on any stream, frames pipelined — the FIRST pass, made
                        with the arguments as given, straight after the scene upload (the GPU's clock ramp after idle included)
  value_steady / ms_per_step_steady  the same W + K frames again after the latency and sampled-roofline passes (what a running
                        frame loop sees; `frames_before_steady_pass` frames were drawn before it)
  latency_ms            one frame alone, swr_draw -> swr_sync
  roofline              dominant kernel k_raster from a SEPARATE sampled pass (two HIP events around every n-th
                        launch, on the raster stream): algorithmic framebuffer bytes of the band / mean duration
                        vs the 8 TB/s HBM peak; .valu = VALU wave-instructions per launch (rocprofv3 SQ counters,
                        profiles/) against the chip's issue peak — the resource the kernel is actually bound by
  extra.host_visible    the same frames with every band copied to ONE page-locked host image (swr_present:
                        hipMemcpyAsync per band, colour/depth on two copy streams, double-buffered framebuffer)
  extra.swr_render_ms   the full drop-in call (upload 120 MB + draw + gather, host pointers in, pixels out)
  cpu_baseline          (rank 0, N = 1 only) the CPU oracle — the C restatement of Renderer.swift's loop
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # measured float4-copy ceiling (same guide)
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD, 2.4 GHz (same guide)
VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2


def shard_rows(swr, height: int, world: int, rank: int):
    """Tile-row band of `rank` (the C-ABI's swr_band_rows)."""
    return swr.band_rows(height, world, rank)


def load_profile_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f)
        except Exception:
            return None
    return None


def csrc_digest():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: profiles/traffic.json and valu.json carry
    the digest of the sources they were measured on (tools/profiles_from_refresh.py), so the bench line can say whether the static
    counters it quotes belong to the code that is running (there is no .git on the GPU box)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "software-renderer_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(scene, budget_s: float = 15.0):
    """The oracle (kind 'port': C restatement of Renderer.swift, per-pixel 2x2 inverse kept,
    single thread like the reference) on whole frames of the same workload."""
    from oracle import oracle
    oracle.build()
    times = []
    t_all = time.perf_counter()
    frames = 0
    while frames < 2 or (time.perf_counter() - t_all < budget_s and frames < 40):   # ~15 s of CPU work
        t0 = time.perf_counter()
        _, _, st, rc = oracle.render_scene(scene)
        assert rc == 0
        times.append(time.perf_counter() - t0)
        frames += 1
    t = float(np.median(times[1:])) if len(times) > 1 else times[0]
    mpix = scene.width * scene.height / t / 1e6
    out = {"value": round(mpix, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
           "sample": f"{frames} whole frames of the same workload ({scene.triangles} triangles, "
                     f"{scene.width}x{scene.height}, flags={scene.flags}); median of frames 2..{frames}; "
                     f"{st.fragments} fragments/frame",
           "ms_per_frame": round(t * 1e3, 2), "host_cpus": os.cpu_count()}
    # SURVEY 8(d) (ii), optional: the same oracle on row bands, one thread per band (the reference itself is single-threaded: `value`
    # above stays the faithful figure).  16 threads: a one-GPU box's share of its host.
    try:
        nthr = max(2, min(16, os.cpu_count() or 2))
        tt = []
        for _ in range(4):
            t0 = time.perf_counter()
            _, _, rcs = oracle.render_threads(scene, nthr)
            assert not any(rcs)
            tt.append(time.perf_counter() - t0)
        tb = float(np.median(tt[1:]))
        out["row_bands"] = {"value": round(scene.width * scene.height / tb / 1e6, 2), "unit": "Mpixels/s", "cores": nthr,
                            "ms_per_frame": round(tb * 1e3, 2),
                            "sample": "4 whole frames, one oracle thread per row band (every thread sets up every triangle); median of frames 2..4"}
    except Exception as e:                                     # optional leg: never fails the bench line
        out["row_bands"] = {"error": str(e)[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--triangles", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--color", action="store_true", help="headline = colour+depth instead of depth-only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)

    import swr_amd
    if local_rank == 0:
        swr_amd.build()          # no-op when the in-tree .so is current
    ndev = swr_amd.device_count()
    if ndev <= 0:
        print("bench.py: no HIP device (the library has no CPU fallback)", file=sys.stderr)
        sys.exit(2)

    dist = None
    torch = None
    device = local_rank
    rehearsal = False
    in_process = 1          # bands driven by this process's context
    if world > 1:
        import torch
        import torch.distributed as dist
        if ndev >= world or os.environ.get("SWR_BENCH_BACKEND", "") == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # rehearsal on a box with fewer GPUs than ranks (e.g. 2 ranks on the 1-GPU dev box):
            # same code path, ranks share devices, gloo carries the barrier / MAX.  Not a result.
            rehearsal = True
            device = local_rank % max(ndev, 1)
            dist.init_process_group("gloo")
        dist.barrier()
    elif args.gpus > 1:
        # no launcher: one process, one multi-device context
        in_process = args.gpus
        device = 0
        if ndev < args.gpus:
            if os.environ.get("SWR_BENCH_ALLOW_SHARED", "") != "1":
                print(f"bench.py: --gpus {args.gpus} but only {ndev} HIP device(s) visible; refusing to report a "
                      f"{args.gpus}-GPU number (SWR_BENCH_ALLOW_SHARED=1 rehearses with bands sharing GPUs)", file=sys.stderr)
                sys.exit(3)
            rehearsal = True
    n_gpus = world if world > 1 else in_process
    S = swr_amd.scenes

    scene = S.cfg4_soup(ntri=args.triangles, width=args.width, height=args.height, depth_only=not args.color)
    W, H = scene.width, scene.height
    r0, r1 = shard_rows(swr_amd, H, world, rank) if world > 1 else (0, H)

    ctx = swr_amd.Context(device if (world > 1 or in_process > 1) else -1, device_count=in_process if in_process > 1 else 0)
    ctx.scene_upload(scene.vertices, scene.indices)
    ctx.target_set(W, H, r0, r1)
    bands = ctx.bands()
    largest_band_px = W * max(b - a for _, a, b in bands)

    def barrier():
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def run(flags, steps, warmup, level=0, sample=1, present=None):
        """W untimed frames, then exactly `steps` timed frames between barrier + sync on both sides.  level 0: no HIP
        event anywhere.  present = (color, depth) host images: every frame is also copied to the host (async)."""
        for _ in range(warmup):
            ctx.draw(scene.transform, flags)
            if present:
                ctx.present(*present)
        ctx.present_wait() if present else ctx.sync()      # also grows the bins if the first frame overflowed them
        if level:
            ctx.timing_sample(sample)
        ctx.timing_enable(level)
        ctx.timing_reset()
        barrier()
        ctx.sync()
        if torch is not None and not rehearsal:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.draw(scene.transform, flags)
            if present:
                ctx.present(*present)
        # a frame whose (triangle,tile) pair list overflowed the bins would raster nothing and time as "fast":
        # sync / present_wait check every frame's pair total and raise (SWR_ERR_FRAME_DROPPED) or redraw
        ctx.present_wait() if present else ctx.sync()
        if torch is not None and not rehearsal:
            torch.cuda.synchronize()
        t1 = time.perf_counter()       # this rank's K steps are complete; the job's time is the MAX over ranks (below) ...
        barrier()                      # ... and the closing barrier brackets the region without its own latency (tens of
        #                                microseconds of collective against 20 x 13 us of work at 8 GPUs) being counted as rendering
        sums, frames = ctx.timing_totals() if level else ({}, 0)
        ctx.timing_enable(0)
        return max_over_ranks(t1 - t0), sums, frames

    flags = scene.flags
    bytes_per_px = 4 if (flags & S.FLAG_NO_COLOR) else 8

    # ---- pass 0: THE HEADLINE — the W + K frames the arguments ask for, as the process's first GPU work ----------------
    # An MI355X that has been idle for ~50 ms runs its first ~150 frames 10-15 % slower (clock / power-state ramp:
    # profiles/r03/gpu_warmup_after_idle.txt — a hot context that sleeps 50 ms shows the same ramp as a fresh one), and
    # the process has just spent seconds on the host building the scene.  `value` / `ms_per_step` are measured in that
    # state, with exactly the requested warm-up (round 3 reported the later, steady pass as `value`: ADVICE r03);
    # the steady state is reported beside it (value_steady / ms_per_step_steady).
    dt, _, _ = run(flags, args.steps, args.warmup, level=0)
    ms_per_step = dt / args.steps * 1e3
    mpix = W * H * args.steps / dt / 1e6
    frames_before_headline = args.steps + args.warmup

    # one frame alone: enqueue -> host sees it finished (no other frame in flight)
    lat = []
    for _ in range(12):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.draw(scene.transform, flags)
        ctx.sync()
        lat.append(time.perf_counter() - t0)
    latency_ms = max_over_ranks(float(np.median(lat[2:]))) * 1e3
    frames_before_headline += 12

    # ---- pass 1: k_raster's launch duration, sampled (an event pair on the raster stream costs a pipelined frame
    # ~15 us, so only every n-th launch is bracketed and this pass never feeds `value`) ---------------------------------
    SAMPLE = max(1, int(os.environ.get("SWR_BENCH_TIMING_SAMPLE", "8")))
    roof_steps = max(args.steps, 128)
    SAMPLE = max(1, min(SAMPLE, roof_steps // 8))
    dt_s, sums, frames = run(flags, roof_steps, 4, level=1, sample=SAMPLE)
    frames_before_headline += roof_steps + 4

    # ---- pass 2: steady state — the same W + K frames again, no HIP event on any stream, GPU clocks up ----------------
    dt_steady, _, _ = run(flags, args.steps, args.warmup, level=0)
    ms_per_step_steady = dt_steady / args.steps * 1e3

    raster_ms = sums["raster_ms"] / max(frames, 1)
    achieved = largest_band_px * bytes_per_px / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else 0.0
    traffic = load_profile_json("traffic.json") or {}
    valu = load_profile_json("valu.json") or {}
    roofline = {
        "bound": "hbm", "kernel": "k_raster_depth (32-bit depth keys)" if (flags & S.FLAG_NO_COLOR) else "k_raster<ztest, colour>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        # frac_upper = bytes of one launch / the steady frame period / peak: one launch per frame, so this is the kernel's
        # aggregate rate in the steady pass (and the upper bound of what a per-launch figure could show without overlap)
        "frac_upper": round(largest_band_px * bytes_per_px / (ms_per_step_steady * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
        "traffic": traffic.get("k_raster_bytes_per_launch") if n_gpus == 1 else None,
        "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of tools/profile.sh on this workload; static, not re-measured in this run)",
        "traffic_measured_on": traffic.get("commit"),
        # do the static counters (traffic, valu.insts) belong to the kernel sources this library was built from?
        "static_counters_match_sources": (traffic.get("csrc_digest") == csrc_digest()) if traffic.get("csrc_digest") else None,
        "algorithmic_bytes_per_launch": largest_band_px * bytes_per_px,
        "avg_launch_ms": round(raster_ms, 5), "launches_timed": frames, "timed_every_nth_launch": SAMPLE,
        "ms_per_step_while_sampling": round(dt_s / roof_steps * 1e3, 4),
        # Frame lanes: up to four frames are in flight, each on its own stream, so the workgroups of several
        # k_raster launches share the chip and ONE launch lasts about launches_in_flight frame periods.  `achieved` / `frac` are
        # the contract's per-launch figures (bytes of one launch / its own duration); the kernel's aggregate rate while the
        # pass runs is frac_aggregate = frac x launches_in_flight = bytes per frame period / peak
        "launches_in_flight": round(raster_ms / (dt_s / roof_steps * 1e3), 2) if dt_s > 0 else None,
        "frac_aggregate": round(largest_band_px * bytes_per_px / (dt_s / roof_steps) / 1e9 / HBM_PEAK_GBS, 5) if dt_s > 0 else None,
        "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 5),
        # the contract's `bound` names the roofline the fraction is quoted against (the HBM-write roofline of north_star);
        # what the kernel is actually limited by is VALU issue: see .valu
        "binding_resource": "valu_issue",
    }
    # the whole frame against its compulsory HBM bytes (SURVEY 8(d) secondary: framebuffer written once + the scene read once)
    compulsory = W * (r1 - r0) * bytes_per_px + 32 * scene.vertices.shape[0] + 8 * scene.indices.size
    roofline["frame"] = {"bytes": compulsory, "achieved": round(compulsory / (ms_per_step_steady * 1e-3) / 1e9, 2), "unit": "GB/s",
                         "frac": round(compulsory / (ms_per_step_steady * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "note": "compulsory bytes of one frame / ms_per_step_steady"}
    if n_gpus == 1 and valu.get("k_raster_valu_wave_insts_per_launch") and raster_ms > 0:
        insts = float(valu["k_raster_valu_wave_insts_per_launch"])
        roofline["valu"] = {
            "insts": insts, "source": valu.get("source", "profiles/valu.json"),
            "peak_wave_insts_per_s": VALU_PEAK_WAVE_INSTS_PER_S,
            "frac": round(insts / (raster_ms * 1e-3) / VALU_PEAK_WAVE_INSTS_PER_S, 4),
            "measured_on": valu.get("commit"),
            "note": "the kernel is VALU-issue bound, not HBM bound: this is the fraction of the binding resource.  insts is a "
                    "static count (SQ_INSTS_VALU of one launch, rocprofv3 pass with frame pipelining off, profiles/); the duration "
                    "it is divided by is this run's pipelined launch average"}

    extra = {"frames_per_s": round(args.steps / dt, 2), "band_rows": [r0, r1], "tile": list(swr_amd.tile_shape()),
             "bands": [{"device": d, "rows": [a, b]} for d, a, b in bands],
             "bin_overflow_checked": True,
             "compulsory_bytes_per_frame": W * (r1 - r0) * bytes_per_px + 32 * scene.vertices.shape[0] + 8 * scene.indices.size,
             "scene": {"degenerate_redrawn": scene.meta.get("degenerate_redrawn"), "degenerate_left": scene.meta.get("degenerate_left")}}

    if not args.no_extra:
        # per-stage breakdown with frame pipelining off (stages serialised on one stream, events around every stage)
        ctx.pipeline_enable(False)
        iso_steps = max(args.steps // 4, 5)
        _, sums_all, frames_all = run(flags, iso_steps, 2, level=2)
        extra["kernel_ms_avg"] = {k: round(v / max(frames_all, 1), 5) for k, v in sums_all.items() if k.endswith("_ms")}
        extra["kernel_ms_avg_note"] = "stages serialised (pipelining off); the timed region overlaps binning of frame N+1 with the raster of frame N"
        _, sums_iso, frames_iso = run(flags, iso_steps, 2, level=1)
        ctx.pipeline_enable(True)
        raster_iso_ms = sums_iso["raster_ms"] / max(frames_iso, 1)
        roofline["isolated"] = {
            "note": "same kernel with frame pipelining off (nothing else on the GPU)",
            "avg_launch_ms": round(raster_iso_ms, 5),
            "achieved": round(largest_band_px * bytes_per_px / (raster_iso_ms * 1e-3) / 1e9, 2) if raster_iso_ms > 0 else None,
            "frac": round(largest_band_px * bytes_per_px / (raster_iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if raster_iso_ms > 0 else None,
        }
        t_last = ctx.timings()
        extra["tile_pairs"], extra["tiles"] = t_last["tile_pairs"], t_last["tiles"]

        if n_gpus == 1:
            # the same scene with the colour store on (8 B/pixel) — reported beside the headline
            steps2 = max(args.steps, 20)          # (a handful of frames would mostly time the pipeline filling up)
            dt2, _, _ = run(S.FLAG_DEPTH_TEST, steps2, 3, level=0)
            _, sums2, frames2 = run(S.FLAG_DEPTH_TEST, steps2, 2, level=1, sample=min(SAMPLE, 4))
            r2 = sums2["raster_ms"] / max(frames2, 1)
            extra["color_plus_depth"] = {
                "Mpixels_per_s": round(W * H * steps2 / dt2 / 1e6, 2), "ms_per_step": round(dt2 / steps2 * 1e3, 4),
                "raster_ms": round(r2, 5), "roofline_frac": round(largest_band_px * 8 / (r2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if r2 > 0 else None}

        # ---- host-visible frames: every band copied into ONE page-locked host image -------------------------------
        hv_steps = max(min(args.steps, 60), 8)
        shm_path = None
        if world > 1:
            # one image shared by the ranks: a /dev/shm file mapped by every rank and page-locked (hipHostRegister)
            shm_path = f"/dev/shm/swr_bench_{os.environ.get('MASTER_PORT', '0')}_depth.bin"
            if rank == 0:
                with open(shm_path, "wb") as f:
                    f.truncate(W * H * 4)
            barrier()
            depth_img = np.memmap(shm_path, dtype=np.float32, mode="r+", shape=(H, W))
            swr_amd.host_register(depth_img)
            color_img = None
        else:
            depth_host = swr_amd.HostImage((H, W), np.float32)
            depth_img = depth_host
            color_img = None if (flags & S.FLAG_NO_COLOR) else swr_amd.HostImage((H, W, 4), np.uint8)
        dth, _, _ = run(flags, hv_steps, 3, level=0, present=(color_img, depth_img))
        hv_bytes = W * H * bytes_per_px
        extra["host_visible"] = {
            "Mpixels_per_s": round(W * H * hv_steps / dth / 1e6, 2), "ms_per_step": round(dth / hv_steps * 1e3, 4),
            "pcie_GBps": round(hv_bytes * hv_steps / dth / 1e9, 2), "bytes_per_frame": hv_bytes, "steps": hv_steps,
            "note": "draw + swr_present per frame (async D2H of every band into one page-locked host image, colour and depth "
                    "on two copy streams, device framebuffer double-buffered), swr_present_wait at the end; PCIe-bound"}
        if world > 1:
            swr_amd.host_unregister(depth_img)
            barrier()
            del depth_img
            if rank == 0:
                os.remove(shm_path)
        else:
            # the full drop-in call, as the reference's caller makes it (App.swift:153-185: the same mesh every frame, a new
            # transform): host pointers in (pageable scene arrays, page-locked images), pixels out.  cold = scene_id 0 (upload
            # every call, ABI 3's only behaviour); cached = the same non-zero scene_id again (resident mesh: one frame + gather)
            def render_calls(scene_id, n):
                rows = []
                for _ in range(n):
                    t0 = time.perf_counter()
                    ctx.render(scene.vertices, scene.indices, scene.transform, W, H, flags,
                               color=None if color_img is None else color_img.array, depth=depth_host.array, scene_id=scene_id)
                    wall = (time.perf_counter() - t0) * 1e3
                    rows.append(dict(ctx.render_timings(), wall_ms=wall))
                med = {k: round(float(np.median([r[k] for r in rows])), 3) for k in rows[0]}
                med["scene_cached"] = int(med["scene_cached"])
                return med
            cold = render_calls(0, 3)
            render_calls(0x5CE4E, 1)                    # makes the mesh resident under an id
            cached = render_calls(0x5CE4E, 5)
            extra["swr_render_ms"] = cold["wall_ms"]
            extra["swr_render"] = {"cold": cold, "cached_scene": cached,
                                   "note": "one synchronous swr_render (medians).  cold: scene_id 0 = H2D of the 120 MB scene from pageable "
                                           "arrays + the tail of the index check / stream build (index order, built behind the index copy) + one "
                                           "frame + gather into page-locked images.  "
                                           "cached_scene: the same non-zero swr_render_pass.scene_id again = one resident frame + the gather "
                                           "(PCIe-bound).  h2d_ms from HIP events, the rest host wall clock"}
            depth_host.free()
            if color_img is not None:
                color_img.free()

    if not args.no_extra and n_gpus == 1 and world == 1:
        # ---- ONE-GPU PROXY of the strong-scaling curve, NOT a scaling result: this GPU renders band k of N (what rank k of an
        # N-GPU run does, host contention and the other N - 1 devices aside); the slowest of the first / middle / last band
        # bounds the N-GPU frame.  Last measurement of the run: it re-targets the context.
        proxy = {"label": "one-GPU proxy, not a scaling result", "unit": "us per frame (worst of the first / middle / last band)", "bands": {}}
        base = None
        for parts in (1, 2, 4, 8):
            worst = 0.0
            for k in sorted({0, parts // 2, parts - 1}):
                b0, b1 = swr_amd.band_rows(H, parts, k)
                ctx.target_set(W, H, b0, b1)
                best = 1e9
                for _ in range(2):
                    for _ in range(10):
                        ctx.draw(scene.transform, flags)
                    ctx.sync()
                    t0 = time.perf_counter()
                    for _ in range(100):
                        ctx.draw(scene.transform, flags)
                    ctx.sync()
                    best = min(best, (time.perf_counter() - t0) / 100)
                worst = max(worst, best)
            base = worst if parts == 1 else base
            proxy["bands"][str(parts)] = {"worst_band_us": round(worst * 1e6, 1), "speedup": round(base / worst, 2),
                                          "efficiency": round(base / worst / parts, 3)}
        extra["band_proxy"] = proxy
        ctx.target_set(W, H, r0, r1)

    out = {
        "metric": "Mpixels/s at 4K on the 1M-triangle synthetic scene (frames/s in extra)",
        "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "value_steady": round(W * H * args.steps / dt_steady / 1e6, 2), "ms_per_step_steady": round(ms_per_step_steady, 4),
        "frames_before_steady_pass": frames_before_headline,
        "pass_order_note": "the same W + K frames are timed twice: value / ms_per_step = the first pass, with the requested warm-up, "
                           "straight out of idle (the GPU's clock ramp after idle costs the first ~150 frames 10-15 %, "
                           "profiles/r03/gpu_warmup_after_idle.txt); value_steady / ms_per_step_steady = again after the latency and "
                           "sampled-roofline passes (what a running frame loop sees).  Round 3's BENCH reported the steady pass as "
                           "`value` (0.0838) and the first as ms_per_step_first_pass (0.0879); rounds 1-2 and this round: first pass",
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_host_visible": (extra.get("host_visible") or {}).get("Mpixels_per_s"),
        "value_note": "value = device-resident frames (inputs and framebuffer in HBM, as the bench contract asks); value_host_visible = "
                      "the same frames with every band copied into one page-locked host image (SURVEY 8(d): 'host-visible image complete'), "
                      "PCIe-bound per GPU",
        "latency_ms": round(latency_ms, 4),
        "latency_note": "ms_per_step is inverse throughput with up to four frames in flight (frame lanes: every frame's binning + raster on "
                        "its own stream); latency_ms is one frame alone, swr_draw -> swr_sync",
        "config": {"workload": f"cfg4: {scene.triangles} random triangles (3 unshared vertices each), {W}x{H}, "
                               f"z-test, {'colour+depth' if args.color else 'depth-only'}, SplitMix64 seed 0x5EED0004, "
                               f"{scene.meta.get('degenerate_redrawn', 0)} degenerate triangles regenerated (SURVEY 8(d))",
                   "sharding": (f"{n_gpus} tile-row band(s), scene replicated, no collective; "
                                + ("one process per GPU (launcher)" if world > 1 else
                                   ("one process, one multi-device context (swr_config.device_count)" if in_process > 1 else "single context"))
                                + (" [REHEARSAL: bands share GPUs — not a result]" if rehearsal else ""))},
        "roofline": roofline, "extra": extra,
    }
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene)
        # vs_baseline stays null: BASELINE.md holds no published number for this metric.  The ratio to the CPU port timed in
        # this run is a reported baseline, not a target (the roofline fraction is the quality measure)
        out["vs_cpu_baseline"] = {"ratio": round(out["value"] / out["cpu_baseline"]["value"], 1),
                                  "note": "value / cpu_baseline.value (1 host core, C port of Renderer.swift): reported baseline, not target"}
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
