#!/usr/bin/env python3
"""bench.py — the BASELINE.json metric on MI355X: Mpixels/s (and frames/s) at 4K on the
1M-triangle synthetic soup (config 4: 3840x2160, z-test, depth-only), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame of the hot path (clear + vertex transform + setup + binning + tile raster
+ framebuffer write) over the resident scene; inputs and the framebuffer stay in HBM.  With
N > 1 the framebuffer is split into N tile-row bands; every rank holds the whole
scene and renders its band; there is no collective on the data path (bands are disjoint), only
the timing barrier.  The total work is fixed as N grows => "scaling": "strong".

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (k_raster): algorithmic
framebuffer bytes of the band / its mean HIP-event duration over the timed region, against the
8 TB/s HBM peak.  `cpu_baseline` (rank 0, N = 1 only) times the CPU oracle — the C restatement
of the reference's Renderer.swift loop — on whole frames of the same workload.
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


def shard_rows(swr, height: int, world: int, rank: int):
    """Tile-row band of `rank` (the C-ABI's swr_band_rows)."""
    return swr.band_rows(height, world, rank)


def load_traffic():
    """HBM bytes per k_raster launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f)
        except Exception:
            return None
    return None


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
    return {"value": round(mpix, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
            "sample": f"{frames} whole frames of the same workload ({scene.triangles} triangles, "
                      f"{scene.width}x{scene.height}, flags={scene.flags}); median of frames 2..{frames}; "
                      f"{st.fragments} fragments/frame",
            "ms_per_frame": round(t * 1e3, 2), "host_cpus": os.cpu_count()}


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
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary colour+depth measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    device = local_rank
    rehearsal = False
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if ndev >= world or os.environ.get("SWR_BENCH_BACKEND", "") == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # rehearsal on a box with fewer GPUs than ranks (e.g. 2 ranks on the 1-GPU dev box):
            # same code path, ranks share devices, gloo carries the barrier / MAX.  Not a result.
            rehearsal = True
            device = local_rank % max(ndev, 1)
            torch.cuda.set_device(device)
            dist.init_process_group("gloo")
    else:
        try:
            import torch
        except Exception:
            torch = None

    import swr_amd
    if local_rank == 0:
        swr_amd.build()          # no-op when the in-tree .so is current
    if dist is not None:
        dist.barrier()
    S = swr_amd.scenes

    scene = S.cfg4_soup(ntri=args.triangles, width=args.width, height=args.height,
                        depth_only=not args.color)
    W, H = scene.width, scene.height
    r0, r1 = shard_rows(swr_amd, H, world, rank)

    ctx = swr_amd.Context(device if world > 1 else -1)
    ctx.scene_upload(scene.vertices, scene.indices)
    ctx.target_set(W, H, r0, r1)

    def sync_all():
        ctx.sync()
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    # HIP events around k_raster on its own stream, live inside the timed region, on every SAMPLE-th launch: an
    # event pair on the raster stream is a synchronisation point that costs a pipelined frame about 15 us, so
    # bracketing every launch would slow the very region it measures (SWR_BENCH_TIMING_SAMPLE=1 does that).
    SAMPLE = max(1, int(os.environ.get("SWR_BENCH_TIMING_SAMPLE", "8")))

    def timed(flags, steps, warmup, level=1, sample=1):
        for _ in range(warmup):
            ctx.draw(scene.transform, flags)
        sync_all()
        ctx.timing_sample(sample)
        ctx.timing_enable(level)
        ctx.timing_reset()
        barrier()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.draw(scene.transform, flags)
        sync_all()
        barrier()
        t1 = time.perf_counter()
        sums, frames = ctx.timing_totals()
        ctx.timing_enable(0)
        dt = t1 - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, sums, frames

    flags = scene.flags
    SAMPLE = max(1, min(SAMPLE, args.steps // 8))     # short runs bracket more (or all) of their launches: >= 8 timed
    dt, sums, frames = timed(flags, args.steps, args.warmup, level=int(os.environ.get("SWR_BENCH_TIMING_LEVEL", "1")),
                             sample=SAMPLE)
    ms_per_step = dt / args.steps * 1e3
    mpix = W * H * args.steps / dt / 1e6

    # roofline of the dominant kernel on this rank's band
    bytes_per_px = 4 if (flags & S.FLAG_NO_COLOR) else 8
    band_px = W * (r1 - r0)
    raster_ms = sums["raster_ms"] / max(frames, 1)
    achieved = band_px * bytes_per_px / (raster_ms * 1e-3) / 1e9 if raster_ms > 0 else 0.0
    traffic = load_traffic()
    roofline = {
        "bound": "hbm", "kernel": "k_raster<ztest>", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": (traffic or {}).get("k_raster_bytes_per_launch") if world == 1 else None,
        "algorithmic_bytes_per_launch": band_px * bytes_per_px,
        "avg_launch_ms": round(raster_ms, 5), "launches_timed": frames, "timed_every_nth_launch": SAMPLE,
        "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 5),
    }
    # per-stage breakdown from a short separate run with frame pipelining off (stages serialised on
    # one stream) and events around every stage — those events cost stream time, and under
    # pipelining the stages of two frames overlap, so this stays out of the timed region above
    ctx.pipeline_enable(False)
    dt_iso, sums_all, frames_all = timed(flags, max(args.steps // 4, 5), 2, level=2)
    kernels = {k: round(v / max(frames_all, 1), 5) for k, v in sums_all.items() if k.endswith("_ms")}
    iso_steps = max(args.steps // 4, 5)
    _, sums_iso, frames_iso = timed(flags, iso_steps, 2, level=1)
    ctx.pipeline_enable(True)
    t_last = ctx.timings()
    raster_iso_ms = sums_iso["raster_ms"] / max(frames_iso, 1)
    roofline["isolated"] = {
        "note": "same kernel with frame pipelining off (nothing else on the GPU)",
        "avg_launch_ms": round(raster_iso_ms, 5),
        "achieved": round(band_px * bytes_per_px / (raster_iso_ms * 1e-3) / 1e9, 2) if raster_iso_ms > 0 else None,
        "frac": round(band_px * bytes_per_px / (raster_iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if raster_iso_ms > 0 else None,
    }

    extra = {"frames_per_s": round(args.steps / dt, 2), "kernel_ms_avg": kernels,
             "kernel_ms_avg_note": "stages serialised (pipelining off); the timed region overlaps binning of frame N+1 with the raster of frame N",
             "tile_pairs": t_last["tile_pairs"], "tiles": t_last["tiles"], "band_rows": [r0, r1],
             "tile": list(swr_amd.tile_shape()),
             "compulsory_bytes_per_frame": band_px * bytes_per_px + 32 * scene.vertices.shape[0] + 8 * scene.indices.size}

    if not args.no_extra and world == 1:
        # the same scene with the colour store on (8 B/pixel) — reported beside the headline
        dt2, sums2, frames2 = timed(S.FLAG_DEPTH_TEST, max(args.steps // 4, 5), 3, sample=min(SAMPLE, 4))
        steps2 = max(args.steps // 4, 5)
        r2 = sums2["raster_ms"] / max(frames2, 1)
        extra["color_plus_depth"] = {
            "Mpixels_per_s": round(W * H * steps2 / dt2 / 1e6, 2), "ms_per_step": round(dt2 / steps2 * 1e3, 4),
            "raster_ms": round(r2, 5), "roofline_frac": round(band_px * 8 / (r2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if r2 > 0 else None}

    out = {
        "metric": "Mpixels/s at 4K on the 1M-triangle synthetic scene (frames/s in extra)",
        "value": round(mpix, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"cfg4: {scene.triangles} random triangles (3 unshared vertices each), {W}x{H}, "
                               f"z-test, {'colour+depth' if args.color else 'depth-only'}, SplitMix64 seed 0x5EED0004",
                   "sharding": f"{world} tile-row band(s), scene replicated, no collective"
                               + (" [REHEARSAL: ranks share GPUs, gloo]" if rehearsal else "")},
        "roofline": roofline, "extra": extra,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
