#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the configuration (default C2: the RTIOW random-sphere
scene at 1200x800, 500 spp, depth 50) through the C ABI (mirt_render), scene already
resident in HBM.  Metric: Msamples/s = width * height * spp * steps / seconds, whole
job.  With N > 1 -- one process per GPU: under torch.distributed.run as the driver starts
it, or from a plain shell, in which case this script starts the N ranks itself -- the
frame is split by interleaved rows across ranks (no data-path collective; rank 0 gathers
the rendered rows), or with --split sample every rank renders all pixels for its share of
the samples and the f64 accumulation images are summed with one RCCL reduce.  Scaling is
strong by default: the configuration's own frame and samples per pixel (BASELINE.json's
1200x800 at 500 spp) shared out over the N GPUs, which is what `value` measures at every N;
with N > 1 the same run then also times the weak case (the same frame at N times the
samples per pixel, every GPU rendering as many samples as the one GPU of N = 1) and
reports it under `weak_scaling` (--scaling weak makes that one `value` instead).
--single-process renders on the N devices from one process through mirt_render_multi
instead; --backend gloo lets ranks share a GPU (a rehearsal).

Rank 0 prints ONE JSON line.  `roofline` names what limits the dominant kernel.  For the
path kernel (default mode: whole paths in registers, 24 bytes of radiance out per sample)
that is the issue of vector instructions: `roofline.frac` = share of the SIMDs' cycles in
which a vector instruction issues x lanes active in it / 64, from the SQ counters of this
configuration committed under profiles/ (`counters_stale` says when they were collected
with another build of the library than the one loaded: mirt_build_id).  The figure the
survey defined -- the bytes a queue-per-bounce pipeline moves for the same work, 272 per path
segment + 24 per sample, per second of this kernel over 8 TB/s -- is kept as
`roofline.pipeline_equivalent` (a work-equivalent: the kernel moves none of those bytes, and
it exceeds 1 on the Cornell box), the HBM traffic the kernel really has as `roofline.traffic`
/ `hbm_frac_measured`, and `roofline.valu_f64` prices the reference traversal of the same
rays against the f64 vector ALU.  With --handover -1 the dominant kernel is the closest-hit
kernel of the per-bounce pipeline, priced against HBM with its queue traffic.
`cpu_baseline` times the CPU oracle (a C++ restatement of the reference algorithm,
not the reference itself: no Rust toolchain exists here) on a bounded sample of the
same workload on the host cores of the same box.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# bytes one closest-hit query moves through the ray queues in HBM: ray in (7 f64) + hit record out (t, ref)
EXTEND_BYTES_PER_SEGMENT = 56 + 16
# bytes per path segment over the whole extend + bin + shade step, and per sample for the accumulator
PIPELINE_BYTES_PER_SEGMENT = 272
PIPELINE_BYTES_PER_SAMPLE = 24
HBM_PEAK_GBS = 8000.0
FP64_VALU_PEAK_TOPS = 39.3  # 78.6 TFLOP/s counts an FMA as 2; this path may not fuse


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--spp", type=int, default=0, help="override the configuration's samples per pixel")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--split", default=None, choices=["tile", "sample"],
                    help="how N > 1 GPUs share the frame; default: tile, and sample for C5 (the split each configuration is specified with)")
    ap.add_argument("--paths-per-pass", type=int, default=0)
    ap.add_argument("--handover", type=int, default=0,
                    help="mirt_render_opts.handover_bounce: 0 = path kernel from the camera ray (default), N = launches per "
                         "bounce below bounce N, -1 = launches per bounce throughout, -2 = probe")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work to spend on the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not time kernel classes with HIP events")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default) = the configuration's own frame and samples per pixel shared out (the weak case is "
                         "timed too and reported as weak_scaling); weak = the same frame at N times the samples per pixel, so "
                         "every GPU renders as many samples as the one GPU of N = 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collectives between the ranks: nccl (RCCL over xGMI, one GPU per rank) or gloo (host memory; ranks "
                         "may then share a GPU: rank r uses device r mod the number of devices -- a rehearsal on a smaller box)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 from ONE process through mirt_render_multi (one host thread per device) instead of one process per GPU")
    return ap.parse_args()


def cpu_baseline(cfg, world, seconds):
    """Times the CPU oracle on the same scene, camera and frame at a reduced sample count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    cam = cfg.camera()
    threads = oracle_lib.lib().oracle_max_threads()
    # calibrate on one whole frame at 1 spp (rows differ in cost: sky vs ground), then
    # spend about `seconds`
    _, _, _, per_frame_spp1 = oracle_lib.render(world.desc, cam, cfg.background, cfg.width, cfg.height, 1)
    spp = int(max(1, min(cfg.spp, seconds / max(per_frame_spp1, 1e-3))))
    _, _, ctr, secs = oracle_lib.render(world.desc, cam, cfg.background, cfg.width, cfg.height, spp)
    samples = cfg.width * cfg.height * spp
    extra = {}
    if cfg.name.startswith("C2"):
        # the reference's own CPU-runnable case, exactly: the same scene at 400x225, 10 spp (best of three: it takes 0.1 s)
        from ray_tracer_amd import configs
        c1 = configs.C1
        runs = [oracle_lib.render(world.desc, c1.camera(), c1.background, c1.width, c1.height, c1.spp)[3] for _ in range(3)]
        extra["c1_400x225x10_msamples_per_s"] = c1.width * c1.height * c1.spp / min(runs) / 1e6
    return {
        **extra,
        "value": samples / secs / 1e6,
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{cfg.width}x{cfg.height} frame of the same scene and camera at {spp} spp ({samples} samples, "
                  f"{secs:.1f} s), f64 C++ restatement of the reference algorithm, OpenMP dynamic over pixels",
        "mrays_per_s": ctr["segments"] / secs / 1e6,
        "box_tests_per_segment": ctr["box_tests"] / max(1, ctr["segments"]),
        "leaf_tests_per_segment": ctr["leaf_tests"] / max(1, ctr["segments"]),
    }


def launch_ranks(n):
    """python bench.py --gpus N from a plain shell: the same command under torch.distributed.run,
    N child processes (this process has not initialised a GPU and only waits)."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def main():
    args = parse_args()
    if args.split is None:
        args.split = "sample" if args.config == "C5" else "tile"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process:
        world_size = 1
    elif "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started from a plain shell: one fresh process per GPU, before this one has touched a GPU
        raise SystemExit(launch_ranks(args.gpus))
    elif world_size != args.gpus:
        args.gpus = world_size
    # ONE line on stdout, the result: whatever else the libraries below write there (gloo announces its
    # connections on stdout) goes to stderr instead, in every rank
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from ray_tracer_amd import _abi as abi
    from ray_tracer_amd import api, configs, distributed

    if not torch.cuda.is_available() or api.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: no HIP device visible")
    n_devices = api.device_count()
    if args.backend == "nccl" and not args.single_process and local_rank >= n_devices:
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this box ({n_devices} visible); --backend gloo lets ranks share one")
    device_index = local_rank % n_devices
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where tensors handed to collectives live
    gloo = None
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
            gloo = dist.new_group(backend="gloo")
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)
    n_ranks = args.gpus if args.single_process else world_size  # shares the frame is cut into
    cfg = configs.ALL[args.config]
    if args.spp or args.width or args.height:
        cfg = cfg.scaled(args.width or cfg.width, args.height or cfg.height, args.spp or cfg.spp)
    spp_one_gpu = cfg.spp
    if n_ranks > 1 and args.scaling == "weak":
        cfg = cfg.scaled(cfg.width, cfg.height, cfg.spp * n_ranks)
    W, H, spp = cfg.width, cfg.height, cfg.spp
    # the other scaling's sample count, timed after the main loop when the main loop is the strong case
    spp_weak = spp_one_gpu * n_ranks if (n_ranks > 1 and args.scaling == "strong") else 0
    if args.split == "sample" and n_ranks > 1 and spp < n_ranks:
        raise SystemExit(f"--split sample needs at least one sample per rank: spp {spp} < {n_ranks} ranks")
    world = cfg.world()
    cam = cfg.camera()
    # the first call into the library pays for the HIP context and for loading its code object, once per process
    t_upload = time.perf_counter()
    api.Scene(world, device=device_index).close()
    first_call_ms = (time.perf_counter() - t_upload) * 1e3
    t_upload = time.perf_counter()
    scene = api.Scene(world, device=device_index)  # scene resident in HBM before the timed region
    scene_upload_ms = (time.perf_counter() - t_upload) * 1e3  # culling-tree build on the host + copies; outside `value`
    rank_scenes = [scene]
    if args.single_process and args.gpus > 1:
        # one scene per rank, on device r (on a box with fewer devices the ranks share them)
        rank_scenes = [scene if r % n_devices == device_index else api.Scene(world, device=r % n_devices) for r in range(args.gpus)]
    profile = not args.no_profile

    out = np.zeros((H, W, 3), dtype=np.uint8)
    accum = torch.zeros(H * W * 3, dtype=torch.float64, device=dev) if args.split == "sample" else None
    step_stats = []

    # Tile split over several ranks: the gather of a finished frame's rows (host memory, gloo) runs on a helper
    # thread while the next frame renders into a second image; every frame is gathered before the clock stops.
    pipelined = world_size > 1 and args.split == "tile" and len(rank_scenes) == 1
    images = [out, np.zeros_like(out)] if pipelined else [out]
    gathers = [None] * len(images)
    gatherer = None
    if pipelined:
        from concurrent.futures import ThreadPoolExecutor

        gatherer = ThreadPoolExecutor(max_workers=1)  # one thread: the ranks issue their gathers in the same order
    frames = [0]

    def drain():
        for k, g in enumerate(gathers):
            if g is not None:
                g.result()
                gathers[k] = None

    def barrier():
        drain()
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step(spp=spp):
        if len(rank_scenes) > 1:
            _, sts = api.render_world_multi(cfg.background, rank_scenes, cam, W, H, spp, split=args.split, paths_per_pass=args.paths_per_pass,
                                            out=out, return_stats=True, handover=(args.handover, 0))
            st = dict(sts[0])
            for k in ("samples", "segments", "segments_paths", "launches_extend", "launches_paths"):
                st[k] = sum(x[k] for x in sts)
            return st
        if args.split == "tile" or world_size == 1:
            # rows interleaved over ranks: tile = one row, rank r owns rows r, r + N, ...
            k = frames[0] % len(images)
            frames[0] += 1
            if gathers[k] is not None:  # the gather that still reads this image
                gathers[k].result()
                gathers[k] = None
            _, st = api.render_world(cfg.background, scene, cam, W, H, spp, paths_per_pass=args.paths_per_pass,
                                     profile=profile, out=images[k], return_stats=True,
                                     tiles=distributed.row_tiles(W, world_size, rank), handover=(args.handover, 0))
            if pipelined:
                gathers[k] = gatherer.submit(distributed.gather_rows, images[k], rank, world_size, gloo)
        else:
            accum.zero_()
            lo, hi = distributed.sample_range(spp, rank, world_size)  # (never empty: spp >= ranks was checked)
            st = api.render_accumulate(scene, cam, cfg.background, W, H, spp, accum.data_ptr(), sample_range=(lo, hi),
                                       paths_per_pass=args.paths_per_pass, profile=profile, handover=(args.handover, 0))
            if args.backend == "nccl":
                distributed.reduce_accum(accum)  # RCCL over xGMI
            else:
                host = accum.cpu()
                distributed.reduce_accum(host)
                if rank == 0:
                    accum.copy_(host)
            if rank == 0:
                api.resolve(accum.data_ptr(), W, H, spp, out=out, device=device_index)
        return st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_stats.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    weak = None
    if spp_weak:
        # the weak case of the same run: the same frame at N times the samples per pixel, same bracket
        weak_steps = max(1, min(args.steps, 3))
        step(spp_weak)
        barrier()
        t0 = time.perf_counter()
        for _ in range(weak_steps):
            step(spp_weak)
        barrier()
        weak_elapsed = time.perf_counter() - t0
        if world_size > 1:
            t = torch.tensor([weak_elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            weak_elapsed = float(t.item())
        weak = {"value": float(W) * H * spp_weak * weak_steps / weak_elapsed / 1e6, "unit": "Msamples/s", "spp": spp_weak,
                "steps": weak_steps, "warmup": 1, "ms_per_step": weak_elapsed * 1e3 / weak_steps,
                "note": "the same frame at n_gpus x the configuration's samples per pixel: every GPU renders as many samples as "
                        "the one GPU of n_gpus = 1"}

    # sums over ranks of what each rank's library calls report
    keys = ("samples", "segments", "segments_paths", "launches_extend", "launches_paths", "ms_extend", "ms_shade", "ms_bin",
            "ms_raygen", "ms_accumulate", "ms_paths")
    local = torch.tensor([sum(st[k] for st in step_stats) for k in keys], dtype=torch.float64, device=cdev)
    dominant = "ms_paths" if step_stats[-1]["handover_bounce"] == 0 else "ms_extend"
    local_max_extend = torch.tensor([sum(st[dominant] for st in step_stats)], dtype=torch.float64, device=cdev)
    if world_size > 1:
        dist.all_reduce(local, op=dist.ReduceOp.SUM)
        dist.all_reduce(local_max_extend, op=dist.ReduceOp.MAX)
    tot = dict(zip(keys, local.tolist()))

    if rank == 0:
        samples = float(W) * H * spp * args.steps
        assert abs(tot["samples"] - samples) < 0.5, (tot["samples"], samples)
        value = samples / elapsed / 1e6
        result = {
            "metric": "Msamples/s (camera paths x spp per second), RTIOW final scene 1200x800" if args.config == "C2"
                      else f"Msamples/s (camera paths x spp per second), {cfg.name}",
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (seeded scene generator, world seed 0x5EED0001, render seed 0x5EED0002)",
            "config": {
                "workload": cfg.name,
                "width": W, "height": H, "spp": spp, "spp_at_one_gpu": spp_one_gpu, "max_depth": api.MAX_DEPTH,
                "shapes": world.n_shapes, "bvh_members": world.n_nodes,
                "split": args.split if n_ranks > 1 else "none",
                "ranks": ("one process, one host thread per device (mirt_render_multi)" if args.single_process and n_ranks > 1
                          else "one process per GPU (torch.distributed)" if n_ranks > 1 else "one"),
                "collectives": args.backend if world_size > 1 else "none",
                "devices_visible": n_devices,
                "traversal": "fast (f32 culling tree, near-first, distance-pruned; reference boxes decide which hits count)",
            },
            "handover": {"bounce": int(step_stats[-1]["handover_bounce"]),
                         "path_kernel_segments_frac": tot["segments_paths"] / max(1.0, tot["segments"])},
            "library_build_id": abi.libmirt().mirt_build_id().decode(),
            "parity": "unpinned by the reference (it holds no fixtures and cannot be built here); bit-exact hits and images within "
                      "one level against the repository's own CPU restatement, tests/ -m gpu",
            "scene_upload_ms": scene_upload_ms,
            "library_first_call_ms": first_call_ms,
            "mrays_per_s": tot["segments"] / elapsed / 1e6,
            "segments_per_sample": tot["segments"] / samples,
        }
        if profile and dominant == "ms_paths" and tot["ms_paths"] > 0:
            launches = tot["launches_paths"]
            avg_ms = tot["ms_paths"] / launches
            # the algorithmic bytes of the path, per segment and per sample, whatever the kernel keeps in
            # registers instead of moving: the same figure prices the per-bounce pipeline below
            bytes_per_launch = (tot["segments_paths"] * PIPELINE_BYTES_PER_SEGMENT + samples * PIPELINE_BYTES_PER_SAMPLE) / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            pipeline = {
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "bytes_per_unit": {"segment": PIPELINE_BYTES_PER_SEGMENT, "sample": PIPELINE_BYTES_PER_SAMPLE},
                "units_per_launch": {"segments": tot["segments_paths"] / launches, "samples": samples / launches},
                "note": "a work-equivalent, not a utilisation: the bytes a queue-per-bounce pipeline moves through HBM for these "
                        "segments (ray 56 + hit 16 + path state 88 in, 88 out + two queue indices 8 = 272 B per segment, 24 B "
                        "per sample) per second of this kernel, which keeps all of it in registers",
            }
            roof = {
                "bound": "valu_issue",
                "kernel": "k_paths (whole paths, one per lane: camera ray, closest hit, shade, next bounce, in registers)",
                "achieved": None,
                "peak": 1.0,
                "unit": "share of the vector ALU's lane issue slots (SIMD cycles in which a vector instruction issues x lanes active / 64)",
                "frac": None,
                "traffic": None,
                "avg_launch_ms": avg_ms,
                "launches": launches,
                "pipeline_equivalent": pipeline,
                "pipeline_equivalent_frac": pipeline["frac"],
            }
            result["roofline"] = roof
            # The counters of this kernel on this workload come from the PMC passes committed under profiles/
            # (separate rocprofv3 --pmc runs of one launch; per path, scaled to this run's paths per launch).
            tags = ("r03", "r02", "r01")
            pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"{tag}_pmc_{args.config}.json") for tag in tags)
                             if os.path.exists(q)), None)
            if pmc_path is None and args.config == "C2":
                pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_k_paths.json")
            if pmc_path and os.path.exists(pmc_path):
                with open(pmc_path) as f:
                    pmc = json.load(f)
                roof["traffic"] = pmc["hbm_bytes_per_path"] * samples / launches
                roof["hbm_frac_measured"] = roof["traffic"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                busy = pmc.get("valu_issue_busy_share_of_simd_cycles")
                lanes = pmc.get("lanes_active_per_valu_instruction")
                roof["valu_issue_busy"] = busy
                roof["lanes_active_per_valu_instruction"] = lanes
                if busy is not None and lanes is not None:
                    # (the busy share's denominator is the mean lifetime of a wave, which is shorter than its SIMD's by
                    # the launch's tail: on a kernel that saturates the vector ALU the share can read 1.01)
                    roof["achieved"] = roof["frac"] = min(1.0, busy) * lanes / 64.0
                roof["counters_source"] = os.path.relpath(pmc_path, ROOT) + " (counters of one launch, per path x this run's paths per launch)"
                roof["counters_build_id"] = pmc.get("library_build_id")
                roof["counters_stale"] = pmc.get("library_build_id") != result["library_build_id"]
        elif profile and tot["ms_extend"] > 0:
            launches = tot["launches_extend"]
            avg_ms = tot["ms_extend"] / launches
            extend_segments = tot["segments"] - tot["segments_paths"]  # the path kernel's are not this kernel's
            bytes_per_launch = extend_segments * EXTEND_BYTES_PER_SEGMENT / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            result["roofline"] = {
                "bound": "hbm",
                "kernel": "k_traverse (closest-hit query of every live path of one bounce)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "bytes_per_unit": EXTEND_BYTES_PER_SEGMENT,
                "units_per_launch": extend_segments / launches,
                "avg_launch_ms": avg_ms,
                "launches": launches,
                "pipeline_gbs": (tot["segments"] * PIPELINE_BYTES_PER_SEGMENT + samples * PIPELINE_BYTES_PER_SAMPLE)
                                / elapsed / 1e9,
            }
        if weak:
            result["weak_scaling"] = weak
        if profile:
            result["kernel_ms"] = {k[3:]: tot[k] / world_size for k in keys if k.startswith("ms_")}
        if args.cpu_seconds > 0 and world_size == 1:
            base = cpu_baseline(cfg, world, args.cpu_seconds)
            result["cpu_baseline"] = base
            if "roofline" in result:
                # FP64 VALU view of the same kernel: operations the reference algorithm needs for
                # these rays (27 per box test, 33 per leaf test, counted by the oracle) per second
                kernel_segments = tot["segments_paths"] if dominant == "ms_paths" else tot["segments"] - tot["segments_paths"]
                ops = kernel_segments * (27 * base["box_tests_per_segment"] + 33 * base["leaf_tests_per_segment"])
                result["roofline"]["valu_f64"] = {
                    "achieved_tops": ops / (local_max_extend.item() * 1e-3) / 1e12,
                    "peak_tops": FP64_VALU_PEAK_TOPS,
                    "note": "reference-traversal op count of this kernel's rays (shading not counted) / its time",
                }
                result["roofline"]["valu_f64"]["frac"] = result["roofline"]["valu_f64"]["achieved_tops"] / FP64_VALU_PEAK_TOPS
        os.write(result_fd, (json.dumps(result) + "\n").encode())

    for sc in set(rank_scenes):
        sc.close()
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
