#!/usr/bin/env python3
"""Benchmark of the MI355X render loop: path samples / s at 1080p.

    python bench.py --gpus N --steps K --warmup W

One step = one full render of the workload (all pixels, all samples) on every
rank.  N = 1 renders the Cornell box with glossy (Phong) boxes and MIS at
1920x1080, 1024 spp -- the configuration the metric is quoted on.  With N > 1
(launched one process per GPU by torch.distributed.run) every rank renders its
own 1024-spp batch of the same frame (rank-specific seed: weak scaling) and the
float sum buffers are combined on rank 0 with one RCCL reduce inside the timed
region.  Rank 0 prints one JSON line.

The line also carries
  roofline      algorithmic bytes of the reference algorithm (per-sample counts
                measured by the instrumented kernel on the same scene) divided
                by the kernel's duration from HIP events on the launch stream,
                against the HBM peak; and the same for FP32 vector flops,
  cpu_baseline  the compiled reference (oracle/_ref/libpt_ref_fast.so: the
                reference's own pathSample / BVH / material code with its
                shipped optimisation level) timed on this machine's host cores
                on a reduced-spp render of the same scene.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # HBM3E peak, GB/s
VALU_PEAK_TFLOPS = 157.3     # FP32 vector peak, TFLOP/s


def workload(name):
    """(scene, camera, label, sqrt_spp, extra): extra may hold width, height, samples (a range of the
    sqrt_spp^2 strata), max_path_segments, frames (list of shutter intervals, one render each per step)
    and the sharding mode for more than one GPU."""
    import path_tracing_amd as pt
    golden = os.path.join(ROOT, "tests", "golden")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if name == "cornell":
        scene = pt.LoadedScene(os.path.join(golden, "cornell_glossy.ptscene"))
        cam = pt.load_camera(os.path.join(golden, "cornell_wide.ptcam"))
        return scene, cam, "cornell-box glossy+MIS 1920x1080 1024spp (BASELINE configs[1])", 32, {}
    if name.startswith("mesh"):
        import scenes
        n = int(name[4:] or 1000)
        s, cam = scenes.big_mesh_scene(n)
        return s, cam, "displaced grid %d triangles 1920x1080 256spp (BASELINE configs[2])" % (2 * n * n + 2), 16, {}
    if name == "textured":
        # every material, image/checker/transformer textures, normal and opacity maps, image environment map
        import scenes
        s, cam, _ = scenes.build("materials")
        return s, cam, "textured multi-material 3840x2160 512spp max depth 32 (BASELINE configs[3])", 23, {
            "width": 3840, "height": 2160, "samples": (0, 512), "max_path_segments": 32, "sharding": "tiles"}
    if name == "frames":
        # two-key animated spheres, meshes, lights and camera; a new tree per frame (shutter 1/25 s)
        import scenes
        s, cam, _ = scenes.build("animated")
        frames = [(f / 25.0, (f + 1) / 25.0) for f in range(8)]
        return s, cam, "animated scene, 8 of 240 frames per step, 1920x1080 128spp (BASELINE configs[4])", 12, {
            "samples": (0, 128), "frames": frames, "sharding": "frames"}
    raise SystemExit("unknown workload " + name)


def algorithmic_per_sample(cnt, n_lights):
    """Bytes and flops per path sample of the reference algorithm, from measured work counts:
    32 B per node visit, 48 B per leaf primitive test, 48 B per light-pdf primitive test, 12 B of
    output; ~30 flops per slab test, ~60 per primitive test, ~300 per shaded vertex."""
    n = max(cnt["path_samples"], 1)
    bytes_ = (cnt["node_visits"] * 32 + cnt["leaf_tests"] * 48 + cnt["light_tests"] * 48) / n + 12
    flops = (cnt["node_visits"] * 30 + (cnt["leaf_tests"] + cnt["light_tests"]) * 60 + cnt["vertices"] * 300) / n
    return bytes_, flops


def measured_traffic(workload_name, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (tools/collect_traffic.sh writes
    profiles/traffic_<workload>.json from two rocprofv3 --pmc runs of this benchmark: FETCH_SIZE and
    WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled as gfx950 needs for 16-byte-per-lane
    streaming reads).  None when no such file is present."""
    path = os.path.join(ROOT, "profiles", "traffic_%s.json" % workload_name)
    if not os.path.exists(path):
        return None, "no PMC pass committed for this workload"
    with open(path) as f:
        t = json.load(f)
    ks = t.get("kernels", {})
    if kernel == "traceKernel":
        # one tracing step of the renderer is one launch of traceKernel (scenes held in LDS), or a launch of
        # traceOrderedKernel followed by one of traceRetryKernel (scenes in memory)
        rows = [v for name, v in ks.items() if name.startswith("trace")]
    else:
        rows = [ks[kernel]] if kernel in ks else []
    if not rows:
        return None, "PMC file has no row for " + kernel
    return sum(r["hbm_bytes_per_launch"] for r in rows), t.get("how", "")


def cpu_baseline(scene, cam, width, height, seed):
    """Time the compiled reference on the host cores on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refbind
    import path_tracing_amd as pt
    lib = refbind.REF_FAST_LIB if os.path.exists(refbind.REF_FAST_LIB) else refbind.REF_LIB
    if not os.path.exists(lib):
        return None
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    R = refbind.Reference(lib)
    d = scene.desc()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference prints progress on stderr
    try:
        rs = R.scene(C.byref(d))
        # calibrate on 1 spp, then pick sqrt_spp for roughly 15 s
        p = pt.make_params(width, height, 1, seed=seed)
        _, t1 = rs.render(cam, p, rng_mode=1, threads=threads)
        rate = width * height / max(t1, 1e-6)
        sqrt_spp = max(1, min(8, int((15.0 * rate / (width * height)) ** 0.5)))
        p = pt.make_params(width, height, sqrt_spp, seed=seed)
        _, secs = rs.render(cam, p, rng_mode=1, threads=threads)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    n = width * height * sqrt_spp * sqrt_spp
    return {"value": n / secs, "unit": "path samples/s", "cores": threads, "kind": "reference",
            "sample": "%dx%d at %d spp (%d path samples, %.1f s), reference's own generator, "
                      "g++ -Ofast -march=x86-64-v3 -fopenmp" % (width, height, sqrt_spp * sqrt_spp, n, secs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sqrt-spp", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bvh", choices=["host", "device"], default="host",
                    help="host: the reference's tree (pt_bvh_build); device: the tree built by pt_scene_build_bvh_device")
    args = ap.parse_args()
    if os.environ.get("PT_BENCH_WATCHDOG"):
        # where every thread stands, should a run stop making progress
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["PT_BENCH_WATCHDOG"]), exit=False, file=sys.stderr)

    import torch
    import torch.distributed as dist
    import path_tracing_amd as pt
    from path_tracing_amd import abi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU machine: PT_BENCH_REHEARSAL=1 puts every rank on GPU 0 and exchanges over gloo
    rehearsal = os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus does not match WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    scene, cam, label, default_sqrt, extra = workload(args.workload)
    sqrt_spp = args.sqrt_spp or default_sqrt
    W, H = extra.get("width", args.width), extra.get("height", args.height)
    samples = extra.get("samples") if not args.sqrt_spp else None
    spp = (samples[1] - samples[0]) if samples else sqrt_spp * sqrt_spp
    frames = extra.get("frames")
    mode = extra.get("sharding", "batches")
    dev = pt.DeviceScene(scene, device=local_rank)
    bvh_build_ms = None
    if args.bvh == "device":
        dev.build_bvh_device(0.0, 0.0)  # first call allocates the work area
        torch.cuda.synchronize()
        tb = time.perf_counter()
        dev.build_bvh_device(0.0, 0.0)
        torch.cuda.synchronize()
        bvh_build_ms = (time.perf_counter() - tb) * 1e3
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    # every rank renders its own batch of samples of the same frame (its own seed)
    params = pt.make_params(W, H, sqrt_spp, seed=42, samples=samples,
                            max_path_segments=extra.get("max_path_segments", 0))

    from path_tracing_amd import distributed as shard
    events = []

    def render(p, buf):
        dev.render_device(cam, p, buf.data_ptr(), stream.cuda_stream)

    def step(timed=False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        if frames:
            # this rank's frames, round robin: tree over the frame's shutter interval (host builder),
            # upload, render; no exchange
            for f in shard.plan_frames(len(frames), world, rank):
                t0f, t1f = frames[f]
                if args.bvh == "device":
                    dev.build_bvh_device(t0f, t1f, stream.cuda_stream)
                else:
                    dev.update_bvh(scene.build_bvh(t0f, t1f))
                render(pt.make_params(W, H, sqrt_spp, seed=42 + f, t0=t0f, t1=t1f, samples=samples), out)
        elif world == 1:
            render(params, out)
        else:
            # this rank's share (its own batch of samples, or its tiles), then one RCCL sum-reduce to rank 0
            if mode == "tiles":
                out.zero_()
            for p in shard.shard_params(params, mode, world, rank, tile=512):
                render(p, out)
        if timed:
            e1.record(stream)
            events.append((e0, e1))
        if world > 1 and not frames:
            if rehearsal:
                # gloo, ranks sharing one GPU: reduce a host copy (gloo's own staging of device tensors was seen
                # to stall now and then when two processes share a device)
                host = out.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                out.copy_(host)
            else:
                dist.reduce(out, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0 and mode == "batches":
                out.div_(float(spp * world))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if rank == 0:
        dev.profile(True)   # events between the kernels, on the launch stream, for the timed steps
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    # duration of the whole launch sequence of one render: HIP events on the launch stream
    last_ms = sum(e0.elapsed_time(e1) for e0, e1 in events) / len(events)
    kernel_times = dev.kernel_times() if rank == 0 else {}
    dev.profile(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        n_renders = len(frames) if frames else (world if mode == "batches" else 1)
        total_samples = float(W) * H * spp * n_renders * args.steps
        value = total_samples / elapsed
        # work counts of the reference algorithm on this scene (instrumented kernel, reduced spp)
        cparams = pt.make_params(W // 4, H // 4, min(sqrt_spp, 4), seed=42)
        _, cnt = dev.render_counted(cam, cparams)
        cnt = cnt.as_dict()
        bytes_ps, flops_ps = algorithmic_per_sample(cnt, 0)
        launch_samples = float(W) * H * spp * (len(shard.plan_frames(len(frames), world, 0)) if frames else 1)
        if world > 1 and mode == "tiles":
            launch_samples /= world
        n_s = max(cnt["path_samples"], 1)
        # the reference algorithm's bytes, split by the kernel that does that part of the work
        kernel_bytes_ps = {
            "traceKernel": (cnt["node_visits"] * 32 + cnt["leaf_tests"] * 48) / n_s,
            "shadeHitsKernel": cnt["light_tests"] * 48 / n_s + 12,
        }
        kernels = {}
        for name, (launches, total_ms) in kernel_times.items():
            if launches:
                kernels[name] = {"launches_per_step": launches / args.steps, "avg_us": total_ms / launches * 1e3,
                                 "ms_per_step": total_ms / args.steps}
                if name in kernel_bytes_ps:
                    per_launch = launch_samples * args.steps / launches
                    kernels[name]["algorithmic_GBps"] = kernel_bytes_ps[name] * per_launch / (total_ms / launches * 1e-3) / 1e9
                    kernels[name]["hbm_bytes_per_launch"] = measured_traffic(args.workload, name)[0]
        dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
        dk = kernels[dominant]
        # one launch of the dominant kernel advances (samples of a render / its launches per render) samples
        samples_per_launch = launch_samples / dk["launches_per_step"]
        ach_gbs = kernel_bytes_ps.get(dominant, bytes_ps) * samples_per_launch / (dk["avg_us"] * 1e-6) / 1e9
        # vector flops are priced over the whole launch sequence (they are spread over all kernels)
        ach_tf = flops_ps * launch_samples / (last_ms * 1e-3) / 1e12
        traffic, traffic_note = measured_traffic(args.workload, dominant)
        if args.bvh == "device":
            traffic, traffic_note = None, "the committed PMC passes were taken with the reference's tree"

        result = {
            "metric": "path samples/sec at 1080p", "value": value, "unit": "path samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if mode == "batches" else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "spp": spp, "bvh": args.bvh, "bvh_build_ms": bvh_build_ms,
                       "samples_per_step": total_samples / args.steps,
                       "sharding": "single GPU" if world == 1 else
                       {"batches": "one %d-spp batch per GPU (rank-specific seed), RCCL sum-reduce to rank 0" % spp,
                        "tiles": "512x512 tiles interleaved over GPUs, RCCL sum-reduce to rank 0",
                        "frames": "frames round robin over GPUs, no exchange"}[mode]},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dominant, "kernel_avg_us": dk["avg_us"],
                         "kernel_launches_per_step": dk["launches_per_step"],
                         "samples_per_launch": samples_per_launch,
                         "algorithmic_bytes_per_sample": kernel_bytes_ps.get(dominant, bytes_ps),
                         "traffic_note": traffic_note,
                         "note": "algorithmic bytes = 32 B per node visit + 48 B per primitive test of the reference "
                                 "algorithm for the rays this kernel traces (work counts measured on this scene); a "
                                 "scene whose tables fit LDS or cache is bound by vector ALU issue instead, see "
                                 "roofline_valu"},
            "roofline_valu": {"bound": "fp32-valu", "achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": ach_tf / VALU_PEAK_TFLOPS, "algorithmic_flops_per_sample": flops_ps,
                              "over": "all kernels of one render (%.1f ms)" % last_ms},
            "kernels": kernels,
            "work_per_sample": {k: cnt[k] / max(cnt["path_samples"], 1) for k in cnt if k != "path_samples"},
        }
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(scene, cam, W, H, 42)
            if base is not None:
                result["cpu_baseline"] = base
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
