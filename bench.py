#!/usr/bin/env python3
"""Benchmark of the MI355X render loop: path samples / s at 1080p.

    python bench.py --gpus N --steps K --warmup W [--workload cornell|mesh1000|textured|frames]

One step = one full render of the workload (all pixels, all samples; `frames`: a set of frames).  N = 1 renders
the Cornell box with glossy (Phong) boxes and MIS at 1920x1080, 1024 spp -- the configuration the metric is quoted
on.  With N > 1 (one process per GPU, launched by torch.distributed.run) the SAME work is split over the ranks
(strong scaling): disjoint sample ranges of the same 1024 spp (cornell, mesh), image tiles (textured), or whole
frames with the frames left over split by samples (frames); the float buffers are combined on rank 0 with one RCCL
sum-reduce inside the timed region.  Rank 0 prints one JSON line.

The line also carries
  roofline      for the kernel with the most device time, against the limit that binds the workload.  Scenes whose
                tables fit LDS (cornell, frames): the reference algorithm needs no memory traffic beyond the pixel
                contribution there, so the limit is FP32 vector issue -- useful flops of the kernel's share of the
                algorithm (30 per box test, 60 per surface test, counted by the instrumented kernel; 300 per shaded
                vertex, 60 per light test) over its duration against the FP32 vector peak.  Scenes in memory
                (mesh, textured): bytes the executed walk has to move (32 per box row, 48 per surface test, the ray
                records in and out; counted by the instrumented kernel) over the duration against the HBM peak.  Each
                beside the HBM bytes rocprofv3's counters saw for the same launches (profiles/traffic_*.json).
  roofline_trace  the same for the tracing kernel when another kernel has the most time.
  state_traffic (LDS scenes) what the wavefront design moves through HBM per path sample -- path records, hit lists,
                shadow-ray queues -- by its own byte model and as measured, against the 12 bytes per sample that the
                algorithm itself needs, and the dominant kernel priced against the HBM peak by that traffic.
  rmse          the rendered image, averaged over 20x20 pixel blocks, against a converged 96x54 image of the same
                view rendered by the reference with its own generator (tests/golden/converged_*.npz).
  cpu_baseline  the compiled reference (oracle/_ref/libpt_ref_fast.so: the reference's own pathSample / BVH /
                material code at its shipped optimisation level) timed on this machine's host cores on a
                reduced-spp render of the same scene, median of three.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # HBM3E peak, GB/s
VALU_PEAK_TFLOPS = 157.3         # FP32 vector peak of the data sheet (packed fused multiply-adds), TFLOP/s
VALU_PEAK_UNFUSED_TFLOPS = 78.6  # half of it: one plain fused multiply-add per lane where the data sheet counts a packed one
# What one-flop instructions can reach: this code is built with contraction off (bit-reproducible results), so an
# instruction is one flop per lane, and a plain vector instruction of a full wave measures 3.04 cycles per SIMD with
# every SIMD busy (tools/micro/pk_rate.hip, profiles/r03_pk_rate.txt; packed ones 5-5.9: no gain per flop):
# 256 CUs x 4 SIMDs x 64 lanes / 3.04 cycles x 2.4 GHz
VALU_MEASURED_ONE_FLOP_TFLOPS = 256 * 4 * 64 / 3.04 * 2.4e9 / 1e12

TRACE_KERNELS = ("traceKernel", "traceOrderedKernel", "traceRetryKernel")


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
    if kernel not in ks:
        return None, "PMC file has no row for " + kernel
    return ks[kernel]["hbm_bytes_per_launch"], t.get("how", "")


def cpu_baseline(scene, cam, width, height, seed, t0=0.0, t1=0.0):
    """Time the compiled reference on the host cores on a bounded sample of the workload: median of three."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refbind
    import path_tracing_amd as pt
    lib = refbind.REF_FAST_LIB if os.path.exists(refbind.REF_FAST_LIB) else refbind.REF_LIB
    if not os.path.exists(lib):
        return None
    # The cores this job may use: a one-GPU job on the box is granted a share of the host (OMP_NUM_THREADS is set to
    # it, 16 there, and wider pools are killed), not the os.cpu_count() CPUs of the whole machine; both are stated.
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(allowed, 16)
    R = refbind.Reference(lib)
    d = scene.desc()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference prints progress on stderr
    try:
        rs = R.scene(C.byref(d), t0, t1)
        # calibrate on 1 spp, then pick sqrt_spp for roughly 7 s per run
        p = pt.make_params(width, height, 1, seed=seed, t0=t0, t1=t1)
        _, t_one = rs.render(cam, p, rng_mode=1, threads=threads)
        rate = width * height / max(t_one, 1e-6)
        sqrt_spp = max(1, min(8, int((7.0 * rate / (width * height)) ** 0.5)))
        p = pt.make_params(width, height, sqrt_spp, seed=seed, t0=t0, t1=t1)
        runs = sorted(rs.render(cam, p, rng_mode=1, threads=threads)[1] for _ in range(3))
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)
    n = width * height * sqrt_spp * sqrt_spp
    return {"value": n / runs[1], "unit": "path samples/s", "cores": threads, "kind": "reference",
            "host_cpus_visible": os.cpu_count(), "host_cpus_allowed": allowed,
            "cores_note": "threads = OMP_NUM_THREADS as set for this job (its share of the host), else min(allowed CPUs, 16)",
            "runs_s": runs,
            "sample": "%dx%d at %d spp (%d path samples), median of 3 runs (%.1f s), reference's own generator, "
                      "g++ -Ofast -march=x86-64-v3 -fopenmp" % (width, height, sqrt_spp * sqrt_spp, n, runs[1])}


def block_average(img, bh, bw):
    h, w, _ = img.shape
    return img.reshape(h // bh, bh, w // bw, bw, 3).mean(axis=(1, 3))


def rmse_against_converged(img, path, spp):
    """RMSE of the block-averaged image against the converged reference image of the same view.  A pixel of the
    small image integrates over exactly the area of its block of pixels of the large one (the camera maps image
    fractions to directions), so the two estimate the same quantity; what is left is the noise of both.  Bound:
    1.5 x the RMS noise expected from the two sample counts."""
    import numpy as np
    if not path or not os.path.exists(path):
        return None
    z = np.load(path)
    ref = z["image"]
    h, w, _ = img.shape
    if h % ref.shape[0] or w % ref.shape[1]:
        return None
    bh, bw = h // ref.shape[0], w // ref.shape[1]
    small = block_average(img.astype(np.float64), bh, bw)
    rmse = float(np.sqrt(np.mean((small - ref) ** 2)))
    noise, ref_spp = float(z["noise_rmse"]), int(z["spp"])
    expected = noise * (1.0 + ref_spp / float(spp * bh * bw)) ** 0.5
    return {"value": rmse, "bound": 1.5 * expected, "within_bound": bool(rmse <= 1.5 * expected),
            "relative_to_mean": rmse / max(float(ref.mean()), 1e-12),
            "reference": "%s: reference's pathSample with its own generator, %d spp at %dx%d, noise rmse %.3g; image "
                         "averaged over %dx%d blocks" % (os.path.basename(path), ref_spp, ref.shape[1], ref.shape[0], noise, bw, bh)}


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
    ap.add_argument("--pmc-pass", action="store_true",
                    help="for a rocprofv3 --pmc run: exactly the launches of the timed steps (no warm-up is needed, no "
                         "instrumented renders, no CPU baseline), so that per-kernel counter averages are over the very "
                         "launches whose durations the normal run reports")
    ap.add_argument("--bvh", choices=["file", "device"], default="device",
                    help="device: the tree built by pt_scene_build_bvh_device (surface-area builder, the faster tree on every "
                         "workload); file: the tree the reference built, as stored in the workload's scene file (cornell, "
                         "textured: the mesh is generated and the animation rebuilds its tree per frame, so neither has one)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    # where every thread stands, should a run stop making progress: written to gpurun_out/, kept only if it trips
    import faulthandler
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    watchdog_path = os.path.join(ROOT, "gpurun_out", "bench_watchdog_rank%d.log" % rank)
    watchdog_file = open(watchdog_path, "w")
    faulthandler.dump_traceback_later(int(os.environ.get("PT_BENCH_WATCHDOG", "240")), repeat=True, exit=False,
                                      file=watchdog_file)

    import numpy as np
    import torch
    import torch.distributed as dist
    import path_tracing_amd as pt
    from path_tracing_amd import workloads
    from path_tracing_amd import distributed as shard

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

    scene, cam, label, default_sqrt, extra = workloads.load(args.workload)
    sqrt_spp = args.sqrt_spp or default_sqrt
    W, H = extra.get("width", args.width), extra.get("height", args.height)
    samples = extra.get("samples") if not args.sqrt_spp else None
    s_begin, s_end = samples if samples else (0, sqrt_spp * sqrt_spp)
    spp = s_end - s_begin
    frames = extra.get("frames")
    mode = extra.get("sharding", "samples")
    max_seg = extra.get("max_path_segments", 0)
    if args.bvh == "file" and (frames or scene.desc().n_bvh_nodes == 0):
        raise SystemExit("--bvh file: this workload's scene has no stored tree for its frames (use --bvh device)")
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
    out8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") if frames else None
    stream = torch.cuda.current_stream()
    # a range of the strata (512 of 23 x 23): the library's mean is over all sqrt_spp^2 strata, so take plain sums and
    # divide by the number of samples rendered
    partial = samples is not None and spp != sqrt_spp * sqrt_spp
    params = pt.make_params(W, H, sqrt_spp, seed=42, samples=samples, max_path_segments=max_seg,
                            flags=pt.abi.PT_RENDER_SUM_ONLY if partial else 0)
    events = []

    def render(p, buf):
        dev.render_device(cam, p, buf.data_ptr(), stream.cuda_stream)

    def reduce_to_root(buf):
        if rehearsal:
            # gloo, ranks sharing one GPU: reduce a host copy
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        else:
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)

    def frame_tree(t0f, t1f):
        dev.build_bvh_device(t0f, t1f, stream.cuda_stream)

    def step(timed=False):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        if frames:
            # whole frames round robin, no exchange; the frames left over when the count is no multiple of the
            # ranks are rendered by all ranks together, each its own sample range, and summed on rank 0
            for f, srange in shard.plan_frames_and_samples(len(frames), spp, world, rank):
                t0f, t1f = frames[f]
                frame_tree(t0f, t1f)
                if srange is None:
                    render(pt.make_params(W, H, sqrt_spp, seed=42 + f, t0=t0f, t1=t1f, samples=samples,
                                          flags=pt.abi.PT_RENDER_SUM_ONLY if partial else 0), out)
                    if partial:
                        out.div_(float(spp))
                    # the frame as the reference's animation program stores it: 8-bit sRGB, top row first
                    # (src/07/pathtracer-frames.cpp:235-236 -> color.hpp:84-98, imgsave.hpp:29-33), on the device
                    pt.image_to_8bit_device(out.data_ptr(), W, H, out8.data_ptr(), device=local_rank,
                                            stream_ptr=stream.cuda_stream)
                else:
                    b, e = s_begin + srange[0], s_begin + srange[1]
                    if e > b:
                        render(pt.make_params(W, H, sqrt_spp, seed=42 + f, t0=t0f, t1=t1f, samples=(b, e),
                                              flags=pt.abi.PT_RENDER_SUM_ONLY), out)
                    else:
                        out.zero_()
                    reduce_to_root(out)
                    if rank == 0:
                        out.div_(float(spp))
        elif world == 1:
            render(params, out)
            if partial:
                out.div_(float(spp))
        else:
            # this rank's share (its sample range, or its tiles), then one sum-reduce to rank 0
            if mode == "tiles":
                out.zero_()
            for p in shard.shard_params(params, mode, world, rank, tile=256):
                render(p, out)
        if timed:
            e1.record(stream)
            events.append((e0, e1))
        if world > 1 and not frames:
            reduce_to_root(out)
            if rank == 0 and (mode == "samples" or partial):
                out.div_(float(spp))

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
    # duration of the whole launch sequence of one step on this rank: HIP events on the launch stream
    last_ms = sum(e0.elapsed_time(e1) for e0, e1 in events) / len(events)
    kernel_times = dev.kernel_times() if rank == 0 else {}
    dev.profile(False)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if args.pmc_pass:
        if rank == 0:
            print(json.dumps({"pmc_pass": True, "workload": args.workload, "steps": args.steps, "ms_per_step": elapsed / args.steps * 1e3,
                              "kernels": {k: v[0] / args.steps for k, v in kernel_times.items() if v[0]}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()
        watchdog_file.close()
        if os.path.getsize(watchdog_path) == 0:
            os.remove(watchdog_path)
        return
    if rank == 0:
        n_renders = len(frames) if frames else 1
        total_samples = float(W) * H * spp * n_renders * args.steps
        value = total_samples / elapsed
        image = out.cpu().numpy()
        # what fraction of the step's samples this rank's kernels advanced
        if frames:
            plan = shard.plan_frames_and_samples(len(frames), spp, world, 0)
            rank_samples = float(W) * H * sum(spp if sr is None else sr[1] - sr[0] for _, sr in plan)
        else:
            rank_samples = float(W) * H * spp / world

        # work counts of the reference algorithm on this scene (instrumented megakernel, reduced size) and of the
        # walk the tracing kernels execute (instrumented tracing kernels, same reduced render)
        tf0, tf1 = frames[0] if frames else (0.0, 0.0)
        if frames:
            frame_tree(tf0, tf1)
        cparams = pt.make_params(W // 4, H // 4, min(sqrt_spp, 4), seed=42, t0=tf0, t1=tf1, max_path_segments=max_seg)
        _, cnt = dev.render_counted(cam, cparams)
        cnt = cnt.as_dict()
        n_s = float(max(cnt["path_samples"], 1))
        dev.count_walk(True)
        dev.render(cam, cparams)
        walk = dev.walk_counters()
        dev.count_walk(False)
        ref_per_sample = {k: cnt[k] / n_s for k in cnt if k != "path_samples"}
        walk_per_sample = {k: walk[k] / n_s for k in walk}

        # per launch of the tracing step: bytes the executed walk has to move, and its vector flops
        row_boxes = 2 if os.environ.get("PT_NO_QUAD_ROWS") else 4  # boxes per row of the walk for scenes in memory
        first_pass_rays = walk_per_sample["path_rays"] + walk_per_sample["shadow_rays"]
        per_sample = {
            "traceKernel": {
                "bytes": 32 * walk_per_sample["box_rows"] + 48 * walk_per_sample["surface_tests"]
                         + 49 * walk_per_sample["path_rays"] + 64 * walk_per_sample["shadow_rays"],
                "flops": 30 * walk_per_sample["box_rows"] + 60 * walk_per_sample["surface_tests"]},
            "traceOrderedKernel": {
                # a row of the nearer-child-first walk holds row_boxes boxes of 12 bytes and their 4-byte links (a slot
                # of a row may be empty: counted all the same); a leaf that is no static triangle reads its node (32)
                "bytes": 16 * row_boxes * walk_per_sample["box_rows"] + 32 * walk_per_sample["leaf_rows"]
                         + 48 * walk_per_sample["surface_tests"]
                         + 49 * walk_per_sample["path_rays"] + 64 * walk_per_sample["shadow_rays"],
                # per box a test (30) and 12 flops of decoding
                "flops": 42 * row_boxes * walk_per_sample["box_rows"] + 30 * walk_per_sample["leaf_rows"]
                         + 60 * walk_per_sample["surface_tests"]},
            "traceRetryKernel": {
                "bytes": 32 * walk_per_sample["retry_node_visits"] + 48 * walk_per_sample["retry_surface_tests"]
                         + 57 * walk_per_sample["retry_rays"],
                "flops": 30 * walk_per_sample["retry_node_visits"] + 60 * walk_per_sample["retry_surface_tests"]},
            "shadeHitsKernel": {
                # path state the kernel gathers and scatters per shaded vertex: its place in the hit list (4 bytes), the
                # record in (four 16-byte rows and the 8-byte key: 72) and, for a path that goes on, out (56), the
                # surface's attributes and plane normal or vertices (32 to 64: 48 taken), a shadow-queue entry (48).
                # (The rows of the lights are the same for every lane: scalar loads that stay in cache.)
                "bytes": ref_per_sample["vertices"] * (4 + 72 + 56 + 48 + 48),
                "flops": 300 * ref_per_sample["vertices"] + 60 * ref_per_sample["light_tests"]},
        }
        n_surf = scene.desc().n_surfaces
        table_bytes = (2 * (2 * n_surf - 1) + 3 * n_surf) * 16
        in_lds = kernel_times.get("traceKernel", (0, 0))[0] > 0 and kernel_times.get("traceOrderedKernel", (0, 0))[0] == 0 \
            and table_bytes <= 16384
        if in_lds:
            # node and surface rows are read from the copy in LDS: only the ray records cross the memory system
            per_sample["traceKernel"]["lds_bytes"] = 32 * walk_per_sample["box_rows"] + 48 * walk_per_sample["surface_tests"]
            per_sample["traceKernel"]["bytes"] -= per_sample["traceKernel"]["lds_bytes"]
        kernels = {}
        for name, (launches, total_ms) in kernel_times.items():
            if not launches:
                continue
            k = {"launches_per_step": launches / args.steps, "avg_us": total_ms / launches * 1e3,
                 "ms_per_step": total_ms / args.steps}
            if name in per_sample:
                samples_per_launch = rank_samples * args.steps / launches
                secs = total_ms / launches * 1e-3
                k["samples_per_launch"] = samples_per_launch
                k["algorithmic_GBps"] = per_sample[name]["bytes"] * samples_per_launch / secs / 1e9
                k["algorithmic_TFLOPs"] = per_sample[name]["flops"] * samples_per_launch / secs / 1e12
            traffic = measured_traffic(args.workload, name)[0] if args.bvh == "device" else None
            k["hbm_bytes_per_launch"] = traffic
            if traffic:
                k["hbm_GBps"] = traffic / (total_ms / launches * 1e-3) / 1e9
            kernels[name] = k
        dominant = max(kernels, key=lambda n: kernels[n]["ms_per_step"])
        traffic_note = measured_traffic(args.workload, dominant)[1]
        if args.bvh != "device":
            traffic_note = "the committed PMC passes were taken with the device-built tree"

        def valu_line(name):
            """A kernel against the FP32 vector peak: useful flops of the reference algorithm's work (30 per box test,
            60 per surface test of the executed walk; 300 per shaded vertex, 60 per light test) over its duration."""
            k = kernels[name]
            ach = k.get("algorithmic_TFLOPs", 0.0)
            return {"kernel": name, "kernel_avg_us": k["avg_us"], "kernel_launches_per_step": k["launches_per_step"],
                    "samples_per_launch": k.get("samples_per_launch"), "bound": "fp32-valu", "achieved": ach,
                    "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / VALU_PEAK_TFLOPS,
                    "peak_unfused": VALU_PEAK_UNFUSED_TFLOPS, "frac_unfused": ach / VALU_PEAK_UNFUSED_TFLOPS,
                    "measured_issue_limit_one_flop": VALU_MEASURED_ONE_FLOP_TFLOPS,
                    "frac_of_measured_issue_limit": ach / VALU_MEASURED_ONE_FLOP_TFLOPS,
                    "flops_per_sample": per_sample.get(name, {}).get("flops"),
                    "traffic": k.get("hbm_bytes_per_launch"), "traffic_note": traffic_note}

        def hbm_line(name, what):
            k = kernels[name]
            ach = k.get("algorithmic_GBps", 0.0)
            alg = per_sample.get(name, {"bytes": 0})["bytes"] * k.get("samples_per_launch", 0)
            t = k.get("hbm_bytes_per_launch")
            return {"kernel": name, "kernel_avg_us": k["avg_us"], "kernel_launches_per_step": k["launches_per_step"],
                    "samples_per_launch": k.get("samples_per_launch"), "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "bytes_per_sample": per_sample.get(name, {}).get("bytes"),
                    "algorithmic_bytes_per_launch": alg, "traffic": t, "traffic_note": traffic_note,
                    "traffic_GBps": k.get("hbm_GBps"), "traffic_frac": (k["hbm_GBps"] / HBM_PEAK_GBS) if k.get("hbm_GBps") else None,
                    "traffic_over_algorithmic": (t / alg) if t and alg else None, "note": what}

        state_traffic = None
        if in_lds:
            # The scene's tables (%d bytes) are in LDS: the reference algorithm needs no memory traffic per sample beyond
            # the pixel contribution, and what binds is vector issue.  The headline is therefore the dominant kernel
            # against the FP32 vector peak; what the kernels do move -- the path state of the wavefront design -- is
            # reported under its own name.
            roofline = valu_line(dominant)
            roofline["note"] = ("scene tables (%d bytes) are held in LDS; useful flops of the reference algorithm's work over the "
                                "kernel's duration; contraction is off for bit-reproducible results, so an instruction is one "
                                "flop per lane and the issue limit of this code is the measured %.1f TFLOP/s "
                                "(frac_of_measured_issue_limit); `traffic` is path-state traffic, see state_traffic"
                                % (table_bytes, VALU_MEASURED_ONE_FLOP_TFLOPS))
            pmc_step = sum(k["hbm_bytes_per_launch"] * k["launches_per_step"] for k in kernels.values() if k.get("hbm_bytes_per_launch"))
            model_step = sum(per_sample[n]["bytes"] * kernels[n]["samples_per_launch"] * kernels[n]["launches_per_step"]
                             for n in kernels if n in per_sample)
            state_traffic = {
                "what": "path records, hit lists and shadow-ray queues of the wavefront design that cross HBM per path sample, "
                        "all kernels of a step; the reference algorithm on an LDS-resident scene needs 12 bytes per sample",
                "algorithmic_hbm_bytes_per_sample": 12,
                "model_bytes_per_sample": model_step / rank_samples,
                "measured_bytes_per_sample": (pmc_step / rank_samples) if pmc_step else None,
                "measured_over_algorithmic": (pmc_step / rank_samples / 12.0) if pmc_step else None,
                "measured_GBps": (pmc_step / (last_ms * 1e-3) / 1e9) if pmc_step else None,
                "measured_frac_of_hbm_peak": (pmc_step / (last_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc_step else None,
                "dominant_kernel": hbm_line(dominant, "path state the kernel has to move by the design's own byte model / duration"),
            }
        else:
            roofline = hbm_line(dominant, "achieved = bytes the executed walk has to move (64 per row of four boxes, 48 per surface test, ray "
                                          "records in and out; counts from the instrumented kernel) / kernel duration; `traffic` = "
                                          "HBM bytes per launch from rocprofv3 counters (profiles/); where traffic is a fraction of "
                                          "the algorithmic bytes the rows come out of L2 and the Infinity Cache, frac can pass 1, "
                                          "and the limit that binds is vector issue: roofline_valu")
        # the tracing kernel against its own limit, whichever kernel has the most time
        trace_name = max((n for n in kernels if n in ("traceKernel", "traceOrderedKernel")), key=lambda n: kernels[n]["ms_per_step"],
                         default=None)
        roofline_trace = None
        if trace_name and trace_name != dominant:
            roofline_trace = valu_line(trace_name) if in_lds else hbm_line(trace_name, "as `roofline`, for the tracing kernel")
        # a scene in memory whose rows are mostly served by the caches: the dominant kernel against vector issue as well
        roofline_valu = valu_line(dominant) if not in_lds else None
        # all kernels of one step together against both limits
        total_flops = sum(per_sample[n]["flops"] * kernels[n]["samples_per_launch"] * kernels[n]["launches_per_step"]
                          for n in kernels if n in per_sample)
        total_bytes = sum(per_sample[n]["bytes"] * kernels[n]["samples_per_launch"] * kernels[n]["launches_per_step"]
                          for n in kernels if n in per_sample)
        whole = {"ms": last_ms,
                 "valu_TFLOPs": total_flops / (last_ms * 1e-3) / 1e12, "valu_frac": total_flops / (last_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                 "valu_frac_unfused": total_flops / (last_ms * 1e-3) / 1e12 / VALU_PEAK_UNFUSED_TFLOPS,
                 "algorithmic_GBps": total_bytes / (last_ms * 1e-3) / 1e9, "hbm_frac": total_bytes / (last_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

        shard_text = {"samples": "disjoint sample ranges of the same %d spp per GPU, one sum-reduce to rank 0" % spp,
                      "tiles": "256x256 tiles interleaved over GPUs, one sum-reduce to rank 0",
                      "frames": "frames round robin over GPUs, no exchange; frames left over split by sample range "
                                "and sum-reduced"}[mode if not frames else "frames"]
        result = {
            "metric": "path samples/sec at 1080p", "value": value, "unit": "path samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": W, "height": H, "spp": spp, "bvh": args.bvh, "bvh_build_ms": bvh_build_ms,
                       "samples_per_step": total_samples / args.steps,
                       "sharding": "single GPU" if world == 1 else shard_text},
            # rays of the reference algorithm settled per second, and those of them that a tracing kernel walks: camera
            # rays that miss the box of the whole scene and shadow rays that can only add zeros are settled without
            "rays_per_s": value * ref_per_sample["rays"],
            "traced_rays_per_s": value * first_pass_rays,
            # (an animation step on several GPUs leaves another frame in rank 0's buffer than the converged one)
            "rmse": rmse_against_converged(image, extra.get("converged"), spp)
                    if not args.sqrt_spp and not (frames and world > 1) else None,
            "roofline": roofline,
            "roofline_trace": roofline_trace,
            "roofline_valu": roofline_valu,
            "state_traffic": state_traffic,
            "whole_step": whole,
            "kernels": kernels,
            "work_per_sample": ref_per_sample,
            "executed_walk_per_sample": walk_per_sample,
        }
        if not args.no_cpu_baseline and world == 1:
            base = cpu_baseline(scene, cam, W, H, 42, tf0, tf1)
            if base is not None:
                result["cpu_baseline"] = base
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    watchdog_file.close()
    if os.path.getsize(watchdog_path) == 0:
        os.remove(watchdog_path)
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
