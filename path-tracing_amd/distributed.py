"""Sharding of one render over the GPUs of a node, one process per GPU.

The reference shards by operating-system process and files: image tiles written as raw float
rows and stitched by a separate program (reference src/07/pathtracer-tiles.cpp:72-139,
tile-composer.cpp:10-37), or whole animation frames (reference src/07/pathtracer-frames.cpp:205-236).
Here every rank renders its share into a full-size float buffer in its own HBM and the buffers are
combined by ONE sum-reduce to rank 0 (RCCL over xGMI on GPUs, gloo in CPU tests):

  tiles     disjoint pixel sets, interleaved over ranks for load balance.  A pixel's value does not
            depend on which rank renders it (the generator is keyed by the pixel's index in the whole
            image), and every other rank contributes exact zeros, so the combined image equals the
            single-GPU image bit for bit.
  samples   disjoint ranges of the sample index.  Each rank writes plain sums (PT_RENDER_SUM_ONLY);
            rank 0 divides by the total sample count after the reduce.  Float addition order differs
            from a sequential sum, so results agree with the single-GPU image to rounding.
  batches   every rank renders the same pixels and a full set of samples with its own seed, the way
            independent renders are averaged; rank 0 divides by spp * world.  This is the weak-scaling
            mode of bench.py.
  frames    whole frames per rank, no combine (plan_frames); plan_frames_and_samples also splits the frames
            left over by sample range.

Nothing here renders: the caller passes a `render(params, out)` callable (the GPU library in
production, the CPU oracle in the gloo tests).
"""
from . import abi


def plan_tiles(width, height, world, rank, tile=32):
    """Tile rectangles (x0, y0, x1, y1) of this rank: tile t of the row-major tile grid goes to rank t % world."""
    rects = []
    t = 0
    for y0 in range(0, height, tile):
        for x0 in range(0, width, tile):
            if t % world == rank:
                rects.append((x0, y0, min(x0 + tile, width), min(y0 + tile, height)))
            t += 1
    return rects


def plan_samples(n_samples, world, rank):
    """Contiguous sample range [begin, end) of this rank; ranges differ in size by at most one."""
    base, extra = divmod(n_samples, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def plan_frames(n_frames, world, rank):
    """Frames of this rank, round robin (frame f goes to rank f % world)."""
    return list(range(rank, n_frames, world))


def plan_frames_and_samples(n_frames, n_samples, world, rank):
    """This rank's work for an animation of n_frames: (frame, None) for a whole frame of its own, or
    (frame, (begin, end)) for its sample range of a frame that all ranks render together.  Whole frames go round
    robin for as long as every rank gets one; the n_frames % world frames left over are split by samples, so that
    no rank idles through the last round (the reference shards by frame only, src/07/pathtracer-frames.cpp:205-207)."""
    full = n_frames - n_frames % world
    plan = [(f, None) for f in range(rank, full, world)]
    if world > 1:
        for f in range(full, n_frames):
            plan.append((f, plan_samples(n_samples, world, rank)))
    return plan


def shard_params(base, mode, world, rank, tile=32):
    """The pt_render_params of every render call this rank makes for one image."""
    def clone(**kw):
        p = abi.pt_render_params.from_buffer_copy(base)
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    if world == 1:
        return [clone()]
    if mode == "tiles":
        out = []
        for (x0, y0, x1, y1) in plan_tiles(base.width, base.height, world, rank, tile):
            out.append(clone(tile_x0=x0, tile_y0=y0, tile_x1=x1, tile_y1=y1))
        return out
    if mode == "samples":
        s0, s1 = plan_samples(base.sqrt_spp * base.sqrt_spp, world, rank)
        if s0 == s1:
            return []
        return [clone(sample_begin=s0, sample_end=s1, flags=base.flags | abi.PT_RENDER_SUM_ONLY)]
    if mode == "batches":
        return [clone(seed=base.seed + rank, flags=base.flags | abi.PT_RENDER_SUM_ONLY)]
    raise ValueError("unknown sharding mode %r" % mode)


def render_sharded(render, base, mode, out, world, rank, reduce_to_root=None, scale_root=None, tile=32):
    """Render this rank's share of one image into `out` (zero-filled by the caller) and combine.

    render(params, out)      renders one call's pixels into the full-size buffer `out`
    reduce_to_root(out)      sum-reduces the buffers of all ranks into rank 0's (None when world == 1)
    scale_root(out, divisor) divides rank 0's buffer in place
    """
    for p in shard_params(base, mode, world, rank, tile):
        render(p, out)
    if world > 1:
        reduce_to_root(out)
        if rank == 0:
            spp = base.sqrt_spp * base.sqrt_spp
            if mode == "samples":
                scale_root(out, float(spp))
            elif mode == "batches":
                scale_root(out, float(spp * world))
    return out
