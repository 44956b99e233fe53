"""Multi-GPU host logic: one process per GPU, no collective inside a batch, two ways to split a batch.

"tiles"   (`render_partitioned`): the frame is row-interleaved across ranks; bit-identical to the single-GPU frame.
"batches" (`render_batches`): every rank renders the WHOLE frame with spp / world samples and its own Seed, from zeroed
          accumulators; the partial accumulators are folded in rank order - slice by slice on all ranks (one all-to-all) - and
          the frame is gathered on rank 0.  This is the reference's own
          notion of accumulation - successive batches with fresh seeds summed into the same buffers
          (Assets/Scripts/Unity/Raytracer.cs:656-661,798-802) - run concurrently instead of back to back.

Why both: the reference RNG stream makes a pixel's samples sequential (one lane per pixel), so with tiles a GPU that owns
about one pixel per resident lane (1080p over 8 GPUs) finishes only when its slowest pixel does: measured 1.5x / 2.5x /
3.2x at 2 / 4 / 8 slices of the 1080p cover scene.  Splitting the SAMPLES keeps ~8 pixels per lane on every GPU.

Pixels are independent given (params, scene, Seed, global pixel index) - the per-pixel RNG seed is a function of the
GLOBAL index (JOBS/SampleBatchJob.cs:91) - so any partition reproduces the single-GPU frame bit for bit.  The partition
is the reference's own slice contract: rank g of G renders the rows with row % G == g (SliceDivider = G,
SliceOffset = g, JOBS/SampleBatchJob.cs:69-70).  There is no exchange inside a batch; after it, each rank's rows are
collected on rank 0 with ONE collective (torch.distributed.gather: RCCL over xGMI with backend "nccl", gloo on CPU).

The functions take the process group explicitly and never touch the GPU themselves, so the same code path is covered
by world_size-2 gloo tests on CPU (tests/test_multigpu_gloo.py) with the CPU checker standing in for the kernel.
"""
import torch
import torch.distributed as dist


def owned_rows(rank, world, height):
    """Rows of the frame rank `rank` renders: row % world == rank."""
    return range(rank, height, world)


def slice_params(params, rank, world):
    """Set the reference's slice fields on an RtowSampleParams (in place) for this rank."""
    params.sliceOffset = rank
    params.sliceDivider = world
    return params


def pack_owned(buffer_hw, rank, world):
    """[H, W, C] accumulator -> contiguous [rows_owned, W, C] block of this rank's rows."""
    return buffer_hw[rank::world].contiguous()


def gather_frame(mine, height, rank, world, group=None, dst=0):
    """Collect every rank's packed rows on `dst` and interleave them back into a [H, W, C] frame (None elsewhere).

    A gather needs equal-sized blocks; when `height` is not a multiple of `world` the ranks with one row less pad their
    block with one dummy row, which `dst` drops again."""
    if world == 1:
        return mine
    max_rows = (height + world - 1) // world
    block = mine
    if mine.shape[0] < max_rows:
        block = torch.cat([mine, mine.new_zeros((max_rows - mine.shape[0],) + tuple(mine.shape[1:]))], dim=0)
    gather_list = None
    if rank == dst:
        gather_list = [torch.empty_like(block) for _ in range(world)]
    dist.gather(block, gather_list, dst=dst, group=group)
    if rank != dst:
        return None
    frame = torch.empty((height,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    for r in range(world):
        frame[r::world] = gather_list[r][: len(owned_rows(r, world, height))]
    return frame


def render_partitioned(render_slice, params, height, width, rank, world, group=None):
    """Run one sample batch on this rank's slice and gather the colour buffer.

    `render_slice(params)` must return the rank's full-frame-sized colour accumulator as a [H*W, 4] torch tensor in
    which only the owned rows are meaningful (skipped pixels write nothing).  Returns the assembled [H, W, 4] frame on
    rank 0, None elsewhere.
    """
    slice_params(params, rank, world)
    color = render_slice(params)
    mine = pack_owned(color.view(height, width, 4), rank, world)
    return gather_frame(mine, height, rank, world, group)


# ---------------------------------------------------------------------------------------------------
# batch-parallel: split the samples of a batch, not the pixels
# ---------------------------------------------------------------------------------------------------
ACCUM_FLOATS = 11  # float4 colour + float3 normal + float3 albedo + float sample-count weight per pixel


def accum_views(flat, n):
    """The four accumulation buffers as views of ONE contiguous [11 * n] tensor (so a rank's partial result travels in a
    single collective): colour [n,4] | normal [n,3] | albedo [n,3] | sampleCountWeight [n]."""
    return [flat[0:4 * n].view(n, 4), flat[4 * n:7 * n].view(n, 3), flat[7 * n:10 * n].view(n, 3), flat[10 * n:11 * n]]


def batch_split(spp, rank, world):
    """Samples per pixel rank `rank` takes of a `spp`-sample batch (remainder to the low ranks)."""
    return spp // world + (1 if rank < spp % world else 0)


def batch_seed(seed, rank, world):
    """Seed of rank `rank`'s sub-batch: batches of one step get consecutive seeds, like consecutive frames do in the host
    (frameSeed = Time.frameCount + 1, UNITY/Raytracer.cs:660)."""
    return (seed - 1) * world + rank + 1


def slice_floats(n, world):
    """Length of one rank's slice of the flat accumulator: 11 floats x ceil(n / world) pixels' worth, so that a slice can be
    added with the 4-buffer device add (rtowAddAccumDevice) however it straddles the colour | normal | albedo | weight sections."""
    return ACCUM_FLOATS * ((n + world - 1) // world)


def padded_floats(n, world):
    """Flat accumulator length that splits into `world` equal slices (>= 11 * n; the tail is padding)."""
    return world * slice_floats(n, world)


def fold_partials(acc_flat, partials, add_flat):
    """acc += partial_0, acc += partial_1, ... in RANK ORDER (deterministic, so the result can be reproduced bit for bit)."""
    for part in partials:
        add_flat(acc_flat, part)
    return acc_flat


def render_batches(render_full, acc_slice, n, rank, world, add_flat, group=None, dst=0, exchange=None, frame=None, via_host=False):
    """One sample batch, batch-parallel, with the fold spread over the ranks.

    `render_full()` returns this rank's partial accumulators (from ZERO inputs, its own seed and sample share) as one flat
    [padded_floats(n, world)] tensor.  The running accumulation is kept DISTRIBUTED: rank r owns floats [r*m, (r+1)*m) of it
    (`acc_slice`, m = slice_floats).  Per batch:
      1. all_to_all: rank r receives slice r of every rank's partial   (each rank sends and receives (world-1)/world of 44 B/pixel,
         spread over all its xGMI links - instead of rank 0 swallowing world-1 whole partials over its own links);
      2. acc_slice += partial_0[r], += partial_1[r], ...  in rank order - element for element the same float adds, in the same
         order, as folding whole partials on one rank, so the frame is bit-identical to that;
      3. gather of the accumulated slices on `dst`: the frame of this batch (1/world of the frame per peer).
    Returns the flat frame [padded_floats] on `dst`, None elsewhere.  `exchange` / `frame` are optional preallocated buffers.
    `via_host`: development only (ranks sharing one GPU over gloo, which has no device all-to-all): the two collectives run on host copies."""
    part = render_full()
    m = slice_floats(n, world)
    if world == 1:
        return fold_partials(acc_slice, [part], add_flat)
    if exchange is None:
        exchange = torch.empty_like(part)
    if via_host:
        host_out = torch.empty(part.shape, dtype=part.dtype)
        dist.all_to_all_single(host_out, part.cpu(), group=group)
        exchange.copy_(host_out)
    else:
        dist.all_to_all_single(exchange, part, group=group)
    fold_partials(acc_slice, [exchange[j * m:(j + 1) * m] for j in range(world)], add_flat)
    if rank == dst and frame is None:
        frame = torch.empty_like(part)
    if via_host:
        host_list = [torch.empty(m, dtype=part.dtype) for _ in range(world)] if rank == dst else None
        dist.gather(acc_slice.cpu(), host_list, dst=dst, group=group)
        if rank == dst:
            frame.view(world, m).copy_(torch.stack(host_list))
        return frame if rank == dst else None
    gather_list = list(frame.view(world, m).unbind(0)) if rank == dst else None
    dist.gather(acc_slice, gather_list, dst=dst, group=group)
    return frame if rank == dst else None


# ---------------------------------------------------------------------------------------------------
# tiles x batches (rtowHybridPlan / rtowExchangeAccumDevice, include/rtow.h): G ranks = T row slices x B seed groups
# ---------------------------------------------------------------------------------------------------
def hybrid_plan(world, rank, tiles, samples_per_batch, step):
    """The arithmetic of rtowHybridPlan (csrc/rtow_api.hip), for the hosts that run without the library (CPU tests): rank = tile + T * group renders
    slice `tile` of T with its share of the batch's samples and the Seed of sub-batch `group` of step `step` (1-based)."""
    if world < 1 or not 0 <= rank < world or tiles < 1 or world % tiles or step < 1:
        raise ValueError("hybrid_plan: world = tiles x groups, 0 <= rank < world, step >= 1")
    groups = world // tiles
    tile, group = rank % tiles, rank // tiles
    return {"tiles": tiles, "groups": groups, "tile": tile, "group": group, "slice_offset": tile, "slice_divider": tiles,
            "samples": samples_per_batch // groups + (1 if group < samples_per_batch % groups else 0), "seed": (step - 1) * groups + group + 1}


def default_tiles(world, samples_per_batch):
    """T of the partition bench.py reports: as many seed groups as there are samples to give each at least one (B = the largest divisor of the
    world size that is <= samples per batch), the rest of the world size as row slices.  256 samples on 8 GPUs: 1 x 8."""
    groups = max(b for b in range(1, world + 1) if world % b == 0 and b <= max(1, samples_per_batch))
    return world // groups


def exchange_accum(partial, accum, height, rank, world, tiles, group=None):
    """torch.distributed mirror of rtowExchangeAccumDevice (same rows, same order of additions; gloo on CPU, RCCL with backend "nccl").
    partial / accum: lists of [H, W, C] tensors (full-frame buffers).  Rank p folds the rows with row % world == p; the ranks of its tile
    (p % tiles) send it those rows of their partial sums, and it adds them to `accum` in group order, its own partial in place."""
    groups, tile, own = world // tiles, rank % tiles, rank // tiles
    recv = {}
    ops = []
    for g in range(groups):
        peer = tile + tiles * g
        if g == own:
            continue
        for k, buf in enumerate(partial):
            rows = buf[peer::world].contiguous()
            if rows.numel():
                ops.append(dist.isend(rows, dst=peer, group=group))
            mine = torch.empty_like(buf[rank::world])
            if mine.numel():
                ops.append(dist.irecv(mine, src=peer, group=group))
            recv[(g, k)] = mine
    for op in ops:
        op.wait()
    for k, (acc, part) in enumerate(zip(accum, partial)):
        rows = acc[rank::world]                              # a view: the adds below land in `accum`
        for g in range(groups):
            rows += part[rank::world] if g == own else recv[(g, k)]
    return accum


def render_hybrid(render_partial, accum, height, width, rank, world, tiles, samples_per_batch, step, group=None, dst=0):
    """One step of the tiles x batches partition over torch.distributed: plan -> render_partial(plan) -> exchange + fold -> gather of colour.
    `render_partial(plan)` returns this rank's partial sums from ZEROED inputs as [color [H*W,4], normal, albedo, scw [H*W]] tensors in which only
    the rows of its tile are meaningful.  `accum`: the running accumulators, same shapes (rows with row % world == rank are this rank's).
    Returns the colour frame [H, W, 4] on `dst` (None elsewhere)."""
    plan = hybrid_plan(world, rank, tiles, samples_per_batch, step)
    part = render_partial(plan)
    shapes = (4, 3, 3, 1)
    exchange_accum([t.view(height, width, c) for t, c in zip(part, shapes)], [t.view(height, width, c) for t, c in zip(accum, shapes)], height, rank, world, tiles, group)
    return gather_frame(pack_owned(accum[0].view(height, width, 4), rank, world), height, rank, world, group, dst)
