"""Multi-GPU host logic: one process per GPU, the frame row-interleaved across ranks, one gather per sample batch.

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
