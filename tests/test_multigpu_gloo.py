"""world_size-2 test of the multi-GPU host path on CPU (gloo): row-interleaved slices + one gather == the full frame.

The partition / pack / gather / interleave code is the product's (raytracing-in-one-weekend_amd/multigpu.py, the same
functions bench.py calls under RCCL); the CPU checker stands in for the kernel because the HIP path needs a GPU."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH = 48, 27, 2, 6   # odd height: ranks own a different number of rows


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    import importlib
    sys.path.insert(0, ROOT)
    rt = importlib.import_module("raytracing-in-one-weekend_amd")
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = rt.scenes.cover_scene()
    osc = ob.OracleScene(scene.desc())
    params = rt.scenes.make_params(scene, W, H, spp=SPP, trace_depth=DEPTH)

    def render_slice(p):
        assert (p.sliceOffset, p.sliceDivider) == (rank, world)
        ins = ob.zero_buffers(W * H)
        ins["color"][:] = -5.0                      # pixels this rank does not own must stay untouched
        r = osc.sample_batch(p, ins, nthreads=2)
        rows = np.repeat(np.arange(H) % world == rank, W)
        assert np.all(r["color"][~rows] == -5.0)
        r["color"][rows] -= 0.0
        return torch.from_numpy(r["color"])

    # inputs are -5 everywhere, so owned pixels hold (-5 + sum); undo the offset after the gather
    frame = mg.render_partitioned(render_slice, params, H, W, rank, world)
    if rank == 0:
        assert frame.shape == (H, W, 4)
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    assert list(mg.owned_rows(rank, world, H)) == list(range(rank, H, world))
    dist.barrier()
    dist.destroy_process_group()
    osc.close()


def test_two_rank_partition_and_gather_reproduces_the_full_frame(rt, oracle, tmp_path):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    frame = np.load(out).reshape(W * H, 4)
    scene = rt.scenes.cover_scene()
    osc = oracle.OracleScene(scene.desc())
    ins = oracle.zero_buffers(W * H)
    ins["color"][:] = -5.0
    full = osc.sample_batch(rt.scenes.make_params(scene, W, H, spp=SPP, trace_depth=DEPTH), ins)
    osc.close()
    assert np.array_equal(frame.view(np.uint32), full["color"].view(np.uint32))   # bit-identical to the single-rank frame


def test_pack_and_gather_single_rank_identity(rt):
    import importlib
    mg = importlib.import_module("raytracing-in-one-weekend_amd.multigpu")
    x = torch.arange(5 * 3 * 4, dtype=torch.float32).view(5, 3, 4)
    assert torch.equal(mg.gather_frame(mg.pack_owned(x, 0, 1), 5, 0, 1), x)
    assert [list(mg.owned_rows(r, 3, 7)) for r in range(3)] == [[0, 3, 6], [1, 4], [2, 5]]
    p = rt.abi.SampleParams()
    mg.slice_params(p, 2, 8)
    assert (p.sliceOffset, p.sliceDivider) == (2, 8)
