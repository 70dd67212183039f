"""The multi-GPU entry points of the C ABI against the REAL RCCL on the one GPU of the test box: a world of one rank. What this can
prove on a single device: librccl is found and bound at run time, ncclGetUniqueId / ncclCommInitRank work from inside libvqhip.so, the edge
rank needs no halo and the composite of a one-tile frame is the tile itself (device copy on the caller's stream, then visible after a stream
sync). Two ranks on one device are refused by RCCL ("duplicate GPU"); the exchange logic itself is covered by tests/test_mgpu_mock.py
(shared-memory mock) and tests/test_gpu_bench_flow.py (mock + real kernels); RCCL over xGMI is the driver's 8-GPU run."""
import numpy as np
import pytest
import torch

from vqengine_amd import abi, capi

pytestmark = pytest.mark.gpu


def test_world_of_one_through_real_rccl(ctx):
    uid = capi.comm_unique_id()
    assert len(uid) == capi.COMM_ID_BYTES and any(uid)
    comm = capi.Comm(uid, 1, 0)
    try:
        H, W = 64, 96
        x = torch.rand((H, W, 4), device="cuda").to(torch.float16)
        comm.exchange_blur_halos(x, abi.FMT_RGBA16F, None, None)                       # rank 0 of 1: both edges are frame borders
        tile = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda")
        frame = torch.zeros_like(tile)
        comm.composite_tiles(tile, abi.FMT_RGBA8_UNORM, H, 0, frame)
        comm.composite_tiles(tile, abi.FMT_RGBA8_UNORM, H, capi.ALL_RANKS, frame)
        torch.cuda.synchronize()
        assert torch.equal(frame, tile)
        with pytest.raises(capi.VQHipError):
            comm.composite_tiles(tile, abi.FMT_RGBA8_UNORM, H, 3, frame)               # root outside the world
    finally:
        comm.close()
    assert np.frombuffer(uid, np.uint8).size == 128
