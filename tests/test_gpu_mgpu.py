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


@pytest.mark.parametrize("nbytes", [1 << 10, 10 * 7680 * 8, 33 << 20])
def test_loopback_send_recv_through_real_rccl(ctx, nbytes):
    """ncclSend + ncclRecv of the bound RCCL, grouped and addressed to the own rank (vqhip_comm_loopback): the point-to-point entry points the
    halo exchange and the composite use — here with the message sizes of a cfg5 halo (10 rows of 7680 RGBA16F pixels) and of a 4K RGBA8 tile —
    run from inside libvqhip.so on a side stream, and the bytes arrive."""
    comm = capi.Comm(capi.comm_unique_id(), 1, 0)
    try:
        q = comm.query()
        assert q["nranks_seen"] == 1 and q["rank_seen"] == 0 and q["version"] > 0 and "rccl" in q["library_path"].lower()
        src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda")
        dst = torch.zeros_like(src)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        for _ in range(3):                                                             # back-to-back groups on one stream
            comm.loopback(src, dst, stream=s.cuda_stream)
        s.synchronize()
        assert torch.equal(src, dst)
        with pytest.raises(capi.VQHipError):
            comm.loopback(src, None)
    finally:
        comm.close()
