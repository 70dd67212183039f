"""Row-tiled multi-GPU mode on CPU: world_size 2 and 3 over the gloo backend (no GPU). The exchange logic in
tests/gloo_tiling.py (blur halo via P2P or one all-gather, composite all-gather; the independent second statement of the tiling) is exercised with the ORACLE doing
the per-tile compute, and the composited frame must equal the single-process full-frame result bit-for-bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import gloo_tiling
from vqengine_amd import abi, synth, tiling

W, TILE_H = 96, 24


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_frame(world):
    from tests import oracle_lib as O
    H = TILE_H * world
    gb = synth.gbuffer(W, H, seed=0xD157)
    pf, _ = synth.per_frame(points=synth.point_lights(12, seed=0xD157))
    pv = synth.per_view(W, H)
    scene = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F)
    x = O.blur_pass(scene, abi.FMT_RGBA16F, 0)
    y = O.blur_pass(x, abi.FMT_RGBA16F, 1)
    return O.tonemap(y, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)


def _worker(rank, world, port, halo_mode, q, comp="allgather"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import oracle_lib as O
        H = TILE_H * world
        tl = tiling.RowTiling(W, H, world, rank)
        gb = synth.gbuffer_rows(W, H, tl.row0, tl.row1, seed=0xD157)          # this rank's tile only
        pf, _ = synth.per_frame(points=synth.point_lights(12, seed=0xD157))    # lights replicated
        pv = synth.per_view(W, H)
        scene = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F)                # no exchange: per-pixel
        x = O.blur_pass(scene, abi.FMT_RGBA16F, 0)                             # no exchange: X pass
        xt = torch.from_numpy(x)
        fn = gloo_tiling.exchange_halos_p2p if halo_mode == "p2p" else gloo_tiling.exchange_halos_allgather
        top, bottom = fn(xt)                                                   # exchange 1: 10-row halos
        assert (top is None) == (rank == 0) and (bottom is None) == (rank == world - 1)
        y = O.blur_pass(x, abi.FMT_RGBA16F, 1, halo_top=top.numpy() if top is not None else None,
                        halo_bottom=bottom.numpy() if bottom is not None else None)
        sdr = torch.from_numpy(O.tonemap(y, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM))
        if comp == "gather":                                                   # exchange 2: composite on rank 0 only
            frame, work = gloo_tiling.composite_to_root(sdr, dst=0, async_op=True)
            work.wait()
            assert (frame is None) == (rank != 0)
            q.put((rank, frame.numpy().copy() if frame is not None else None))
        else:                                                                  # exchange 2: all-gather composite
            frame, work = gloo_tiling.composite(sdr, async_op=True)
            work.wait()
            q.put((rank, frame.numpy().copy()))
        dist.barrier()
    except Exception as e:                                                     # surface failures immediately instead of a queue timeout
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,halo_mode,comp", [(2, "p2p", "allgather"), (2, "allgather", "allgather"), (3, "p2p", "allgather"),
                                                  (2, "p2p", "gather"), (3, "allgather", "gather")])
def test_row_tiled_chain_equals_full_frame(world, halo_mode, comp):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, halo_mode, q, comp)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, frame = q.get(timeout=180)
        assert not isinstance(frame, str), f"rank {r} failed: {frame}"
        got[r] = frame
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _full_frame(world)
    for r in range(world):
        if comp == "gather" and r != 0:
            assert got[r] is None
            continue
        assert got[r].shape == ref.shape and np.array_equal(got[r], ref), f"rank {r}: composited frame differs from the full-frame result"


def test_row_tiling_geometry():
    t = tiling.RowTiling(3840, 2160 * 8, 8, 3)
    assert (t.tile_rows, t.row0, t.row1) == (2160, 6480, 8640)
    t = tiling.RowTiling(64, 100, 3, 0)        # uneven heights: the first frame_height % world ranks own one row more (vqhip_rowtile)
    assert (t.row0, t.tile_rows) == (0, 34) and tiling.RowTiling(64, 100, 3, 2).row1 == 100
    with pytest.raises(ValueError):
        tiling.RowTiling(64, 16, 2, 0)         # tiles shorter than the 10-row halo
    assert tiling.HALO_ROWS == 10              # KERNEL_RANGE - 1, GaussianBlur.hlsl:54-55


def test_synthetic_tiles_are_row_range_independent():
    full = synth.gbuffer(64, 100, seed=9)
    part = synth.gbuffer_rows(64, 100, 37, 81, seed=9)
    for k in range(4):
        assert np.array_equal(full[k][37:81], part[k])
