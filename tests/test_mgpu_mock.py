"""The row-tiled multi-GPU data path THROUGH THE C ABI (vqhip_rowtile, vqhip_comm_create, vqhip_exchange_blur_halos,
vqhip_composite_tiles; vqengine_amd/csrc/mgpu.hip) on a box without GPUs: RCCL is replaced by tests/cpp/libmock_rccl.so
($VQHIP_RCCL_LIBRARY), which moves the bytes of ncclSend / ncclRecv through shared memory, buffers are host arrays and the per-tile
compute is the oracle's. What is verified is everything the product adds around the RCCL calls: tile bounds (uneven heights too), which
rows go to which neighbour, where received rows land, edge ranks, root-only and all-ranks composites, pitched tiles —
the composited frame must equal the single-process full-frame result bit for bit. RCCL itself over xGMI is the driver's 8-GPU run."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from vqengine_amd import abi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "cpp", "libmock_rccl.so")
W = 96


def _full_frame(H):
    from tests import oracle_lib as O
    gb = synth.gbuffer(W, H, seed=0xD157)
    pf, _ = synth.per_frame(points=synth.point_lights(12, seed=0xD157))
    scene = O.forward_lighting(gb, pf, synth.per_view(W, H), abi.FMT_RGBA16F)
    return O.tonemap(O.gaussian_blur(scene, abi.FMT_RGBA16F), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)


def _worker(rank, world, H, root, pitch_pad, conn, q):
    try:
        os.environ["VQHIP_RCCL_LIBRARY"] = MOCK
        from tests import oracle_lib as O
        from vqengine_amd import capi, tiling
        uid = capi.comm_unique_id() if rank == 0 else None
        if rank == 0:
            for c in conn:
                c.send(uid)
        else:
            uid = conn.recv()
        fr = tiling.RowTiledFrame(uid, W, H, world, rank)
        tl = fr.tiling
        info = fr.comm.query()                               # read back from the communicator, not from the arguments
        assert (info["nranks_seen"], info["rank_seen"], info["version"]) == (world, rank, 0) and info["library_path"].endswith("libmock_rccl.so"), info
        lb_src = np.arange(3001, dtype=np.uint8) + rank      # vqhip_comm_loopback: a grouped send + receive addressed to the own rank
        lb_dst = np.zeros_like(lb_src)
        fr.comm.loopback(lb_src, lb_dst)
        assert np.array_equal(lb_src, lb_dst)
        gb = synth.gbuffer_rows(W, H, tl.row0, tl.row1, seed=0xD157)
        pf, _ = synth.per_frame(points=synth.point_lights(12, seed=0xD157))
        scene = O.forward_lighting(gb, pf, synth.per_view(W, H), abi.FMT_RGBA16F)
        x = O.blur_pass(scene, abi.FMT_RGBA16F, 0)
        top, bottom = np.full((10, W, 4), 7, np.float16), np.full((10, W, 4), 7, np.float16)
        if pitch_pad:                                        # a tile whose rows are pitch_pad pixels apart: packed into ONE dense message per neighbour
            xp = np.zeros((tl.tile_rows, W + pitch_pad, 4), np.float16)
            xp[:, :W] = x
            rc = fr.comm.lib.vqhip_exchange_blur_halos(fr.comm._h, None, xp.ctypes.data, W, tl.tile_rows, W + pitch_pad, abi.FMT_RGBA16F,
                                                      top.ctypes.data if rank > 0 else None, bottom.ctypes.data if rank < world - 1 else None)
            assert rc == 0
            top, bottom = (top if rank > 0 else None), (bottom if rank < world - 1 else None)
        else:
            top, bottom = fr.exchange_blur_halos(x, abi.FMT_RGBA16F, top, bottom)
        assert (top is None) == (rank == 0) and (bottom is None) == (rank == world - 1)
        y = O.blur_pass(x, abi.FMT_RGBA16F, 1, halo_top=top, halo_bottom=bottom)
        sdr = O.tonemap(y, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
        receives = root == capi.ALL_RANKS or root == rank
        frame = np.zeros((H, W, 4), np.uint8) if receives else None
        fr.composite(sdr, abi.FMT_RGBA8_UNORM, frame, root=root)
        q.put((rank, frame))
        fr.close()
    except Exception as e:                                   # surface failures instead of a queue timeout
        q.put((rank, repr(e)))
        raise


@pytest.mark.parametrize("world,H,root,pitch_pad", [(2, 48, 0, 0), (3, 72, 0, 0), (3, 70, -1, 0), (4, 97, 2, 0), (2, 51, -1, 5)])
def test_row_tiled_frame_through_the_c_abi(world, H, root, pitch_pad):
    if not os.path.exists(MOCK):
        pytest.fail("tests/cpp/libmock_rccl.so not built (make -C tests/cpp)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pipes = [ctx.Pipe() for _ in range(world - 1)]
    procs = [ctx.Process(target=_worker, args=(0, world, H, root, pitch_pad, [p[0] for p in pipes], q))]
    procs += [ctx.Process(target=_worker, args=(r, world, H, root, pitch_pad, pipes[r - 1][1], q)) for r in range(1, world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, f = q.get(timeout=120)
        assert not isinstance(f, str), f"rank {r}: {f}"
        got[r] = f
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = _full_frame(H)
    for r in range(world):
        if root in (-1, r):
            assert np.array_equal(got[r], want), f"rank {r}: composite differs from the untiled frame"
        else:
            assert got[r] is None


def test_rowtile_partition_and_errors():
    from vqengine_amd import capi, tiling
    for H, world in ((4320, 8), (2160, 7), (97, 4), (10, 1), (1, 1)):
        rows = [capi.rowtile(H, world, r) for r in range(world)]
        assert rows == [tiling.rowtile(H, world, r) for r in range(world)]          # the pure-Python split == vqhip_rowtile
        assert rows[0][0] == 0 and sum(n for _, n in rows) == H
        assert all(rows[r][0] + rows[r][1] == rows[r + 1][0] for r in range(world - 1))
        assert max(n for _, n in rows) - min(n for _, n in rows) <= 1
    with pytest.raises(ValueError):
        tiling.rowtile(79, 8, 0)
    with pytest.raises(capi.VQHipError):
        capi.rowtile(79, 8, 0)                               # 9-row tiles: shorter than the halo
    with pytest.raises(capi.VQHipError):
        capi.rowtile(100, 4, 4)


def _mismatched(rank, conn):
    """ten 1-row sends against one 10-row receive: what a per-row send of a pitched tile would do to a neighbour with a dense halo buffer"""
    import ctypes as C
    os.environ["VQHIP_RCCL_LIBRARY"] = MOCK
    from vqengine_amd import capi
    uid = capi.comm_unique_id() if rank == 0 else conn.recv()
    if rank == 0:
        conn.send(uid)
    comm = capi.Comm(uid, 2, rank)
    mock = C.CDLL(MOCK)
    buf = np.zeros(10 * 64, np.uint8)
    mock.ncclSend.argtypes = mock.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    h = C.cast(comm._h, C.POINTER(C.c_void_p))[0]            # vqhip_comm::comm is the first member
    if rank == 0:
        for y in range(10):
            mock.ncclSend(buf.ctypes.data + 64 * y, 64, 1, 1, h, None)
    else:
        mock.ncclRecv(buf.ctypes.data, 640, 1, 0, h, None)


def test_mock_rccl_refuses_mismatched_message_sizes():
    """The stand-in enforces RCCL's matching rule (equal counts, one send per receive): the receiver of a mismatched message aborts."""
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    procs = [ctx.Process(target=_mismatched, args=(0, a)), ctx.Process(target=_mismatched, args=(1, b))]
    for p in procs:
        p.start()
    procs[1].join(60)
    assert procs[1].exitcode not in (0, None), "the mismatched receive must abort"
    procs[0].kill()
    procs[0].join(10)
