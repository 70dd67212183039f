"""Row-tiled multi-GPU mode: one process per GPU, the frame is cut into contiguous row tiles,
light arrays / env cubemaps / LUT are replicated, and only two exchanges touch the data path:

  1. blur halo  — the Y pass needs KERNEL_RANGE-1 = 10 rows of the X-blurred image above and below the tile
                  (GaussianBlur.hlsl:54-55,176-180): point-to-point send/recv with ranks r-1 / r+1 (each GPU pair
                  has its own xGMI link), or — as BASELINE.json words it — one small all-gather of boundary rows;
  2. composite  — all-gather of the tonemapped tiles so every rank holds the whole frame.

The PRODUCT data path is the C ABI: vqhip_exchange_blur_halos / vqhip_composite_tiles (vqengine_amd/csrc/mgpu.hip, RCCL
send/recv on the caller's stream), bound here as `RowTiledFrame`; bench.py uses nothing else between its kernels. The
torch.distributed functions further down state the same two exchanges on any backend ("gloo" with CPU tensors): they are the
independent second statement the CPU tests compare the tiling against (tests/test_distributed_cpu.py), not the product path.
There is no collective in the shade / X-blur / tonemap stages."""
import torch
import torch.distributed as dist

from . import capi

HALO_ROWS = capi.HALO_ROWS  # KERNEL_RANGE_MINUS1, Shaders/GaussianBlur.hlsl:54-55


class RowTiling:
    """Rows of rank `rank` == vqhip_rowtile: frame_height // world each, the first frame_height % world ranks one more."""

    def __init__(self, width, frame_height, world_size, rank):
        self.width, self.frame_height, self.world, self.rank = width, frame_height, world_size, rank
        self.row0, self.tile_rows = capi.rowtile(frame_height, world_size, rank)
        self.row1 = self.row0 + self.tile_rows


class RowTiledFrame:
    """The two exchanges of a row-tiled frame through the C ABI (RCCL). `unique_id` comes from capi.comm_unique_id() on rank 0 and
    reaches the other ranks out of band (bench.py broadcasts it with torch.distributed — control plane only). Buffers are the caller's:
    device tensors, or numpy arrays with the shared-memory mock RCCL of the tests (tests/cpp/mock_rccl.cpp)."""

    def __init__(self, unique_id, width, frame_height, world, rank):
        self.tiling = RowTiling(width, frame_height, world, rank)
        self.comm = capi.Comm(unique_id, world, rank)

    def close(self):
        self.comm.close()

    def exchange_blur_halos(self, x_tile, fmt, halo_top, halo_bottom, stream=None):
        """halo_top / halo_bottom: caller-owned [10, W, 4] buffers; returns the pair with None for a frame edge."""
        t = self.tiling
        top = halo_top if t.rank > 0 else None
        bottom = halo_bottom if t.rank < t.world - 1 else None
        self.comm.exchange_blur_halos(x_tile, fmt, top, bottom, stream)
        return top, bottom

    def composite(self, tile, fmt, frame, root=0, stream=None):
        self.comm.composite_tiles(tile, fmt, self.tiling.frame_height, root, frame, stream)
        return frame


def exchange_halos_p2p(x_tile, group=None):
    """x_tile: [rows, W, C] X-blurred tile. Returns (halo_top, halo_bottom); None at the frame border."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    top = bottom = None
    ops = []
    if rank > 0:
        top = torch.empty_like(x_tile[:HALO_ROWS])
        ops.append(dist.P2POp(dist.isend, x_tile[:HALO_ROWS].contiguous(), dist.get_global_rank(group, rank - 1) if group else rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, top, dist.get_global_rank(group, rank - 1) if group else rank - 1, group))
    if rank < world - 1:
        bottom = torch.empty_like(x_tile[:HALO_ROWS])
        ops.append(dist.P2POp(dist.isend, x_tile[-HALO_ROWS:].contiguous(), dist.get_global_rank(group, rank + 1) if group else rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, bottom, dist.get_global_rank(group, rank + 1) if group else rank + 1, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return top, bottom


def exchange_halos_allgather(x_tile, group=None):
    """Same result through ONE all-gather of each rank's 2x10 boundary rows (BASELINE.json's wording)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1:
        return None, None
    mine = torch.cat([x_tile[:HALO_ROWS], x_tile[-HALO_ROWS:]], 0).contiguous()
    flat = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(flat, mine, group=group)          # concatenated along dim 0 (the form gloo and nccl share)
    allb = flat.view((world,) + tuple(mine.shape))
    top = allb[rank - 1, HALO_ROWS:].contiguous() if rank > 0 else None
    bottom = allb[rank + 1, :HALO_ROWS].contiguous() if rank < world - 1 else None
    return top, bottom


def composite_to_root(tile, out=None, dst=0, group=None, async_op=False):
    """Gather the row tiles into the full frame [world*rows, W, C] on rank `dst` only (the GPU that presents the frame, like
    the reference's single swap chain). 1/world of the all-gather's traffic: `dst` receives world-1 tiles over its world-1
    direct xGMI links, every other rank sends one. Returns (frame on dst | None elsewhere, work|None)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    root = dist.get_global_rank(group, dst) if group else dst
    parts = None
    if rank == dst:
        if out is None:
            out = torch.empty((world * tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        parts = list(out.chunk(world, 0))                               # contiguous row-tile views of the frame
    else:
        out = None
    work = dist.gather(tile.contiguous(), parts, dst=root, group=group, async_op=async_op)
    return out, work


def composite(tile, out=None, group=None, async_op=False):
    """All-gather the row tiles into the full frame [world*rows, W, C] on every rank. Returns (frame, work|None)."""
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * tile.shape[0],) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    work = dist.all_gather_into_tensor(out, tile.contiguous(), group=group, async_op=async_op)
    return out, work
