"""Row-tiled multi-GPU mode: one process per GPU, the frame is cut into contiguous row tiles,
light arrays / env cubemaps / LUT are replicated, and only two exchanges touch the data path:

  1. blur halo  — the Y pass needs KERNEL_RANGE-1 = 10 rows of the X-blurred image above and below the tile
                  (GaussianBlur.hlsl:54-55,176-180): point-to-point send/recv with ranks r-1 / r+1 (each GPU pair
                  has its own xGMI link), or — as BASELINE.json words it — one small all-gather of boundary rows;
  2. composite  — all-gather of the tonemapped tiles so every rank holds the whole frame.

The data path is the C ABI: vqhip_exchange_blur_halos / vqhip_composite_tiles (vqengine_amd/csrc/mgpu.hip, RCCL send/recv on the
caller's stream), bound here as `RowTiledFrame`; bench.py uses nothing else between its kernels. There is no collective in the
shade / X-blur / tonemap stages. (An independent torch.distributed / gloo statement of the same two exchanges lives with the tests:
tests/gloo_tiling.py, tests/test_distributed_cpu.py.)"""
from . import capi

HALO_ROWS = capi.HALO_ROWS  # KERNEL_RANGE_MINUS1, Shaders/GaussianBlur.hlsl:54-55


def rowtile(frame_height, world, rank):
    """(row0, rows) of rank `rank`: frame_height // world rows each, the first frame_height % world ranks one more — the same split as
    vqhip_rowtile (csrc/mgpu.hip), stated here in plain Python so that host-side tiling logic and the CPU tests do not need the built
    library; tests/test_mgpu_mock.py::test_rowtile_partition_and_errors holds the two together."""
    if frame_height <= 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("rowtile: bad arguments")
    q, rem = divmod(frame_height, world)
    if world > 1 and q < HALO_ROWS:
        raise ValueError("rowtile: tiles must be at least 10 rows tall (the blur halo comes from the direct neighbour only)")
    return rank * q + min(rank, rem), q + (1 if rank < rem else 0)


class RowTiling:
    """Rows of rank `rank` (rowtile above == vqhip_rowtile)."""

    def __init__(self, width, frame_height, world_size, rank):
        self.width, self.frame_height, self.world, self.rank = width, frame_height, world_size, rank
        self.row0, self.tile_rows = rowtile(frame_height, world_size, rank)
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
