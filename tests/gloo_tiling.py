"""TEST INFRASTRUCTURE: the two exchanges of the row-tiled frame (blur halos, composite) stated a second time with plain torch.distributed
collectives / point-to-point ops on any backend ("gloo" with CPU tensors) — independent of the product's C-ABI path (vqengine_amd/csrc/mgpu.hip
through vqengine_amd/tiling.RowTiledFrame) and of libvqhip.so. tests/test_distributed_cpu.py checks the tiling against these with world sizes
2 and 3; nothing in the product or in bench.py uses them."""
import torch
import torch.distributed as dist

HALO_ROWS = 10  # KERNEL_RANGE_MINUS1, Shaders/GaussianBlur.hlsl:54-55


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
