#!/usr/bin/env python3
"""bench.py — forward-PBR hot path on MI355X (BASELINE.json metric: Mpixels/s forward-PBR @4K, 64 lights; 1/2/4/8 GPUs).

A "step" is one pass of the hot path over one synthetic frame (tile) that is already resident in HBM:
    forward lighting (point lights [+ IBL sample])   -> RGBA16F scene colour   [vqhip_forward_lighting]
    (N > 1) 10-row halo exchange of scene colour with the neighbours            [vqhip_exchange_blur_halos, RCCL send/recv]
    21-tap Gaussian blur X, blur Y, tonemap (Reinhard + sRGB OETF), ONE kernel -> RGBA8_UNORM   [vqhip_post_process_tile]
    (N > 1) composite of the RGBA8 tiles on rank 0                               [vqhip_composite_tiles, RCCL send/recv]
(--post fused: blur X [vqhip_gaussian_blur_x], exchange of X-blurred rows, blur Y + tonemap in one kernel [vqhip_gaussian_blur_y_tonemap] — rounds 1-4's chain, reported
as `other_post_form`; --post split: three dispatches.)

The HEADLINE (`metric`, `value`, `ms_per_step`, `roofline`) is --config cfg3, BASELINE config 3, the configuration the metric is quoted on:
a 3840x2160 tile per GPU, 64 point lights + the full-size cfg4 IBL; N > 1 is WEAK scaling (frame 3840 x 2160*N). K timed steps, exactly.

Every invocation (N = 1 included) ALSO times, outside the headline's timed region and reported as extra objects of the same JSON line:
  cfg5_strong    BASELINE config 5, the configuration the ">= 6x at 8 GPUs" target is defined on: ONE 7680x4320 frame, 256 point lights
                 (100 cbuffer + 156 extension), 4320/N rows per GPU — STRONG scaling; with shade / halo / composite / latency figures
  (N = 1 only)
  cfg2           BASELINE config 2: 1920x1080, 16 point lights, shade kernel (the size where HBM is the roof that matters)
  sustained      the headline's own step for ~2 s without interruption (thousands of steps): a long-window cross-check of `value`, visible to rocm-smi
  ibl_load       BASELINE config 4: the load-time IBL stages (min-filter mip chain, diffuse irradiance, specular prefilter, BRDF LUT), timed
  widened        the SURVEY 8f kernels at 4K: G-buffer producer (textured / texture-less), PSMain as one kernel, skydome, .hdr decode, FSR EASU / RCAS,
                 SSR environment fallback — ms, algorithmic bytes per pixel, fraction of the HBM spec
  coherent_scene the cfg3 frame on surface-coherent content (synth.gbuffer_rows_coherent) instead of white noise — never the headline
  tile_curve     per-tile step time of the cfg5 frame at 4320/N rows, N = 1, 2, 4, 8, on this one GPU + a labelled MODELLED speed-up
--config cfg5 makes cfg5 the headline instead (then `scaling` is "strong"); --no-extras skips everything but the headline.

Every byte that crosses GPUs goes through the C ABI (include/vqhip.h, vqengine_amd/csrc/mgpu.hip); torch.distributed is the control
plane only (communicator-id broadcast, barriers, the max-over-ranks reduction of the wall time).
`value` = pixels of the whole frame * steps / max-over-ranks wall time. Prints ONE JSON line on rank 0.

This file holds the workload (CONFIGS, Pipeline.step), the ranks (Dist, Comms) and main(); the reporting parts live in benchlib/: consts (peaks), timing (the stage
timer), pmc (counter constants), cpu (the CPU legs: the only code here that runs oracle/), casters / ibl / widened / tiles (the extra objects), line (key order + digest)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import casters as bl_casters, ibl as bl_ibl  # noqa: E402
from vqengine_amd import abi, capi, synth, tiling  # noqa: E402
from benchlib.consts import F16, HBM_PEAK_GBPS, R8, SHADE_BYTES_PER_PX, VALU_ISSUE_CEILING_TLIS, VALU_PEAK_TFLOPS  # noqa: E402
from benchlib.cpu import cpu_baseline, cpu_reference_source, host_cores  # noqa: E402
from benchlib.ibl import ibl_load_report  # noqa: E402
from benchlib.line import finish_line  # noqa: E402
from benchlib.pmc import PMC_FILE, PMC_SOURCES, kernel_source_hash, load_pmc_constants, load_time_kernel_counters  # noqa: E402,F401
from benchlib.tiles import tile_curve  # noqa: E402
from benchlib.timing import _ev, _stage_ms, _stage_stats, _time_loop  # noqa: E402,F401
from benchlib.widened import widened_report  # noqa: E402

CONFIGS = {
    "cfg2": dict(width=1920, height=1080, lights=16, env=False, seed=0xC0FFEE, light_seed=0x1600, scaling="weak",
                 metric="Mpixels/s forward-PBR @1080p,16 lights (BASELINE cfg2)",
                 workload="BASELINE cfg2: 1920x1080 float4 G-buffer, 16 point lights, no IBL -> RGBA16F"),
    "cfg3": dict(width=3840, height=2160, lights=64, env=True, seed=0x6400, light_seed=0x6400, scaling="weak",
                 metric="Mpixels/s forward-PBR @4K,64 lights",
                 workload="BASELINE cfg3: 3840x2160 float4 G-buffer tile per GPU, 64 point lights + IBL sample -> RGBA16F, 21-tap blur X/Y, Reinhard+sRGB tonemap -> RGBA8"),
    "cfg5": dict(width=7680, height=4320, lights=256, env=False, seed=0x2560, light_seed=0x2560, scaling="strong",
                 metric="Mpixels/s forward-PBR @8K,256 lights (BASELINE cfg5, one frame row-tiled over the GPUs)",
                 workload="BASELINE cfg5: ONE 7680x4320 float4 G-buffer, 256 point lights (100 cbuffer + 156 extension) -> RGBA16F, 21-tap blur X/Y, Reinhard+sRGB tonemap -> RGBA8"),
}
SPINUP_STEPS = int(os.environ.get("VQ_BENCH_SPINUP", "200"))     # untimed steady-state spin-up before the W warm-up steps (~0.25 s of GPU work)
COLD_STEPS = 20                 # the first steps after the idle set-up phase, timed on their own ("cold_start")
WATCHDOG_S = float(os.environ.get("VQ_BENCH_WATCHDOG_S", "30"))
SUSTAINED_S = float(os.environ.get("VQ_BENCH_SUSTAINED_S", "2.0"))   # length of the `sustained` companion run (0: off)
# Test hooks live OUTSIDE this file: tests/bench_fault_harness.py subclasses Pipeline (a dropped stream wait, a forced watchdog timeout) and runs main() with it


def build_ibl(ctx, timings=None):
    """Load-time inputs (outside the timed region): BASELINE config 4 — 2048^2 equirect -> min-filter mips ->
    diffuse 64^2 (step 0.010) + blur + 7-mip specular 128^2, and the 1024^2 x 2048 BRDF LUT. With `timings` (a dict) each stage is
    bracketed by HIP events on the stream it runs on and, after the product call, re-run on its own for a per-stage figure."""
    eq = torch.from_numpy(synth.equirect(2048, 2048)).cuda()
    e = [_ev() for _ in range(4)]
    e[0].record()
    chain, n = ctx.mip_chain(eq)
    e[1].record()
    pre = ctx.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128, abi.CONV_SEQUENTIAL)
    e[2].record()
    lut = ctx.brdf_lut(1024, 2048, abi.FMT_RG16F)
    e[3].record()
    torch.cuda.synchronize()
    if timings is not None:
        timings.update(mip_chain_ms=round(e[0].elapsed_time(e[1]), 4), prefilter_ms=round(e[1].elapsed_time(e[2]), 4), brdf_lut_ms=round(e[2].elapsed_time(e[3]), 4))
        for name, fn in (("conv_diffuse_ms", lambda: ctx.conv_diffuse(chain, 2048, 2048, n, 64, 0.010, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F)),
                         ("conv_specular_ms", lambda: ctx.conv_specular(chain, 2048, 2048, n, 128, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F)),
                         ("brdf_lut_warm_ms", lambda: ctx.brdf_lut(1024, 2048, abi.FMT_RG16F)),
                         ("mip_chain_warm_ms", lambda: ctx.mip_chain(eq)),
                         ("prefilter_warm_ms", lambda: ctx.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128, abi.CONV_SEQUENTIAL))):
            a, b = _ev(), _ev()
            a.record(); fn(); b.record(); b.synchronize()
            timings[name.replace("_ms", "_single_call_ms")] = round(a.elapsed_time(b), 4)      # one call, clocks as the set-up phase left them
            st = _stage_stats(fn)                                    # >= 0.25 s spin-up, median of 7 batches: the timer of every per-stage figure of the line
            timings[name] = round(st["ms"], 4)
            timings[name.replace("_ms", "_ms_spread")] = [round(st["ms_min"], 4), round(st["ms_max"], 4)]
    return pre, lut


def upload_tile(cfg, frame_h, row0, row1, coherent=False):
    W = cfg["width"]
    gen = synth.gbuffer_rows_coherent if coherent else synth.gbuffer_rows
    gb = [torch.empty((row1 - row0, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    for r in range(row0, row1, 240):
        part = gen(W, frame_h, r, min(r + 240, row1), seed=cfg["seed"])
        for k in range(4):
            gb[k][r - row0:r - row0 + part[k].shape[0]].copy_(torch.from_numpy(part[k]))
    return gb


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: bench.py starts its own N ranks (one per GPU) by replacing itself with
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`;
    rank 0 prints the one JSON line, the exit status is the launcher's. Refuses — non-zero, with a message — when the node has fewer than N GPUs
    (VQ_BENCH_SHARE_GPU=1, the single-GPU debug aid, lifts that: the ranks then share the visible GPUs)."""
    import socket
    have = torch.cuda.device_count()
    if have < n and os.environ.get("VQ_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py: --gpus {n} but this node shows {have} GPU(s); one rank per GPU is the only supported layout")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: no launcher in the environment (WORLD_SIZE unset): starting {n} ranks: {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


class Dist:
    """Control plane: world / rank, the barrier + synchronize bracket, max-over-ranks reductions, and CPU-side agreement (gloo) that keeps
    working when a GPU stream does not progress."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        # VQ_BENCH_SHARE_GPU=1 (debug aid for single-GPU boxes): every rank uses the visible GPUs round-robin, the control plane runs over
        # gloo and the C ABI's RCCL calls are served by tests/cpp/libmock_rccl.so (shared memory), so that the N > 1 control flow — tiles,
        # halo exchange, double-buffered composite, drain — can be exercised on real kernels without N GPUs. Never set by the driver.
        self.share = os.environ.get("VQ_BENCH_SHARE_GPU") == "1"
        self.device_ordinal = local_rank % torch.cuda.device_count() if self.share else local_rank
        torch.cuda.set_device(self.device_ordinal)
        self.cpu_group = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.share:
                os.environ["VQHIP_RCCL_LIBRARY"] = os.path.join(ROOT, "tests", "cpp", "libmock_rccl.so")
                dist.init_process_group("gloo")
                self.cpu_group = dist.group.WORLD
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device_ordinal))
                self.cpu_group = dist.new_group(backend="gloo")
                dist.barrier()                               # one RCCL collective of the control plane, before any data-path communicator exists
                torch.cuda.synchronize()

    def barrier(self):
        """Device idle on this rank, then every rank here (a CPU barrier over gloo), so that no collective of the control plane is ever in
        flight on a GPU next to the data path's own RCCL transfers (two communicators whose kernels reach different GPUs in different orders
        are only safe while both kernels can be co-resident)."""
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier(group=self.cpu_group)

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.cpu_group)
        return float(t.item())

    def min_max_over_ranks(self, x):
        """[min, max] of a per-rank figure: the first real multi-GPU run shows the load imbalance across row tiles without a second run"""
        if self.world == 1:
            return [float(x), float(x)]
        lo, hi = torch.tensor([float(x)], dtype=torch.float64), torch.tensor([float(x)], dtype=torch.float64)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.cpu_group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.cpu_group)
        return [float(lo.item()), float(hi.item())]

    def all_true(self, flag):
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.cpu_group)
        return bool(t.item())

    def share_ids(self, n):
        ids = [capi.comm_unique_id() for _ in range(n)] if self.rank == 0 else [None] * n
        dist.broadcast_object_list(ids, src=0, group=self.cpu_group)          # control plane: n x 128 bytes
        return ids


class Comms:
    """The communicators of the data path (C ABI) and how the composite overlaps the next frame.
       one-comm : ONE communicator; the halo exchange runs on the main stream, the composite on a second stream. Every rank issues the
                  operations of the communicator in the same host order (composite n, halo n+1, composite n+1, ...), RCCL orders them on
                  the device, and composite n has the whole shading of frame n+1 to finish in before halo n+1 needs the communicator.
       two-comms: round 2's form — a second communicator for the composite, nothing orders the two on the device.
       off      : one communicator, one stream order: the composite sits on the critical path.
    --composite-overlap auto = one-comm under a watchdog: the wiring check and the first overlapped steps must complete within
    VQ_BENCH_WATCHDOG_S seconds on every rank, else every rank aborts its communicator (ncclCommAbort), builds a new one and runs `off`."""

    def __init__(self, d, mode):
        self.d, self.requested = d, mode
        self.mode = "one-comm" if mode in ("auto", "on") else mode
        self.fallback = None
        self.halo = self.comp = None
        if d.world > 1:
            self._create()

    def _create(self):
        d = self.d
        ids = d.share_ids(2 if self.mode == "two-comms" else 1)
        self.halo = capi.Comm(ids[0], d.world, d.rank)
        self.comp = capi.Comm(ids[1], d.world, d.rank) if self.mode == "two-comms" else self.halo

    @property
    def overlap(self):
        return self.d.world > 1 and self.mode != "off"

    def fall_back(self, why):
        for c in {id(self.halo): self.halo, id(self.comp): self.comp}.values():
            c.abort()
        torch.cuda.synchronize()
        self.fallback, self.mode = why, "off"
        self._create()

    def close(self):
        for c in {id(self.halo): self.halo, id(self.comp): self.comp}.values():
            if c is not None:
                c.close()

    def info(self):
        if self.halo is None:
            return {"nranks_seen": 1, "note": "N = 1: no communicator is created, RCCL is not loaded"}
        q = self.halo.query()
        return {"version": q["version"], "nranks_seen": q["nranks_seen"], "rank_seen": q["rank_seen"], "library_path": q["library_path"],
                "communicators": 2 if self.mode == "two-comms" else 1, "composite_overlap_mode": self.mode, "requested": self.requested,
                "fallback": self.fallback, "note": "read back from the communicator (ncclGetVersion / ncclCommCount / ncclCommUserRank, dladdr of ncclSend); version 0 = the "
                                                   "shared-memory test stand-in"}


class Pipeline:
    """One workload resident on this rank's GPU: the G-buffer tile, the intermediate images, and `step(i)` = one pass of the hot path."""

    def __init__(self, ctx, d, comms, cfg, args, env=None, max_env_lod=0, coherent=False, composite_root=0, rows_limit=None):
        self.ctx, self.d, self.comms, self.cfg, self.args, self.env = ctx, d, comms, cfg, args, env
        W, world = cfg["width"], d.world
        self.W = W
        self.frame_h = cfg["height"] * world if cfg["scaling"] == "weak" else cfg["height"]
        self.tl = tiling.RowTiling(W, self.frame_h, world, d.rank)
        self.rows = self.tl.tile_rows if rows_limit is None else rows_limit
        self.pf, self.extra = synth.per_frame(points=synth.point_lights(cfg["lights"], seed=cfg["light_seed"]), hdri_offset=0.3 if env is not None else 0.0)
        self.pv = synth.per_view(W, self.frame_h, max_env_lod=max_env_lod)
        self.gb = upload_tile(cfg, self.frame_h, self.tl.row0, self.tl.row0 + self.rows, coherent)
        dev = ctx.device
        self.scene = [capi.empty_image(self.rows, W, F16, dev) for _ in range(2)]
        self.xblur = capi.empty_image(self.rows, W, F16, dev)
        self.yblur = capi.empty_image(self.rows, W, F16, dev) if args.post == "split" else None
        self.sdr = [capi.empty_image(self.rows, W, R8, dev) for _ in range(2)]
        self.root = composite_root
        need_frame = world > 1 and (self.root == capi.ALL_RANKS or d.rank == 0)
        self.frame = [torch.empty((self.frame_h, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if need_frame else [None, None]
        self.halo_top = capi.empty_image(capi.HALO_ROWS, W, F16, dev) if world > 1 and d.rank > 0 else None
        self.halo_bottom = capi.empty_image(capi.HALO_ROWS, W, F16, dev) if world > 1 and d.rank < world - 1 else None
        self.s_main = torch.cuda.current_stream(dev)
        self.s_comp = torch.cuda.Stream(dev) if world > 1 else self.s_main       # used only while comms.overlap
        self.e_post = [torch.cuda.Event(), torch.cuda.Event()]
        self.e_comp = [None, None]
        self.s_post = torch.cuda.Stream(dev) if world == 1 else None
        self.s_frame = [self.s_main, self.s_post] if world == 1 else None     # two frames in flight reuse the two streams of the default mode: HIP multiplexes streams onto 4
                                                                               # hardware queues, and two more streams may land on ONE queue (then nothing overlaps)
        self.e_shade = [torch.cuda.Event(), torch.cuda.Event()]
        self.e_pdone = [None, None]

    def free(self):
        self.gb = self.scene = self.sdr = self.frame = self.xblur = self.yblur = None
        torch.cuda.empty_cache()

    def step(self, i, ev=None):
        """ev: dict of events to record: t0, shade (end of the shade kernel), x (end of the X pass), halo (end of the halo exchange),
        post (end of the last post kernel) on the main stream; comp0 / comp1 around the composite on the stream it runs on."""
        ctx, world, overlap = self.ctx, self.d.world, self.comms.overlap
        s_main = self.s_main
        s_comp = self.s_comp if overlap else s_main
        b = i & 1
        rec = (lambda k, s=s_main: ev[k].record(s)) if ev else (lambda k, s=None: None)
        if overlap and self.e_comp[b] is not None:   # sdr[b] / frame[b] were last touched by the composite of step i-2
            s_main.wait_event(self.e_comp[b])
        if self.args.post == "chain" and world == 1 and getattr(self.args, "post_stream", "main") == "frames":
            # TWO FRAMES IN FLIGHT — frame i runs shade + post chain on stream i & 1 with its own buffer pair: no event between the streams at all, the tail of one
            # frame's shade kernel runs under the head of the next frame's
            st = self.s_frame[b]
            if ev and "t0" in ev:
                ev["t0"].record(st)
            ctx.forward_lighting(self.gb, self.pf, self.pv, out=self.scene[b], out_fmt=F16, extra_point=self.extra, env=self.env, stream=st)
            for k in ("shade", "x", "halo"):
                if ev and k in ev:
                    ev[k].record(st)
            ctx.post_process_tile(self.scene[b], F16, R8, out=self.sdr[b], stream=st)
            if ev and "post" in ev:
                ev["post"].record(st)
            return
        if self.e_pdone[b] is not None:             # --post-stream own: shade(i) overwrites scene[b], which post(i - 2) reads on the other stream
            s_main.wait_event(self.e_pdone[b])
            self.e_pdone[b] = None
        if ev and "t0" in ev:
            rec("t0")
        ctx.forward_lighting(self.gb, self.pf, self.pv, out=self.scene[b], out_fmt=F16, extra_point=self.extra, env=self.env)
        if ev and "shade" in ev:
            rec("shade")
        if self.args.post == "chain" and world == 1 and getattr(self.args, "post_stream", "main") == "frames":
            return                                           # handled above (two frames in flight)
        if self.args.post == "chain" and world == 1 and getattr(self.args, "post_stream", "main") == "own":
            # N = 1 experiment: the post chain of frame i on its own stream, so that it runs next to the shade kernel of frame i + 1 (double-buffered scene colour)
            if ev and "x" in ev:
                rec("x")
            if ev and "halo" in ev:
                rec("halo")
            self.e_shade[b].record(s_main)
            self.s_post.wait_event(self.e_shade[b])
            ctx.post_process_tile(self.scene[b], F16, R8, out=self.sdr[b], stream=self.s_post)
            self.e_pdone[b] = torch.cuda.Event()
            self.e_pdone[b].record(self.s_post)
            if ev and "post" in ev:
                ev["post"].record(self.s_post)
            return
        if self.args.post == "chain":
            # X, Y and the tonemapper in ONE kernel (k_post_chain): neither BlurIntermediate nor BlurOutput exists. Row tiles exchange 10 rows of SCENE COLOUR
            # right behind the shade kernel (the X pass is horizontal: the kernel filters the neighbour's rows like its own)
            if ev and "x" in ev:
                rec("x")
            if world > 1:
                self.comms.halo.exchange_blur_halos(self.scene[b], F16, self.halo_top, self.halo_bottom, stream=C.c_void_p(s_main.cuda_stream))
            if ev and "halo" in ev:
                rec("halo")
            ctx.post_process_tile(self.scene[b], F16, R8, out=self.sdr[b], halo_top=self.halo_top, halo_bottom=self.halo_bottom)
            if ev and "post" in ev:
                rec("post")
            return self._composite(b, ev, rec, s_main, s_comp, overlap, world)
        ctx.gaussian_blur_x(self.scene[b], F16, out=self.xblur)
        if ev and "x" in ev:
            rec("x")
        if world > 1:
            self.comms.halo.exchange_blur_halos(self.xblur, F16, self.halo_top, self.halo_bottom, stream=C.c_void_p(s_main.cuda_stream))
        if ev and "halo" in ev:
            rec("halo")
        if self.args.post == "fused":
            # CSMain_Y + Tonemapper in one kernel: bit-identical to the two dispatches, BlurOutput never touches HBM
            ctx.gaussian_blur_y_tonemap(self.xblur, F16, R8, out=self.sdr[b], halo_top=self.halo_top, halo_bottom=self.halo_bottom)
        else:
            ctx.gaussian_blur_y(self.xblur, F16, out=self.yblur, halo_top=self.halo_top, halo_bottom=self.halo_bottom)
            if ev and "y" in ev:
                rec("y")
            ctx.tonemap(self.yblur, F16, R8, out=self.sdr[b])
        if ev and "post" in ev:
            rec("post")
        self._composite(b, ev, rec, s_main, s_comp, overlap, world)

    def _composite(self, b, ev, rec, s_main, s_comp, overlap, world):
        if world > 1:
            if overlap:
                self.e_post[b].record(s_main)
                self.order_composite_behind_post(s_comp, b)
            if ev and "comp0" in ev:
                rec("comp0", s_comp)
            self.comms.comp.composite_tiles(self.sdr[b], R8, self.frame_h, self.root, self.frame[b], stream=C.c_void_p(s_comp.cuda_stream))
            if ev and "comp1" in ev:
                rec("comp1", s_comp)
            if overlap:
                self.e_comp[b] = torch.cuda.Event()
                self.e_comp[b].record(s_comp)

    def order_composite_behind_post(self, s_comp, b):
        """the composite's stream waits for the post kernel of this frame (tests/bench_fault_harness.py overrides this to inject the fault)"""
        s_comp.wait_event(self.e_post[b])

    def drain(self):
        if self.comms.overlap:
            self.s_main.wait_stream(self.s_comp)
        if self.s_post is not None:
            self.s_main.wait_stream(self.s_post)

    def verify_step(self, i):
        """One step whose output buffers were zeroed first (and the device drained): what the composite delivers can only be this step's pixels if every wait
        between the streams is in place."""
        b = i & 1
        self.drain()
        torch.cuda.synchronize()
        self.sdr[b].zero_()
        if self.frame[b] is not None:
            self.frame[b].zero_()
        torch.cuda.synchronize()
        self.d.barrier()
        self.e_comp[b] = None
        self.step(i)
        self.drain()
        torch.cuda.synchronize()
        self.d.barrier()
        return b

    def timed(self, n_steps, evs=None, first=0):
        self.d.barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            self.step(first + i, evs[i] if evs else None)
        self.drain()
        self.d.barrier()
        return self.d.max_over_ranks(time.perf_counter() - t0)

    def latency(self, n=5):
        """one step at a time, nothing in flight before or after it (the throughput figure pipelines the composite)"""
        lat = []
        for i in range(n):
            self.d.barrier()
            t0 = time.perf_counter()
            self.step(i)
            self.drain()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        return self.d.max_over_ranks(float(np.median(lat)))

    def completes_within(self, n_steps, seconds):
        """Watchdog: enqueue n_steps and poll (no blocking call) until both streams have drained or `seconds` have passed."""
        for i in range(n_steps):
            self.step(i)
        marks = [torch.cuda.Event(), torch.cuda.Event()]
        marks[0].record(self.s_main); marks[1].record(self.s_comp)
        deadline = time.perf_counter() + seconds
        while not (marks[0].query() and marks[1].query()):
            if time.perf_counter() > deadline:
                return False
            time.sleep(0.005)
        return True


def wiring_check(d, comms, ctx):
    """Before anything is timed: every rank sends rows that name their sender and their row, so a transfer that lands in the wrong place,
    comes from the wrong neighbour or does not arrive at all is a loud failure here instead of a wrong frame later."""
    world, rank = d.world, d.rank
    chk = tiling.RowTiling(64, 16 * world + (world // 2), world, rank)          # uneven tiles on purpose
    stream = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
    t_x = torch.empty((chk.tile_rows, 64, 4), dtype=torch.float16, device=ctx.device)
    t_x[:] = (torch.arange(chk.row0, chk.row1, device=ctx.device, dtype=torch.float32) + 1000.0 * rank).to(torch.float16)[:, None, None]
    h_t = torch.zeros((capi.HALO_ROWS, 64, 4), dtype=torch.float16, device=ctx.device) if rank > 0 else None
    h_b = torch.zeros((capi.HALO_ROWS, 64, 4), dtype=torch.float16, device=ctx.device) if rank < world - 1 else None
    comms.halo.exchange_blur_halos(t_x, F16, h_t, h_b, stream=stream)
    t_c = torch.full((chk.tile_rows, 64, 4), rank + 1, dtype=torch.uint8, device=ctx.device)
    f_c = torch.zeros((chk.frame_height, 64, 4), dtype=torch.uint8, device=ctx.device)
    comms.comp.composite_tiles(t_c, R8, chk.frame_height, capi.ALL_RANKS, f_c, stream=stream)
    torch.cuda.synchronize()
    if h_t is not None:
        want = (torch.arange(chk.row0 - capi.HALO_ROWS, chk.row0, device=ctx.device, dtype=torch.float32) + 1000.0 * (rank - 1)).to(torch.float16)
        assert torch.equal(h_t[:, 0, 0], want), f"rank {rank}: top halo rows are not the last 10 rows of rank {rank - 1}"
    if h_b is not None:
        want = (torch.arange(chk.row1, chk.row1 + capi.HALO_ROWS, device=ctx.device, dtype=torch.float32) + 1000.0 * (rank + 1)).to(torch.float16)
        assert torch.equal(h_b[:, 0, 0], want), f"rank {rank}: bottom halo rows are not the first 10 rows of rank {rank + 1}"
    for k in range(world):
        r0, n = capi.rowtile(chk.frame_height, world, k)
        assert bool((f_c[r0:r0 + n] == k + 1).all()), f"rank {rank}: rows of rank {k} are wrong in the composited frame"
    q = comms.halo.query()
    assert q["nranks_seen"] in (world, -1) and q["rank_seen"] in (rank, -1), f"rank {rank}: the communicator reports {q}"


def mean_ms(evs, a, b):
    return float(np.mean([e[a].elapsed_time(e[b]) for e in evs]))


def measure_strong(pipe, steps, warmup):
    """A shorter protocol for the extra objects: warm-up, K timed steps (barrier + synchronize bracket, max over ranks), a detail pass with
    per-stage events, the frame latency."""
    for i in range(warmup):
        pipe.step(i)
    pipe.drain()
    keys = ["t0", "shade", "x", "halo", "post"] + (["comp0", "comp1"] if pipe.d.world > 1 else [])
    dt = pipe.timed(steps)
    evd = [{k: _ev() for k in keys} for _ in range(min(steps, 6))]
    for i in range(len(evd)):
        pipe.step(steps + i + (steps & 1), evd[i])
    pipe.drain()
    pipe.d.barrier()
    px = pipe.W * pipe.frame_h
    out = {"value": round(px * steps / dt / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup,
           "frame": [pipe.W, pipe.frame_h], "tile_rows": pipe.rows, "lights": pipe.cfg["lights"], "scaling": pipe.cfg["scaling"],
           "shade_ms": round(pipe.d.max_over_ranks(mean_ms(evd, "t0", "shade")), 4),
           "shade_ms_over_ranks": [round(v, 4) for v in pipe.d.min_max_over_ranks(mean_ms(evd, "t0", "shade"))],
           **({"post_chain_ms": round(pipe.d.max_over_ranks(mean_ms(evd, "halo", "post")), 4)} if pipe.args.post == "chain" else
              {"blur_x_ms": round(pipe.d.max_over_ranks(mean_ms(evd, "shade", "x")), 4),
               "blur_y_tonemap_ms": round(pipe.d.max_over_ranks(mean_ms(evd, "halo", "post")), 4)}),
           "halo_ms": round(pipe.d.max_over_ranks(mean_ms(evd, "x", "halo")), 4) if pipe.d.world > 1 else 0.0,
           "composite_ms": round(pipe.d.max_over_ranks(mean_ms(evd, "comp0", "comp1")), 4) if pipe.d.world > 1 else 0.0,
           "composite_overlapped": pipe.comms.overlap,
           "frame_latency_ms": round(pipe.latency() * 1e3, 4),
           "note": "max over ranks of each rank's mean; halo_ms includes waiting for the slower neighbour's X pass; composite_ms is measured on the "
                   "stream the composite runs on and overlaps the next frame's shading when composite_overlapped"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # a step is ~1 ms: 0.2 s of timed GPU work; the run stays dominated by set-up and the CPU baselines
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=["cfg3", "cfg5"], default="cfg3", help="the HEADLINE workload (the other BASELINE configs are reported as extra objects)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip cfg5_strong / cfg2 / ibl_load / coherent_scene / tile_curve")
    ap.add_argument("--post", choices=["chain", "fused", "split"], default="chain",
                    help="post chain: blur X + blur Y + tonemapper in ONE kernel (chain: vqhip_post_process_tile, the library's default for frames of >= 2^20 pixels), "
                         "blur X then Y blur + tonemapper in one kernel (fused), or three dispatches (split); identical bits")
    ap.add_argument("--post-stream", choices=["main", "own", "frames"], default="own",
                    help="N = 1 with --post chain: the post chain of frame i on a stream of its own, next to the shade kernel of frame i + 1 (double-buffered scene colour; + 1.5 %: "
                         "profiles/r5y_post_stream_ab.txt), or on the main stream (main). N > 1 keeps the main stream (the composite is what overlaps the next frame there)")
    ap.add_argument("--composite", choices=["root", "all"], default="root",
                    help="final composite of the RGBA8 tiles: on rank 0 only (the presenting GPU; it receives over its N-1 direct xGMI links) or on every rank")
    ap.add_argument("--composite-overlap", choices=["auto", "on", "off", "two-comms"], default="auto",
                    help="auto / on: ONE communicator, the composite of frame n on a second stream so that it overlaps the shading of frame n+1 (drained inside "
                         "the timed region); auto adds a watchdog that falls back to `off` (one stream order) if the first steps do not complete; "
                         "two-comms: round 2's form with a second communicator for the composite")
    ap.add_argument("--fresnel-pow", choices=["product", "exp2_log2"], default="product",
                    help="pow(1 - cos, 5) of the Fresnel terms: the product x*((x*x)*(x*x)) (default, contract v4) or exp2(5*log2 x), the engine's own "
                         "DXC lowering (vqhip_set_fresnel_pow; DESIGN.md 3.2). The other mode is timed too and reported as `engine_lowering`.")
    ap.add_argument("--no-second-mode", action="store_true", help="skip the timing of the other Fresnel-pow mode")
    ap.add_argument("--content", choices=["noise", "coherent"], default="noise",
                    help="noise: the BASELINE workload (white-noise G-buffer, SURVEY.md 8d). coherent: the same frame size on surface-coherent content; the line is "
                         "then NOT the headline (metric and data say so) — used to collect counters for `coherent_scene` (scripts/pmc_refresh.sh)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                              # does not return
    d = Dist(args)
    world, rank = d.world, d.rank
    ctx = capi.Context(d.device_ordinal)
    ctx.set_fresnel_pow(args.fresnel_pow == "exp2_log2")
    comms = Comms(d, args.composite_overlap)
    if world > 1:
        wiring_check(d, comms, ctx)

    ibl_t = {}
    pre, lut = build_ibl(ctx, ibl_t if (world == 1 and not args.no_extras) else None)     # cfg4: the headline's IBL inputs (cfg3) — and, timed, `ibl_load`
    env = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], 128, pre["spec_mips"], lut)
    root = 0 if args.composite == "root" else capi.ALL_RANKS
    pipe = Pipeline(ctx, d, comms, cfg, args, env=env if cfg["env"] else None, max_env_lod=pre["spec_mips"] if cfg["env"] else 0, composite_root=root,
                    coherent=args.content == "coherent")
    W, L, rows, frame_h = cfg["width"], cfg["lights"], pipe.rows, pipe.frame_h

    if world > 1 and args.composite_overlap == "auto":
        ok = pipe.completes_within(3, WATCHDOG_S)
        if not d.all_true(ok):
            if rank == 0:
                print(f"bench.py: the overlapped composite did not complete within {WATCHDOG_S:.0f} s on every rank: falling back to one stream order", file=sys.stderr)
            comms.fall_back(f"first 3 overlapped steps did not complete within {WATCHDOG_S:.0f} s on every rank")
            pipe.e_comp = [None, None]
            wiring_check(d, comms, ctx)
    overlap = comms.overlap

    # 1. cold start: the very first steps after the idle set-up phase, timed on their own (the chip has not ramped its clocks yet)
    dt_cold = pipe.timed(COLD_STEPS)
    # 2. untimed spin-up: the chip needs ~0.2-0.3 s of sustained load to reach its steady-state clocks, so a fixed number of extra
    #    untimed steps precedes the W warm-up steps whatever W is. The timed region is still exactly K steps.
    for i in range(max(0, SPINUP_STEPS - COLD_STEPS)):
        pipe.step(i)
    pipe.drain()
    # 2b. the post kernels on their own, after the spin-up and before the warm-up: 20 back-to-back launches of each between two events, on the
    #     frame's own buffers (no event, no other kernel in between). In the frame loop each of them follows a kernel that has just filled the
    #     caches with other data, and the per-stage events sit inside the intervals they measure; both figures are reported. (Taken here rather
    #     than after the timed region: after ~0.4 s of sustained shading a burst of X passes runs 2.5-3x slower — the chip's power limiter,
    #     profiles/r2k_frame_loop.md — which says nothing about the kernel.)
    iso = None
    if args.post in ("chain", "fused"):
        iso = {}
        if args.post == "chain":
            st = _stage_stats(lambda: ctx.post_process_tile(pipe.scene[0], F16, R8, out=pipe.sdr[0], halo_top=pipe.halo_top, halo_bottom=pipe.halo_bottom), spin_s=0.01)
            iso.update({"post_chain": st["ms"] * 1e-3, "post_chain_spread": [round(st["ms_min"], 4), round(st["ms_max"], 4)]})
        for name, fn in (("blur_x", lambda: ctx.gaussian_blur_x(pipe.scene[0], F16, out=pipe.xblur)),
                         ("blur_y_tonemap", lambda: ctx.gaussian_blur_y_tonemap(pipe.xblur, F16, R8, out=pipe.sdr[0], halo_top=pipe.halo_top, halo_bottom=pipe.halo_bottom))):
            st = _stage_stats(fn, spin_s=0.01)               # the frame loop has just spun the chip up
            iso[name] = st["ms"] * 1e-3
            iso[name + "_spread"] = [round(st["ms_min"], 4), round(st["ms_max"], 4)]
        pipe.drain()
        d.barrier()
        # the isolated passes above are ~0.5 s of light post-kernel load: the chip leaves the clock / power state the frame loop had reached, and the W warm-up steps
        # (5 for the driver) are too few to bring it back (measured: `value` 6 % under `sustained` without this). Half a spin-up of the frame loop again, untimed.
        for i in range(SPINUP_STEPS // 2):
            pipe.step(i)
        pipe.drain()
    for i in range(args.warmup):
        pipe.step(i)
    pipe.drain()
    # 3. timed region: only the dominant kernel is bracketed by HIP events (2 records per step); the per-stage timings of the
    #    HBM-bound post kernels are taken in a separate, untimed pass afterwards so their instrumentation does not sit in `value`
    evs = [{"t0": _ev(), "shade": _ev()} for _ in range(args.steps)]
    dt = pipe.timed(args.steps, evs)

    n_detail = min(args.steps, 10)
    keys = ["t0", "shade", "x", "halo", "post"] + (["y"] if args.post == "split" else []) + (["comp0", "comp1"] if world > 1 else [])
    evd = [{k: _ev() for k in keys} for _ in range(n_detail)]
    for i in range(n_detail):
        pipe.step(args.steps + i + (args.steps & 1), evd[i])
    pipe.drain()
    d.barrier()
    # the chain as a whole, without an event between its kernels: `shade` after the shade kernel ... `post` after the last post kernel
    evc = [{"shade": _ev(), "post": _ev()} for _ in range(n_detail)]
    for i in range(n_detail):
        pipe.step(args.steps + i + (args.steps & 1), evc[i])
    pipe.drain()
    d.barrier()
    # 4. frame latency: one step at a time, nothing in flight before or after it
    frame_latency = pipe.latency()

    # 4b. sustained run: the same step for ~SUSTAINED_S seconds without interruption — a long-window cross-check of `value` (thousands of steps instead
    #     of K) and a stretch of GPU activity an outside observer (rocm-smi, the driver's clock) can see; never the headline
    sustained = None
    if not args.no_extras and SUSTAINED_S > 0:
        n_sus = int(min(20000, max(200, SUSTAINED_S / (dt / args.steps))))         # the same on every rank: dt is the max over ranks
        n_sus += n_sus & 1                                                         # the double-buffered composite alternates buffers
        dt_sus = pipe.timed(n_sus)
        sustained = {"steps": n_sus, "seconds": round(dt_sus, 4), "ms_per_step": round(dt_sus / n_sus * 1e3, 4),
                     "value": round(W * frame_h * n_sus / dt_sus / 1e6, 2), "unit": "Mpix/s",
                     "note": "same step, same buffers, one barrier + synchronize bracket around all of it; after the timed region (power limiter and clocks in their long-run state)"}

    # 4c. the OTHER form of the post chain (two kernels when the headline runs the one-kernel chain, and vice versa), same frame loop, same clocks: a companion figure
    chain_alt = None
    if args.post in ("chain", "fused") and not args.no_extras:
        mine, other = args.post, ("fused" if args.post == "chain" else "chain")
        args.post = other
        for i in range(10):
            pipe.step(i)
        pipe.drain()
        n_alt = max(20, min(args.steps, 100)); n_alt += n_alt & 1
        dt_alt = pipe.timed(n_alt)
        evc2 = [{"t0": _ev(), "shade": _ev(), "post": _ev()} for _ in range(10)]
        for i in range(10):
            pipe.step(n_alt + i, evc2[i])
        pipe.drain()
        d.barrier()
        args.post = mine
        chain_alt = {"form": other, "steps": n_alt, "ms_per_step": round(dt_alt / n_alt * 1e3, 4), "value": round(W * frame_h * n_alt / dt_alt / 1e6, 2), "unit": "Mpix/s",
                     "shade_ms": round(mean_ms(evc2, "t0", "shade"), 4), "post_chain_ms": round(mean_ms(evc2, "shade", "post"), 4), "bytes_per_px": 12 if other == "chain" else 28,
                     "note": "bench.py --post " + other + ": " + ("blur X, blur Y and the tonemapper in one kernel (8 B read + 4 B written per pixel; row tiles exchange scene-colour halos)"
                                                                 if other == "chain" else "blur X (16 B/px), then blur Y + tonemapper in one kernel (12 B/px); row tiles exchange X-blurred halos") +
                             "; identical bits; measured right after the sustained run (compare with sustained.ms_per_step, not with ms_per_step: profiles/r5g_post_forms.md)"}

    # 4d. N = 1: TWO FRAMES IN FLIGHT (--post-stream frames): frame i on stream i & 1 with its own buffer pair, so that the tail of one frame's shade kernel runs under the head of
    #     the next frame's — the reference's own habit (three back buffers, SwapChain.h). A companion figure, not the headline: with two shade kernels on the chip at once a launch
    #     lasts twice as long, and the per-kernel roofline / rocprofv3 figures of the line would stop describing the kernel
    frames2 = None
    if world == 1 and args.post == "chain" and args.post_stream != "frames" and not args.no_extras:
        mine = args.post_stream
        args.post_stream = "frames"
        for i in range(10):
            pipe.step(i)
        pipe.drain()
        n2 = max(20, min(args.steps, 100)); n2 += n2 & 1
        dt2f = pipe.timed(n2)
        ev2f = [{"t0": _ev(), "post": _ev()} for _ in range(10)]
        for i in range(10):
            pipe.step(n2 + i, ev2f[i])
        pipe.drain()
        d.barrier()                                          # device idle: the events of both frame streams have completed
        args.post_stream = mine
        frames2 = {"steps": n2, "ms_per_step": round(dt2f / n2 * 1e3, 4), "value": round(W * frame_h * n2 / dt2f / 1e6, 2), "unit": "Mpix/s", "frames_in_flight": 2,
                   "frame_interval_ms": round(mean_ms(ev2f, "t0", "post"), 4),
                   "note": "bench.py --post-stream frames: frame i runs shade + post chain on stream i & 1 with its own scene-colour / SDR buffers (no event between the streams); "
                           "frame_interval_ms = first kernel's start to last kernel's end of ONE frame while the other is in flight; identical bytes (VQ_BENCH_VERIFY); measured "
                           "right after other_post_form (compare with sustained.ms_per_step)"}

    verify = None
    if world > 1 and os.environ.get("VQ_BENCH_VERIFY") == "1":
        # debug aid: rank 0 recomputes the WHOLE frame on its own GPU (no tiles, no halos) and compares it byte for byte with the
        # composite of the last step — the row tiling + halo exchange + composite on real kernels (tests/test_gpu_bench_flow.py)
        torch.cuda.synchronize()
        last = pipe.verify_step(4)
        if rank == 0:
            gb_full = upload_tile(cfg, frame_h, 0, frame_h)
            sc = ctx.forward_lighting(gb_full, pipe.pf, pipe.pv, out_fmt=F16, extra_point=pipe.extra, env=pipe.env)
            xb = ctx.gaussian_blur_x(sc, F16)
            want = ctx.gaussian_blur_y_tonemap(xb, F16, R8)
            torch.cuda.synchronize()
            verify = {"mismatching_bytes": int((want != pipe.frame[last]).sum().item()), "frame": [W, frame_h]}
            del gb_full, sc, xb, want
        d.barrier()
    if world == 1 and os.environ.get("VQ_BENCH_VERIFY") == "1":
        # N = 1: the frame loop's own double buffering (and, with --post-stream own, the ordering between the two streams): both buffer pairs zeroed on a drained
        # device, four steps, then each SDR image against the two-kernel chain computed in stream order — a step that read scene colour before its shade kernel had
        # written it, or a shade kernel that overwrote what a post chain was still reading, leaves other bytes
        pipe.drain(); torch.cuda.synchronize()
        for t in pipe.scene + pipe.sdr:
            t.zero_()
        pipe.e_pdone = [None, None]
        torch.cuda.synchronize()
        for i in range(4):
            pipe.step(i)
        pipe.drain(); torch.cuda.synchronize()
        sc = ctx.forward_lighting(pipe.gb, pipe.pf, pipe.pv, out_fmt=F16, extra_point=pipe.extra, env=pipe.env)
        want = ctx.gaussian_blur_y_tonemap(ctx.gaussian_blur_x(sc, F16), F16, R8)
        torch.cuda.synchronize()
        verify = {"mismatching_bytes": int((want != pipe.sdr[0]).sum().item()) + int((want != pipe.sdr[1]).sum().item()), "frame": [W, frame_h], "buffers": 2}
        del sc, want

    # 5. the other Fresnel-pow lowering, same invocation, same clocks: `engine_lowering` is the engine-faithful exp2(5*log2 x) form
    second = None
    if not args.no_second_mode:
        other = "exp2_log2" if args.fresnel_pow == "product" else "product"
        ctx.set_fresnel_pow(other == "exp2_log2")
        for i in range(20):
            pipe.step(i)
        pipe.drain()
        evs2 = [{"t0": _ev(), "shade": _ev()} for _ in range(args.steps)]
        dt2 = pipe.timed(args.steps, evs2)
        ctx.set_fresnel_pow(args.fresnel_pow == "exp2_log2")
        second = {"fresnel_pow": other, "value": round(W * frame_h * args.steps / dt2 / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(dt2 / args.steps * 1e3, 4),
                  "shade_ms": round(mean_ms(evs2, "t0", "shade"), 4)}
    # 5b. the other READING of dot / normalize (vqhip_set_arithmetic): DXC's lowering — FMA-chain dot, v * correctly rounded rsqrt — with the exp2/log2
    # Fresnel power, i.e. the second build of the reference's sources (tests/golden/ref_outputs_dxc.npz); same invocation, same clocks
    dxc = None
    if not args.no_second_mode:
        ctx.set_arithmetic(True); ctx.set_fresnel_pow(True)
        for i in range(20):
            pipe.step(i)
        pipe.drain()
        evs3 = [{"t0": _ev(), "shade": _ev()} for _ in range(args.steps)]
        dt3 = pipe.timed(args.steps, evs3)
        ctx.set_arithmetic(False); ctx.set_fresnel_pow(args.fresnel_pow == "exp2_log2")
        dxc = {"arithmetic": "dxc", "fresnel_pow": "exp2_log2", "value": round(W * frame_h * args.steps / dt3 / 1e6, 2), "unit": "Mpix/s",
               "ms_per_step": round(dt3 / args.steps * 1e3, 4), "shade_ms": round(mean_ms(evs3, "t0", "shade"), 4),
               "note": "vqhip_set_arithmetic(VQHIP_ARITH_DXC) + VQHIP_FRESNEL_POW_EXP2_LOG2: within one RGBA16F ulp of the reference's HLSL in DXC's reading of dot / "
                       "normalize / pow (tests/test_gpu_arith_modes.py); the headline runs the literal reading. normalize = v * rsqrt(dot) with a correctly rounded rsqrt: a binary32 "
                       "second-order sequence from the v_rsq_f32 seed (vq_devmath.h:rsqrt_cr_fast, validated on [2^-100, 2^100]; the binary64 definition is the out-of-range fallback)"}

    # 6. the other BASELINE configs, outside the headline's timed region (every rank takes part in the distributed ones)
    extras = {}
    pf_head, extra_head, pv_head = pipe.pf, pipe.extra, pipe.pv
    if not args.no_extras:
        x_steps = max(4, min(args.steps, 20))
        if args.config != "cfg5":
            pipe.free()
            p5 = Pipeline(ctx, d, comms, CONFIGS["cfg5"], args, composite_root=root)
            extras["cfg5_strong"] = measure_strong(p5, x_steps, 3)
            extras["cfg5_strong"]["workload"] = CONFIGS["cfg5"]["workload"] + f"; {p5.rows} of 4320 rows on each of {world} GPU(s)"
            if world == 1:
                extras["tile_curve"] = tile_curve(ctx, d, comms, args, p5, extras["cfg5_strong"])
            p5.free()
        if world == 1:
            extras["cfg2"] = shade_only(ctx, d, comms, args, CONFIGS["cfg2"], None, 0)
            c2 = load_pmc_constants("cfg2", "product")[0]          # counter-derived: VALU instructions per wave of the cfg2 launch -> the roof that binds there too
            if c2:
                waves2 = CONFIGS["cfg2"]["width"] * CONFIGS["cfg2"]["height"] / 64.0
                t2 = extras["cfg2"]["shade_ms"] * 1e-3
                extras["cfg2"]["valu_issue"] = {"valu_instr_per_wave": c2["valu_instr_per_wave"], "achieved_T_lane_instr_s": round(c2["valu_instr_per_wave"] * 64 * waves2 / t2 / 1e12, 2),
                                                "ceiling": VALU_ISSUE_CEILING_TLIS, "frac": round(c2["valu_instr_per_wave"] * 64 * waves2 / t2 / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                                                "traffic": c2.get("hbm_bytes_per_launch"),
                                                "note": "SQ_INSTS_VALU / SQ_WAVES of the cfg2 launch (profiles/pmc_constants.json) x live kernel time against the fast issue rate (see valu_issue.note)"}
            extras["ibl_load"] = ibl_load_report(ibl_t)
            extras["ibl_load"]["engine_default"] = bl_ibl.engine_default_report(ctx, _stage_stats)      # 4096x2048 .hdr -> 13 mips -> 512^2 x 9: the engine's own default sizes
            # the spot-light + PCF shadow-caster path (A4 / A7): BASELINE cfg1 with its own CPU baseline, and the cbuffer's limits at 4K, on noise and on coherent content
            cas = bl_casters.casters_report(ctx, _stage_stats, HBM_PEAK_GBPS, cores=None if args.no_cpu_baseline else host_cores())
            extras["cfg1"], extras["engine_max"] = cas["cfg1"], cas["engine_max"]
            extras["widened"] = widened_report(ctx, env, pre["spec_mips"])
            if args.config == "cfg3":
                extras["coherent_scene"] = coherent_scene(ctx, d, comms, args, cfg, env, pre["spec_mips"])

    shade_span = [round(v, 4) for v in d.min_max_over_ranks(mean_ms(evs, "t0", "shade"))]      # [min, max] over the ranks' own means: tile imbalance, visible in the first multi-GPU run
    if rank == 0:
        px_tile, px_frame = W * rows, W * frame_h
        t_shade = mean_ms(evs, "t0", "shade") * 1e-3
        if args.post == "split":
            t_blur, t_tm = mean_ms(evd, "shade", "y") * 1e-3, mean_ms(evd, "y", "post") * 1e-3
        else:
            t_blur, t_tm = mean_ms(evd, "shade", "halo") * 1e-3, mean_ms(evd, "halo", "post") * 1e-3
        t_chain = mean_ms(evc, "shade", "post") * 1e-3
        own_stream = args.post == "chain" and world == 1 and args.post_stream == "own"
        t_chain_alone = iso["post_chain"] if (own_stream and iso and "post_chain" in iso) else t_chain
        ach = SHADE_BYTES_PER_PX * px_tile / t_shade / 1e9
        flops_px = 170 * L + 160
        pmc, pmc_meta = load_pmc_constants(args.config + ("_coherent" if args.content == "coherent" else ""), args.fresnel_pow)
        if pmc_meta.get("stale"):
            print(f"bench.py: PMC constants not used: {pmc_meta.get('why')}", file=sys.stderr)
        out = {
            "metric": cfg["metric"] + (" [NOT the BASELINE workload: surface-coherent content]" if args.content == "coherent" else ""),
            "value": round(px_frame * args.steps / dt / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None, "dtype": "f32", "data": "synthetic" if args.content == "noise" else "synthetic-coherent",
            "config": {"workload": cfg["workload"] + ("" if world == 1 else f"; frame {W}x{frame_h} row-tiled {rows} rows per GPU, RCCL p2p halo exchange + composite on "
                                                       f"{'rank 0' if root == 0 else 'every rank'} through the C ABI"),
                       "name": args.config, "width": W, "frame_height": frame_h, "tile_rows": rows, "lights": L, "parallelism": f"rows{world}",
                       "composite_overlap": overlap, "untimed_spinup_steps": SPINUP_STEPS + (SPINUP_STEPS // 2 if args.post in ("chain", "fused") else 0), "fresnel_pow": args.fresnel_pow,
                       "post_stream": ("own: the post chain of frame i runs on a second stream next to the shade kernel of frame i + 1 (double-buffered); frame_latency_ms is one frame alone"
                                       if (args.post == "chain" and world == 1 and args.post_stream == "own") else "main"),
                       "post": {"chain": "blur X, blur Y and the tonemapper in ONE kernel (identical bits to three dispatches)",
                                "fused": "blur X, then blur Y + tonemap in one kernel (identical bits to three dispatches)", "split": "blur X, blur Y, tonemap"}[args.post]},
            "roofline": {"bound": "hbm", "kernel": f"k_forward_lighting<{'env' if cfg['env'] else 'noenv'},nocasters,RGBA16F>", "achieved": round(ach, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                         "traffic": pmc["hbm_bytes_per_launch"] if pmc else None, "traffic_unit": "bytes/launch",
                         "traffic_source": "rocprofv3 PMC, separate passes, 2*FETCH_SIZE + WRITE_SIZE (profiles/pmc_constants.json); algorithmic = %d" % (SHADE_BYTES_PER_PX * px_tile),
                         "bytes_per_px": SHADE_BYTES_PER_PX, "ms": round(t_shade * 1e3, 4),
                         "note": f"{L}-light shading is VALU-bound by construction (SURVEY.md 8d): see valu / valu_issue; the post kernels are under stages, cfg2 is its own object" +
                                 ("; --post-stream own: this kernel shares the chip with the previous frame's post chain, its launch lasts ~2-3 % longer than alone "
                                  "(other_post_form.shade_ms: alone, main stream)" if own_stream else "")},
            "valu": {"achieved_tflops_model": round(flops_px * px_tile / t_shade / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                     "frac": round(flops_px * px_tile / t_shade / 1e12 / VALU_PEAK_TFLOPS, 4), "flops_per_px_model": flops_px},
            "pmc_constants": pmc_meta,
            "rccl": comms.info(),
            "stages": {"shade_Mpix_s": round(px_tile / t_shade / 1e6, 1), "shade_ms": round(t_shade * 1e3, 4),
                       **({"blur_xy_ms": round(t_blur * 1e3, 4), "blur_xy_GBps": round(px_tile * 32 / t_blur / 1e9, 1),
                           "tonemap_ms": round(t_tm * 1e3, 4), "tonemap_GBps": round(px_tile * 12 / t_tm / 1e9, 1)} if args.post == "split" else
                          {"post_chain_ms": round(t_chain * 1e3, 4), "post_chain_in_loop_ms": round(t_chain * 1e3, 4), "post_chain_alone_ms": round(t_chain_alone * 1e3, 4),
                           "post_chain_bytes_per_px": 12,
                           **({"post_chain_co_runs_with": "the shade kernel of the next frame (--post-stream own): post_chain_ms is the stretched interval on the second stream, the *_frac "
                                                          "figures below price the kernel ALONE (stages.isolated.post_chain_ms)"} if own_stream else {}),
                           "post_chain_GBps": round(px_tile * 12 / t_chain_alone / 1e9, 1),
                           "post_chain_frac_of_hbm_peak": round(px_tile * 12 / t_chain_alone / 1e9 / HBM_PEAK_GBPS, 4),
                           "post_chain_frac_at_28_B_per_px": round(px_tile * 28 / t_chain_alone / 1e9 / HBM_PEAK_GBPS, 4),
                           "post_chain_note": "ONE kernel (k_post_chain: blur X, blur Y, tonemapper; 8 B read + 4 B written per pixel, no intermediate image), from the end of the shade "
                                              "kernel to its end inside the frame loop. The kernel is bound by VALU issue (126 mads + 6 conversions + 3 table lookups per pixel), not by "
                                              "HBM: *_frac_at_28_B_per_px prices the same time at the 28 B/px the two-kernel chain of rounds 1-4 moved, for comparison with their figures",
                           "isolated": {"post_chain_ms": round(iso["post_chain"] * 1e3, 4), "post_chain_ms_spread": iso["post_chain_spread"],
                                        "post_chain_frac_of_hbm_peak": round(px_tile * 12 / iso["post_chain"] / 1e9 / HBM_PEAK_GBPS, 4),
                                        "two_kernels": {"blur_x_ms": round(iso["blur_x"] * 1e3, 4), "blur_x_frac_of_hbm_peak": round(px_tile * 16 / iso["blur_x"] / 1e9 / HBM_PEAK_GBPS, 4),
                                                        "blur_y_tonemap_ms": round(iso["blur_y_tonemap"] * 1e3, 4),
                                                        "blur_y_tonemap_frac_of_hbm_peak": round(px_tile * 12 / iso["blur_y_tonemap"] / 1e9 / HBM_PEAK_GBPS, 4),
                                                        "note": "the kernels of --post fused alone (halos of the tiled path here: whatever the buffers hold)"},
                                        "note": "median of 7 batches of back-to-back launches of the one kernel, after the spin-up and before the warm-up steps"}} if args.post == "chain" else
                          {"blur_x_ms": round(t_blur * 1e3, 4), "blur_x_GBps": round(px_tile * 16 / t_blur / 1e9, 1),
                           "blur_y_tonemap_ms": round(t_tm * 1e3, 4), "blur_y_tonemap_GBps": round(px_tile * 12 / t_tm / 1e9, 1),
                           "post_chain_ms": round(t_chain * 1e3, 4), "post_chain_GBps": round(px_tile * 28 / t_chain / 1e9, 1),
                           "post_chain_frac_of_hbm_peak": round(px_tile * 28 / t_chain / 1e9 / HBM_PEAK_GBPS, 4),
                           "post_chain_note": "from the end of the shade kernel to the end of the last post kernel inside the frame loop, no event in between"
                                              " (blur_x_ms / blur_y_tonemap_ms above come from steps that record one)",
                           "isolated": {"blur_x_ms": round(iso["blur_x"] * 1e3, 4), "blur_x_GBps": round(px_tile * 16 / iso["blur_x"] / 1e9, 1),
                                        "blur_x_frac_of_hbm_peak": round(px_tile * 16 / iso["blur_x"] / 1e9 / HBM_PEAK_GBPS, 4),
                                        "blur_y_tonemap_ms": round(iso["blur_y_tonemap"] * 1e3, 4),
                                        "blur_y_tonemap_GBps": round(px_tile * 12 / iso["blur_y_tonemap"] / 1e9, 1),
                                        "blur_y_tonemap_frac_of_hbm_peak": round(px_tile * 12 / iso["blur_y_tonemap"] / 1e9 / HBM_PEAK_GBPS, 4),
                                        "blur_x_ms_spread": iso["blur_x_spread"], "blur_y_tonemap_ms_spread": iso["blur_y_tonemap_spread"],
                                        "note": "median of 7 batches of back-to-back launches of the one kernel (spread = fastest / slowest batch), after the spin-up and before the warm-up steps; "
                                                "the figures above are taken inside the frame loop with an event record between the stages"}}),
                       **({"blur_x_includes": "halo exchange", "composite_ms": round(mean_ms(evd, "comp0", "comp1"), 4)} if world > 1 else {})},
            "frame_latency_ms": round(frame_latency * 1e3, 4),
            **({"sustained": sustained} if sustained else {}),
            **({"other_post_form": chain_alt} if chain_alt else {}),
            **({"two_frames_in_flight": frames2} if frames2 else {}),
            "cold_start": {"steps": COLD_STEPS, "ms_per_step": round(dt_cold / COLD_STEPS * 1e3, 4), "value": round(px_frame * COLD_STEPS / dt_cold / 1e6, 2),
                           "note": "the first steps after the idle set-up phase, before the clocks ramp; `value` is the steady-state figure"},
        }
        if pmc:
            vw, tw = pmc["valu_instr_per_wave"], pmc.get("quarter_rate_instr_per_wave", 0)
            waves = px_tile / 64.0
            out["valu_issue"] = {"achieved_T_lane_instr_s": round(vw * 64 * waves / t_shade / 1e12, 2), "ceiling": VALU_ISSUE_CEILING_TLIS,
                                 "frac": round(vw * 64 * waves / t_shade / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                                 "frac_slot_weighted": round((vw + 3 * tw) * 64 * waves / t_shade / 1e12 / VALU_ISSUE_CEILING_TLIS, 4),
                                 "valu_instr_per_wave": vw, "quarter_rate_instr_per_wave": tw,
                                 "ceiling_slow_class": 35.4,
                                 "note": "the binding roof: VALU instructions issued per second (PMC count x live kernel time) vs the FAST issue rate of the chip (plain fp32 "
                                         "add / mul / fma on registers; scripts/ubench/valu_ceiling.hip, mix_rate.hip). Instructions with an SGPR source, conversions, compares, min / max "
                                         "issue at ceiling_slow_class, v_rcp / v_rsq at a quarter of the fast rate (profiles/r5f_issue_classes.md): the light loop's mix (91 fast, 18 slow, "
                                         "3 transcendental of 112) cannot reach frac 1 whatever its schedule; slot-weighted counts each v_rcp / v_rsq as 4 slots"}
        if second is not None:
            out["engine_lowering" if second["fresnel_pow"] == "exp2_log2" else "product_lowering"] = second
        if dxc is not None:
            out["dxc_lowering"] = dxc
        if verify is not None:
            out["verify"] = verify
        out.update(extras)
        out["stages"]["shade_ms_over_ranks"] = shade_span
        # second-tier kernels inside `roofline` (the driver's record keeps that object whole): algorithmic bytes / live time / 8 TB/s for each, and the VALU fractions of the headline kernel
        others = []
        if iso and "post_chain" in iso:
            others.append({"kernel": "k_post_chain (blur X + blur Y + tonemap, 4K, alone)", "ms": round(iso["post_chain"] * 1e3, 4), "bytes": 12 * px_tile,
                           "frac": round(12 * px_tile / iso["post_chain"] / 1e9 / HBM_PEAK_GBPS, 4)})
        if "cfg2" in extras:
            c2x = extras["cfg2"]
            others.append({"kernel": "k_forward_lighting<noenv,nocasters> cfg2 (1920x1080, 16 lights)", "ms": c2x["shade_ms"], "bytes": SHADE_BYTES_PER_PX * 1920 * 1080, "frac": c2x["hbm_frac"],
                           "valu_issue_frac": c2x.get("valu_issue", {}).get("frac")})
        for key, label in (("cfg1", "k_forward_lighting<noenv,casters> cfg1 (1280x720, Default scene lights, PCF)"), ("engine_max", "k_forward_lighting<noenv,casters> engine_max (4K, 100+20 lights, 5+5+1 casters)")):
            if key in extras:
                others.append({"kernel": label, "ms": extras[key]["shade_ms"], "bytes": SHADE_BYTES_PER_PX * extras[key]["pixels"], "frac": extras[key]["hbm_frac"]})
        if "ibl_load" in extras:
            ib = extras["ibl_load"]
            ltk = load_time_kernel_counters()

            def issue(kernel, ms):                           # PMC wave-instruction count x 64 lanes / live time / the fast issue rate; L1 tag lookups per clock per CU at 2.4 GHz
                c = ltk.get(kernel) if ltk else None
                if not c:
                    return {}
                r = {"valu_issue_frac": round(c["valu_wave_instr_per_launch"] * 64 / (ms * 1e-3) / 1e12 / VALU_ISSUE_CEILING_TLIS, 4)}
                if "tcp_accesses_per_launch" in c:
                    r["l1_lookups_per_clock_per_cu"] = round(c["tcp_accesses_per_launch"] / (ms * 1e-3) / 2.4e9 / 256, 3)      # at the 2.4 GHz peak clock; the L1 does one tag lookup per clock
                return r
            others.append({"kernel": "k_conv_diffuse_ordered (cfg4: 6x64^2 texels x 99 382 taps; L1-tag / VALU-bound, no HBM stream)", "ms": ib["conv_diffuse_ms"], "bytes": None, "frac": None,
                           "valu_frac_model": ib.get("conv_diffuse_valu_frac_model"), **issue("k_conv_diffuse_ordered", ib["conv_diffuse_ms"])})
            others.append({"kernel": "k_conv_specular_ordered (cfg4: 128^2 x 7)", "ms": ib["conv_specular_ms"], "bytes": None, "frac": None, "valu_frac_model": ib.get("conv_specular_valu_frac_model"),
                           **issue("k_conv_specular_ordered", ib["conv_specular_ms"])})
            if "engine_default" in ib:
                others.append({"kernel": "k_conv_specular_ordered (engine default: 512^2 x 9)", "ms": ib["engine_default"]["conv_specular_ms"], "bytes": None, "frac": None})
            others.append({"kernel": "k_brdf_lut (1024^2 x 2048)", "ms": ib["brdf_lut_warm_ms"], "bytes": 4 * 1024 * 1024, "frac": None, "valu_frac_model": ib.get("brdf_lut_warm_valu_frac_model"),
                           **issue("k_brdf_lut", ib["brdf_lut_warm_ms"])})
        out["roofline"]["others"] = others
        out["roofline"]["valu"] = {"frac_spec": out["valu"]["frac"], "frac_issue": out.get("valu_issue", {}).get("frac"), "frac_issue_slot_weighted": out.get("valu_issue", {}).get("frac_slot_weighted"),
                                   "note": "the roof that binds the headline kernel: flop model / 157.3 TFLOP/s, and PMC instruction count x live time / 66.7 T lane-instructions/s"}
        if "coherent_scene" in out:
            cc = load_pmc_constants("cfg3_coherent", args.fresnel_pow)[0]
            if cc:
                out["coherent_scene"].update(traffic=cc.get("hbm_bytes_per_launch"), valu_instr_per_wave=cc["valu_instr_per_wave"],
                                             traffic_ratio_to_algorithmic=round(cc["hbm_bytes_per_launch"] / (SHADE_BYTES_PER_PX * px_tile), 3) if cc.get("hbm_bytes_per_launch") else None)
        if world == 1 and not args.no_cpu_baseline:
            env_np = (pre["diffuse_blurred"].cpu().numpy(), pre["specular"].cpu().numpy(), 128, pre["spec_mips"], lut.cpu().numpy()) if cfg["env"] else None
            if args.fresnel_pow == "exp2_log2":
                from tests import oracle_lib as _O
                _O.load().vqo_set_fresnel_pow(1)             # keeps the cpu_baseline leg on the same arithmetic
            out["cpu_baseline"] = cpu_baseline(cfg, env_np, pf_head, extra_head, pv_head, frame_h)
            try:                                             # optional second baseline: only where oracle/_ref exists
                ref_line = cpu_reference_source(cfg, env_np, pf_head, extra_head, pv_head, frame_h)
                if ref_line is not None:
                    out["cpu_reference_source"] = ref_line
            except Exception as e:                           # never let the optional leg break the bench line
                out["cpu_reference_source"] = {"error": repr(e)[:200]}
        print(json.dumps(finish_line(out)), flush=True)
    if world > 1:
        d.barrier()
        comms.close()
        dist.destroy_process_group()
    ctx.close()


# ---- the extra objects ------------------------------------------------------------------------------------------------------------------
def shade_only(ctx, d, comms, args, cfg, env, max_env_lod, coherent=False):
    """The shade kernel of `cfg` alone (N = 1): back-to-back launches between two events after a spin-up, with both roofs."""
    p = Pipeline(ctx, d, comms, cfg, args, env=env, max_env_lod=max_env_lod, coherent=coherent)
    out_img = p.scene[0]
    fn = lambda i: ctx.forward_lighting(p.gb, p.pf, p.pv, out=out_img, out_fmt=F16, extra_point=p.extra, env=p.env)  # noqa: E731
    st = _stage_stats(lambda: fn(0))
    ms = st["ms"]
    px, L = p.W * p.rows, cfg["lights"]
    res = {"workload": cfg["workload"] + (" [surface-coherent content]" if coherent else ""), "shade_ms": round(ms, 4), "shade_ms_min": round(st["ms_min"], 4),
           "shade_ms_max": round(st["ms_max"], 4), "shade_Mpix_s": round(px / ms / 1e3, 1),
           "hbm_GBps": round(SHADE_BYTES_PER_PX * px / ms / 1e6, 1), "hbm_frac": round(SHADE_BYTES_PER_PX * px / ms / 1e6 / HBM_PEAK_GBPS, 4),
           "valu_frac_model": round((170 * L + 160) * px / ms / 1e9 / VALU_PEAK_TFLOPS, 4), "bytes_per_px": SHADE_BYTES_PER_PX, "flops_per_px_model": 170 * L + 160,
           "note": f"shade kernel only: median of {st['batches']} batches of {st['launches_per_batch']} back-to-back launches after a {st['spinup_launches']}-launch spin-up (>= 0.25 s of load)"}
    if coherent:
        r = p.gb[1][..., 3]
        res["slow_path_pixel_fraction_round2"] = round(float((r < 0.04).float().mean().item()), 4)
    p.free()
    return res


def coherent_scene(ctx, d, comms, args, cfg, env, spec_mips):
    """The headline frame on surface-coherent content (terrain normals, material regions, 12 % polished regions): what real frames cost. The
    white-noise frame of the headline never takes the wave-uniform skip of back-facing lights and has no pixel below roughness 0.04."""
    res = shade_only(ctx, d, comms, args, cfg, env, spec_mips, coherent=True)
    res["note"] = ("the cfg3 shade kernel on synth.gbuffer_rows_coherent; slow_path_pixel_fraction_round2 = pixels with roughness < 0.04, which round 2 sent "
                   "through the IEEE light loop wave by wave; now whole waves choose the EPSILON-select / back-facing-skip forms (shade.hip)")
    return res


if __name__ == "__main__":
    main()
