"""The CPU legs of the line (`cpu_baseline`, `cpu_reference_source`): the oracle and the reference's own shader source on the host cores, on a bounded sample of the
headline's workload. The ONLY place besides tests/ and smoke() where anything under oracle/ is executed — as the reported baseline, never as the thing measured."""
import os
import time

import numpy as np

from vqengine_amd import abi, synth


def host_cores():
    cores = len(os.sched_getaffinity(0))
    try:                                                     # honour a cgroup CPU quota if the box has one
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(cfg, env_np, pf, extra, pv, frame_h, target_s=10.0):
    """The CPU oracle (a scalar C++ port of the HLSL, OpenMP over rows) timed on the host cores on a bounded row
    band of the SAME workload. Reported baseline only — never the thing measured as `value`."""
    from tests import oracle_lib as O
    O.load()
    cores = host_cores()
    W = cfg["width"]
    env = O.host_envmap(*env_np) if env_np is not None else None
    band = 540 if cfg["lights"] <= 64 else 135               # bands of the SAME synthetic frame

    def run(row0, rows):
        gb = synth.gbuffer_rows(W, frame_h, row0, row0 + rows, seed=cfg["seed"])
        t0 = time.perf_counter()
        sc = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, extra_point=extra, env=env, nthreads=cores)
        x = O.blur_pass(sc, abi.FMT_RGBA16F, 0, nthreads=cores)
        y = O.blur_pass(x, abi.FMT_RGBA16F, 1, nthreads=cores)
        O.tonemap(y, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM, nthreads=cores)
        return time.perf_counter() - t0
    run(0, 64)                                               # warm-up (thread pool, page faults)
    t, rows, k = 0.0, 0, 0
    while t < target_s and k < 64:
        t += run((k % 4) * band, band)
        rows += band
        k += 1
    return {"value": round(W * rows / t / 1e6, 4), "unit": "Mpix/s", "cores": int(cores), "kind": "port",
            "sample": f"oracle (scalar C++ port of the HLSL, OpenMP static over rows, {cores} threads) on {k} bands of {W}x{band} rows of the same "
                      f"frame ({W * rows / 1e6:.1f} Mpix): shade {cfg['lights']} lights{' + IBL' if env is not None else ''}, blur X/Y, tonemap; {t:.1f} s"}


def cpu_reference_source(cfg, env_np, pf, extra, pv, frame_h, target_s=6.0):
    """The REFERENCE'S OWN shader source (ForwardLighting.hlsl:PSMain, GaussianBlur.hlsl, Tonemapper.hlsl) run on ALL host cores through oracle/_ref
    (oracle/ref_src/hlsl_shim.h) on bands of rows of the same frame, when that library travelled with the tree. A second reported baseline next to
    `cpu_baseline`: scalar, literal IEEE; one band per worker thread at a time, every thread on its own copy of the library (the translated shaders keep
    their cbuffers in globals; ctypes releases the GIL inside the calls)."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import oracle_lib as O, ref_lib as R
    if not R.available("shaders") or (extra is not None and not R.available("shaders_l256")):
        return None
    W = cfg["width"]
    env = O.host_envmap(*env_np) if env_np is not None else None
    cores = host_cores()
    rows_per = 22 if cfg["lights"] <= 64 else 4              # one blur kernel height: the band is a (small) image of its own

    def band(job):
        worker, k = job
        R.use_private_copy(f"w{worker}")
        r0 = (k * 97) % (frame_h - rows_per)
        gb = synth.gbuffer_rows(W, frame_h, r0, r0 + rows_per, seed=cfg["seed"])
        t0 = time.perf_counter()
        sc = R.forward_from_gbuffer(gb, pf, pv, env=env, extra=extra).astype(np.float16).astype(np.float32)
        x = R.blur_pass(sc, 0).astype(np.float16).astype(np.float32)
        y = R.blur_pass(x, 1).astype(np.float16).astype(np.float32)
        R.tonemap(y, abi.TonemapperParams.default())
        return time.perf_counter() - t0

    def worker_loop(worker):                                 # each worker runs bands until the wall-clock budget is spent
        n, busy = 0, 0.0
        while time.perf_counter() - start < target_s and n < 64:
            busy += band((worker, worker * 64 + n))
            n += 1
        return n, busy
    band((0, 0))                                             # load + first-touch outside the timed window
    start = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(worker_loop, range(cores)))
    wall = time.perf_counter() - start
    bands = sum(n for n, _ in res)
    rows = bands * rows_per
    return {"value": round(W * rows / wall / 1e6, 4), "unit": "Mpix/s", "cores": int(cores), "kind": "reference",
            "sample": f"the reference's HLSL (PSMain {cfg['lights']} lights{' + IBL' if env is not None else ''}, CSMain_X/_Y, tonemapper CSMain) compiled to C++ through "
                      f"oracle/ref_src/hlsl_shim.h, {cores} threads (one private copy of the library each), {bands} bands of {W}x{rows_per} rows of the same frame "
                      f"({W * rows / 1e6:.2f} Mpix); {wall:.1f} s wall, {sum(b for _, b in res):.1f} thread-seconds"}
