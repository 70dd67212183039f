"""The timers of the bench line's per-stage figures: HIP events on torch's current stream (every call of the C ABI made through vqengine_amd.capi without a
`stream` argument runs there)."""
import os

import torch


def _ev():
    return torch.cuda.Event(enable_timing=True)


def _time_loop(fn, n, spin):
    for i in range(spin):
        fn(i)
    a, b = _ev(), _ev()
    a.record()
    for i in range(n):
        fn(i)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


STAGE_SPIN_S = float(os.environ.get("VQ_BENCH_STAGE_SPIN_S", "0.25"))     # the chip needs ~0.25 s of sustained load to reach its clocks
STAGE_BATCHES = 7
STAGE_BATCH_S = 0.03


def _stage_stats(fn, spin_s=None, batches=STAGE_BATCHES, batch_s=STAGE_BATCH_S):
    """THE timer of every per-stage figure of the line (widened.*, ibl_load.*_warm_ms, cfg2, coherent_scene, stages.isolated): a spin-up sized by TIME
    (>= STAGE_SPIN_S of back-to-back calls of the same fn), then `batches` back-to-back timed batches of ~batch_s each, every batch between its own
    two HIP events with no host synchronisation in between (all events are recorded first, read afterwards). The figure is the MEDIAN batch; the
    spread (min / max batch) is reported with it."""
    spin_s = STAGE_SPIN_S if spin_s is None else spin_s
    probe = _time_loop(lambda i: fn(), 3, 2)                                   # ms per call, cold: only sizes the loops
    spin = int(min(20000, max(3, spin_s * 1e3 / probe)))
    n = int(min(4000, max(2, batch_s * 1e3 / probe)))
    for _ in range(spin):
        fn()
    ev = [_ev() for _ in range(batches + 1)]
    ev[0].record()
    for b in range(batches):
        for _ in range(n):
            fn()
        ev[b + 1].record()
    ev[-1].synchronize()
    per = sorted(ev[b].elapsed_time(ev[b + 1]) / n for b in range(batches))
    return {"ms": per[len(per) // 2], "ms_min": per[0], "ms_max": per[-1], "batches": batches, "launches_per_batch": n, "spinup_launches": spin}


def _stage_ms(fn):
    return _stage_stats(fn)["ms"]
