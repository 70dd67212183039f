#!/usr/bin/env python3
"""One-off, larger than the committed fixtures: oracle vs the reference's own PSMain (oracle/_ref, this container only) over MANY bands of the
BASELINE frames, in RGBA16F storage ulps. Prints max ulp / differing fraction per config. Usage: python scripts/ulp_sweep.py [bands] [rows]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib as O, ref_cases, ref_lib as R  # noqa: E402
from vqengine_amd import abi, synth  # noqa: E402


def main():
    bands = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    env = ref_cases.cfg4_env()
    for name, W, H, L, seed, use_env in (("cfg3", 3840, 2160, 64, 0x6400, True), ("cfg2", 1920, 1080, 16, 0x1600, False), ("cfg5", 7680, 4320, 256, 0x2560, False)):
        pf, extra = synth.per_frame(points=synth.point_lights(L, seed=seed), hdri_offset=0.3 if use_env else 0.0)
        pv = synth.per_view(W, H, max_env_lod=env["spec_mips"] if use_env else 0)
        e = ref_cases.host_env(env) if use_env else None
        mx, nd, n1, n, nnan = 0, 0, 0, 0, 0
        for b in range(bands):
            r0 = (b * (H - rows)) // max(bands - 1, 1)
            raw, gb = ref_cases.band_gbuffer(W, H, r0, rows, seed)
            ref = R.forward_from_gbuffer(raw, pf, pv, env=e, extra=extra)[..., :3]
            got = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, extra_point=extra, env=e)[..., :3]
            nan = ~np.isfinite(ref).all(-1)          # the reference's own NaN pixels: pow(1 - dot(H,V), 5) = exp2(5*log2(negative)) where the dot rounds above 1
            nnan += int(nan.sum())
            assert np.isfinite(got.astype(np.float32)[nan]).all()       # the product's x*((x*x)*(x*x)) is finite there
            d = ref_cases.ulp16_distance(got, ref)[~nan]
            mx = max(mx, int(d.max())); nd += int((d > 0).sum()); n1 += int((d > 1).sum()); n += d.size
        print(f"{name}: {bands} bands x {rows} rows = {n // 3 / 1e6:.2f} Mpix: max {mx} ulp, channels differing {nd / n:.2e}, > 1 ulp {n1 / n:.2e}; "
              f"{nnan} pixels are NaN in the REFERENCE (pow of a negative base) and finite here", flush=True)


if __name__ == "__main__":
    main()
