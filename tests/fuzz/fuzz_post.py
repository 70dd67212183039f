#!/usr/bin/env python3
"""Randomised parity of the post chain (21-tap blur X, blur Y, tonemapper): python tests/fuzz/fuzz_post.py [--seconds 120] [--seed 1] (needs a GPU; the oracle is the checker).

Every case draws the image size (1 x 1 .. a few hundred squared for the two-kernel forms, and frames of >= 2^20 pixels with widths that are no multiple of the 64-column strips
for the one-kernel chain), the content (smooth, white noise, constant blocks; magnitudes up to the fp16 maximum; a few non-finite, negative, zero and subnormal channels — the
zero-weight outer taps of the kernel turn an infinity 10 pixels away into NaN), the tonemapper's parameters (every display curve, both colour spaces, gamma on / off, brightness),
input / output formats, the form option (one kernel / two kernels, strip heights) and whether the image is a row tile with scene-colour halos above / below. The HIP product's
bytes must equal the oracle's. Exit status 1 if a case differed."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))          # the fuzzers import each other

from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi  # noqa: E402

F32, F16, R8 = abi.FMT_RGBA32F, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
SPECIALS16 = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 65504.0, -65504.0, 6e-8, -6e-8, 6.1e-5, -1.0, 1e-3], np.float16)


def content(r, h, w):
    kind = r.integers(0, 4)
    scale = np.float32(r.choice([0.05, 1.0, 20.0, 3000.0, 60000.0]))
    if kind == 0:
        img = r.random((h, w, 4), dtype=np.float32) * scale
    elif kind == 1:
        yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        img = np.stack([(0.5 + 0.5 * np.sin(xx * 0.07 + c) * np.cos(yy * 0.05 - c)) * scale for c in range(4)], axis=-1).astype(np.float32)
    elif kind == 2:
        bs = int(r.choice([1, 3, 8, 21, 64]))
        blocks = r.random(((h + bs - 1) // bs, (w + bs - 1) // bs, 4), dtype=np.float32) * scale
        img = np.repeat(np.repeat(blocks, bs, axis=0), bs, axis=1)[:h, :w]
    else:
        img = np.zeros((h, w, 4), np.float32)
        n = max(1, h * w // 50)
        img[r.integers(0, h, n), r.integers(0, w, n), r.integers(0, 4, n)] = r.random(n, dtype=np.float32) * scale
    img = np.ascontiguousarray(img).astype(np.float16)
    if r.random() < 0.6:
        n = int(r.integers(1, 12))
        img[r.integers(0, h, n), r.integers(0, w, n), r.integers(0, 4, n)] = r.choice(SPECIALS16, n)
    if r.random() < 0.2:                                             # specials on the borders and corners (the clamp repeats them)
        for y, x in ((0, 0), (h - 1, w - 1), (0, w - 1), (h - 1, 0), (h // 2, 0), (0, w // 2)):
            if r.random() < 0.5:
                img[y, x, int(r.integers(0, 3))] = r.choice(SPECIALS16)
    return img


def case(seed):
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF2]))
    if r.random() < 0.25:                                            # frames of >= 2^20 pixels: the one-kernel chain
        w = int(r.choice([1024, 1088, 1100, 1999, 2048, 2600, 3840, 4097]))
        h = (1 << 20) // w + 1 + int(r.integers(0, 90))
    else:
        w = int(r.choice([1, 2, 3, 9, 10, 11, 20, 21, 22, 63, 64, 65, 100, 255, 256, 257, 300, 1023, 1024, 1025, 1045, 2050]))
        h = int(r.choice([1, 2, 3, 9, 10, 11, 15, 16, 17, 20, 21, 35, 36, 37, 64, 100, 135, 270]))
    tile = r.random() < 0.35 and h >= 1
    hr = int(r.choice([10, 11, 16])) if tile else 0
    halos = (bool(r.integers(0, 2)), bool(r.integers(0, 2))) if tile else (False, False)
    img = content(r, h + 2 * hr, w)
    curve = int(r.choice([abi.DISPLAY_CURVE_SRGB, abi.DISPLAY_CURVE_SRGB, abi.DISPLAY_CURVE_ST2084, abi.DISPLAY_CURVE_LINEAR, 7]))
    params = abi.TonemapperParams(int(r.integers(0, 2)), curve, float(r.choice([80.0, 200.0, 1000.0, 10000.0, 0.0])), int(r.integers(0, 2)))
    in_fmt = F16 if r.random() < 0.8 else F32
    out_fmt = R8 if r.random() < 0.8 else F16
    form = r.choice(["default", "default", "chain", "two"])
    strips = int(r.choice([0, 0, 1, 3, 7]))
    return dict(w=w, h=h, hr=hr, halos=halos, img=img, params=params, in_fmt=in_fmt, out_fmt=out_fmt, form=str(form), strips=strips, tile=tile, blur=r.random() < 0.95)


def run_case(ctx, seed, dev):
    c = case(seed)
    hr, h = c["hr"], c["h"]
    img = c["img"] if c["in_fmt"] == F16 else c["img"].astype(np.float32)
    p = c["params"]
    what = (f"seed {seed}: {c['w']}x{h} tile {c['tile']} halo rows {hr} {c['halos']} in {c['in_fmt']} out {c['out_fmt']} form {c['form']} strips {c['strips']} "
            f"curve {p.OutputDisplayCurveEnum} space {p.ContentColorSpaceEnum} nits {p.DisplayReferenceBrightnessLevel} gamma {p.ToggleGammaCorrection} blur {c['blur']}")
    ctx.set_option("post_form", None if c["form"] == "default" else c["form"])
    ctx.set_option("post_strips", c["strips"] if c["strips"] else None)
    try:
        if c["tile"]:
            lo = 0 if c["halos"][0] else hr                          # the rows the oracle sees: the tile, plus the halos that exist (a missing one = the frame's border: clamp)
            hi = h + 2 * hr if c["halos"][1] else h + hr
            with np.errstate(all="ignore"):
                full = O.tonemap(O.gaussian_blur(img[lo:hi], c["in_fmt"]), c["in_fmt"], c["out_fmt"], params=p)
            want = full[hr - lo:hr - lo + h]
            top = dev(img[:hr].copy()) if c["halos"][0] else None
            bottom = dev(img[hr + h:].copy()) if c["halos"][1] else None
            got = ctx.post_process_tile(dev(img[hr:hr + h].copy()), c["in_fmt"], c["out_fmt"], params=p, halo_top=top, halo_bottom=bottom).cpu().numpy()
        else:
            with np.errstate(all="ignore"):
                src = O.gaussian_blur(img, c["in_fmt"]) if c["blur"] else img
                want = O.tonemap(src, c["in_fmt"], c["out_fmt"], params=p)
            got = ctx.post_process(dev(img), c["in_fmt"], c["out_fmt"], params=p, blur=c["blur"]).cpu().numpy()
    finally:
        ctx.set_option("post_form", None)
        ctx.set_option("post_strips", None)
    n, idx = O.bits_equal(got, want)
    return n, idx, what, got, want


def main():
    import torch
    from vqengine_amd import capi
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: by time)")
    a = ap.parse_args()
    ctx = capi.Context(0)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    t0, n, fails = time.time(), 0, []
    while (a.cases and n < a.cases) or (not a.cases and time.time() - t0 < a.seconds):
        seed = a.seed * 1000003 + n
        bad, idx, what, got, want = run_case(ctx, seed, dev)
        if bad:
            fails.append(seed)
            y, x = (int(v) for v in idx[0][:2])
            print(f"MISMATCH {what}: {bad} channels, first at (y {y}, x {x}): got {got[y, x]} want {want[y, x]}", flush=True)
        n += 1
    print(f"fuzz_post: {n} cases, {len(fails)} failed {fails[:20]}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
