#!/usr/bin/env python3
"""Randomised parity of the load-time IBL path (min-filter mip chain, diffuse irradiance, GGX specular prefilter, face blur, BRDF LUT):
python tests/fuzz/fuzz_ibl.py [--seconds 120] [--seed 1] (needs a GPU; the oracle is the checker).

Every case draws the equirect's size (2:1, square, tall; 8 .. 512 wide), its content (smooth sky, white noise, a few suns of 1e4 .. 6e4, zeros, a few non-finite texels), the cube
sizes (diffuse 1 .. 16, specular 4 .. 64 (the product takes powers of two >= 4) with every mip), the integration step, the summation order (the reference's / wave-parallel), the kernel-form options
(diffuse_form, diffuse_seq_form, specular_form, lut_form) and the output format, and runs ONE of: the product's whole prefilter call, the diffuse pass, the specular pass, the mip
chain, the LUT (random size / sample count / rows). The HIP product's bits must equal the oracle's. Exit status 1 if a case differed."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))          # the fuzzers import each other

from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, synth  # noqa: E402

F32, F16 = abi.FMT_RGBA32F, abi.FMT_RGBA16F


def equirect(r, w, h):
    kind = int(r.integers(0, 4))
    if kind == 0:
        eq = synth.equirect(w, h, seed=int(r.integers(0, 1 << 30)))
    elif kind == 1:
        eq = r.random((h, w, 4), dtype=np.float32) * np.float32(r.choice([1.0, 50.0, 6e4]))
    elif kind == 2:
        eq = np.full((h, w, 4), np.float32(r.choice([0.0, 0.25, 1.0])), np.float32)
        for _ in range(int(r.integers(1, 5))):
            eq[int(r.integers(0, h)), int(r.integers(0, w)), :3] = np.float32(r.choice([1e4, 6e4, 3e38, 1e-30]))
    else:
        yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        eq = np.stack([0.5 + 0.5 * np.sin(xx * 0.3 + c) * np.cos(yy * 0.2) for c in range(4)], axis=-1).astype(np.float32)
    eq = np.ascontiguousarray(eq, np.float32)
    eq[..., 3] = 1.0
    if r.random() < 0.15:
        n = int(r.integers(1, 4))
        eq[r.integers(0, h, n), r.integers(0, w, n), r.integers(0, 3, n)] = r.choice(np.array([np.inf, -np.inf, np.nan, -1.0, -0.0], np.float32), n)
    return eq


def case(seed):
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0xF3]))
    w, h = [(8, 8), (16, 8), (32, 16), (64, 32), (64, 64), (128, 64), (256, 128), (512, 256), (32, 64), (128, 128)][int(r.integers(0, 10))]
    what = str(r.choice(["prefilter", "diffuse", "diffuse", "specular", "specular", "mips", "lut"]))
    return dict(w=w, h=h, eq=equirect(r, w, h), what=what, dres=int(r.choice([1, 2, 3, 4, 8, 16])), step=float(r.choice([0.5, 0.25, 0.1, 0.07, 0.05])),
                sres=int(r.choice([4, 8, 16, 32, 64])), order=int(r.choice([abi.CONV_SEQUENTIAL, abi.CONV_WAVE64])), fmt=int(r.choice([F16, F32])),
                diffuse_form=r.choice([None, None, "texels", "general"]), diffuse_seq_form=r.choice([None, None, "lane"]), specular_form=r.choice([None, None, "general"]),
                lut_form=r.choice([None, "general"]), lut_size=int(r.choice([1, 2, 7, 16, 33, 64])), lut_samples=int(r.choice([1, 2, 16, 64, 100, 512])),
                lut_fmt=int(r.choice([abi.FMT_RG16F, abi.FMT_RG32F])) if hasattr(abi, "FMT_RG32F") else abi.FMT_RG16F)


def run_case(ctx, seed, dev):
    c = case(seed)
    what = (f"seed {seed}: {c['what']} equirect {c['w']}x{c['h']} diffuse {c['dres']} step {c['step']} specular {c['sres']} order {c['order']} fmt {c['fmt']} forms "
            f"{c['diffuse_form']}/{c['diffuse_seq_form']}/{c['specular_form']}/{c['lut_form']} lut {c['lut_size']}x{c['lut_samples']} fmt {c['lut_fmt']}")
    opts = {k: c[k] for k in ("diffuse_form", "diffuse_seq_form", "specular_form", "lut_form")}
    for k, v in opts.items():
        ctx.set_option(k, None if v is None else str(v))
    try:
        with np.errstate(all="ignore"):
            if c["what"] == "lut":
                want = [O.brdf_lut(c["lut_size"], c["lut_samples"], c["lut_fmt"])]
                got = [ctx.brdf_lut(c["lut_size"], c["lut_samples"], c["lut_fmt"])]
            else:
                chain_o, n = O.mip_chain(c["eq"])
                chain_g, n_g = ctx.mip_chain(dev(c["eq"]))
                assert n == n_g
                if c["what"] == "mips":
                    want, got = [chain_o], [chain_g]
                elif c["what"] == "diffuse":
                    want = [O.conv_diffuse(chain_o, c["w"], c["h"], n, c["dres"], c["step"], c["order"], c["fmt"])]
                    got = [ctx.conv_diffuse(chain_g, c["w"], c["h"], n, c["dres"], c["step"], c["order"], c["fmt"])]
                elif c["what"] == "specular":
                    want = [O.conv_specular(chain_o, c["w"], c["h"], n, c["sres"], c["order"], c["fmt"])[0]]
                    got = [ctx.conv_specular(chain_g, c["w"], c["h"], n, c["sres"], c["order"], c["fmt"])[0]]
                else:
                    po = O.envmap_prefilter(chain_o, c["w"], c["h"], n, c["dres"], c["step"], c["sres"], c["order"])
                    pg = ctx.envmap_prefilter(chain_g, c["w"], c["h"], n, c["dres"], c["step"], c["sres"], c["order"])
                    keys = ("diffuse_unblurred", "diffuse_blurred", "specular")
                    want, got = [po[k] for k in keys], [pg[k] for k in keys]
    finally:
        for k in opts:
            ctx.set_option(k, None)
    bad, first = 0, []
    for a, b in zip(got, want):
        a = a.cpu().numpy()
        n_, idx = O.bits_equal(a.reshape(np.asarray(b).shape), np.asarray(b))
        bad += n_
        if n_ and not len(first):
            first = idx
    return bad, first, what, got, want


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
            print(f"MISMATCH {what}: {bad} channels, first at {np.asarray(idx).tolist()[:2]}", flush=True)
        n += 1
    print(f"fuzz_ibl: {n} cases, {len(fails)} failed {fails[:20]}", flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
