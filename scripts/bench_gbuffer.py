#!/usr/bin/env python3
"""Measurement of the G-buffer producer and the skydome on one MI355X: 3840x2160 interpolant planes,
12 materials (up to 7 RGBA8 mip-chained maps each) + SSAO -> the four float4 G-buffer planes. Algorithmic HBM bytes per
pixel: 48 in (3 float4 planes) + 1 (SSAO) + 64 out = 113; textures are cache-resident. Prints one JSON line per variant."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, capi, scene, synth  # noqa: E402


def gpu_ms(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    ctx = capi.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    W, H, NM = 3840, 2160, 12
    px = W * H
    ip = synth.interpolants(W, H, NM)
    ipd = [dev(p) for p in ip]
    ssao = dev(synth.ssao_image(W, H))
    out = tuple(torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4))
    for same in (True, False):
        datas, texsets = synth.material_set(NM, max_dim=1024, same_size=same)
        dmats = (abi.MaterialDesc * NM)()
        keep, host_chains, nmaps = [], [], 0
        for i, (d, ts) in enumerate(zip(datas, texsets)):
            dmats[i].data = d
            cs = {}
            for slot, img in ts.items():
                chain_g, nm = ctx.mip_chain_rgba8(dev(img))
                keep.append(chain_g)
                setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
                cs[slot] = (chain_g.cpu().numpy(), img.shape[1], img.shape[0], nm)
                nmaps += 1
            host_chains.append(cs)
        ms = gpu_ms(lambda: ctx.gbuffer_from_materials(ipd, dmats, 0.055, ssao, out=out))
        rows = 64
        hm = O.host_materials(datas, host_chains)
        t0 = time.perf_counter(); O.gbuffer_from_materials([p[1000:1000 + rows] for p in ip], hm, 0.055, None); tc = time.perf_counter() - t0
        what = "one texture size per material" if same else "random size per map"
        print(json.dumps({"stage": f"F1 G-buffer producer 3840x2160, 12 materials, {nmaps} maps ({what}), SSAO", "units": px, "unit": "pixel",
                          "gpu_ms": round(ms, 4), "M_units_per_s": round(px / ms / 1e3, 1), "algorithmic_GBps": round(px * 113 / ms / 1e6, 1),
                          "hbm_frac": round(px * 113 / ms / 1e6 / 8000.0, 4),
                          "cpu_oracle_M_units_per_s": round(rows * W / tc / 1e6, 3), "cpu_threads": O.load().vqo_max_threads(),
                          "note": f"cpu sample = {rows} rows"}), flush=True)
    # texture-less materials: the pure streaming floor of the kernel
    dm0 = (abi.MaterialDesc * NM)()
    for i, d in enumerate(datas):
        dm0[i].data = d
        dm0[i].data.textureConfig = 0.0
    ms0 = gpu_ms(lambda: ctx.gbuffer_from_materials(ipd, dm0, 0.055, None, out=out))
    print(json.dumps({"stage": "F1 G-buffer producer 3840x2160, texture-less materials", "units": px, "unit": "pixel", "gpu_ms": round(ms0, 4),
                      "M_units_per_s": round(px / ms0 / 1e3, 1), "algorithmic_GBps": round(px * 112 / ms0 / 1e6, 1),
                      "hbm_frac": round(px * 112 / ms0 / 1e6 / 8000.0, 4)}), flush=True)
    # skydome: all-sky frame (worst case) and the sky band of the synthetic view, RGBA16F target, 2048^2 equirect
    eq = dev(synth.equirect(2048, 2048))
    sp = scene.skydome_params(0.9, -0.2, 0.5, 60.0 * math.pi / 180.0, W, H)
    col = torch.zeros((H, W, 4), dtype=torch.float16, device="cuda")
    ms_all = gpu_ms(lambda: ctx.skydome(eq, sp, col, abi.FMT_RGBA16F))
    ms_band = gpu_ms(lambda: ctx.skydome(eq, sp, col, abi.FMT_RGBA16F, coverage_ip=ipd))
    nsky = int((np.ascontiguousarray(ip[2][..., 3]).view(np.int32) < 0).sum())
    print(json.dumps({"stage": "F2 skydome 3840x2160 all-sky, RGBA16F", "units": px, "unit": "pixel", "gpu_ms": round(ms_all, 4),
                      "M_units_per_s": round(px / ms_all / 1e3, 1), "algorithmic_GBps": round(px * 8 / ms_all / 1e6, 1),
                      "hbm_frac": round(px * 8 / ms_all / 1e6 / 8000.0, 4)}), flush=True)
    print(json.dumps({"stage": "F2 skydome 3840x2160 composite over geometry (coverage plane read)", "units": px, "unit": "pixel",
                      "sky_pixels": nsky, "gpu_ms": round(ms_band, 4), "M_units_per_s": round(px / ms_band / 1e3, 1),
                      "algorithmic_GBps": round((px * 16 + nsky * 8) / ms_band / 1e6, 1),
                      "hbm_frac": round((px * 16 + nsky * 8) / ms_band / 1e6 / 8000.0, 4)}), flush=True)


    # FSR 1.0: 2560x1440 RGBA8 (tonemapper output) -> EASU -> 3840x2160 -> RCAS
    iw, ih = 2560, 1440
    src = torch.randint(0, 256, (ih, iw, 4), dtype=torch.uint8, device="cuda")
    up = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    fin = torch.empty_like(up)
    econ, rcon = capi.fsr_easu_con(iw, ih, W, H), capi.fsr_rcas_con(0.2)
    ms_e = gpu_ms(lambda: ctx.fsr_easu(src, abi.FMT_RGBA8_UNORM, W, H, con=econ, out=up))
    ms_r = gpu_ms(lambda: ctx.fsr_rcas(up, abi.FMT_RGBA8_UNORM, con=rcon, out=fin))
    print(json.dumps({"stage": "F4 FSR1 EASU 2560x1440 -> 3840x2160 RGBA8", "units": px, "unit": "output pixel", "gpu_ms": round(ms_e, 4),
                      "M_units_per_s": round(px / ms_e / 1e3, 1), "algorithmic_GBps": round((px * 4 + iw * ih * 4) / ms_e / 1e6, 1),
                      "hbm_frac": round((px * 4 + iw * ih * 4) / ms_e / 1e6 / 8000.0, 4), "note": "VALU-bound: ~450 operations per output pixel"}), flush=True)
    print(json.dumps({"stage": "F4 FSR1 RCAS 3840x2160 RGBA8", "units": px, "unit": "pixel", "gpu_ms": round(ms_r, 4),
                      "M_units_per_s": round(px / ms_r / 1e3, 1), "algorithmic_GBps": round(px * 8 / ms_r / 1e6, 1),
                      "hbm_frac": round(px * 8 / ms_r / 1e6 / 8000.0, 4)}), flush=True)


if __name__ == "__main__":
    main()
