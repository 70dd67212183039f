"""The spot-light + PCF shadow-caster path of the shade kernel, timed on the two workloads that exercise it:

  cfg1        BASELINE config 1: 1280x720, the Default scene's light set (Data/Levels/Default.xml:202-308 — a shadowing directional light and two spot
              CASTERS, every light with its own view-projection matrix, Light.cpp:133-233) against shadow maps of the engine's sizes (SceneRendering.cpp:439-441)
  engine_max  3840x2160, the most lights the cbuffer holds (LightingConstantBufferData.h:39-44): 100 point + 20 spot lights, directional shadowing,
              5 spot casters (5 x 1024^2), 5 point casters (5 x 6 x 1024^2), directional 2048^2

k_forward_lighting<noenv, casters, RGBA16F> — Lighting.hlsl:57-73,110-272,323-333, ForwardLighting.hlsl:314-377. Outside the headline's timed region."""
import time

import numpy as np
import torch

from vqengine_amd import abi, scene, synth

F16 = abi.FMT_RGBA16F
BYTES_PER_PX = 64 + 8
WORKLOADS = {
    "cfg1": dict(width=1280, height=720, seed=0xC0FFEE, frame=scene.default_scene_frame,
                 workload="BASELINE cfg1: 1280x720 float4 G-buffer, Default scene lights (directional shadowing 2048^2 + 2 spot casters 1024^2, 5x5 PCF each) -> RGBA16F"),
    "engine_max": dict(width=3840, height=2160, seed=0x6400, frame=scene.engine_max_frame,
                       workload="engine limits: 3840x2160, 100 point + 20 spot lights, directional shadowing (2048^2, 5x5 PCF), 5 spot casters (1024^2, 5x5 PCF), "
                                "5 point casters (6 x 1024^2, 20-tap PCF) -> RGBA16F"),
}


def light_counts(pf):
    L = pf.Lights
    return {"point": int(L.numPointLights), "spot": int(L.numSpotLights), "point_casters": int(L.numPointCasters), "spot_casters": int(L.numSpotCasters),
            "directional": int(L.directional.enabled), "directional_shadowing": int(L.directional.shadowing)}


def device_gbuffer(name, coherent=False):
    w = WORKLOADS[name]
    W, H = w["width"], w["height"]
    gb = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    for r in range(0, H, 240):
        part = (synth.gbuffer_rows_coherent if coherent else synth.gbuffer_rows)(W, H, r, min(r + 240, H), seed=w["seed"])
        for k in range(4):
            gb[k][r:r + part[k].shape[0]].copy_(torch.from_numpy(part[k]))
    return gb


def device_inputs(name, gb=None, coherent=False):
    """(gb planes on the device, PerFrameData, PerViewLightingData, abi.ShadowMaps, keep-alive list, host maps)"""
    w = WORKLOADS[name]
    pf, maps = w["frame"]()
    keep = [torch.from_numpy(maps[k]).cuda() for k in ("dir", "spot", "point")]
    sm = abi.ShadowMaps(keep[0].data_ptr(), maps["dims"][0], keep[1].data_ptr(), maps["dims"][1], keep[2].data_ptr(), maps["dims"][2])
    if gb is None:
        gb = device_gbuffer(name, coherent)
    return gb, pf, synth.per_view(w["width"], w["height"]), sm, keep, maps


def cpu_baseline_cfg1(cores, target_s=8.0):
    """cfg1 is the configuration BASELINE defines as the CPU reference (plumbing): the oracle (scalar C++ port of the HLSL, OpenMP over rows) on bands of the same frame."""
    from tests import oracle_lib as O
    O.load()
    w = WORKLOADS["cfg1"]
    W, H = w["width"], w["height"]
    pf, maps = w["frame"]()
    sm = scene.shadow_maps_struct(maps, lambda a: a.ctypes.data)
    pv = synth.per_view(W, H)
    band = 240

    def run(r0):
        gb = synth.gbuffer_rows(W, H, r0, r0 + band, seed=w["seed"])
        t0 = time.perf_counter()
        O.forward_lighting(gb, pf, pv, F16, shadow=sm, nthreads=cores)
        return time.perf_counter() - t0
    run(0)
    t, rows, k = 0.0, 0, 0
    while t < target_s and k < 4000:
        t += run((k % 3) * band)
        rows += band
        k += 1
    return {"value": round(W * rows / t / 1e6, 4), "unit": "Mpix/s", "cores": int(cores), "kind": "port",
            "sample": f"oracle (scalar C++ port of the HLSL, OpenMP static over rows, {cores} threads): {k} bands of {W}x{band} rows of the cfg1 frame "
                      f"({W * rows / 1e6:.1f} Mpix), forward lighting with the Default scene's lights and PCF casters; {t:.1f} s"}


def casters_report(ctx, stage_stats, hbm_peak_gbps, gb_4k=None, cores=None):
    out = {}
    for name, w in WORKLOADS.items():
        gb, pf, pv, sm, keep, _ = device_inputs(name, gb_4k if name == "engine_max" else None)
        W, H = w["width"], w["height"]
        img = torch.empty((H, W, 4), dtype=torch.float16, device="cuda")
        st = stage_stats(lambda: ctx.forward_lighting(gb, pf, pv, out=img, out_fmt=F16, shadow=sm))
        px, ms = W * H, st["ms"]
        out[name] = {"workload": w["workload"], "lights": light_counts(pf), "pixels": px, "shade_ms": round(ms, 4), "shade_ms_min": round(st["ms_min"], 4), "shade_ms_max": round(st["ms_max"], 4),
                     "shade_Mpix_s": round(px / ms / 1e3, 1), "bytes_per_px": BYTES_PER_PX, "hbm_GBps": round(BYTES_PER_PX * px / ms / 1e6, 1),
                     "hbm_frac": round(BYTES_PER_PX * px / ms / 1e6 / hbm_peak_gbps, 4),
                     "kernel": "k_forward_lighting<noenv,casters,RGBA16F>"}
        if name == "engine_max":                         # the same lights on surface-coherent positions / normals: whole waves skip spots and back-facing casters
            gbc = device_gbuffer(name, coherent=True)
            stc = stage_stats(lambda: ctx.forward_lighting(gbc, pf, pv, out=img, out_fmt=F16, shadow=sm))
            out[name]["coherent_content"] = {"shade_ms": round(stc["ms"], 4), "shade_Mpix_s": round(px / stc["ms"] / 1e3, 1),
                                             "note": "synth.gbuffer_rows_coherent instead of the white-noise G-buffer; the shadow maps' fetch addresses follow the positions"}
            del gbc
        del gb, keep, img
        torch.cuda.empty_cache()
    if cores:
        out["cfg1"]["cpu_baseline"] = cpu_baseline_cfg1(cores)
    return out
