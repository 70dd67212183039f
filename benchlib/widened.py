"""`widened`: the kernels of SURVEY.md 8f (the callers and data formats either side of the hot path) at 4K, outside the headline's timed region."""
import math
import time

import numpy as np
import torch

from vqengine_amd import abi, capi, scene as scene_mod, synth

from .consts import F16, HBM_PEAK_GBPS, R8, VALU_PEAK_TFLOPS
from .timing import _stage_stats


def widened_report(ctx, env, spec_mips):
    """The kernels of SURVEY.md 8f (the callers and data formats either side of the hot path) at 4K on this GPU, each with its algorithmic HBM bytes
    per pixel and the fraction of the 8 TB/s spec they amount to; VALU-bound ones say so. Inputs: 540-row synthetic bands tiled to 2160 rows."""
    W, H, NM, BAND = 3840, 2160, 12, 540
    px = W * H
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    tile = lambda a: np.tile(a, (H // BAND,) + (1,) * (a.ndim - 1))   # noqa: E731
    res = {"frame": [W, H], "note": "every figure: >= 0.25 s spin-up of the one call, then the median of 7 back-to-back batches of ~30 ms (ms_min / ms_max = the fastest / slowest batch); *_hbm_frac = algorithmic bytes / time / 8 TB/s. producer / "
                                    "skydome / RCAS / SSR fallback are HBM-shaped; the fused PSMain and EASU are VALU-bound (see their notes)"}

    def entry(st, bytes_px, **kw):
        ms = st["ms"]
        return dict(ms=round(ms, 4), ms_min=round(st["ms_min"], 4), ms_max=round(st["ms_max"], 4), batches=st["batches"], launches_per_batch=st["launches_per_batch"],
                    Mpix_s=round(px / ms / 1e3, 1), bytes_per_px=bytes_px, GBps=round(px * bytes_px / ms / 1e6, 1),
                    hbm_frac=round(px * bytes_px / ms / 1e6 / HBM_PEAK_GBPS, 4), **kw)
    # ---- 8f.1: G-buffer producer, alone and fused with the lighting (PSMain as one kernel)
    ipd = [dev(tile(p_)) for p_ in synth.interpolants(W, BAND, NM)]
    ssao = dev(tile(synth.ssao_image(W, BAND)))
    datas, texsets = synth.material_set(NM, max_dim=1024, same_size=False)
    dmats, dm0, keep, nmaps = (abi.MaterialDesc * NM)(), (abi.MaterialDesc * NM)(), [], 0
    for i, (dd, ts) in enumerate(zip(datas, texsets)):
        dmats[i].data = dd
        dm0[i].data = dd
        dm0[i].data.textureConfig = 0.0
        for slot, img in ts.items():
            chain_g, nm = ctx.mip_chain_rgba8(dev(img))
            keep.append(chain_g)
            setattr(dmats[i], slot, abi.Texture2D(chain_g.data_ptr(), img.shape[1], img.shape[0], nm, 0))
            nmaps += 1
    gb = tuple(torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4))
    res["gbuffer_producer_textured"] = entry(_stage_stats(lambda: ctx.gbuffer_from_materials(ipd, dmats, 0.055, ssao, out=gb)), 113,
                                             what=f"vqhip_gbuffer_from_materials: 3 interpolant planes + SSAO -> 4 float4 planes, {NM} materials, {nmaps} RGBA8 mip-chained maps (cache resident)")
    res["gbuffer_producer_textureless"] = entry(_stage_stats(lambda: ctx.gbuffer_from_materials(ipd, dm0, 0.055, None, out=gb)), 112,
                                                what="the same call with texture-less materials: the streaming floor of the kernel")
    pf, extra = synth.per_frame(points=synth.point_lights(64, seed=0x6400), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=spec_mips)
    scene = capi.empty_image(H, W, F16, ctx.device)
    st_f = _stage_stats(lambda: ctx.forward_lighting_from_materials(ipd, dmats, pf, pv, ssao=ssao, out=scene, out_fmt=F16, extra_point=extra, env=env))
    ms = st_f["ms"]
    res["psmain_fused"] = entry(st_f, 57, valu_frac_model=round((170 * 64 + 160) * px / ms / 1e9 / VALU_PEAK_TFLOPS, 4),
                                what="vqhip_forward_lighting_from_materials: PSMain as ONE kernel (producer + 64 point lights + IBL), 48 + 1 B in, 8 B out; VALU-bound like the headline's shade kernel")
    # the same draw with its other render targets bound (OUTPUT_ALBEDO + OUTPUT_MOTION_VECTORS, ForwardLighting.hlsl:382-389), and the Z pre-pass's normals
    svc, svp = (dev(tile(a_)) for a_ in synth.clip_positions(W, BAND))
    tg, _alb, _mv = ctx._psmain_targets(H, W, F16, abi.FMT_RG16F, svc, svp)
    st_m = _stage_stats(lambda: ctx.forward_lighting_from_materials(ipd, dmats, pf, pv, ssao=ssao, out=scene, out_fmt=F16, extra_point=extra, env=env, _targets=tg))
    ms_mrt = st_m["ms"]
    res["psmain_fused_mrt"] = entry(st_m, 101, extra_ms_over_psmain_fused=round(ms_mrt - ms, 4),
                                    extra_ms_spread=[round(st_m["ms_min"] - st_f["ms_max"], 4), round(st_m["ms_max"] - st_f["ms_min"], 4)],
                                    what="vqhip_forward_lighting_from_materials_mrt: the same kernel also writing SV_TARGET1 (albedo, metalness: RGBA16F) and the motion "
                                         "vectors (RG16F, from two float4 clip-position planes): + 32 B in, + 12 B out per pixel")
    nrm = torch.empty((H, W), dtype=torch.int32, device="cuda")
    res["scene_normals_prepass"] = entry(_stage_stats(lambda: ctx.scene_normals_from_materials(ipd, dmats, out=nrm)), 52,
                                         what="vqhip_scene_normals_from_materials (DepthPrePass.hlsl:PSMain): 3 interpolant planes -> Tex_SceneNormals R10G10B10A2, normal maps "
                                              "(+ diffuse alpha of masked materials) cache resident: 48 B in, 4 B out")
    del gb, ipd, ssao, keep, svc, svp, _alb, _mv, nrm
    # ---- 8f.2: skydome, all-sky frame (worst case), RGBA16F target, 2048^2 equirect
    eq = dev(synth.equirect(2048, 2048))
    sp = scene_mod.skydome_params(0.9, -0.2, 0.5, 60.0 * math.pi / 180.0, W, H)
    res["skydome_all_sky"] = entry(_stage_stats(lambda: ctx.skydome(eq, sp, scene, F16)), 8, what="vqhip_skydome over a frame without geometry: 8 B/px written, equirect taps cache resident")
    # ---- 8f.3: Radiance .hdr ingest, 2048^2 (256 run-length coded rows, repeated): host header parse + walk of the run headers, device run expansion + RGBE -> RGBA32F
    rgbe = synth.float_to_rgbe(synth.equirect(2048, 256)[..., :3])
    part = synth.hdr_file_bytes(rgbe)
    body = part[part.index(b"+X 2048\n") + 8:]
    data = part[:part.index(b"-Y ")] + b"-Y 2048 +X 2048\n" + body * 8
    ctx.load_hdr(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ctx.load_hdr(data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    res["hdr_decode_2048"] = {"ms": round(ms, 3), "file_bytes": len(data), "Mpix_s": round(2048 * 2048 / ms / 1e3, 1),
                              "kernel_ms": None,
                              "what": "vqhip_hdr_decode_rgba32f, wall time of the call + stream sync: host walk of the run headers (count bytes only), upload of the encoded file from "
                                      "pageable memory, ONE kernel that expands the runs (a workgroup per scanline, a wave per byte plane, through LDS) and converts RGBE -> RGBA32F"}
    # ---- 8f.4: FSR 1.0 (2560x1440 -> 3840x2160, RGBA8), SSR environment fallback
    iw, ih = 2560, 1440
    src = torch.randint(0, 256, (ih, iw, 4), dtype=torch.uint8, device="cuda")
    up, fin = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda"), torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    econ, rcon = capi.fsr_easu_con(iw, ih, W, H), capi.fsr_rcas_con(0.2)
    res["fsr_easu_1440p_to_4k"] = entry(_stage_stats(lambda: ctx.fsr_easu(src, R8, W, H, con=econ, out=up)), round(4 + iw * ih * 4 / px, 2),
                                        what="vqhip_fsr_easu RGBA8 -> RGBA8; VALU-bound: ~700 VALU per output pixel (profiles/r3i_conv_kernels.md addendum)")
    res["fsr_rcas_4k"] = entry(_stage_stats(lambda: ctx.fsr_rcas(up, R8, con=rcon, out=fin)), 8, what="vqhip_fsr_rcas RGBA8 -> RGBA8; VALU-bound: ~300 VALU per pixel")
    sc, depth, packed, _ = synth.ssr_surfaces(W, BAND)
    scd, dpd, nmd = dev(tile(sc.astype(np.float16))), dev(tile(depth)), dev(tile(packed.view(np.int32)))
    cb = synth.ssr_constants(W, H, spec_mips)
    rad = capi.empty_image(H, W, F16, ctx.device)
    res["ssr_env_fallback_4k"] = entry(_stage_stats(lambda: ctx.ssr_environment_fallback(scd, F16, dpd, nmd, abi.FMT_R10G10B10A2_UNORM, cb, env, F16, out=rad)), 24,
                                       what="vqhip_ssr_environment_fallback on white-noise surfaces (72 % of the pixels take the fallback): 8 + 4 + 4 B in, 8 B out; "
                                            "fractional-LOD seamless cube fetch + LUT per pixel, cache resident")
    return res
