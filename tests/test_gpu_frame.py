"""End-to-end GPU parity of the widened path (SURVEY.md §8 rows A–D + §8f rows 1–4), every hand-over staying in HBM:

  .hdr file -> level 0 -> min-filter mips -> env-map prefilter + BRDF LUT            (load time)
  interpolant planes + materials -> G-buffer -> forward lighting + IBL -> skydome
  -> blur X/Y -> tonemap (RGBA8) -> FSR EASU 1.5x -> FSR RCAS                          (per frame)

against the same chain evaluated by the CPU oracle: identical bits in the final RGBA8 image and in every intermediate."""
import math

import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from tests.test_gpu_gbuffer import build_materials, dev
from vqengine_amd import abi, capi, scene, synth

pytestmark = pytest.mark.gpu


def test_full_frame_chain_matches_oracle(ctx):
    W, H, NM = 256, 144, 5
    EW, EH = 128, 64
    # ---- load time
    hdr = synth.hdr_file_bytes(synth.float_to_rgbe(synth.equirect(EW, EH)[..., :3]))
    eq_o = O.hdr_decode(hdr)
    eq_g = ctx.load_hdr(hdr)
    chain_o, n = O.mip_chain(eq_o)
    chain_g, n_g = ctx.mip_chain(eq_g)
    pre_o = O.envmap_prefilter(chain_o, EW, EH, n, 16, 0.05, 32, abi.CONV_SEQUENTIAL)
    pre_g = ctx.envmap_prefilter(chain_g, EW, EH, n_g, 16, 0.05, 32, abi.CONV_SEQUENTIAL)
    lut_o, lut_g = O.brdf_lut(64, 128, abi.FMT_RG16F), ctx.brdf_lut(64, 128, abi.FMT_RG16F)
    env_o = O.host_envmap(pre_o["diffuse_blurred"], pre_o["specular"], 32, pre_o["spec_mips"], lut_o)
    env_g = capi.make_envmap(pre_g["diffuse_blurred"], pre_g["specular"], 32, pre_g["spec_mips"], lut_g)
    # ---- frame inputs
    ip = synth.interpolants(W, H, NM)
    datas, host_chains, hmats, dmats, keep = build_materials(ctx, NM, max_dim=128)
    ssao = synth.ssao_image(W, H)
    pf, extra = synth.per_frame(points=synth.point_lights(12, seed=0xF00D), spots=synth.spot_lights(2, seed=0xF00D),
                                directional=synth.directional_light(), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=pre_o["spec_mips"])
    sp = scene.skydome_params(0.6, -0.15, 0.3 / (2 * math.pi) * 2 * math.pi, 60.0 * math.pi / 180.0, W, H)
    F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
    # ---- oracle chain
    gb_o = O.gbuffer_from_materials(ip, hmats, pf.fAmbientLightingFactor, ssao)
    col_o = O.forward_lighting(gb_o, pf, pv, F16, env=env_o)
    col_o = O.skydome(eq_o, sp, col_o, F16, ip[2])
    sdr_o = O.tonemap(O.gaussian_blur(col_o, F16), F16, R8)
    OW, OH = W * 3 // 2, H * 3 // 2
    fin_o = O.fsr_rcas(O.fsr_easu(sdr_o, R8, OW, OH), R8)
    # ---- product chain
    ipd = [dev(p) for p in ip]
    gb_g = ctx.gbuffer_from_materials(ipd, dmats, pf.fAmbientLightingFactor, dev(ssao))
    col_g = ctx.forward_lighting(gb_g, pf, pv, out_fmt=F16, env=env_g)
    col_g = ctx.skydome(eq_g, sp, col_g, F16, coverage_ip=ipd)
    sdr_g = ctx.tonemap(ctx.gaussian_blur(col_g, F16), F16, R8)
    fin_g = ctx.fsr_rcas(ctx.fsr_easu(sdr_g, R8, OW, OH), R8)
    torch.cuda.synchronize()
    for name, g, o in (("hdr level 0", eq_g, eq_o), ("scene colour", col_g, col_o), ("sdr", sdr_g, sdr_o), ("final", fin_g, fin_o)):
        n_bad, idx = O.bits_equal(g.cpu().numpy(), o)
        assert n_bad == 0, (name, n_bad, idx)
    sky = np.ascontiguousarray(ip[2][..., 3]).view(np.int32) < 0
    assert sky.sum() > 1000 and np.isfinite(col_o[sky].astype(np.float32)).all()       # sky pixels were overwritten (no NaN records left)
    assert fin_o.shape == (OH, OW, 4) and fin_o[..., :3].std() > 5


def test_ssr_frame_chain_matches_oracle(ctx):
    """The frame with reflections on, in the engine's order (SceneRendering.cpp:563-755): Z pre-pass normals -> lit draws as ONE PSMain kernel with the extra render
    targets bound (EnvironmentMapDiffuseOnlyIllumination = 1: the specular IBL comes from the reflections pass, :464) -> SSR's environment fallback -> CompositeReflections
    (with a light-bounds image) -> blur -> tonemap; and the debug views of the intermediate targets. Product chain == oracle chain, every intermediate, bit for bit."""
    W, H, NM = 192, 96, 5
    F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
    eq = synth.equirect(128, 64)
    chain_o, n = O.mip_chain(eq)
    chain_g, n_g = ctx.mip_chain(dev(eq))
    pre_o = O.envmap_prefilter(chain_o, 128, 64, n, 16, 0.05, 32, abi.CONV_SEQUENTIAL)
    pre_g = ctx.envmap_prefilter(chain_g, 128, 64, n_g, 16, 0.05, 32, abi.CONV_SEQUENTIAL)
    lut_o, lut_g = O.brdf_lut(64, 128, abi.FMT_RG16F), ctx.brdf_lut(64, 128, abi.FMT_RG16F)
    env_o = O.host_envmap(pre_o["diffuse_blurred"], pre_o["specular"], 32, pre_o["spec_mips"], lut_o)
    env_g = capi.make_envmap(pre_g["diffuse_blurred"], pre_g["specular"], 32, pre_g["spec_mips"], lut_g)
    ip = synth.interpolants(W, H, NM)
    _, _, hmats, dmats, keep = build_materials(ctx, NM, max_dim=128)
    ssao = synth.ssao_image(W, H)
    cur, prev = synth.clip_positions(W, H)
    pf, extra = synth.per_frame(points=synth.point_lights(9, seed=0xBEE), hdri_offset=0.3)
    pv = synth.per_view(W, H, max_env_lod=pre_o["spec_mips"])
    pv.EnvironmentMapDiffuseOnlyIllumination = 1
    _, depth, _, _ = synth.ssr_surfaces(W, H, seed=5)
    cb = synth.ssr_constants(W, H, pre_o["spec_mips"], hdri_yaw=0.3)
    r = np.random.default_rng(2)
    bv = np.zeros((H, W, 4), np.float16)                     # light bounds drawn over a corner of the frame (RenderLightBounds: rasterised, the caller's)
    bv[10:40, 20:90] = (0.1, 0.8, 0.2, 0.35)
    bv[..., 3] *= (r.random((H, W)) > 0.2)
    # ---- oracle chain
    nrm_o = O.scene_normals_from_materials(ip, hmats)
    gb_o = O.gbuffer_from_materials([p.copy() for p in ip], hmats, pf.fAmbientLightingFactor, ssao)
    col_o = O.forward_lighting(gb_o, pf, pv, F16, env=env_o)
    alb_o, mv_o = O.psmain_extra_targets(gb_o, cur, prev)
    rad_o = O.ssr_environment_fallback(col_o, F16, depth, nrm_o, abi.FMT_R10G10B10A2_UNORM, cb, env_o, F16)
    comp_o = O.composite_reflections(rad_o, col_o, F16, bv)
    sdr_o = O.tonemap(O.gaussian_blur(comp_o, F16), F16, R8)
    # ---- product chain
    ipd = [dev(p) for p in ip]
    nrm_g = ctx.scene_normals_from_materials(ipd, dmats)
    col_g, alb_g, mv_g = ctx.forward_lighting_from_materials_mrt(ipd, dmats, pf, pv, motion_fmt=abi.FMT_RG16F, sv_curr=dev(cur), sv_prev=dev(prev), ssao=dev(ssao), env=env_g)
    rad_g = ctx.ssr_environment_fallback(col_g, F16, dev(depth), nrm_g, abi.FMT_R10G10B10A2_UNORM, cb, env_g, F16)
    lit_g = col_g.clone()
    comp_g = ctx.composite_reflections(rad_g, col_g, F16, dev(bv))
    sdr_g = ctx.tonemap(ctx.gaussian_blur(comp_g, F16), F16, R8)
    torch.cuda.synchronize()
    idx = ipd[2][..., 3].contiguous().view(torch.int32).cpu().numpy()
    holes = ~((idx >= 0) & (idx < NM))                       # pixels no fragment reached: the one-kernel PSMain leaves SV_TARGET1 / the motion vectors at their clear value 0,
    alb_o[holes] = 0                                         # like the rasteriser (the G-buffer form of the oracle chain writes every pixel)
    mv_o[holes] = 0
    for name, g, o in (("scene normals", nrm_g.cpu().numpy().view(np.uint32), nrm_o), ("scene colour", lit_g, col_o), ("albedo / metalness", alb_g, alb_o), ("motion vectors", mv_g, mv_o),
                       ("reflection radiance", rad_g, rad_o), ("composite", comp_g, comp_o), ("sdr", sdr_g, sdr_o)):
        n_bad, idx = O.bits_equal(g.cpu().numpy() if hasattr(g, "cpu") else g, o)
        assert n_bad == 0, (name, n_bad, idx)
    # the debug views read the targets as they are (SceneRendering.cpp:2555-2566)
    for src_g, src_o, fmt, p in ((nrm_g, nrm_o, abi.FMT_R10G10B10A2_UNORM, abi.VizParams(2, 1, 1.0)), (mv_g, mv_o, abi.FMT_RG16F, abi.VizParams(8, 0, 30.0)),
                                 (alb_g, alb_o, F16, abi.VizParams(6, 0, 1.0)), (rad_g, rad_o, F16, abi.VizParams(7, 0, 1.0)), (lit_g, col_o, F16, abi.VizParams(3, 0, 1.0))):
        n_bad, idx = O.bits_equal(ctx.visualize(src_g, fmt, p, R8).cpu().numpy(), O.visualize(src_o, fmt, p, R8))
        assert n_bad == 0, (fmt, p.iDrawMode, n_bad, idx)
    assert (rad_o[..., :3].astype(np.float32).sum(-1) > 0).mean() > 0.3 and (comp_o != col_o).any()


def test_hdri_downsize_matches_oracle_and_reports_unsupported(ctx):
    """vqhip_hdr_downsize_rgba32f: the engine's 8k -> 4k / 2k / 1k fallback shapes at 1/8 scale (1024x512 -> 512x256 / 256x128 / 128x64) bit for
    bit against the oracle; any non-integer or anisotropic ratio is VQHIP_ERR_UNSUPPORTED, not an approximation."""
    import torch
    from tests import oracle_lib as O
    from vqengine_amd import abi, capi, synth
    img = synth.equirect(1024, 512)
    dev = torch.from_numpy(img).cuda()
    for k in (1, 2, 4, 8):
        got = ctx.hdr_downsize(dev, 1024 // k, 512 // k).cpu().numpy()
        n, idx = O.bits_equal(got, O.hdr_downsize(img, 1024 // k, 512 // k))
        assert n == 0, (k, n, idx)
    for ow, oh in ((1000, 500), (512, 128), (2048, 1024)):
        with pytest.raises(capi.VQHipError) as e:
            ctx.hdr_downsize(dev, ow, oh)
        assert e.value.code == abi.VQHIP_ERR_UNSUPPORTED
