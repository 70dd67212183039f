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
