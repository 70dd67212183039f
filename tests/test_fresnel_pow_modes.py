"""vqhip_set_fresnel_pow: pow(1 - cos, 5.0) of the Fresnel terms as the product (default, contract v4) or as exp2(5*log2 x) — the engine's own
DXC lowering. Both modes have a self-made golden (tests/golden/make_golden.py); the HIP path is bit-exact in both."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests.golden import make_golden
from vqengine_amd import abi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def exp2_log2_oracle():
    lib = O.load()
    lib.vqo_set_fresnel_pow(1)
    yield
    lib.vqo_set_fresnel_pow(0)


def test_oracle_mode1_reproduces_its_golden(exp2_log2_oracle):
    got = make_golden.shade_small()
    want = np.load(os.path.join(GOLD, "shade_small_exp2log2.npz"))
    for k in ("noenv_rgba32f", "env_rgba16f"):
        n, where = O.bits_equal(got[k], want[k])
        assert n == 0, (k, n, where)


def test_the_two_forms_differ_by_ulps_and_by_the_nan_corner():
    now = np.load(os.path.join(GOLD, "shade_small.npz"))["noenv_rgba32f"]
    v3 = np.load(os.path.join(GOLD, "shade_small_exp2log2.npz"))["noenv_rgba32f"]
    both = np.isfinite(now) & np.isfinite(v3)
    rel = np.abs(now[both].astype(np.float64) - v3[both]) / np.maximum(np.abs(v3[both]), 1e-6)
    assert rel.max() < 2e-6 and (now != v3).any()            # a few binary32 ulps of a sum over 28 lights
    assert np.isfinite(now).all()                            # the product form never manufactures a NaN from dot(H,V) = 1 + ulp


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 0])
def test_hip_path_matches_oracle_in_both_modes(ctx, mode):
    import torch
    lib = O.load()
    try:
        lib.vqo_set_fresnel_pow(mode)
        ctx.set_fresnel_pow(bool(mode))
        W, H, gb, pf, extra = make_golden.shade_inputs()
        eq, chain, n, pre, lut = make_golden.ibl_inputs()
        lut_g = ctx.brdf_lut(lut.shape[0], 64, abi.FMT_RG16F)       # the LUT integrates the same Fresnel term (BRDF.hlsl:274)
        lut_o = O.brdf_lut(lut.shape[0], 64, abi.FMT_RG16F)
        assert O.bits_equal(lut_g.cpu().numpy(), lut_o)[0] == 0
        env_o = O.host_envmap(pre["diffuse_blurred"], pre["specular"], 16, pre["spec_mips"], lut_o)
        from vqengine_amd import capi
        d, s = torch.from_numpy(pre["diffuse_blurred"]).cuda(), torch.from_numpy(pre["specular"]).cuda()
        env_g = capi.make_envmap(d, s, 16, pre["spec_mips"], lut_g)
        pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
        ref = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F, env=env_o)
        got = ctx.forward_lighting([torch.from_numpy(g).cuda() for g in gb], pf, pv, out_fmt=abi.FMT_RGBA32F, env=env_g).cpu().numpy()
        n_bad, where = O.bits_equal(got, ref)
        assert n_bad == 0, (mode, n_bad, where)
    finally:
        lib.vqo_set_fresnel_pow(0)
        ctx.set_fresnel_pow(False)
    assert ctx.lib.vqhip_set_fresnel_pow(ctx._h, 7) == abi.VQHIP_ERR_INVALID_ARG
