"""The engine's DEFAULT image-based-lighting configuration end to end (tests/engine_default.py: Data/EngineSettings.ini:11 EnvironmentMapResolution=512 -> a 512^2,
9-mip specular cube from a 2:1 4096 x 2048 .hdr; EnvironmentMapRendering.cpp:55-63,431-435; EnvironmentMap.cpp:164-165) against tests/golden/engine_default.npz
(tests/golden/make_engine_default_fixture.py: the reference's own HLSL on ~2 100 texels of the cube and on a shade band, the oracle's full cubes as sha256).

CPU: the oracle's specular pass on the sampled texels of all 9 mips (the 2^2 one included) vs the reference's output, in RGBA16F ulps.
GPU: .hdr -> vqhip_hdr_decode_rgba32f -> 13-level min chain -> diffuse 64^2 @ 0.010 -> specular 512^2 x 9 through the C ABI: every intermediate and BOTH WHOLE CUBES equal
the oracle's bit for bit (sha256), the sampled texels and the shade band (MaxEnvMapLODLevels = 9: the mip-offset arithmetic of vq_shade.h / conv.hip past 7 mips) within one
ulp of the reference's outputs, the band bit-exact against the oracle."""
import os

import numpy as np
import pytest

from tests import engine_default as E
from tests import oracle_lib as O
from tests import ref_cases
from vqengine_amd import abi

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "engine_default.npz")
# measured: the oracle (= the HIP product, bit for bit) differs from the reference's HLSL in 0.15 % of the 6 603 sampled channels, never by more than one ulp (mips 5-8: identical);
# the band in 3e-5 of its channels. No filter-step tail on this environment (cfg4's 26 channels sit next to 2.6e4-radiance suns on a 128^2 cube)
SPEC_TOL = ("ulp16", 1, 0.01)
BAND_TOL = ("ulp16", 1, 5e-4)


@pytest.fixture(scope="module")
def fx():
    z = np.load(FIX)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def hdr_bytes(fx):
    data = E.hdr_file()
    assert E.sha(np.frombuffer(data, np.uint8)) == bytes(fx["sha_file"]).decode(), "the synthetic .hdr drifted from the one the fixture was made with"
    return data


def _sha(fx, k):
    return bytes(fx["sha_" + k]).decode()


def test_oracle_specular_sample_over_nine_mips(fx, hdr_bytes):
    img = O.hdr_decode(hdr_bytes)
    assert img.shape == (E.H0, E.W0, 4) and E.sha(img) == _sha(fx, "image")
    chain, n = O.mip_chain(img)
    assert n == 13 and E.sha(chain) == _sha(fx, "chain")
    tex = E.sample_texels()
    assert np.array_equal(tex, fx["spec_sample_idx"]) and len(tex) >= 2000
    per_mip = [len(g[4]) for g in E.split_texels(tex)]
    assert len(per_mip) == 9 and min(per_mip) >= 24 and per_mip[8] == 24          # every mip, all 24 texels of the 2^2 one
    got = O.conv_specular_texels(chain, E.W0, E.H0, n, E.SPEC_RES0, abi.CONV_SEQUENTIAL, tex)[:, :3]
    ref_cases.check("engine_default_specular_512x9_sample (oracle)", got, fx["spec_sample_ref"], SPEC_TOL)


@pytest.mark.gpu
def test_hip_engine_default_ibl_end_to_end(fx, hdr_bytes, ctx):
    import torch
    img = ctx.load_hdr(hdr_bytes)
    assert tuple(img.shape) == (E.H0, E.W0, 4) and E.sha(img.cpu().numpy()) == _sha(fx, "image"), "decoded .hdr differs from the oracle's"
    chain, n = ctx.mip_chain(img)
    assert n == 13 and E.sha(chain.cpu().numpy()) == _sha(fx, "chain"), "13-level min-filter chain differs from the oracle's"
    pre = ctx.envmap_prefilter(chain, E.W0, E.H0, n, E.DIFF_RES, E.DIFF_STEP, E.SPEC_RES0, abi.CONV_SEQUENTIAL)
    assert pre["spec_mips"] == E.SPEC_MIPS == 9
    spec = pre["specular"].cpu().numpy()
    assert spec.shape[0] == abi.cube_px(E.SPEC_RES0, E.SPEC_MIPS)
    assert E.sha(pre["diffuse_blurred"].cpu().numpy()) == _sha(fx, "diffuse"), "blurred diffuse cube (64^2, from the 4096 x 2048 chain) differs from the oracle's"
    assert E.sha(spec) == _sha(fx, "specular"), "the 512^2 x 9-mip specular cube differs from the oracle's (every texel is compared)"
    # the sampled texels vs the REFERENCE'S HLSL
    ref_cases.check("engine_default_specular_512x9_sample (HIP)", spec[fx["spec_sample_idx"]][:, :3], fx["spec_sample_ref"], SPEC_TOL)
    # the shade band: MaxEnvMapLODLevels = 9, mip offsets up to 8 * (512^2 - 2^2) texels
    lut = ctx.brdf_lut(1024, 2048, abi.FMT_RG16F)
    assert np.array_equal(lut.cpu().numpy().view(np.uint16), ref_cases.cfg4_env()["lut"].view(np.uint16))
    from vqengine_amd import capi
    env_g = capi.make_envmap(pre["diffuse_blurred"], pre["specular"], E.SPEC_RES0, E.SPEC_MIPS, lut)
    _, gb, pf, extra, pv = E.band_inputs()
    got = ctx.forward_lighting([torch.from_numpy(g).cuda() for g in gb], pf, pv, out_fmt=abi.FMT_RGBA16F, extra_point=extra, env=env_g).cpu().numpy()
    ref_cases.check("engine_default_band_3840x24 (HIP vs the reference's PSMain)", got[..., :3], fx["band_ref"], BAND_TOL)
    diff_h, lut_h = pre["diffuse_blurred"].cpu().numpy(), lut.cpu().numpy()      # host_envmap keeps POINTERS: the arrays must outlive the oracle call
    env_o = O.host_envmap(diff_h, spec, E.SPEC_RES0, E.SPEC_MIPS, lut_h)
    want = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, extra_point=extra, env=env_o)
    nbad, idx = O.bits_equal(got, want)
    assert nbad == 0, f"engine-default shade band differs from the oracle in {nbad} elements, first {idx.tolist()}"
    used = np.unique(np.clip((gb[1][..., 3] * np.float32(E.SPEC_MIPS)).astype(np.int32), 0, E.SPEC_MIPS - 1))
    assert set(used.tolist()) == set(range(E.SPEC_MIPS)), "the band did not read every mip"
