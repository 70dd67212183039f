#!/usr/bin/env python3
"""Generates the committed golden fixtures tests/golden/*.npz from the CPU ORACLE on seeded synthetic inputs.

The reference (vilbeyli/VQEngine) holds no golden vectors for this path and cannot run here, so
these fixtures pin OUR oracle (and, through the -m gpu tests, the HIP kernels) rather than the reference's bits:
these are SELF-MADE goldens (they guard against drift); the fixtures made from the reference's own sources are
tests/golden/ref_outputs.npz (make_ref_fixtures.py). Regenerate with:  python -m tests.golden.make_golden
Each function returns {name: array}; inputs are regenerated from seeds by the tests, only outputs are stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, synth  # noqa: E402


def ibl_inputs():
    eq = synth.equirect(64, 32)
    chain, n = O.mip_chain(eq)
    pre = O.envmap_prefilter(chain, 64, 32, n, 8, 0.1, 16, abi.CONV_SEQUENTIAL)
    lut = O.brdf_lut(32, 64, abi.FMT_RG16F)
    return eq, chain, n, pre, lut


def ibl_small():
    eq, chain, n, pre, lut = ibl_inputs()
    seq = O.conv_diffuse(chain, 64, 32, n, 4, 0.1, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F)
    return {"mip_tail": chain[64 * 32:], "diffuse_unblurred": pre["diffuse_unblurred"], "diffuse_blurred": pre["diffuse_blurred"],
            "specular": pre["specular"], "lut": lut, "diffuse_sequential_4": seq}


def shade_inputs():
    W, H = 128, 16
    gb = synth.gbuffer(W, H, seed=0x601D)
    pf, extra = synth.per_frame(points=synth.point_lights(24, seed=0x601D), spots=synth.spot_lights(3, seed=0x601D),
                                directional=synth.directional_light(), hdri_offset=0.3)
    return W, H, gb, pf, extra


def shade_small():
    W, H, gb, pf, extra = shade_inputs()
    eq, chain, n, pre, lut = ibl_inputs()
    env = O.host_envmap(pre["diffuse_blurred"], pre["specular"], 16, pre["spec_mips"], lut)
    pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
    return {"noenv_rgba32f": O.forward_lighting(gb, pf, synth.per_view(W, H), abi.FMT_RGBA32F),
            "env_rgba16f": O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, env=env)}


def post_inputs():
    return synth.hdr_image(96, 40, seed=0x905).astype(np.float16)


def post_small():
    img = post_inputs()
    bl = O.gaussian_blur(img, abi.FMT_RGBA16F)
    return {"blur_rgba16f": bl, "sdr_rgba8": O.tonemap(bl, abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM),
            "pq_rgba16f": O.tonemap(bl, abi.FMT_RGBA16F, abi.FMT_RGBA16F, abi.TonemapperParams(0, abi.DISPLAY_CURVE_ST2084, 200.0, 1))}


def gbuffer_inputs():
    """Interpolant planes 96x48, 4 materials with <= 64^2 maps, SSAO. Returns (ip, datas, level-0 texture sets, ssao)."""
    W, H, NM = 96, 48, 4
    datas, texsets = synth.material_set(NM, seed=0x601D, max_dim=64)
    return synth.interpolants(W, H, NM, seed=0x601D), datas, texsets, synth.ssao_image(W, H, seed=0x601D)


def gbuffer_small():
    ip, datas, texsets, ssao = gbuffer_inputs()
    chains = []
    for ts in texsets:
        cs = {}
        for slot, img in ts.items():
            chain, nm = O.mip_chain_rgba8(img)
            cs[slot] = (chain, img.shape[1], img.shape[0], nm)
        chains.append(cs)
    gb = O.gbuffer_from_materials(ip, O.host_materials(datas, chains), 0.055, ssao)
    first = next(iter(chains[0].values()))
    return {"gb0": gb[0], "gb1": gb[1], "gb2": gb[2], "gb3": gb[3], "mat0_first_chain_tail": first[0][first[1] * first[2]:]}


def shade_small_exp2log2():
    """shade_small with the Fresnel pow evaluated as exp2(5*log2 x), the engine's own DXC lowering (vqo_set_fresnel_pow(1))"""
    lib = O.load()
    lib.vqo_set_fresnel_pow(1)
    try:
        return shade_small()
    finally:
        lib.vqo_set_fresnel_pow(0)


if __name__ == "__main__":
    for fn in (ibl_small, shade_small, shade_small_exp2log2, post_small, gbuffer_small):
        out = fn()
        path = os.path.join(HERE, fn.__name__ + ".npz")
        np.savez_compressed(path, **out)
        print(path, {k: (v.shape, str(v.dtype)) for k, v in out.items()}, os.path.getsize(path), "bytes")
