"""vqhip_set_arithmetic / vqo_set_arithmetic: the TWO readings of dot / normalize / length / reflect (include/vqhip.h, oracle/vqo_math.h).
CPU side: the oracle in DXC mode (+ the exp2/log2 Fresnel power: together the second build of the reference's sources, hlsl_shim.h VQ_SHIM_DXC) against
the stored outputs of that build, tests/golden/ref_outputs_dxc.npz — within ONE RGBA16F ulp on the four BASELINE-shape bands, like the literal mode
against the literal build (tests/test_ref_fixtures.py); and the correctly rounded rsqrt the reading rests on. The HIP product in that mode:
tests/test_gpu_arith_modes.py."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import ref_cases
from tests.test_ref_readings import distance
from vqengine_amd import abi

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_outputs_dxc.npz")
TAGS = sorted(ref_cases.DXC_SCENES)
# measured (oracle, dxc + exp2_log2): max 1 ulp; fraction of differing channels 3.3e-5 / 6.5e-5 / 4.5e-5 / 1.2e-4 (cfg1 / cfg2 / cfg3 / cfg5)
MAX_FRACTION = 4e-4


class dxc_mode:
    """oracle (and, given a context, the product) in the DXC reading with the exp2/log2 Fresnel power, restored on exit"""
    def __init__(self, ctx=None, fresnel=True):
        self.ctx, self.fresnel = ctx, fresnel

    def __enter__(self):
        lib = O.load()
        lib.vqo_set_arithmetic(1); lib.vqo_set_fresnel_pow(1 if self.fresnel else 0)
        if self.ctx is not None:
            self.ctx.set_arithmetic(True); self.ctx.set_fresnel_pow(self.fresnel)

    def __exit__(self, *a):
        lib = O.load()
        lib.vqo_set_arithmetic(0); lib.vqo_set_fresnel_pow(0)
        if self.ctx is not None:
            self.ctx.set_arithmetic(False); self.ctx.set_fresnel_pow(False)


def boundary_gbuffer(inp):
    """the G-buffer at the product's boundary in the CURRENT reading: plane 1 = normalize(In.WorldSpaceNormal) (ForwardLighting.hlsl:264)"""
    lib = O.load()
    lib.vqo_normalize_lit_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    gb = [g.copy() for g in inp["gb_raw"]]
    n = np.ascontiguousarray(gb[1][..., :3])
    out = np.empty_like(n)
    lib.vqo_normalize_lit_array(n.ctypes.data, out.ctypes.data, n.size // 3)
    gb[1][..., :3] = out
    return gb


def test_rsqrt_cr_is_the_correctly_rounded_reciprocal_square_root():
    lib = O.load()
    lib.vqo_rsqrt_cr_check.restype = C.c_long
    assert lib.vqo_rsqrt_cr_check() == 0          # every significand, both exponent parities, against x87 extended precision
    lib.vqo_rsqrt_cr_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    x = np.array([0.0, -0.0, np.inf, -1.0, np.nan, 1.0, 4.0, 0.25, 1e-45, 3.4e38], np.float32)
    out = np.empty_like(x)
    with np.errstate(all="ignore"):
        lib.vqo_rsqrt_cr_array(x.ctypes.data, out.ctypes.data, x.size)
    assert out[0] == np.inf and out[1] == -np.inf and out[2] == 0 and np.isnan(out[3]) and np.isnan(out[4]) and tuple(out[5:8]) == (1.0, 0.5, 2.0) and np.isfinite(out[8:]).all()


def test_modes_are_two_different_functions():
    build, _, _ = ref_cases.DXC_SCENES["cfg2_band_1920x32"]
    inp = build()
    lit = O.forward_lighting(inp["gb"], inp["pf"], inp["pv"], abi.FMT_RGBA32F, extra_point=inp["extra"])
    with dxc_mode(fresnel=False):
        assert O.load().vqo_get_arithmetic() == 1
        dxc = O.forward_lighting(boundary_gbuffer(inp), inp["pf"], inp["pv"], abi.FMT_RGBA32F, extra_point=inp["extra"])
    assert O.load().vqo_get_arithmetic() == 0
    rel = np.abs(lit[..., :3] - dxc[..., :3]) / np.maximum(np.abs(lit[..., :3]), 1e-6)
    assert 0.2 < np.mean(lit.view(np.uint32) != dxc.view(np.uint32)) and np.median(rel) < 1e-6 and rel.max() < 0.05     # most pixels move in the last bits, highlights by percents


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_in_dxc_mode_matches_the_dxc_build_of_the_reference(tag):
    fixtures = np.load(FIX)
    build, _, _ = ref_cases.DXC_SCENES[tag]
    inp = build()
    assert ref_cases.checksum(inp) == bytes(fixtures[tag + "/scene/inputs"]).decode(), "inputs drifted: rerun tests/golden/make_ref_fixtures.py"
    with dxc_mode():
        sh = ref_cases.host_shadow_dims(inp["shadow"]) if inp["shadow"] is not None else None
        scene = O.forward_lighting(boundary_gbuffer(inp), inp["pf"], inp["pv"], abi.FMT_RGBA16F, extra_point=inp["extra"], env=ref_cases.host_env(inp["env"]), shadow=sh)
    d = distance(scene[..., :3], fixtures[tag + "/scene"])
    assert d["max"] <= 1 and d["frac_gt0"] <= MAX_FRACTION and d["nonfinite_mismatch"] == 0, d
