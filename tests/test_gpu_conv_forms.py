"""GPU parity of the round-3 forms of the load-time kernels (conv.hip). Each fast form is the same operations with range tests / branches
removed where they cannot fire; the general form stays in the binary (VQHIP_*_FORM=general) and is what runs wherever a precondition
fails. Checked here, through the C ABI:
  * every form gives identical bits on whole outputs (fast == general == the round-1/2 per-sample kernel), at the reference's sizes;
  * the fast forms == the CPU oracle on inputs built to hit the rare paths: pole texels (odd cube resolution: a NaN frame), taps on the
    equirect seam and the poles (zero x / z components, |y| = 1), non-power-of-two chains (fast tap disabled), chains shorter than the
    sampled mip level, sample counts that are not a multiple of the 512-sample table, both Fresnel lowerings."""
import numpy as np
import pytest
import torch

from tests import oracle_lib as O
from tests.test_gpu_parity import assert_bits, dev
from vqengine_amd import abi, synth

pytestmark = pytest.mark.gpu


def _same(a, b, what):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    n, idx = O.bits_equal(a, b)
    assert n == 0, f"{what}: {n} of {a.size} elements differ, first at {idx.tolist()}"


@pytest.mark.parametrize("size,samples,fmt", [(1024, 2048, abi.FMT_RG16F), (96, 700, abi.FMT_RG32F), (33, 100, abi.FMT_RG32F)])
def test_brdf_lut_forms_identical(ctx, set_opt, size, samples, fmt):
    """Shared-H table + unchecked sample body == the same kernel with every range test left in (option lut_form = general), in both Fresnel-pow modes.
    (The per-sample kernel of rounds 1-2 gave the same bits until it was removed in round 4.)"""
    for explog in (False, True):
        ctx.set_fresnel_pow(explog)
        try:
            set_opt("lut_form", None)
            fast = ctx.brdf_lut(size, samples, fmt)
            set_opt("lut_form", "general")
            _same(ctx.brdf_lut(size, samples, fmt), fast, f"BRDF LUT {size}^2 x {samples} fast vs general, exp2/log2 Fresnel {explog}")
        finally:
            ctx.set_fresnel_pow(False)


@pytest.mark.parametrize("size,samples", [(33, 100), (64, 513), (16, 1500)])
def test_brdf_lut_fast_vs_oracle_ragged(ctx, size, samples):
    """Sizes / sample counts off the 256-texel block and the 512-sample table, against the CPU oracle."""
    assert_bits(ctx.brdf_lut(size, samples, abi.FMT_RG32F), O.brdf_lut(size, samples, abi.FMT_RG32F), f"BRDF LUT {size}^2 x {samples}")


def _chain(w, h, seed=0xE9):
    eq = synth.equirect(w, h, seed=seed)
    chain_o, n = O.mip_chain(eq)
    return eq, chain_o, dev(chain_o), n


@pytest.mark.parametrize("order", [abi.CONV_WAVE64, abi.CONV_SEQUENTIAL])
def test_conv_diffuse_forms_identical_cfg4(ctx, set_opt, order):
    """2048^2 equirect -> 6 x 64^2 at step 0.010 (99 382 taps per texel): whole cube, the default (branch-free tap on footprint records) vs the other
    two forms. The sequential order runs
    one lane per texel for ~0.1 s: a coarser step keeps it short."""
    _, _, chain_g, n = _chain(2048, 2048)
    step = 0.010 if order == abi.CONV_WAVE64 else 0.05
    fast = ctx.conv_diffuse(chain_g, 2048, 2048, n, 64, step, order, abi.FMT_RGBA32F)
    for form in ("general", "texels"):                      # every tap with its branches / the branch-free tap gathering from the level itself
        set_opt("diffuse_form", form)
        _same(ctx.conv_diffuse(chain_g, 2048, 2048, n, 64, step, order, abi.FMT_RGBA32F), fast, f"cfg4 diffuse records vs {form}, order {order}")


@pytest.mark.parametrize("res", [1, 3, 5, 8])
@pytest.mark.parametrize("order", [abi.CONV_WAVE64, abi.CONV_SEQUENTIAL])
def test_conv_diffuse_pole_texels_and_axis_taps(ctx, res, order):
    """Odd resolutions put texel centres ON the axes: N = (0, +-1, 0) has no frame (NaN right / up: every tap of those texels takes the general
    form and yields the oracle's NaN), N = (+-1, 0, 0) / (0, 0, +-1) make taps land exactly on x = 0 / z = 0 / |y| = 1 (atan2_'s axis cases,
    asin_'s pole): the wave redoes those taps in the general form."""
    _, chain_o, chain_g, n = _chain(256, 128, seed=0x51)
    for step in (0.25, 0.05):
        with np.errstate(all="ignore"):
            ref = O.conv_diffuse(chain_o, 256, 128, n, res, step, order, abi.FMT_RGBA32F)
        assert_bits(ctx.conv_diffuse(chain_g, 256, 128, n, res, step, order, abi.FMT_RGBA32F), ref, f"diffuse res={res} step={step} order={order}")


@pytest.mark.parametrize("order", [abi.CONV_SEQUENTIAL, abi.CONV_WAVE64])
@pytest.mark.parametrize("w,h", [(96, 48), (100, 50), (16, 8), (8, 4), (4, 2)])
def test_conv_diffuse_other_chains(ctx, w, h, order):
    """Non-power-of-two chains (the fast tap is off: integer-modulo wrap) and chains whose last level is above mip 3 (the sampled level is
    clamped to the chain: 1 x 1 ... 2 x 1 images, every tap wraps)."""
    _, chain_o, chain_g, n = _chain(w, h, seed=0x52)
    ref = O.conv_diffuse(chain_o, w, h, n, 4, 0.1, order, abi.FMT_RGBA32F)
    assert_bits(ctx.conv_diffuse(chain_g, w, h, n, 4, 0.1, order, abi.FMT_RGBA32F), ref, f"diffuse from a {w}x{h} chain ({n} levels), order {order}")


@pytest.mark.parametrize("fmt", [abi.FMT_RGBA32F, abi.FMT_RGBA16F])
@pytest.mark.parametrize("res,step", [(1, 0.3), (2, 0.11), (3, 0.7), (5, 0.05), (6, 1.2), (12, 0.031), (16, 0.2)])
def test_conv_diffuse_ordered_ragged_shapes(ctx, set_opt, res, step, fmt):
    """k_conv_diffuse_ordered where nothing divides evenly: cubes of 6 ... 1 536 texels (partial blocks; resolutions that are no multiple of 4 use the
    row-major texel list instead of 4 x 4 patches), steps whose theta count is below the 16 taps of a step (a lane's tap wraps over several phis at once),
    a last round of fewer than 32 taps, both storage formats — against the oracle and against the one-lane-per-texel kernel of the same order."""
    _, chain_o, chain_g, n = _chain(256, 128, seed=0x57)
    with np.errstate(all="ignore"):
        ref = O.conv_diffuse(chain_o, 256, 128, n, res, step, abi.CONV_SEQUENTIAL, fmt)
    got = ctx.conv_diffuse(chain_g, 256, 128, n, res, step, abi.CONV_SEQUENTIAL, fmt)
    assert_bits(got, ref, f"ordered diffuse res={res} step={step}")
    set_opt("diffuse_seq_form", "lane")
    _same(ctx.conv_diffuse(chain_g, 256, 128, n, res, step, abi.CONV_SEQUENTIAL, fmt), got, f"ordered vs lane form, res={res} step={step}")


def test_conv_diffuse_step_too_small_for_the_lds_tables_falls_back(ctx):
    """a step whose (sin, cos) tables of both loops exceed the ordered kernel's LDS budget runs the one-lane-per-texel kernel: same bits as the oracle"""
    _, chain_o, chain_g, n = _chain(64, 32, seed=0x58)
    step = 0.0012                                            # 5 236 phis + 1 309 thetas = 52 KB of tables next to 17 KB of tap buffers: over 64 KB
    ref = O.conv_diffuse(chain_o, 64, 32, n, 1, step, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)
    assert_bits(ctx.conv_diffuse(chain_g, 64, 32, n, 1, step, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F), ref, "diffuse, step 0.0012 (lane fallback)")


def test_conv_diffuse_nonfinite_texels(ctx):
    """inf / NaN texels in the sampled level flow through the same fma chain in both forms."""
    eq = synth.equirect(128, 64, seed=0x53)
    eq[10:14, 20:24, 0] = np.inf
    eq[40, 100, 1] = np.nan
    chain_o, n = O.mip_chain(eq)
    with np.errstate(all="ignore"):
        ref = O.conv_diffuse(chain_o, 128, 64, n, 4, 0.1, abi.CONV_WAVE64, abi.FMT_RGBA32F)
    assert_bits(ctx.conv_diffuse(dev(chain_o), 128, 64, n, 4, 0.1, abi.CONV_WAVE64, abi.FMT_RGBA32F), ref, "diffuse over inf / NaN texels")


@pytest.mark.parametrize("order", [abi.CONV_SEQUENTIAL, abi.CONV_WAVE64])
@pytest.mark.parametrize("w,h,res0,fmt", [(128, 64, 32, abi.FMT_RGBA32F), (64, 32, 4, abi.FMT_RGBA32F), (100, 50, 8, abi.FMT_RGBA16F), (256, 256, 64, abi.FMT_RGBA16F)])
def test_conv_specular_all_mips_in_one_launch(ctx, w, h, res0, fmt, order):
    """Every mip in one launch with the per-block table of tangent-space half vectors (k_conv_specular_all / k_conv_specular_ordered) against the CPU
    oracle, both summation orders, odd chains and the smallest cube. (The per-mip kernels of rounds 1-3 gave the same bits until they were removed.)"""
    _, chain_o, chain_g, n = _chain(w, h, seed=0x54)
    one, mips = ctx.conv_specular(chain_g, w, h, n, res0, order, fmt)
    ref, mips_o = O.conv_specular(chain_o, w, h, n, res0, order, fmt)
    assert mips == mips_o
    assert_bits(one, ref, f"specular {res0}^2 from {w}x{h}, order {order} vs oracle")


@pytest.mark.parametrize("order", [abi.CONV_SEQUENTIAL, abi.CONV_WAVE64])
@pytest.mark.parametrize("w,h,res0", [(2048, 2048, 128), (512, 256, 64), (64, 32, 8), (100, 50, 8)])
def test_conv_specular_forms_identical(ctx, set_opt, order, w, h, res0):
    """Round 6: the branch-free specular sample (unchecked reciprocals + one validity flag, log2 without special cases, branch-free DirectionToEquirectUV, table-driven
    power-of-two trilinear fetch) == the same kernel with every sample in the general form (option specular_form = general), whole cubes, both summation orders: the BASELINE
    cfg4 chain, a 2:1 chain, the smallest cube, and a chain that is no power of two (the fast sample is off: both runs are the general form)."""
    _, _, chain_g, n = _chain(w, h, seed=0x55)
    set_opt("specular_form", None)
    fast, _ = ctx.conv_specular(chain_g, w, h, n, res0, order, abi.FMT_RGBA32F)
    set_opt("specular_form", "general")
    gen, _ = ctx.conv_specular(chain_g, w, h, n, res0, order, abi.FMT_RGBA32F)
    _same(fast, gen, f"specular {res0}^2 from {w}x{h}, order {order}: fast vs general sample")


def test_conv_specular_fast_sample_special_cases(ctx):
    """Inputs that raise the fast sample's validity flag or its rare selects, against the CPU oracle: inf / NaN / zero / denormal texels (NaN filtered colours, not NaN
    addresses), a 1-level chain (lod clamps to 0 everywhere), texel directions on the axes (cube face centres of an odd-free 2^k cube hit x = 0 / z = 0 exactly at res 2)."""
    eq = synth.equirect(256, 128, seed=0x56)
    eq[30:34, 60:70, 0] = np.inf
    eq[80, 200, 1] = np.nan
    eq[100:104, :, 2] = 0.0
    eq[110, 10:20, :3] = 1e-42
    chain_o, n = O.mip_chain(eq)
    for res0 in (16, 4):
        with np.errstate(all="ignore"):
            ref, _ = O.conv_specular(chain_o, 256, 128, n, res0, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)
        got, _ = ctx.conv_specular(dev(chain_o), 256, 128, n, res0, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)
        assert_bits(got, ref, f"specular {res0}^2 over inf / NaN / zero texels")
    one = synth.equirect(1, 1, seed=0x57)
    c1, n1 = O.mip_chain(one)
    ref, _ = O.conv_specular(c1, 1, 1, n1, 4, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)
    assert_bits(ctx.conv_specular(dev(c1), 1, 1, n1, 4, abi.CONV_SEQUENTIAL, abi.FMT_RGBA32F)[0], ref, "specular from a 1 x 1 chain")


def test_conv_diffuse_records_across_streams_and_chains(ctx):
    """The footprint records live in one buffer of the context and are rewritten by every call: calls for DIFFERENT chains on different streams must
    not read each other's records (the second call's stream waits for the first call's kernel)."""
    _, co_a, cg_a, n_a = _chain(256, 128, seed=0x61)
    _, co_b, cg_b, n_b = _chain(512, 256, seed=0x62)
    ref_a = O.conv_diffuse(co_a, 256, 128, n_a, 8, 0.05, abi.CONV_WAVE64, abi.FMT_RGBA32F)
    ref_b = O.conv_diffuse(co_b, 512, 256, n_b, 8, 0.05, abi.CONV_WAVE64, abi.FMT_RGBA32F)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        outs.append((ctx.conv_diffuse(cg_a, 256, 128, n_a, 8, 0.05, abi.CONV_WAVE64, abi.FMT_RGBA32F, stream=s1), ref_a))
        outs.append((ctx.conv_diffuse(cg_b, 512, 256, n_b, 8, 0.05, abi.CONV_WAVE64, abi.FMT_RGBA32F, stream=s2), ref_b))
    torch.cuda.synchronize()
    for k, (got, ref) in enumerate(outs):
        assert_bits(got, ref, f"interleaved conv_diffuse call {k}")
