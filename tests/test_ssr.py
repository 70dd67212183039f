"""SSR's environment-map fallback (ClassifyReflectionTiles.hlsl:78-94,146-153): the CPU oracle's statement — its classification
condition, its input formats, the fractional-LOD cube fetch it introduces — and, where oracle/_ref exists, the reference's own HLSL beside it.
The stored reference outputs (tests/golden/ref_outputs.npz, cases ssr_env_fallback_*) are checked by tests/test_ref_fixtures.py."""
import numpy as np
import pytest

from tests import oracle_lib as O
from tests import ref_cases, ref_lib
from vqengine_amd import abi, synth

W, H = 96, 20


@pytest.fixture(scope="module")
def setup():
    e = ref_cases.small_env()
    scene, depth, packed, n01 = synth.ssr_surfaces(W, H, seed=0x51)
    return {"env": e, "henv": ref_cases.host_env(e), "scene": scene.astype(np.float16), "depth": depth, "packed": packed, "n01": n01,
            "cb": synth.ssr_constants(W, H, e["spec_mips"])}


def _run(s, **kw):
    a = dict(scene=s["scene"], scene_fmt=abi.FMT_RGBA16F, depth=s["depth"], normals=s["packed"], normal_fmt=abi.FMT_R10G10B10A2_UNORM, cb=s["cb"], env=s["henv"])
    a.update(kw)
    return O.ssr_environment_fallback(a["scene"], a["scene_fmt"], a["depth"], a["normals"], a["normal_fmt"], a["cb"], a["env"], a.get("out_fmt", abi.FMT_RGBA32F),
                                      extract_roughness=a.get("extract_roughness", False))


def test_constants_layout():
    assert abi.SSSRConstants.invView.offset == 256 and abi.SSSRConstants.envMapRotation.offset == 384 and abi.SSSRConstants.inverseBufferDimensions.offset == 456


def test_classification_condition(setup):
    """ClassifyTiles :146-152: only surfaces in front of the far plane (depth < 1) that are NOT glossy (roughness >= threshold) get the fallback; everything else,
    and every alpha, is 0."""
    out = _run(setup)
    rough = setup["scene"][..., 3].astype(np.float32)
    takes = (setup["depth"] < 1.0) & ~(rough < np.float32(0.2))
    assert 0.5 < takes.mean() < 0.9
    assert (out[~takes] == 0).all() and (out[..., 3] == 0).all()
    assert (out[takes][:, :3].sum(-1) > 0).mean() > 0.99 and np.isfinite(out).all()
    # the threshold is the cbuffer's: 0 sends every reflective pixel to the fallback, 2 none
    cb0, cb2 = synth.ssr_constants(W, H, setup["env"]["spec_mips"], roughness_threshold=0.0), synth.ssr_constants(W, H, setup["env"]["spec_mips"], roughness_threshold=2.0)
    assert ((_run(setup, cb=cb0)[..., :3].sum(-1) > 0) == (setup["depth"] < 1.0)).mean() > 0.99
    assert (_run(setup, cb=cb2) == 0).all()


def test_normal_and_scene_formats(setup):
    """R10G10B10A2_UNORM normals decode as c / 1023 (the same values handed over as floats give the same bits); the roughness is the scene colour's alpha in either format"""
    a = _run(setup)
    assert np.array_equal(a.view(np.uint32), _run(setup, normals=setup["n01"], normal_fmt=abi.FMT_RGBA32F).view(np.uint32))
    assert np.array_equal(a.view(np.uint32), _run(setup, scene=setup["scene"].astype(np.float32), scene_fmt=abi.FMT_RGBA32F).view(np.uint32))
    h16 = _run(setup, out_fmt=abi.FMT_RGBA16F)
    with np.errstate(over="ignore"):
        assert np.array_equal(h16.view(np.uint16), a.astype(np.float16).view(np.uint16))


def test_extracted_roughness(setup):
    _, r8 = _run(setup, extract_roughness=True)                   # g_extracted_roughness, R8_UNORM store of CSMain :196
    assert np.array_equal(r8, ref_cases.to_unorm8(setup["scene"][..., 3].astype(np.float32)))


def test_lod_is_fractional(setup):
    """roughness * (mip_count - 1) selects BETWEEN cube mips: a constant-roughness frame at lod 1.5 is the mean of the frames at lod 1 and 2 (weights 1/2 are exact)"""
    mips = setup["env"]["spec_mips"]
    assert mips >= 3

    def frame(r):
        sc = setup["scene"].astype(np.float32).copy()
        sc[..., 3] = r
        return _run(setup, scene=sc, scene_fmt=abi.FMT_RGBA32F)
    # the LUT factor (Ks * A + B) depends on the roughness too: compare pre-LUT ratios through a LUT of constant (A, B) = (0, 1)
    lut1 = np.zeros_like(setup["env"]["lut"]); lut1[..., 1] = 1.0
    e = dict(setup["env"], lut=lut1)
    s = dict(setup, henv=ref_cases.host_env(e))

    def frame1(r):
        sc = setup["scene"].astype(np.float32).copy()
        sc[..., 3] = r
        return _run(s, scene=sc, scene_fmt=abi.FMT_RGBA32F)
    lo, mid, hi = frame1(1.0 / (mips - 1)), frame1(1.5 / (mips - 1)), frame1(2.0 / (mips - 1))
    m = (lo[..., :3].sum(-1) > 0)
    assert np.allclose(mid[m], 0.5 * (lo[m] + hi[m]), rtol=2e-2, atol=1e-4)        # 8-bit fraction of 1.5 (+- an ulp of the product) = 128/256 or one step beside it
    assert not np.allclose(mid[m], lo[m], rtol=1e-3)


@pytest.mark.skipif(not ref_lib.available("shaders"), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_vs_reference_hlsl(setup):
    """the reference's own SampleEnvironmentMap / InvProjectPosition / EnvironmentBRDF through the shim, literal reading"""
    ref = ref_lib.ssr_environment_fallback(setup["scene"].astype(np.float32), setup["depth"], setup["n01"], setup["cb"], setup["henv"])
    got = _run(setup)
    assert np.array_equal(got == 0, ref == 0)
    ref_cases.check("ssr fallback, fp32", got[..., :3], ref[..., :3], (1e-7, 2e-6, 1e-3, 1e-4))
    ref_cases.check("ssr fallback, RGBA16F", got[..., :3], ref[..., :3], ("ulp16", 1, 0.002))
