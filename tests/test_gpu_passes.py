"""The IRenderPass-shaped C++ adaptors (include/vqhip_passes.hpp) driven from a C++ program the way VQRenderer would
drive them (tests/cpp/test_passes.cpp), checked bit-for-bit against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import oracle_lib as O
from vqengine_amd import abi, synth

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "test_passes")


def test_cpp_pass_adaptors(tmp_path, ctx):
    W, H, EW, EH = 160, 24, 64, 32
    gb = synth.gbuffer(W, H, seed=0xCAFE)
    pf, _ = synth.per_frame(points=synth.point_lights(20, seed=0xCAFE), spots=synth.spot_lights(2, seed=0xCAFE),
                            directional=synth.directional_light(), hdri_offset=0.3)
    eq = synth.equirect(EW, EH)
    chain, n = O.mip_chain(eq)
    pre = O.envmap_prefilter(chain, EW, EH, n, 8, 0.1, 16, abi.CONV_SEQUENTIAL)
    pv = synth.per_view(W, H, max_env_lod=pre["spec_mips"])
    for k in range(4):
        gb[k].tofile(tmp_path / f"gb{k}.bin")
    eq.tofile(tmp_path / "equirect.bin")
    (tmp_path / "perframe.bin").write_bytes(bytes(pf))
    (tmp_path / "perview.bin").write_bytes(bytes(pv))
    # §8f.1 inputs: interpolant planes + 3 texture-less materials (descriptor pointers NULL, so the table is position independent)
    ip = synth.interpolants(W, H, 3, seed=0xCAFE)
    for k in range(3):
        ip[k].tofile(tmp_path / f"ip{k}.bin")
    datas, _ = synth.material_set(3, seed=0xCAFE, max_dim=16)
    mats = (abi.MaterialDesc * 3)()
    for i, d in enumerate(datas):
        mats[i].data = d
        mats[i].data.textureConfig = 0.0
    (tmp_path / "materials.bin").write_bytes(bytes(mats))
    # §8f.4 inputs: FFX_SSSRConstants + depth + packed normals (the roughness comes from the lit frame's alpha)
    _, depth, packed, _ = synth.ssr_surfaces(W, H, seed=0xCAFE)
    cb = synth.ssr_constants(W, H, pre["spec_mips"])
    (tmp_path / "ssr_cb.bin").write_bytes(bytes(cb))
    depth.tofile(tmp_path / "ssr_depth.bin")
    packed.tofile(tmp_path / "ssr_normals.bin")
    sv_curr, sv_prev = synth.clip_positions(W, H)             # PSInput.svPositionCurr / Prev for the pass's motion-vector target
    sv_curr.tofile(tmp_path / "sv_curr.bin"); sv_prev.tofile(tmp_path / "sv_prev.bin")
    r = subprocess.run([EXE, str(tmp_path), str(W), str(H), str(EW), str(EH), "rccl"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    diff = np.fromfile(tmp_path / "diffuse_blurred.bin", np.float16).reshape(6, 8, 8, 4)
    n_bad, idx = O.bits_equal(diff, pre["diffuse_blurred"])
    assert n_bad == 0, idx
    # full-size LUT from the product (already checked row-wise against the oracle in test_gpu_parity) feeds the oracle shade
    lut_g = ctx.brdf_lut(1024, 2048, abi.FMT_RG16F).cpu().numpy()
    env = O.host_envmap(pre["diffuse_blurred"], pre["specular"], 16, pre["spec_mips"], lut_g)
    scene = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA16F, env=env)
    got = np.fromfile(tmp_path / "scene_rgba16f.bin", np.float16).reshape(H, W, 4)
    n_bad, idx = O.bits_equal(got, scene)
    assert n_bad == 0, (n_bad, idx)
    rad, r8 = O.ssr_environment_fallback(scene, abi.FMT_RGBA16F, depth, packed, abi.FMT_R10G10B10A2_UNORM, cb, env, abi.FMT_RGBA16F, extract_roughness=True)
    n_bad, idx = O.bits_equal(np.fromfile(tmp_path / "ssr_radiance_rgba16f.bin", np.float16).reshape(H, W, 4), rad)
    assert n_bad == 0, ("ssr radiance", n_bad, idx)
    assert np.array_equal(np.fromfile(tmp_path / "ssr_roughness_r8.bin", np.uint8).reshape(H, W), r8)
    gb_ip = O.gbuffer_from_materials(ip, mats, pf.fAmbientLightingFactor, None)
    scene_ip = O.forward_lighting(gb_ip, pf, pv, abi.FMT_RGBA16F, env=env)
    got = np.fromfile(tmp_path / "scene_ip_rgba16f.bin", np.float16).reshape(H, W, 4)
    n_bad, idx = O.bits_equal(got, scene_ip)
    assert n_bad == 0, (n_bad, idx)
    assert np.array_equal(np.fromfile(tmp_path / "scene_normals_r10g10b10a2.bin", np.uint32).reshape(H, W), O.scene_normals_from_materials(ip, mats))
    viz, mv = O.psmain_extra_targets(gb_ip, sv_curr, sv_prev)                    # Tex_SceneVisualization / Tex_SceneMotionVectors of the same draw
    n_bad, idx = O.bits_equal(np.fromfile(tmp_path / "scene_viz_rgba16f.bin", np.float16).reshape(H, W, 4), viz)
    assert n_bad == 0, ("albedo / metalness target", n_bad, idx)
    n_bad, idx = O.bits_equal(np.fromfile(tmp_path / "scene_mv_rg16f.bin", np.float16).reshape(H, W, 2), mv)
    assert n_bad == 0, ("motion vectors", n_bad, idx)
    sdr = O.tonemap(O.gaussian_blur(scene, abi.FMT_RGBA16F), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    got = np.fromfile(tmp_path / "sdr_rgba8.bin", np.uint8).reshape(H, W, 4)
    assert np.array_equal(got, sdr)
    # the post pass in row-tiled mode over a world of one rank (real RCCL), run on the scene colour of the §8f.1 path (the last one rendered):
    # its composite frame is the post chain of that image
    sdr_ip = O.tonemap(O.gaussian_blur(scene_ip, abi.FMT_RGBA16F), abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM)
    assert np.array_equal(np.fromfile(tmp_path / "frame_rgba8.bin", np.uint8).reshape(H, W, 4), sdr_ip)


def test_adaptors_derive_from_the_engine_interface():
    """include/vqhip_passes.hpp with VQHIP_ENGINE_RENDERPASS_H defined (tests/cpp/test_passes_engine.cpp against tests/cpp/mock_engine/): the
    adaptors are ::IRenderPass objects that VQRenderer::mRenderPasses can hold. Host-only program; run here because it links libvqhip.so."""
    exe = os.path.join(os.path.dirname(EXE), "test_passes_engine")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "engine-interface passes OK" in r.stdout, r.stdout + r.stderr
