"""CPU tests of the host-side producers (vqengine_amd/scene.py) that mirror Light::GetGPUData, Scene::GatherSceneLightData
and Material::GetCBufferData, ending in an oracle render of the reference's own visual unit
test: the 8 x 4 roughness x metalness sphere grid of Source/Scenes/EnvironmentMapUnitTestScene.cpp:50-72."""
import ctypes as C
import math

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import ref64
from vqengine_amd import abi, scene, synth


def test_light_defaults_and_gpu_data():
    l = scene.Light()                                                    # Light.cpp:58-73
    assert (l.Range, l.Brightness, l.DepthBias, l.bEnabled, l.bCastingShadows) == (1000.0, 300.0, 0.00005, True, False)
    p = scene.Light(Type=scene.Light.POINT, Position=(1, 2, 3), Color=(0.5, 0.25, 1.0), Brightness=1500.0, Range=35.0).get_gpu_data()
    assert isinstance(p, abi.PointLight) and (p.position.x, p.position.y, p.position.z, p.range, p.brightness) == (1, 2, 3, 35.0, 1500.0)
    assert (p.color.x, p.color.y, p.color.z) == (0.5, 0.25, 1.0) and abs(p.depthBias - 5e-5) < 1e-10
    s = scene.Light(Type=scene.Light.SPOT, SpotInnerConeAngleDegrees=22.0, SpotOuterConeAngleDegrees=32.0).get_gpu_data()
    assert abs(s.innerConeAngle - math.radians(22)) < 1e-7 and abs(s.outerConeAngle - math.radians(32)) < 1e-7
    assert (s.spotDir.x, s.spotDir.y, s.spotDir.z) == (0.0, 0.0, 1.0) and s.range == 0.0          # range left unset (Light.cpp:108-121)
    d = scene.Light(Type=scene.Light.DIRECTIONAL, Brightness=0.9, bCastingShadows=True).get_gpu_data()
    assert (d.lightDirection.x, d.lightDirection.y, d.lightDirection.z) == (0.0, -1.0, 0.0) and d.shadowing == 1 and d.enabled == 1
    q = (math.cos(math.pi / 4), math.sin(math.pi / 4), 0.0, 0.0)         # +90 degrees about X: (0,-1,0) -> (0,0,-1)
    d = scene.Light(Type=scene.Light.DIRECTIONAL, RotationQuaternion=q).get_gpu_data()
    assert np.allclose((d.lightDirection.x, d.lightDirection.y, d.lightDirection.z), (0, 0, -1), atol=1e-6)


def test_gather_scene_light_data_default_scene_shape():
    """Data/Levels/Default.xml:202-308: shadowing directional (0.9), two shadow-casting spots (1500/1000), two DISABLED points."""
    vp = np.eye(4, dtype=np.float32) * 2
    lights = [scene.Light(Type=scene.Light.DIRECTIONAL, Brightness=0.9, bCastingShadows=True, ViewProjection=vp, Mobility=scene.Light.STATIC),
              scene.Light(Type=scene.Light.SPOT, Brightness=1500.0, Range=35.0, bCastingShadows=True, ViewProjection=vp * 3),
              scene.Light(Type=scene.Light.POINT, bEnabled=False), scene.Light(Type=scene.Light.POINT, bEnabled=False),
              scene.Light(Type=scene.Light.SPOT, Brightness=1000.0, bCastingShadows=True, Mobility=scene.Light.STATIONARY, ViewProjection=vp * 5),
              scene.Light(Type=scene.Light.POINT, Brightness=7.0, Mobility=scene.Light.STATIC),
              scene.Light(Type=scene.Light.POINT, Brightness=9.0)]
    d = scene.gather_scene_light_data(lights)
    assert (d.numPointLights, d.numSpotLights, d.numPointCasters, d.numSpotCasters) == (2, 0, 0, 2)
    assert d.directional.shadowing == 1 and abs(d.directional.brightness - 0.9) < 1e-7 and d.shadowViewDirectional.m[2][2] == 2.0
    # mobility order: static first, then stationary, then dynamic (Scene.cpp:1017-1019)
    assert d.point_lights[0].brightness == 7.0 and d.point_lights[1].brightness == 9.0
    assert d.spot_casters[0].brightness == 1000.0 and d.spot_casters[1].brightness == 1500.0
    assert d.shadowViews[0].m[0][0] == 10.0 and d.shadowViews[1].m[0][0] == 6.0
    with pytest.raises(ValueError):
        scene.gather_scene_light_data([scene.Light() for _ in range(101)])


def test_material_cbuffer_data():
    m = scene.Material()
    assert (m.roughness, m.metalness, m.emissiveIntensity, m.alpha) == (0.8, 0.0, 0.0, 1.0) and m.get_texture_config() == 0
    d = m.get_cbuffer_data()
    assert C.sizeof(d) == 80 and d.textureConfig == 0.0 and (d.uvScaleOffset.x, d.uvScaleOffset.y, d.uvScaleOffset.z, d.uvScaleOffset.w) == (1, 1, 0, 0)
    m = scene.Material(TexDiffuseMap=3, TexRoughnessMap=9, TexOcclusionRoughnessMetalnessMap=1, TexEmissiveMap=0, tiling=(2, 3), uv_bias=(0.5, 0.25))
    cfg = m.get_texture_config()
    assert cfg == (1 << 0) | (1 << 4) | (1 << 8) | (1 << 7)              # Material.cpp:26-34
    d = m.get_cbuffer_data()
    assert d.textureConfig == float(cfg) and (d.uvScaleOffset.x, d.uvScaleOffset.w) == (2.0, 0.25)
    assert [scene.has_map(cfg, b) for b in range(9)] == [1, 0, 0, 0, 1, 0, 0, 1, 1]
    with pytest.raises(ValueError, match="gbuffer_from_materials"):      # textured materials are the producer kernel's, and the message says so
        scene.gbuffer_from_material(d, np.zeros((1, 3)), np.ones((1, 3)), 0.055)


def test_environment_map_unit_test_scene_grid_render():
    """EnvironmentMapUnitTestScene.cpp:50-72: 8 x 4 spheres, roughness 0..1 along x (clamped to >= 0.04 like :62), metalness 0..1
    along y, diffuse (0, 0.05, 0.45). Each sphere is rendered as a 12 x 12 disc of hemisphere normals facing the camera, lit by
    lights gathered through scene.Light; the oracle output is checked against the float64 restatement."""
    res = 12
    yy, xx = np.mgrid[0:res, 0:res]
    u = (xx + 0.5) / res * 2 - 1
    v = 1 - (yy + 0.5) / res * 2
    r2 = u * u + v * v
    inside = r2 < 0.95
    nz = -np.sqrt(np.clip(1 - r2, 0, 1))                                 # normals face the camera at -Z
    lights = [scene.Light(Type=scene.Light.POINT, Position=(0, 30, -40), Brightness=4000.0, Range=500.0),
              scene.Light(Type=scene.Light.POINT, Position=(-60, 10, -30), Color=(1.0, 0.6, 0.3), Brightness=6000.0, Range=500.0),
              scene.Light(Type=scene.Light.DIRECTIONAL, Brightness=0.9, RotationQuaternion=(math.cos(0.3), math.sin(0.3), 0.0, 0.0))]
    pf = abi.PerFrameData()
    pf.Lights = scene.gather_scene_light_data(lights)
    pf.fAmbientLightingFactor = 0.055                                    # SceneViews.h:61
    planes = [[], [], [], []]
    for ix in range(8):
        for iy in range(4):
            mat = scene.Material(diffuse=(0.0, 0.05, 0.45), roughness=max(0.04, ix / 7.0), metalness=iy / 3.0).get_cbuffer_data()
            centre = np.array([(ix - 3.5) * 10.0, (iy - 1.5) * 10.0, 0.0], np.float32)
            N = np.stack([u, v, nz], -1)[inside]
            P = centre + 4.0 * N
            g = scene.gbuffer_from_material(mat, P, N, pf.fAmbientLightingFactor)
            for k in range(4):
                planes[k].append(g[k])
    gb = [np.concatenate(p)[None] for p in planes]                       # one row of pixels
    W = gb[0].shape[1]
    pv = synth.per_view(W, 1, camera=(0.0, 0.0, -120.0))
    out = O.forward_lighting(gb, pf, pv, abi.FMT_RGBA32F)
    pts = [dict(pos=l.Position, color=l.Color, brightness=l.Brightness, range=l.Range) for l in lights[:2]]
    d = pf.Lights.directional
    ref = ref64.shade(gb, (0.0, 0.0, -120.0), pts, (), dict(dir=(d.lightDirection.x, d.lightDirection.y, d.lightDirection.z),
                                                               color=(d.color.x, d.color.y, d.color.z), brightness=d.brightness))
    rel = np.abs(out - ref) / np.maximum(np.abs(ref), 1e-3)
    assert np.isfinite(out).all() and np.quantile(rel, 0.999) < 2e-4 and rel.max() < 5e-2, (np.quantile(rel, 0.999), rel.max())
    # rough dielectric spheres are dimmer at the highlight than smooth metallic ones; blue channel dominates the diffuse term
    assert out[0, :, 2].mean() > out[0, :, 0].mean()
