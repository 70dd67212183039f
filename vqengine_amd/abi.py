"""ctypes mirror of include/vqhip.h (the C ABI structs == VQ_SHADER_DATA of
Shaders/LightingConstantBufferData.h:50-186). Pure layout, no compute; importable without a GPU.
Sizes/offsets are asserted against SURVEY.md §8(b) at import time."""
import ctypes as C

VQHIP_OK = 0
VQHIP_ERR_INVALID_ARG = -1
VQHIP_ERR_HIP = -2
VQHIP_ERR_UNSUPPORTED = -3
VQHIP_ERR_NO_DEVICE = -4
VQHIP_ERR_RCCL = -5

FMT_RGBA32F, FMT_RGBA16F, FMT_RGBA8_UNORM, FMT_RG16F, FMT_RG32F = 0, 1, 2, 3, 4
FMT_R10G10B10A2_UNORM = 5          # Tex_SceneNormals: input of ssr_environment_fallback only
FMT_BPP = {FMT_RGBA32F: 16, FMT_RGBA16F: 8, FMT_RGBA8_UNORM: 4, FMT_RG16F: 4, FMT_RG32F: 8}
CONV_SEQUENTIAL, CONV_WAVE64 = 0, 1
ARITH_LITERAL, ARITH_DXC = 0, 1
ABI_VERSION = 3
COLOR_SPACE_REC_709, COLOR_SPACE_REC_2020 = 0, 1
DISPLAY_CURVE_SRGB, DISPLAY_CURVE_ST2084, DISPLAY_CURVE_LINEAR = 0, 1, 2

NUM_LIGHTS__POINT = 100
NUM_LIGHTS__SPOT = 20
NUM_SHADOWING_LIGHTS__POINT = 5
NUM_SHADOWING_LIGHTS__SPOT = 5


class float2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def set(self, v):
        self.x, self.y, self.z = float(v[0]), float(v[1]), float(v[2])


class float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class matrix(C.Structure):  # DirectX::XMMATRIX, row-major, 16-byte aligned (padding is explicit in the parents)
    _fields_ = [("m", (C.c_float * 4) * 4)]


class PointLight(C.Structure):
    _fields_ = [("position", float3), ("range", C.c_float), ("color", float3), ("brightness", C.c_float),
                ("attenuation", float3), ("depthBias", C.c_float)]


class SpotLight(C.Structure):
    _fields_ = [("position", float3), ("outerConeAngle", C.c_float), ("color", float3), ("brightness", C.c_float),
                ("spotDir", float3), ("depthBias", C.c_float), ("innerConeAngle", C.c_float), ("range", C.c_float),
                ("dummy1", C.c_float), ("dummy2", C.c_float)]


class DirectionalLight(C.Structure):
    _fields_ = [("lightDirection", float3), ("brightness", C.c_float), ("color", float3), ("depthBias", C.c_float),
                ("shadowing", C.c_int32), ("enabled", C.c_int32)]


class SceneLighting(C.Structure):
    _fields_ = [("numPointLights", C.c_int32), ("numSpotLights", C.c_int32), ("numPointCasters", C.c_int32),
                ("numSpotCasters", C.c_int32), ("directional", DirectionalLight), ("_pad0", C.c_byte * 8),
                ("shadowViewDirectional", matrix),
                ("point_lights", PointLight * NUM_LIGHTS__POINT), ("point_casters", PointLight * NUM_SHADOWING_LIGHTS__POINT),
                ("spot_lights", SpotLight * NUM_LIGHTS__SPOT), ("spot_casters", SpotLight * NUM_SHADOWING_LIGHTS__SPOT),
                ("shadowViews", matrix * NUM_SHADOWING_LIGHTS__SPOT)]


class PerFrameData(C.Structure):
    _fields_ = [("Lights", SceneLighting), ("f2PointLightShadowMapDimensions", float2),
                ("f2SpotLightShadowMapDimensions", float2), ("f2DirectionalLightShadowMapDimensions", float2),
                ("fAmbientLightingFactor", C.c_float), ("fHDRIOffsetInRadians", C.c_float)]


class PerViewLightingData(C.Structure):
    _fields_ = [("matView", matrix), ("matViewToWorld", matrix), ("matProjInverse", matrix),
                ("WorldFrustumPlanes", float4 * 6), ("CameraPosition", float3), ("MaxEnvMapLODLevels", C.c_float),
                ("ScreenDimensions", float2), ("EnvironmentMapDiffuseOnlyIllumination", C.c_int32), ("pad1", C.c_float)]


class MaterialData(C.Structure):
    _fields_ = [("diffuse", float3), ("alpha", C.c_float), ("emissiveColor", float3), ("emissiveIntensity", C.c_float),
                ("specular", float3), ("normalMapMipBias", C.c_float), ("uvScaleOffset", float4),
                ("roughness", C.c_float), ("metalness", C.c_float), ("displacement", C.c_float), ("textureConfig", C.c_float)]


class TonemapperParams(C.Structure):  # FPostProcessParameters::FTonemapper defaults, PostProcess.h:84-91
    _fields_ = [("ContentColorSpaceEnum", C.c_int32), ("OutputDisplayCurveEnum", C.c_int32),
                ("DisplayReferenceBrightnessLevel", C.c_float), ("ToggleGammaCorrection", C.c_int32)]

    @staticmethod
    def default():
        return TonemapperParams(COLOR_SPACE_REC_709, DISPLAY_CURVE_SRGB, 200.0, 1)


class BlurParams(C.Structure):
    _fields_ = [("iImageSizeX", C.c_int32), ("iImageSizeY", C.c_int32)]


class EnvMap(C.Structure):
    _fields_ = [("diffuse_cube", C.c_void_p), ("diffuse_res", C.c_int32), ("specular_cube", C.c_void_p),
                ("spec_res0", C.c_int32), ("spec_mips", C.c_int32), ("brdf_lut", C.c_void_p), ("lut_size", C.c_int32)]


class ShadowMaps(C.Structure):
    _fields_ = [("directional", C.c_void_p), ("dir_dim", C.c_int32), ("spot", C.c_void_p), ("spot_dim", C.c_int32),
                ("point", C.c_void_p), ("point_dim", C.c_int32)]


class GBuffer(C.Structure):
    _fields_ = [("gb0", C.c_void_p), ("gb1", C.c_void_p), ("gb2", C.c_void_p), ("gb3", C.c_void_p),
                ("width", C.c_int32), ("height", C.c_int32), ("row_pitch_px", C.c_int32)]


class PsmainTargets(C.Structure):    # vqhip_psmain_targets: the lit draw's other render targets (ForwardLighting.hlsl:57-68,382-389)
    _fields_ = [("albedo_metallic", C.c_void_p), ("albedo_fmt", C.c_int32), ("albedo_pitch_px", C.c_int32),
                ("motion_vectors", C.c_void_p), ("motion_fmt", C.c_int32), ("motion_pitch_px", C.c_int32),
                ("svPositionCurr", C.c_void_p), ("svPositionPrev", C.c_void_p), ("sv_pitch_px", C.c_int32), ("pad_", C.c_int32)]


assert C.sizeof(PsmainTargets) == 56

class CommInfo(C.Structure):    # vqhip_comm_info
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("nranks_seen", C.c_int32), ("rank_seen", C.c_int32),
                ("rccl_version", C.c_int32), ("reserved", C.c_int32), ("library_path", C.c_char * 232)]


class EnvMapOut(C.Structure):
    _fields_ = [("diffuse_unblurred", C.c_void_p), ("diffuse_blurred", C.c_void_p), ("blur_tmp", C.c_void_p),
                ("specular", C.c_void_p)]


class Texture2D(C.Structure):  # vqhip_texture2d: RGBA8_UNORM mip chain, texels NULL == null SRV
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("mips", C.c_int32), ("reserved", C.c_int32)]


MATERIAL_ALPHA_MASKED = 1   # VQHIP_MATERIAL_ALPHA_MASKED: flag in vqhip_material.texDiffuse.reserved (the "_AlphaMasked" PSO permutation)
MATERIAL_TEXTURE_SLOTS = ("texDiffuse", "texNormals", "texEmissive", "texMetalness", "texRoughness", "texOcclRoughMetal", "texLocalAO")


class MaterialDesc(C.Structure):  # vqhip_material
    _fields_ = [("data", MaterialData)] + [(s, Texture2D) for s in MATERIAL_TEXTURE_SLOTS] + [("_tail_pad", C.c_int32 * 2)]  # alignas(16)


class Interpolants(C.Structure):  # vqhip_interpolants
    _fields_ = [("ip0", C.c_void_p), ("ip1", C.c_void_p), ("ip2", C.c_void_p),
                ("width", C.c_int32), ("height", C.c_int32), ("row_pitch_px", C.c_int32)]


class SSAO(C.Structure):  # vqhip_ssao
    _fields_ = [("texels", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32)]


class VizParams(C.Structure):  # VQ_VizParams == FPostProcessParameters::FVizualizationParams (Visualization.hlsl:26-31)
    _fields_ = [("iDrawMode", C.c_int32), ("iUnpackNormals", C.c_int32), ("fInputStrength", C.c_float)]


class SkydomeParams(C.Structure):  # VQ_SkydomeParams
    _fields_ = [("right", float3), ("tanHalfFovX", C.c_float), ("up", float3), ("tanHalfFovY", C.c_float),
                ("forward", float3), ("pad", C.c_float)]


class SSSRConstants(C.Structure):  # VQ_SSSRConstants == FFX_SSSRConstants (ScreenSpaceReflections.h:43-65)
    _fields_ = [(k, matrix) for k in ("invViewProjection", "projection", "invProjection", "view", "invView", "prevViewProjection", "envMapRotation")] + [
        ("bufferDimensions", C.c_uint32 * 2), ("inverseBufferDimensions", C.c_float * 2),
        ("temporalStabilityFactor", C.c_float), ("depthBufferThickness", C.c_float), ("roughnessThreshold", C.c_float), ("varianceThreshold", C.c_float),
        ("frameIndex", C.c_uint32), ("maxTraversalIntersections", C.c_uint32), ("minTraversalOccupancy", C.c_uint32), ("mostDetailedMip", C.c_uint32),
        ("samplesPerQuad", C.c_uint32), ("temporalVarianceGuidedTracingEnabled", C.c_uint32), ("envMapSpecularIrradianceCubemapMipLevelCount", C.c_uint32),
        ("pad_", C.c_uint32)]


def _chk(t, size, **offs):
    assert C.sizeof(t) == size, (t.__name__, C.sizeof(t), size)
    for k, v in offs.items():
        assert getattr(t, k).offset == v, (t.__name__, k, getattr(t, k).offset, v)


_chk(PointLight, 48, range=12, color=16, brightness=28, attenuation=32, depthBias=44)
_chk(SpotLight, 64, outerConeAngle=12, spotDir=32, depthBias=44, innerConeAngle=48, range=52)
_chk(DirectionalLight, 40, shadowing=32, enabled=36)
_chk(SceneLighting, 7088, directional=16, shadowViewDirectional=64, point_lights=128, point_casters=4928,
     spot_lights=5168, spot_casters=6448, shadowViews=6768)
_chk(PerFrameData, 7120, f2PointLightShadowMapDimensions=7088, fAmbientLightingFactor=7112, fHDRIOffsetInRadians=7116)
_chk(PerViewLightingData, 320, WorldFrustumPlanes=192, CameraPosition=288, MaxEnvMapLODLevels=300,
     ScreenDimensions=304, EnvironmentMapDiffuseOnlyIllumination=312)
_chk(MaterialData, 80, uvScaleOffset=48, roughness=64, textureConfig=76)
_chk(Texture2D, 24, width=8, mips=16)
_chk(MaterialDesc, 256, texDiffuse=80, texLocalAO=224)
_chk(TonemapperParams, 16)
_chk(BlurParams, 8)
_chk(SSSRConstants, 512, bufferDimensions=448, roughnessThreshold=472, envMapSpecularIrradianceCubemapMipLevelCount=504)


def mip_level_count(w, h):
    """Image::CalculateMipLevelCount (VQUtils, absent; semantics from EnvironmentMapRendering.cpp:63)."""
    m, n = max(w, h), 1
    while m > 1:
        m >>= 1
        n += 1
    return n


def mip_dim(d0, level):
    return max(1, d0 >> level)


def mip_chain_px(w0, h0, n_mips):
    return sum(mip_dim(w0, l) * mip_dim(h0, l) for l in range(n_mips))


def specular_mip_count(res0):
    return mip_level_count(res0, res0) - 1


def cube_px(res0, n_mips):
    return sum(6 * (res0 >> m) ** 2 for m in range(n_mips))
