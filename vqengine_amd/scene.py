"""Host-side mirror of the reference's CPU producers of the hot path's inputs:

  Light            Source/Engine/Scene/Light.h:45-187, Light.cpp:58-121   -> abi.PointLight / SpotLight / DirectionalLight
  gather_scene_light_data   Scene::GatherSceneLightData, Scene.cpp:978-1027 -> abi.SceneLighting
  Material         Source/Engine/Scene/Material.h:44-133, Material.cpp:23-36 -> abi.MaterialData (+ textureConfig bits)
  gbuffer_from_material     the texture-less branch of ForwardLighting.hlsl:PSMain :247-281 (HasXMap == 0)

Same names, defaults and packing rules as the reference so tests read like the engine's own call sites. Pure Python /
numpy; no compute kernels here."""
import math
from dataclasses import dataclass, field

import numpy as np

from . import abi

DEG2RAD = math.pi / 180.0


def _quat_rotate(q, v):
    """Rotate v by the unit quaternion q = (w, x, y, z) (Transform::NormalMatrix of a rigid transform == its rotation)."""
    w, x, y, z = q
    u = np.array([x, y, z], np.float64)
    v = np.asarray(v, np.float64)
    return v + 2.0 * np.cross(u, np.cross(u, v) + w * v)


@dataclass
class Light:
    """Light.h:45-187 with the constructor defaults of Light.cpp:58-73."""
    POINT, SPOT, DIRECTIONAL = 0, 1, 2
    STATIC, STATIONARY, DYNAMIC = 0, 1, 2

    Type: int = 0
    Mobility: int = 2
    Position: tuple = (0.0, 0.0, 0.0)
    Range: float = 1000.0
    RotationQuaternion: tuple = (1.0, 0.0, 0.0, 0.0)       # (w, x, y, z), identity
    bEnabled: bool = True
    bCastingShadows: bool = False
    Color: tuple = (1.0, 1.0, 1.0)
    Brightness: float = 300.0
    DepthBias: float = 0.00005                               # FShadowData(0.00005f, 0.01f, 1500.0f)
    SpotInnerConeAngleDegrees: float = 25.0                  # MakeSpotLight, Light.cpp:47-55
    SpotOuterConeAngleDegrees: float = 35.0
    ViewProjection: object = None                            # GetViewProjectionMatrix() result when casting shadows (4x4 row-major)

    def _common(self, dst):                                  # COPY_COMMON_LIGHT_DATA, Light.cpp:75-78
        dst.brightness = self.Brightness
        dst.color.set(self.Color)
        dst.depthBias = self.DepthBias

    def get_gpu_data(self):
        """Light::GetGPUData overloads, Light.cpp:81-121."""
        if self.Type == Light.DIRECTIONAL:
            d = abi.DirectionalLight()
            self._common(d)
            d.enabled = int(self.bEnabled)
            d.shadowing = int(self.bCastingShadows)
            d.lightDirection.set(_quat_rotate(self.RotationQuaternion, (0.0, -1.0, 0.0)))   # default orientation looks down (:90)
            return d
        if self.Type == Light.POINT:
            p = abi.PointLight()
            self._common(p)
            p.position.set(self.Position)
            p.range = self.Range                             # attenuation is left untouched (:104)
            return p
        s = abi.SpotLight()
        self._common(s)
        s.spotDir.set(_quat_rotate(self.RotationQuaternion, (0.0, 0.0, 1.0)))               # default orientation looks forward (:112)
        s.position.set(self.Position)
        s.innerConeAngle = self.SpotInnerConeAngleDegrees * DEG2RAD
        s.outerConeAngle = self.SpotOuterConeAngleDegrees * DEG2RAD
        return s                                             # `range` is NOT set by the CPU side (:108-121)


def quat_mul(a, b):
    """Quaternion::operator* (Quaternion.cpp:151-164): (s1 s2 - v1.v2, s1 v2 + s2 v1 + v1 x v2), quaternions as (w, x, y, z)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    v = a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:])
    return (float(a[0] * b[0] - a[1:] @ b[1:]), float(v[0]), float(v[1]), float(v[2]))


def quat_from_axis_angle(axis, angle_rad):
    """Quaternion::FromAxisAngle (Quaternion.cpp:54-61): (cos(a/2), axis * sin(a/2)); the axis is NOT normalised by the engine."""
    h = 0.5 * float(angle_rad)
    return (math.cos(h),) + tuple(float(c) * math.sin(h) for c in axis)


def rotation_from_xml_euler_degrees(x_deg, y_deg, z_deg):
    """A scene file's `<Rotation> x y z </Rotation>` (FileParser.cpp:543-549): RotateAroundGlobalX, then Y, then Z axis, each
    `_rotation = q * _rotation` (Transform.h:62-74) starting from the identity."""
    q = (1.0, 0.0, 0.0, 0.0)
    for axis, deg in (((1, 0, 0), x_deg), ((0, 1, 0), y_deg), ((0, 0, 1), z_deg)):
        q = quat_mul(quat_from_axis_angle(axis, deg * DEG2RAD), q)
    return q


def default_scene_lights():
    """The <Light> elements of Data/Levels/Default.xml:202-308 (BASELINE config 1's light set) as the parser leaves them
    (FileParser.cpp:676-720): every element has a <Shadows> child, whose mere presence sets bCastingShadows (:710-713), so the
    directional light is shadowing and both spot lights land in spot_casters[]; the two point lights are disabled and
    GatherSceneLightData skips them. Mobility defaults to DYNAMIC where the element has none."""
    rot = rotation_from_xml_euler_degrees
    return [
        Light(Type=Light.DIRECTIONAL, Mobility=Light.STATIONARY, Color=(1.0, 1.0, 1.0), Range=100.0, Brightness=0.90, DepthBias=0.00045,
              RotationQuaternion=rot(0, 0, 40), bCastingShadows=True),                                                   # :202-221
        Light(Type=Light.POINT, Mobility=Light.DYNAMIC, Position=(12.5, 5.0, 5.0), bEnabled=False, Color=(0.4, 0.4, 0.85), Range=20.0,
              Brightness=1500.0, DepthBias=0.001, bCastingShadows=True),                                                 # :223-242
        Light(Type=Light.POINT, Mobility=Light.DYNAMIC, Position=(-12.5, 3.0, 0.0), bEnabled=False, Color=(0.4, 0.4, 0.15), Range=200.0,
              Brightness=35.0, DepthBias=0.05, bCastingShadows=True),                                                    # :243-262
        Light(Type=Light.SPOT, Mobility=Light.STATIC, Color=(0.9, 0.9, 0.9), Range=35.0, Brightness=1500.0, DepthBias=0.000009,
              Position=(22.0, 26.0, 4.0), RotationQuaternion=rot(90, 0, -15), SpotOuterConeAngleDegrees=22.0,
              SpotInnerConeAngleDegrees=20.0, bCastingShadows=True),                                                     # :264-284
        Light(Type=Light.SPOT, Mobility=Light.STATIC, Color=(0.4, 0.4, 0.15), Range=35.0, Brightness=1000.0, DepthBias=0.000005,
              Position=(-18.0, 20.0, 4.0), RotationQuaternion=rot(90, 0, 15), SpotOuterConeAngleDegrees=32.0,
              SpotInnerConeAngleDegrees=28.0, bCastingShadows=True),                                                     # :285-305
    ]


def _set_matrix(dst, m):
    m = np.asarray(m, np.float32).reshape(4, 4)
    for i in range(4):
        for j in range(4):
            dst.m[i][j] = float(m[i, j])


def gather_scene_light_data(lights):
    """Scene::GatherSceneLightData (Scene.cpp:978-1027): static, then stationary, then dynamic lights; disabled point/spot
    lights are skipped, the directional light is copied even when disabled (its `enabled` flag tells the shader)."""
    data = abi.SceneLighting()
    i_spot = i_spot_sh = i_point = i_point_sh = 0
    for mobility in (Light.STATIC, Light.STATIONARY, Light.DYNAMIC):
        for l in lights:
            if not l.bEnabled and l.Type != Light.DIRECTIONAL:
                continue
            if l.Mobility != mobility:
                continue
            if l.Type == Light.DIRECTIONAL:
                data.directional = l.get_gpu_data()
                if l.bCastingShadows and l.ViewProjection is not None:
                    _set_matrix(data.shadowViewDirectional, l.ViewProjection)
            elif l.Type == Light.SPOT:
                if l.bCastingShadows:
                    if i_spot_sh >= abi.NUM_SHADOWING_LIGHTS__SPOT:
                        raise ValueError("more than NUM_SHADOWING_LIGHTS__SPOT spot casters")
                    data.spot_casters[i_spot_sh] = l.get_gpu_data()
                    if l.ViewProjection is not None:
                        _set_matrix(data.shadowViews[i_spot_sh], l.ViewProjection)
                    i_spot_sh += 1
                else:
                    if i_spot >= abi.NUM_LIGHTS__SPOT:
                        raise ValueError("more than NUM_LIGHTS__SPOT spot lights")
                    data.spot_lights[i_spot] = l.get_gpu_data()
                    i_spot += 1
            else:
                if l.bCastingShadows:
                    if i_point_sh >= abi.NUM_SHADOWING_LIGHTS__POINT:
                        raise ValueError("more than NUM_SHADOWING_LIGHTS__POINT point casters")
                    data.point_casters[i_point_sh] = l.get_gpu_data()
                    i_point_sh += 1
                else:
                    if i_point >= abi.NUM_LIGHTS__POINT:
                        raise ValueError("more than NUM_LIGHTS__POINT point lights (use the extraPoint extension of the C ABI)")
                    data.point_lights[i_point] = l.get_gpu_data()
                    i_point += 1
    data.numPointCasters, data.numPointLights = i_point_sh, i_point
    data.numSpotCasters, data.numSpotLights = i_spot_sh, i_spot
    return data


INVALID_ID = -1


@dataclass
class Material:
    """Material.h:44-70 defaults; texture IDs are INVALID_ID (-1) when the map is absent."""
    diffuse: tuple = (1.0, 1.0, 1.0)
    alpha: float = 1.0
    emissiveColor: tuple = (1.0, 1.0, 1.0)
    emissiveIntensity: float = 0.0
    specular: tuple = (1.0, 1.0, 1.0)
    normalMapMipBias: float = 0.0
    tiling: tuple = (1.0, 1.0)
    uv_bias: tuple = (0.0, 0.0)
    roughness: float = 0.8
    metalness: float = 0.0
    displacement: float = 0.0
    TexDiffuseMap: int = INVALID_ID
    TexNormalMap: int = INVALID_ID
    TexEmissiveMap: int = INVALID_ID
    TexAlphaMaskMap: int = INVALID_ID
    TexMetallicMap: int = INVALID_ID
    TexRoughnessMap: int = INVALID_ID
    TexOcclusionRoughnessMetalnessMap: int = INVALID_ID
    TexAmbientOcclusionMap: int = INVALID_ID
    TexHeightMap: int = INVALID_ID

    def get_texture_config(self):
        """Material::GetTextureConfig, Material.cpp:23-36 (bit order must match HasXMap, LightingConstantBufferData.h:116-124)."""
        cfg = 0
        for bit, tex in enumerate((self.TexDiffuseMap, self.TexNormalMap, self.TexAmbientOcclusionMap, self.TexAlphaMaskMap,
                                   self.TexRoughnessMap, self.TexMetallicMap, self.TexHeightMap, self.TexEmissiveMap,
                                   self.TexOcclusionRoughnessMetalnessMap)):
            if tex != INVALID_ID:
                cfg |= 1 << bit
        return cfg

    def is_alpha_masked(self, diffuse_map_uses_alpha_channel=False):
        """Material::IsAlphaMasked (Material.cpp:39): an alpha-mask map is bound, or the diffuse map's alpha channel is in use
        (TextureManager.cpp:835 HasAlphaValuesSIMD). Selects the "_AlphaMasked" PSO == abi.MATERIAL_ALPHA_MASKED in texDiffuse.reserved."""
        return self.TexAlphaMaskMap != INVALID_ID or (self.TexDiffuseMap != INVALID_ID and bool(diffuse_map_uses_alpha_channel))

    def get_cbuffer_data(self):
        """Material::GetCBufferData, Material.h:120-127: memcpy of the first 80 bytes + textureConfig as a FLOAT."""
        d = abi.MaterialData()
        d.diffuse.set(self.diffuse); d.alpha = self.alpha
        d.emissiveColor.set(self.emissiveColor); d.emissiveIntensity = self.emissiveIntensity
        d.specular.set(self.specular); d.normalMapMipBias = self.normalMapMipBias
        d.uvScaleOffset = abi.float4(self.tiling[0], self.tiling[1], self.uv_bias[0], self.uv_bias[1])
        d.roughness, d.metalness, d.displacement = self.roughness, self.metalness, self.displacement
        d.textureConfig = float(self.get_texture_config())
        return d


def has_map(texture_config, bit):
    """HasDiffuseMap .. HasOcclusionRoughnessMetalnessMap, LightingConstantBufferData.h:116-124."""
    return 1 if (int(texture_config) & (1 << bit)) > 0 else 0


def gbuffer_from_material(material_data, world_pos, world_normal, ambient_factor, ssao=None):
    """G-buffer planes for pixels of ONE texture-less material == ForwardLighting.hlsl:PSMain :247-281 with every
    HasXMap() == 0 and a zero normal-map sample (length(Normal) < 0.01 -> Surface.N = normalize(WorldSpaceNormal), :266-267):
      gb0 = (P, ao) with ao = fAmbientLightingFactor * ssao (:247,280-281; SSAO target cleared to 1 when off, SceneRendering.cpp:1543-1553)
      gb1 = (normalize(N), roughness)   gb2 = (diffuse, metalness)   gb3 = (emissiveColor, emissiveIntensity)
    world_pos / world_normal: float arrays [..., 3]. Returns 4 float32 arrays [..., 4]."""
    if int(material_data.textureConfig) != 0:         # a host-side helper for TEXTURE-LESS materials only; materials with maps go through the producer kernel:
        raise ValueError("gbuffer_from_material() covers texture-less materials; pass textured materials to capi.Context.gbuffer_from_materials "
                         "(vqhip_gbuffer_from_materials) or capi.Context.forward_lighting_from_materials, which sample the maps on the GPU")
    P = np.asarray(world_pos, np.float32)
    N = np.asarray(world_normal, np.float32)
    shape = P.shape[:-1]
    ssao = np.ones(shape, np.float32) if ssao is None else np.asarray(ssao, np.float32)
    n = (N / np.sqrt((N.astype(np.float32) ** 2).sum(-1, keepdims=True))).astype(np.float32)
    g = [np.empty(shape + (4,), np.float32) for _ in range(4)]
    g[0][..., :3] = P
    g[0][..., 3] = np.float32(ambient_factor) * ssao
    g[1][..., :3] = n
    g[1][..., 3] = np.float32(material_data.roughness)
    g[2][..., :3] = np.array([material_data.diffuse.x, material_data.diffuse.y, material_data.diffuse.z], np.float32)
    g[2][..., 3] = np.float32(material_data.metalness)
    g[3][..., :3] = np.array([material_data.emissiveColor.x, material_data.emissiveColor.y, material_data.emissiveColor.z], np.float32)
    g[3][..., 3] = np.float32(material_data.emissiveIntensity)
    return g


def skydome_params(camera_yaw, camera_pitch, hdri_yaw_offset, fov_y, viewport_width, viewport_height):
    """abi.SkydomeParams of the sky camera built at Scene.cpp:573-584: position 0, yaw = MainViewCameraYaw + HDRIYawOffset,
    pitch = MainViewCameraPitch, the main camera's perspective projection (vertical FoV `fov_y` in radians,
    Camera::SetProjectionMatrix, Camera.cpp:84-91). Basis as Camera::UpdateViewMatrix (Camera.cpp:94-110): lookAt (0,0,1) and
    up (0,1,0) rotated by XMMatrixRotationRollPitchYaw(pitch, yaw, 0), then XMMatrixLookAtLH: z = forward,
    x = normalize(cross(up, z)), y = cross(z, x). Angles in radians; computed in float64 and rounded to float32."""
    yaw, p = float(camera_yaw) + float(hdri_yaw_offset), float(camera_pitch)
    sy, cy, sp, cp = math.sin(yaw), math.cos(yaw), math.sin(p), math.cos(p)
    fwd = np.array([cp * sy, -sp, cp * cy])                 # (0,0,1) * Rx(pitch) * Ry(yaw), row-vector convention
    up0 = np.array([sp * sy, cp, sp * cy])                  # (0,1,0) * Rx(pitch) * Ry(yaw)
    right = np.cross(up0, fwd); right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    t = math.tan(0.5 * float(fov_y))
    sp_ = abi.SkydomeParams()
    sp_.right.set(tuple(right)); sp_.up.set(tuple(up)); sp_.forward.set(tuple(fwd))
    sp_.tanHalfFovY = t
    sp_.tanHalfFovX = t * float(viewport_width) / float(viewport_height)
    return sp_


# ---- which HDRI file an environment map is loaded from (Source/Engine/EnvironmentMap.cpp) ----------------------
HDRI_DIMENSIONS = {8: (8192, 4096), 4: (4096, 2048), 2: (2048, 1024), 1: (1024, 512)}   # LookupResolutionX/Y, EnvironmentMap.cpp:163-164


def determine_resolution_hdri(file_path, monitor_resolution_y):
    """DetermineResolution_HDRI (EnvironmentMap.cpp:69-91): the `%resolution%` token of an environment-map path is replaced by
    1k / 2k / 4k / 8k after the height of the monitor the swap chain is on (<720, <1080, <=1440, else 8k). A path without a token is
    returned unchanged with the default "1k". Returns (resolution, path)."""
    resolution = "1k"
    i = file_path.find("%")
    if i >= 0:
        if monitor_resolution_y < 720:
            resolution = "1k"
        elif monitor_resolution_y < 1080:
            resolution = "2k"
        elif monitor_resolution_y <= 1440:
            resolution = "4k"
        else:
            resolution = "8k"
        j = file_path.rfind("%")
        file_path = file_path[:i] + resolution + file_path[j + 1:]
    return resolution, file_path


def find_environment_map_to_downsize_from(files_in_folder, env_map_name, target_resolution):
    """FindEnvironmentMapToDownsizeFrom (EnvironmentMap.cpp:92-136): when the file of the chosen resolution is missing, the first file of
    the HDRI folder whose name contains the map's name and ends in a `_<n>k` resolution token is the source to downsize from ("" when
    the next higher resolution would exceed 8k or nothing matches). The resize itself (`Image::CreateResizedImage`, stb_image_resize in
    the un-vendored VQUtils submodule) and the write-back to disk are outside this build — DESIGN.md §7.3."""
    assert len(target_resolution) >= 2
    if (ord(target_resolution[0]) - ord("0")) * 2 > 8:
        return ""
    for path in files_in_folder:
        name = path.replace("\\", "/").rsplit("/", 1)[-1].rsplit(".", 1)[0]
        if env_map_name in name:
            tok = name.split("_")[-1]
            if len(tok) >= 2 and tok[-2].isalnum() and tok[0].isalnum() and tok[-1] == "k":
                return path
    return ""


# ---- shadow views of the lights (Light.cpp:133-233) and the two caster workloads the bench times ----------------------------------
def _quat_to_matrix(q):
    """Rotation matrix (row-vector convention, like DirectXMath) of the unit quaternion (w, x, y, z): rows are the images of the axes."""
    return np.stack([_quat_rotate(q, e) for e in ((1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))])


def orthographic_lh(view_w, view_h, zn, zf):
    """DirectX::XMMatrixOrthographicLH (float64)."""
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1], m[2, 2], m[3, 2], m[3, 3] = 2.0 / view_w, 2.0 / view_h, 1.0 / (zf - zn), -zn / (zf - zn), 1.0
    return m


def light_view_projection(light, near_plane, far_plane, viewport=(2048, 2048), distance_from_origin=500.0):
    """Light::GetViewProjectionMatrix for a spot or directional light (Light.cpp:133-233): spot = LookAtLH(pos, pos + rotated forward, rotated up) x
    PerspectiveFovLH(pi/2, 1, near, far); directional = LookAtLH(-direction * distance, origin, up) x OrthographicLH(ViewportX, ViewportY, near, far).
    float64 here, rounded to binary32 when stored in the cbuffer (the engine computes in binary32 SIMD: these are INPUTS of the path, not results)."""
    from . import synth
    if light.Type == Light.SPOT:
        rot = _quat_to_matrix(light.RotationQuaternion)
        pos = np.asarray(light.Position, np.float64)
        view = synth._look_at_lh(pos, pos + rot[2], rot[1])
        return view @ synth._perspective_fov_lh(math.pi / 2.0, 1.0, near_plane, far_plane)
    if light.Type == Light.DIRECTIONAL:
        direction = _quat_rotate(light.RotationQuaternion, (0.0, -1.0, 0.0))
        pos = -direction * distance_from_origin
        up = np.array([0.0, 1.0, 0.0])
        ldu = float((-pos / np.linalg.norm(pos)) @ up)
        if ldu in (1.0, -1.0):                                # Light.cpp:199-205: nudge the up vector when it is parallel to the light
            up = up + np.array([0.001, 0.0, 0.0]); up /= np.linalg.norm(up)
        view = synth._look_at_lh(pos, (0.0, 0.0, 0.0), up)
        return view @ orthographic_lh(float(viewport[0]), float(viewport[1]), near_plane, far_plane)
    raise ValueError("point lights render six faces: CubemapUtility::CalculateViewMatrix")


def _quat_look_down(tilt_x_deg, tilt_z_deg):
    return rotation_from_xml_euler_degrees(90.0 + tilt_x_deg, 0.0, tilt_z_deg)


def synthetic_shadow_maps(dims=(2048, 1024, 1024), n_spot=5, n_point=5, seed=0x5AD0):
    """Depth maps at the engine's sizes (SceneRendering.cpp:439-441: 2048^2 directional, n x 1024^2 spot, n x 6 x 1024^2 point faces): a lit half (depth 1) with a wavy
    border and structured occluders + grain in the other, so the PCF kernels see lit, shadowed and penumbra taps. Content does not change the kernel's work (every tap is
    fetched and compared), only the fetch addresses' locality, which comes from the pixels' positions."""
    rng = np.random.Generator(np.random.Philox(key=[int(seed), 0x55]))

    def depth(shape, lo, hi, fx, fy):
        h, w = shape[-2:]
        yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
        base = (lo + (hi - lo) * (0.5 + 0.5 * np.sin(xx * fx) * np.cos(yy * fy))).astype(np.float32)
        out = np.empty(shape, np.float32)
        for n, idx in enumerate(np.ndindex(*shape[:-2])):
            k = np.float32(1.0 + 0.13 * n)
            out[idx] = base * np.float32((k % 1.0) * 0.2 + 0.9) + rng.random((h, w), dtype=np.float32) * np.float32(0.02)
            out[idx][xx > w * 0.5 + 0.1 * w * np.sin(yy * 0.01 * k)] = 1.0
        return out
    return {"dir": depth((dims[0], dims[0]), 0.4, 0.6, 0.011, 0.017), "spot": depth((n_spot, dims[1], dims[1]), 0.35, 0.65, 0.02, 0.013),
            "point": depth((n_point, 6, dims[2], dims[2]), 0.05, 0.55, 0.015, 0.019), "dims": tuple(dims)}


def _set_shadow_dims(pf, dims):
    pf.f2DirectionalLightShadowMapDimensions = abi.float2(float(dims[0]), float(dims[0]))
    pf.f2SpotLightShadowMapDimensions = abi.float2(float(dims[1]), float(dims[1]))
    pf.f2PointLightShadowMapDimensions = abi.float2(float(dims[2]), float(dims[2]))


def default_scene_frame(map_dims=(2048, 1024, 1024)):
    """BASELINE config 1's light set as the engine would hand it to PSMain: Data/Levels/Default.xml:202-308 through the parser (default_scene_lights) and
    Scene::GatherSceneLightData, every caster with ITS OWN view-projection matrix (Light::GetViewProjectionMatrix: the directional light's 256 x 256 orthographic
    volume at distance 120, the two spots' 90-degree perspective frusta, near / far planes of the file) and shadow maps of the engine's sizes.
    Returns (PerFrameData, maps) with maps = {"dir", "spot", "point", "dims"} (host float32 arrays)."""
    lights = default_scene_lights()
    shadows = {0: (0.1, 15000.0), 3: (0.001, 1500.0), 4: (0.001, 1500.0)}           # <NearPlane>, <FarPlane> of the three enabled casters
    for i, (zn, zf) in shadows.items():
        lights[i].ViewProjection = light_view_projection(lights[i], zn, zf, viewport=(256, 256), distance_from_origin=120.0)
    pf = abi.PerFrameData()
    pf.Lights = gather_scene_light_data(lights)
    pf.fAmbientLightingFactor = 0.055
    _set_shadow_dims(pf, map_dims)
    return pf, synthetic_shadow_maps(map_dims, n_spot=abi.NUM_SHADOWING_LIGHTS__SPOT, n_point=1)


def engine_max_frame(map_dims=(2048, 1024, 1024)):
    """The most lights the engine's cbuffer holds (LightingConstantBufferData.h:39-44): 100 point + 20 spot lights, the directional light shadowing, 5 spot casters and
    5 point casters, shadow maps at the engine's sizes. Lights come from the same distributions as the BASELINE configs (synth.point_lights / spot_lights), the casters get
    real view-projection matrices. Returns (PerFrameData, maps)."""
    from . import synth
    pf, _ = synth.per_frame(points=synth.point_lights(abi.NUM_LIGHTS__POINT, seed=0xE7A0), spots=synth.spot_lights(abi.NUM_LIGHTS__SPOT, seed=0xE7A1))
    lights = [Light(Type=Light.DIRECTIONAL, Mobility=Light.STATIONARY, Brightness=0.9, DepthBias=0.00045, RotationQuaternion=rotation_from_xml_euler_degrees(0, 0, 40),
                    bCastingShadows=True)]
    lights[0].ViewProjection = light_view_projection(lights[0], 0.1, 15000.0, viewport=(256, 256), distance_from_origin=120.0)
    r = np.random.Generator(np.random.Philox(key=[0xE7A2, 0x66]))
    for i in range(abi.NUM_SHADOWING_LIGHTS__SPOT):
        u = r.random(8)
        l = Light(Type=Light.SPOT, Mobility=Light.STATIC, Position=(-35.0 + 70.0 * u[0], 18.0 + 12.0 * u[1], -35.0 + 70.0 * u[2]),
                  Color=(0.3 + 0.7 * u[3], 0.3 + 0.7 * u[4], 0.3 + 0.7 * u[5]), Brightness=1000.0 + 1000.0 * u[6], DepthBias=0.000009,
                  RotationQuaternion=_quat_look_down(0.0, -20.0 + 40.0 * u[7]), SpotOuterConeAngleDegrees=30.0 + 2.0 * i, SpotInnerConeAngleDegrees=24.0 + 2.0 * i,
                  bCastingShadows=True)
        l.ViewProjection = light_view_projection(l, 0.001, 1500.0)
        lights.append(l)
    pc = synth.point_lights(abi.NUM_SHADOWING_LIGHTS__POINT, seed=0xE7A3)
    for i in range(abi.NUM_SHADOWING_LIGHTS__POINT):
        p = pc[i].position
        lights.append(Light(Type=Light.POINT, Mobility=Light.DYNAMIC, Position=(p.x, p.y, p.z), Range=float(np.float32(120.0 + 30.0 * i)),
                            Color=(pc[i].color.x, pc[i].color.y, pc[i].color.z), Brightness=pc[i].brightness, DepthBias=0.00005, bCastingShadows=True))
    g = gather_scene_light_data(lights)
    L = pf.Lights
    L.directional, L.shadowViewDirectional = g.directional, g.shadowViewDirectional
    L.numSpotCasters, L.numPointCasters = g.numSpotCasters, g.numPointCasters
    for i in range(g.numSpotCasters):
        L.spot_casters[i] = g.spot_casters[i]
        L.shadowViews[i] = g.shadowViews[i]
    for i in range(g.numPointCasters):
        L.point_casters[i] = g.point_casters[i]
    _set_shadow_dims(pf, map_dims)
    return pf, synthetic_shadow_maps(map_dims, n_spot=abi.NUM_SHADOWING_LIGHTS__SPOT, n_point=abi.NUM_SHADOWING_LIGHTS__POINT, seed=0x5AD1)


def shadow_maps_struct(maps, to_ptr):
    """abi.ShadowMaps over the three arrays of `maps` (to_ptr: array -> address; the caller keeps the arrays alive)."""
    d = maps["dims"]
    return abi.ShadowMaps(to_ptr(maps["dir"]), d[0], to_ptr(maps["spot"]), d[1], to_ptr(maps["point"]), d[2])
