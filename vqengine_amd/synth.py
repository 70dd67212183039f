"""Seeded synthetic inputs of the BASELINE.json configs. numpy only, CPU side;
every generator is keyed so that any row range of a frame can be produced independently (row-tiled
multi-GPU ranks generate just their tile and still agree with the single-GPU frame).

Parameter sources in the reference: light defaults Source/Engine/Scene/Light.cpp:58-73, brightness range
Data/Levels/Default.xml:202-308, fAmbientLightingFactor 0.055 Source/Engine/Scene/SceneViews.h:61,
sun radiance cf. MaxCLL values in Data/EnvironmentMaps.ini."""
import numpy as np

from . import abi

ROW_CHUNK = 32


def _chunk_rng(seed, chunk):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(chunk)]))


def gbuffer_rows(width, frame_height, row0, row1, seed=0xC0FFEE):
    """Rows [row0,row1) of the 4 float4 G-buffer planes of a width x frame_height frame.
    Returns 4 float32 arrays [row1-row0, width, 4]."""
    planes = [np.empty((row1 - row0, width, 4), np.float32) for _ in range(4)]
    c0, c1 = row0 // ROW_CHUNK, (row1 + ROW_CHUNK - 1) // ROW_CHUNK
    for ch in range(c0, c1):
        r = _chunk_rng(seed, ch)
        n = ROW_CHUNK
        rows = np.arange(ch * ROW_CHUNK, ch * ROW_CHUNK + n)
        u = r.random((n, width, 16), dtype=np.float32)
        g = [np.empty((n, width, 4), np.float32) for _ in range(4)]
        # gb0 = (P, ao): P on a height field over [-50,50]^2, y in [-5,5]
        cols = np.arange(width, dtype=np.float32)
        g[0][..., 0] = (-50.0 + 100.0 * (cols + 0.5) / width)[None, :]
        g[0][..., 1] = -5.0 + 10.0 * u[..., 0]
        g[0][..., 2] = (-50.0 + 100.0 * (rows.astype(np.float32) + 0.5) / frame_height)[:, None]
        g[0][..., 3] = 0.055 * (0.3 + 0.7 * u[..., 1])
        # gb1 = (N, roughness): uniform upper hemisphere (+Y up), perturbed by +-1e-3 and NOT renormalised
        z = u[..., 2]
        phi = 2.0 * np.pi * u[..., 3]
        rad = np.sqrt(np.maximum(0.0, 1.0 - z * z))
        g[1][..., 0] = rad * np.cos(phi) + (u[..., 4] - 0.5) * 2e-3
        g[1][..., 1] = z + (u[..., 5] - 0.5) * 2e-3
        g[1][..., 2] = rad * np.sin(phi) + (u[..., 6] - 0.5) * 2e-3
        g[1][..., 3] = 0.05 + 0.95 * u[..., 7]
        # gb2 = (diffuse linear, metalness in {0,1} w.p. 1/2 else U[0,1])
        g[2][..., 0:3] = u[..., 8:11]
        m = u[..., 11]
        g[2][..., 3] = np.where(m < 0.25, 0.0, np.where(m < 0.5, 1.0, (m - 0.5) * 2.0))
        # gb3 = (emissive colour, intensity): zero for 95 % of the pixels
        em = u[..., 12] > 0.95
        g[3][..., 0:3] = np.where(em[..., None], u[..., 13:16], 0.0)
        g[3][..., 3] = np.where(em, 5.0 * ((u[..., 12] - 0.95) * 20.0), 0.0)
        lo, hi = max(row0, ch * ROW_CHUNK), min(row1, (ch + 1) * ROW_CHUNK)
        for k in range(4):
            planes[k][lo - row0:hi - row0] = g[k][lo - ch * ROW_CHUNK:hi - ch * ROW_CHUNK].astype(np.float32)
    return planes


def gbuffer(width, height, seed=0xC0FFEE, coherent=False):
    return (gbuffer_rows_coherent if coherent else gbuffer_rows)(width, height, 0, height, seed)


def _hash01(a, b, c):
    """Integer hash of (a, b, c) -> float32 in [0,1): material parameters of a region, independent of how the frame is cut into rows."""
    h = (np.asarray(a, np.uint64) * np.uint64(0x9E3779B1) + np.asarray(b, np.uint64) * np.uint64(0x85EBCA77) + np.uint64(c) * np.uint64(0xC2B2AE3D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15); h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12); h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return (h >> np.uint64(8)).astype(np.float32) / np.float32(1 << 24)


def gbuffer_rows_coherent(width, frame_height, row0, row1, seed=0xC0FFEE):
    """A SURFACE-COHERENT frame, next to the white-noise one of gbuffer_rows (which stays the BASELINE workload): what the engine's rasteriser
    hands to PSMain on real content. Rolling terrain over [-50,50]^2 seen from above (P on a smooth height field, N = the field's analytic normal
    with a fine normal-map ripple, NOT renormalised), materials in 96 x 64-pixel regions: per-region albedo / metalness / roughness with a
    little per-pixel grain, 12 % of the regions POLISHED (roughness 0, 0.01, 0.02 or 0.03: the GGX EPSILON early-out range, shade.hip), 3 % emissive.
    Any row range can be produced on its own (row-tiled ranks agree with the untiled frame)."""
    planes = [np.empty((row1 - row0, width, 4), np.float32) for _ in range(4)]
    c0, c1 = row0 // ROW_CHUNK, (row1 + ROW_CHUNK - 1) // ROW_CHUNK
    cols = np.arange(width, dtype=np.float32)
    X = (-50.0 + 100.0 * (cols + 0.5) / width)[None, :].astype(np.float32)
    for ch in range(c0, c1):
        r = _chunk_rng(seed ^ 0x5EED, ch)
        n = ROW_CHUNK
        rows = np.arange(ch * ROW_CHUNK, ch * ROW_CHUNK + n)
        u = r.random((n, width, 6), dtype=np.float32)
        Z = (-50.0 + 100.0 * (rows.astype(np.float32) + 0.5) / frame_height)[:, None].astype(np.float32)
        g = [np.empty((n, width, 4), np.float32) for _ in range(4)]
        # height field and its gradient
        hgt = 3.0 * np.sin(0.11 * X) * np.cos(0.07 * Z) + 1.2 * np.sin(0.31 * X + 0.23 * Z) + 0.4 * np.cos(0.9 * Z)
        dhx = 3.0 * 0.11 * np.cos(0.11 * X) * np.cos(0.07 * Z) + 1.2 * 0.31 * np.cos(0.31 * X + 0.23 * Z)
        dhz = -3.0 * 0.07 * np.sin(0.11 * X) * np.sin(0.07 * Z) + 1.2 * 0.23 * np.cos(0.31 * X + 0.23 * Z) - 0.4 * 0.9 * np.sin(0.9 * Z)
        # fine ripple (a tiling normal map) on top of the geometric normal
        dhx = dhx + 0.15 * np.sin(5.3 * X) * np.cos(4.1 * Z)
        dhz = dhz + 0.15 * np.cos(4.7 * X) * np.sin(5.9 * Z)
        inv = (1.0 / np.sqrt(dhx * dhx + dhz * dhz + 1.0)).astype(np.float32)
        g[0][..., 0] = np.broadcast_to(X, (n, width)); g[0][..., 1] = hgt; g[0][..., 2] = np.broadcast_to(Z, (n, width))
        g[0][..., 3] = 0.055 * (0.55 + 0.45 * np.cos(0.5 * X) * np.sin(0.4 * Z) ** 2)
        g[1][..., 0] = -dhx * inv; g[1][..., 1] = inv; g[1][..., 2] = -dhz * inv
        # material regions
        bx = (np.arange(width) // 96)[None, :] + np.zeros((n, 1), np.int64)
        by = (rows // 64)[:, None] + np.zeros((1, width), np.int64)
        hk = lambda c: _hash01(bx, by, c + (int(seed) & 0xFFFF) * 16)      # noqa: E731
        kind = hk(0)
        rough = np.where(kind < 0.12, np.floor(hk(1) * 4.0) * 0.01, 0.05 + 0.95 * hk(1)).astype(np.float32)
        rough = np.where(kind < 0.12, rough, np.clip(rough + 0.02 * (u[..., 0] - 0.5), 0.05, 1.0)).astype(np.float32)
        g[1][..., 3] = rough
        metal = np.where(kind < 0.12, 1.0, np.where(hk(2) < 0.5, 0.0, np.where(hk(2) < 0.7, 1.0, hk(3)))).astype(np.float32)
        for c in range(3):
            g[2][..., c] = np.clip(0.1 + 0.85 * hk(4 + c) + 0.06 * (u[..., 1 + c] - 0.5), 0.0, 1.0)
        g[2][..., 3] = metal
        em = hk(7) > 0.97
        for c in range(3):
            g[3][..., c] = np.where(em, hk(8 + c), 0.0)
        g[3][..., 3] = np.where(em, 1.0 + 4.0 * hk(11), 0.0)
        lo, hi = max(row0, ch * ROW_CHUNK), min(row1, (ch + 1) * ROW_CHUNK)
        for k in range(4):
            planes[k][lo - row0:hi - row0] = g[k][lo - ch * ROW_CHUNK:hi - ch * ROW_CHUNK].astype(np.float32)
    return planes


def point_lights(n, seed=0x1600):
    """n PointLight records: pos U([-50,50]x[0,20]x[-50,50]), colour U[0.2,1]^3, brightness U[100,1500], range U[20,200]."""
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x11]))
    arr = (abi.PointLight * n)()
    for i in range(n):
        u = r.random(8, dtype=np.float32)
        l = arr[i]
        l.position.set((-50 + 100 * u[0], 20 * u[1], -50 + 100 * u[2]))
        l.range = float(np.float32(20 + 180 * u[3]))
        l.color.set((0.2 + 0.8 * u[4], 0.2 + 0.8 * u[5], 0.2 + 0.8 * u[6]))
        l.brightness = float(np.float32(100 + 1400 * u[7]))
        l.attenuation.set((1.0, 0.0, 0.0))
        l.depthBias = 0.0
    return arr


def spot_lights(n, seed=0x5907):
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x22]))
    arr = (abi.SpotLight * n)()
    for i in range(n):
        u = r.random(12, dtype=np.float32)
        l = arr[i]
        l.position.set((-40 + 80 * u[0], 10 + 20 * u[1], -40 + 80 * u[2]))
        l.color.set((0.2 + 0.8 * u[3], 0.2 + 0.8 * u[4], 0.2 + 0.8 * u[5]))
        l.brightness = float(np.float32(500 + 1500 * u[6]))
        d = np.array([u[7] - 0.5, -1.0, u[8] - 0.5], np.float32)
        l.spotDir.set(d * 3.0)                       # deliberately not unit length: the shader normalises (Lighting.hlsl:60)
        outer = np.float32(np.deg2rad(20 + 25 * u[9]))
        l.outerConeAngle = float(outer)
        l.innerConeAngle = float(np.float32(outer * (0.6 + 0.3 * u[10])))
        l.depthBias = 5e-5
        l.range = 0.0                                # unset by the CPU side, Light.cpp:108-121
    return arr


def per_frame(points=None, spots=None, directional=None, ambient=0.055, hdri_offset=0.0):
    """PerFrameData with up to 100 point lights in the cbuffer array; returns (PerFrameData, extra PointLight array or None)."""
    pf = abi.PerFrameData()
    n = len(points) if points is not None else 0
    k = min(n, abi.NUM_LIGHTS__POINT)
    pf.Lights.numPointLights = k
    for i in range(k):
        pf.Lights.point_lights[i] = points[i]
    extra = None
    if n > k:
        extra = (abi.PointLight * (n - k))(*[points[i] for i in range(k, n)])
    if spots is not None:
        pf.Lights.numSpotLights = len(spots)
        for i in range(len(spots)):
            pf.Lights.spot_lights[i] = spots[i]
    if directional is not None:
        pf.Lights.directional = directional
    pf.f2PointLightShadowMapDimensions = abi.float2(1024.0, 1024.0)          # SceneRendering.cpp:439-441
    pf.f2SpotLightShadowMapDimensions = abi.float2(1024.0, 1024.0)
    pf.f2DirectionalLightShadowMapDimensions = abi.float2(2048.0, 2048.0)
    pf.fAmbientLightingFactor = ambient
    pf.fHDRIOffsetInRadians = hdri_offset
    return pf, extra


def per_view(width, height, camera=(0.0, 10.0, -60.0), max_env_lod=0, diffuse_only=0):
    pv = abi.PerViewLightingData()
    for m in (pv.matView, pv.matViewToWorld, pv.matProjInverse):
        for i in range(4):
            m.m[i][i] = 1.0
    pv.CameraPosition.set(camera)
    pv.MaxEnvMapLODLevels = float(max_env_lod)
    pv.ScreenDimensions = abi.float2(float(width), float(height))
    pv.EnvironmentMapDiffuseOnlyIllumination = diffuse_only
    return pv


def directional_light(direction=(0.3, -1.0, 0.2), color=(1.0, 0.95, 0.9), brightness=0.9, shadowing=0, enabled=1, depth_bias=5e-5):
    d = abi.DirectionalLight()
    d.lightDirection.set(direction)
    d.color.set(color)
    d.brightness = brightness
    d.depthBias = depth_bias
    d.shadowing = shadowing
    d.enabled = enabled
    return d


def equirect(width, height, seed=0xE9):
    """RGBA32F equirect [H,W,4]: smooth sky gradient + 8 Gaussian 'suns' with peak radiance up to 2.6e4."""
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x33]))
    v = (np.arange(height, dtype=np.float32) + 0.5) / height
    u = (np.arange(width, dtype=np.float32) + 0.5) / width
    uu, vv = np.meshgrid(u, v)
    sky = np.stack([0.25 + 0.35 * (1 - vv), 0.35 + 0.4 * (1 - vv), 0.55 + 0.45 * (1 - vv)], -1)
    ground = np.stack([0.18 + 0.05 * np.sin(12 * uu), 0.15 + 0.05 * np.cos(9 * uu), 0.10 + 0.0 * uu], -1)
    t = np.clip((vv - 0.5) * 8.0 + 0.5, 0, 1)[..., None]
    img = sky * (1 - t) + ground * t
    for _ in range(8):
        cu, cv = r.random(), 0.05 + 0.4 * r.random()
        sig = 0.004 + 0.02 * r.random()
        peak = 10 ** (1.5 + 2.9 * r.random())          # 30 .. 2.5e4
        col = 0.6 + 0.4 * r.random(3)
        du = np.minimum(np.abs(uu - cu), 1 - np.abs(uu - cu))
        img += (peak * np.exp(-(du * du + (vv - cv) ** 2) / (2 * sig * sig)))[..., None] * col
    out = np.empty((height, width, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    out[..., 3] = 1.0
    return out


def hdr_image(width, height, seed=0x70E, scale=4.0):
    """RGBA32F scene-colour-like image for blur/tonemap tests: log-uniform radiance with a few hot pixels."""
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x44]))
    img = np.exp(r.uniform(-6, np.log(scale), (height, width, 4))).astype(np.float32)
    hot = r.random((height, width)) > 0.999
    img[hot, :3] *= 500.0
    img[..., 3] = r.random((height, width), dtype=np.float32)
    return img


# ---- inputs of the G-buffer producer -------------------------------------------------------
def interpolants(width, height, n_materials, seed=0x1A7E):
    """PSInput planes (ForwardLighting.hlsl:42-53) of a synthetic view: a ground plane receding towards the top of the
    image (so the uv footprint — hence the mip level — grows with the row), material indices in 97x61-pixel blocks
    (odd sizes: pixel quads straddle material borders), a sky band of no-geometry pixels and a few stray / invalid indices.
    Returns 3 float32 arrays [H,W,4]: (P, u), (N, v), (T, asfloat(int32 index))."""
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x1F]))
    ys, xs = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    t = (ys + 0.5) / height
    s = (xs + 0.5) / width
    z = (4.0 + 120.0 * (1.0 - t) ** 3).astype(np.float32)
    px = ((s - 0.5) * z * 1.4).astype(np.float32)
    py = (0.25 * np.sin(px * 0.7) * np.cos(z * 0.3)).astype(np.float32)
    ip0 = np.empty((height, width, 4), np.float32)
    ip1 = np.empty_like(ip0)
    ip2 = np.empty_like(ip0)
    ip0[..., 0], ip0[..., 1], ip0[..., 2] = px, py, z
    ip0[..., 3] = px * 0.37 + 0.11                      # uv.x
    ip1[..., 3] = z * 0.37 - 0.05                       # uv.y
    # interpolated (unnormalised) normal / tangent, deliberately not orthogonal
    ip1[..., 0] = 0.35 * np.sin(px * 1.3) + 0.02 * (r.random((height, width), dtype=np.float32) - 0.5)
    ip1[..., 1] = 1.3
    ip1[..., 2] = 0.35 * np.cos(z * 0.9) + 0.02 * (r.random((height, width), dtype=np.float32) - 0.5)
    ip2[..., 0] = 0.9
    ip2[..., 1] = 0.15 * np.sin(z * 0.5)
    ip2[..., 2] = 0.2 * np.cos(px * 0.4)
    bx, by = (xs.astype(np.int64) // 97), (ys.astype(np.int64) // 61)
    idx = ((bx * 5 + by * 3) % max(n_materials, 1)).astype(np.int32)
    idx[ys < height * 0.08] = -1                        # sky
    stray = r.random((height, width)) > 0.9995
    idx[stray] = np.where(r.random(int(stray.sum())) > 0.5, -7, n_materials + 3).astype(np.int32)
    ip2[..., 3] = idx.view(np.float32)
    return ip0, ip1, ip2


def clip_positions(width, height, seed=0x5C11):
    """PSInput.svPositionCurr / svPositionPrev (ForwardLighting.hlsl:49-52) of a synthetic view: clip-space positions whose xy / w land near the pixel's NDC
    position, w = view depth in [0.3, 120] growing towards the top of the image like synth.interpolants; the previous frame is the same geometry under a
    slightly different camera (a few pixels of motion, more for near geometry). Two float32 arrays [H,W,4]."""
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x2B]))
    ys, xs = np.meshgrid(np.arange(height, dtype=np.float32), np.arange(width, dtype=np.float32), indexing="ij")
    t, s = (ys + 0.5) / height, (xs + 0.5) / width
    w = (0.3 + 120.0 * (1.0 - t) ** 3).astype(np.float32)
    cur = np.empty((height, width, 4), np.float32)
    cur[..., 0] = (2.0 * s - 1.0) * w
    cur[..., 1] = (1.0 - 2.0 * t) * w
    cur[..., 2] = w - 0.1
    cur[..., 3] = w
    prev = cur.copy()
    prev[..., 0] += np.float32(0.03) + 0.002 * (r.random((height, width), dtype=np.float32) - 0.5)
    prev[..., 1] -= np.float32(0.011) + 0.002 * (r.random((height, width), dtype=np.float32) - 0.5)
    prev[..., 3] *= np.float32(1.004)
    return cur, prev


def material_set(n, seed=0x3A7, max_dim=256, same_size=False):
    """n materials: (list of abi.MaterialData, list of {slot: uint8 [H,W,4] level 0}) with power-of-two texture sizes
    (non-square allowed), random scalar parameters and uv tiling. textureConfig normally mirrors the bound maps
    (Material::GetTextureConfig, Material.cpp:23-36); materials 3k+1 carry a deliberately inconsistent config (bit set
    without a map / map without its bit) to exercise the null-SRV and ignored-map branches."""
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x77]))
    bits = {"texDiffuse": 0, "texNormals": 1, "texLocalAO": 2, "texRoughness": 4, "texMetalness": 5, "texEmissive": 7, "texOcclRoughMetal": 8}
    datas, texsets = [], []
    for i in range(n):
        d = abi.MaterialData()
        d.diffuse.set(tuple(r.uniform(0.2, 1.0, 3))); d.alpha = 1.0
        d.emissiveColor.set(tuple(r.uniform(0.0, 1.0, 3))); d.emissiveIntensity = float(r.uniform(0, 3)) if i % 3 == 0 else 0.0
        d.specular.set((1.0, 1.0, 1.0)); d.normalMapMipBias = float(r.choice([0.0, -0.5, 0.75, 1.0]))
        d.uvScaleOffset = abi.float4(float(r.uniform(0.3, 6.0)), float(r.uniform(0.3, 6.0)), float(r.uniform(-1, 1)), float(r.uniform(-1, 1)))
        d.roughness, d.metalness, d.displacement = float(r.uniform(0.05, 1.0)), float(r.uniform(0, 1)), 0.0
        texs, cfg = {}, 0
        if same_size:                                   # one square size per material (how material sets are normally authored)
            mw = mh = int(2 ** r.integers(max(3, int(np.log2(max_dim)) - 1), int(np.log2(max_dim)) + 1))
        for slot in abi.MATERIAL_TEXTURE_SLOTS:
            if i == 0 or r.random() < 0.6:              # material 0 binds every map
                w = int(2 ** r.integers(3, int(np.log2(max_dim)) + 1)); h = int(2 ** r.integers(3, int(np.log2(max_dim)) + 1))
                if same_size:
                    w, h = mw, mh
                yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
                img = np.empty((h, w, 4), np.float32)
                for c in range(4):
                    fx, fy, ph = r.integers(1, 5), r.integers(1, 5), r.uniform(0, 6.28)
                    img[..., c] = 0.5 + 0.35 * np.sin(2 * np.pi * (fx * xx / w + fy * yy / h) + ph) + 0.15 * (r.random((h, w)) - 0.5)
                if slot == "texNormals":                # mostly +Z tangent-space normals
                    img[..., 2] = 0.75 + 0.25 * img[..., 2]
                    if i % 4 == 2:
                        img[: h // 2] = 0.0            # a region of all-zero normals: the length(Normal) < 0.01 branch
                texs[slot] = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
                cfg |= 1 << bits[slot]
        if i % 3 == 1:
            cfg ^= (1 << 0) | (1 << 4) | (1 << 8)
        d.textureConfig = float(cfg)
        datas.append(d)
        texsets.append(texs)
    return datas, texsets


def ssao_image(width, height, seed=0x55A0):
    r = np.random.Generator(np.random.Philox(key=[int(seed), 0x0A]))
    return r.integers(96, 256, (height, width), dtype=np.uint8)


# ---- Radiance .hdr files (the reference's HDRI assets, Data/EnvironmentMaps.ini) ----------------
def float_to_rgbe(rgb):
    """float [...,3] -> uint8 [...,4] RGBE (Ward's float2rgbe: mantissas scaled by 256/2^e of the largest channel)."""
    rgb = np.asarray(rgb, np.float64)
    v = rgb.max(-1)
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    ok = v >= 1e-32
    m, e = np.frexp(np.where(ok, v, 1.0))
    scale = np.where(ok, m * 256.0 / np.where(ok, v, 1.0), 0.0)[..., None]
    out[..., :3] = np.clip(np.floor(rgb * scale), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(ok, e + 128, 0).astype(np.uint8)
    out[~ok] = 0
    return out


def _rle_plane(row):
    """New-style Radiance run-length coding of one byte plane of a scanline: runs of >= 4 equal bytes, else literals <= 128."""
    out = bytearray()
    n, i = len(row), 0
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 4:
            out += bytes((128 + run, int(row[i])))
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            r = 1
            while j + r < n and r < 4 and row[j + r] == row[j]:
                r += 1
            if r >= 4:
                break
            j += 1
        out += bytes((j - i,)) + bytes(int(b) for b in row[i:j])
        i = j
    return bytes(out)


def hdr_file_bytes(rgbe, rle=True, magic=b"#?RADIANCE", extra_header=(b"# synthetic", b"EXPOSURE=1.0")):
    """A complete .hdr file for the uint8 [H,W,4] RGBE image: header + run-length coded (or flat) scanlines."""
    h, w = rgbe.shape[:2]
    head = magic + b"\n" + b"".join(l + b"\n" for l in extra_header) + b"FORMAT=32-bit_rle_rgbe\n\n" + b"-Y %d +X %d\n" % (h, w)
    if not rle or w < 8 or w >= 32768:
        return head + np.ascontiguousarray(rgbe).tobytes()
    body = bytearray()
    for y in range(h):
        body += bytes((2, 2, w >> 8, w & 255))
        for k in range(4):
            body += _rle_plane(rgbe[y, :, k])
    return head + bytes(body)


# ---- SSR environment fallback: the inputs of ClassifyReflectionTiles.hlsl ---------------------------------------
def _look_at_lh(eye, at, up):
    """DirectX::XMMatrixLookAtLH: row-major, row-vector convention (float64)."""
    eye, at, up = (np.asarray(v, np.float64) for v in (eye, at, up))
    z = at - eye
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2] = x, y, z
    m[3, :3] = [-x @ eye, -y @ eye, -z @ eye]
    return m


def _perspective_fov_lh(fov_y, aspect, zn, zf):
    """DirectX::XMMatrixPerspectiveFovLH (float64)."""
    h = 1.0 / np.tan(0.5 * fov_y)
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1], m[2, 2], m[2, 3], m[3, 2] = h / aspect, h, zf / (zf - zn), 1.0, -zn * zf / (zf - zn)
    return m


def _set_matrix(dst, m):
    for i in range(4):
        for j in range(4):
            dst.m[i][j] = float(np.float32(m[i, j]))


def ssr_constants(width, height, spec_mips, hdri_yaw=0.3, roughness_threshold=0.2, camera=(3.0, 10.0, -60.0), look_at=(0.0, 2.0, 0.0)):
    """FFX_SSSRConstants as VQRenderer::RenderReflections fills it (SceneRendering.cpp:2221-2242) for a LookAtLH / PerspectiveFovLH camera;
    envMapRotation == GetHDRIRotationMatrix (:2185-2195: float cos / sin of -yaw)."""
    cb = abi.SSSRConstants()
    view = _look_at_lh(camera, look_at, (0.0, 1.0, 0.0))
    proj = _perspective_fov_lh(np.pi / 3.0, width / height, 0.1, 1500.0)
    _set_matrix(cb.view, view)
    _set_matrix(cb.invView, np.linalg.inv(view))
    _set_matrix(cb.projection, proj)
    _set_matrix(cb.invProjection, np.linalg.inv(proj))
    _set_matrix(cb.invViewProjection, np.linalg.inv(view @ proj))
    _set_matrix(cb.prevViewProjection, view @ proj)
    c, s = float(np.cos(np.float32(-hdri_yaw), dtype=np.float32)), float(np.sin(np.float32(-hdri_yaw), dtype=np.float32))
    rot = np.zeros((4, 4))
    rot[0, :3], rot[1, :3], rot[2, :3] = [c, 0, s], [0, 1, 0], [-s, 0, c]
    _set_matrix(cb.envMapRotation, rot)
    cb.bufferDimensions[0], cb.bufferDimensions[1] = width, height
    cb.inverseBufferDimensions[0], cb.inverseBufferDimensions[1] = float(np.float32(1.0) / np.float32(width)), float(np.float32(1.0) / np.float32(height))
    cb.temporalStabilityFactor, cb.depthBufferThickness, cb.roughnessThreshold, cb.varianceThreshold = 0.7, 0.015, roughness_threshold, 0.0
    cb.maxTraversalIntersections, cb.minTraversalOccupancy, cb.mostDetailedMip, cb.samplesPerQuad = 128, 4, 0, 1
    cb.envMapSpecularIrradianceCubemapMipLevelCount = spec_mips
    return cb


def ssr_surfaces(width, height, seed=0x55E7, sky_fraction=0.1):
    """(scene colour float32 [H,W,4] with the roughness in alpha, NDC depth float32 [H,W], normals as R10G10B10A2_UNORM uint32 [H,W] and as the decoded
    float32 [H,W,4] values): white-noise surfaces — roughness U[0,1] (a fifth below the 0.2 ray threshold), unit normals on the whole sphere encoded
    n * 0.5 + 0.5 and quantised to 10 bits, depth U[0.9, 1) with `sky_fraction` of the pixels on the far plane (1.0)."""
    r = _chunk_rng(seed, 0)
    scene = r.random((height, width, 4), dtype=np.float32) * np.array([4.0, 4.0, 4.0, 1.0], np.float32)
    depth = (0.9 + 0.0999 * r.random((height, width), dtype=np.float32)).astype(np.float32)
    depth[r.random((height, width), dtype=np.float32) < sky_fraction] = 1.0
    n = r.normal(size=(height, width, 3))
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    q = np.clip(np.floor((n * 0.5 + 0.5) * 1023.0 + 0.5), 0, 1023).astype(np.uint32)
    packed = (q[..., 0] | (q[..., 1] << 10) | (q[..., 2] << 20) | (np.uint32(3) << 30)).astype(np.uint32)
    n01 = np.concatenate([q.astype(np.float32) / np.float32(1023.0), np.ones((height, width, 1), np.float32)], axis=-1)
    return scene, depth, packed, n01
