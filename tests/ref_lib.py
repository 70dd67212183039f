"""ctypes bindings of oracle/_ref/*.so — the reference's OWN sources compiled here (oracle/Makefile target `ref`):
  libvqref_fsr.so      ffx_a.h + ffx_fsr1.h with A_CPU (FsrEasuCon / FsrRcasCon, the CPU half packing)
  libvqref_shaders.so  ForwardLighting.hlsl, BRDF.hlsl, Lighting.hlsl, ShadingMath.hlsl, ... through oracle/ref_src/hlsl_shim.h
They exist only where /root/reference does (this container). available() gates the tests that need them; the fixtures they
produce (tests/golden/ref_*.npz, made by tests/golden/make_ref_fixtures.py) travel to the GPU box instead."""
import ctypes as C
import os
import shutil
import tempfile
import threading

import numpy as np

from vqengine_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
_libs = {}
_tls = threading.local()
_copy_dir = None


def use_private_copy(copy_id):
    """The translated shaders keep their cbuffers / SRVs in globals, so ONE loaded library serves one thread. A thread that calls this first gets its own
    copy of every library it loads afterwards (the .so copied under a temporary name: a second dlopen of the same file would share the globals) —
    how bench.py's cpu_reference_source runs the reference's HLSL on all host cores."""
    _tls.copy_id = copy_id



def available(name="shaders"):
    return os.path.exists(os.path.join(REF_DIR, f"libvqref_{name}.so"))


def load(name="shaders"):
    global _copy_dir
    copy_id = getattr(_tls, "copy_id", None)
    key, path = name, os.path.join(REF_DIR, f"libvqref_{name}.so")
    if copy_id is not None:
        key = (name, copy_id)
        if key not in _libs:
            if _copy_dir is None:
                _copy_dir = tempfile.mkdtemp(prefix="vqref_copies_")
            path = shutil.copy(path, os.path.join(_copy_dir, f"libvqref_{name}.{copy_id}.so"))
    name_key = key
    if name_key not in _libs:
        lib = C.CDLL(path)
        vp, f32, i32, u32 = C.c_void_p, C.c_float, C.c_int32, C.c_uint32
        if name == "mip":
            lib.vqref_mip_image.argtypes = [vp, vp, C.c_uint, C.c_uint, C.c_uint]
        elif name == "fsr":
            lib.vqref_fsr_easu_con.argtypes = [vp] + [f32] * 6
            lib.vqref_fsr_rcas_con.argtypes = [vp, f32]
            lib.vqref_half_bits.argtypes = [f32]
            lib.vqref_half_bits.restype = u32
        elif name in ("shaders_l256", "shaders_dxc_l256"):
            lib.vqref_forward_from_gbuffer.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
        elif name == "shaders_am":
            lib.vqref_forward_psmain.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
            lib.vqref_prepass_normals.argtypes = [vp, vp, i32, vp]
        elif name == "shaders_mrt":
            lib.vqref_forward_psmain_mrt.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
        else:
            lib.vqref_forward_psmain.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
            lib.vqref_forward_from_gbuffer.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
            lib.vqref_brdf.argtypes = [vp, f32, vp, f32, vp, vp, vp]
            lib.vqref_integrate_brdf.argtypes = [f32, f32, i32, vp]
            lib.vqref_importance_sample_ggx.argtypes = [f32, f32, vp, f32, vp]
            lib.vqref_hammersley.argtypes = [u32, u32, vp]
            lib.vqref_direction_to_equirect_uv.argtypes = [vp, vp]
            lib.vqref_unpack_normal.argtypes = [vp, vp, vp, vp]
            lib.vqref_conv_diffuse.argtypes = [vp, i32, i32, i32, i32, vp, i32, i32]
            lib.vqref_conv_specular.argtypes = [vp, i32, i32, i32, i32, f32, f32, f32, i32, vp]
            lib.vqref_conv_specular_texels.argtypes = [vp, i32, i32, i32, i32, f32, f32, f32, i32, vp, vp, vp, i32, vp]
            lib.vqref_conv_specular_taps.argtypes = [vp, i32, i32, i32, i32, f32, f32, f32, i32, i32, i32, i32, vp, i32, vp]
            lib.vqref_brdf_lut_texels.argtypes = [vp, vp, i32, vp]
            lib.vqref_blur_pass.argtypes = [vp, i32, i32, i32, vp]
            lib.vqref_tonemap.argtypes = [vp, i32, i32, vp, vp]
            lib.vqref_skydome.argtypes = [vp, i32, i32, vp, i32, i32, vp]
            lib.vqref_visualize.argtypes = [vp, i32, i32, vp, vp]
            lib.vqref_apply_reflections.argtypes = [vp, vp, i32, i32]
            lib.vqref_apply_reflections_bv.argtypes = [vp, vp, vp, i32, i32]
            lib.vqref_ssr_environment_fallback.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
            lib.vqref_prepass_normals.argtypes = [vp, vp, i32, vp]
            lib.vqref_unlit_color.argtypes = [vp, vp]
            lib.vqref_fsr_easu.argtypes = [vp, i32, i32, vp, vp, i32, i32]
            lib.vqref_fsr_rcas.argtypes = [vp, i32, i32, vp, vp]
        _libs[name_key] = lib
    return _libs[name_key]


def _ref(x):
    return C.byref(x) if x is not None else None


def fsr_easu_con(in_w, in_h, out_w, out_h):
    con = np.zeros(16, np.uint32)
    load("fsr").vqref_fsr_easu_con(con.ctypes.data, in_w, in_h, in_w, in_h, out_w, out_h)
    return con


def fsr_rcas_con(stops):
    con = np.zeros(4, np.uint32)
    load("fsr").vqref_fsr_rcas_con(con.ctypes.data, stops)
    return con


def forward_from_gbuffer(gb, per_frame, per_view, env=None, shadow=None, extra=None, reading="literal"):
    """PSMain per G-buffer pixel. `extra` (point lights beyond the cbuffer's 100, BASELINE cfg5) needs the build whose cap is raised
    to 256 (libvqref_shaders_l256.so, oracle/Makefile). reading: "literal" (the intrinsics as the HLSL is written: the reading the product
    follows) or "dxc" (libvqref_shaders_dxc*.so: dot as an FMA chain, normalize = v * rsqrt(dot), contract pow — hlsl_shim.h VQ_SHIM_DXC)."""
    gb = [np.ascontiguousarray(g, np.float32) for g in gb]
    h, w = gb[0].shape[:2]
    out = np.empty((h, w, 4), np.float32)
    g = abi.GBuffer(gb[0].ctypes.data, gb[1].ctypes.data, gb[2].ctypes.data, gb[3].ctypes.data, w, h, w)
    n_extra = len(extra) if extra is not None else 0
    base = "shaders" if reading == "literal" else "shaders_dxc"
    lib = load(base + "_l256") if n_extra else load(base)
    rc = lib.vqref_forward_from_gbuffer(C.byref(g), C.byref(per_frame), C.byref(per_view), extra if n_extra else None, n_extra,
                                        _ref(env), _ref(shadow), out.ctypes.data)
    assert rc == 0, rc
    return out


def forward_psmain(ip, materials, per_frame, per_view, ssao=None, env=None, shadow=None, alpha_masked=False):
    """PSMain per pixel of the interpolant planes. alpha_masked: the ENABLE_ALPHA_MASK permutation (libvqref_shaders_am.so) — every material
    is then drawn with the alpha-masked PSO; discarded pixels come back as -1 in all four channels."""
    ip = [np.ascontiguousarray(p, np.float32) for p in ip]
    h, w = ip[0].shape[:2]
    out = np.empty((h, w, 4), np.float32)
    inter = abi.Interpolants(ip[0].ctypes.data, ip[1].ctypes.data, ip[2].ctypes.data, w, h, w)
    s = None
    if ssao is not None:
        ssao = np.ascontiguousarray(ssao, np.uint8)
        s = abi.SSAO(ssao.ctypes.data, ssao.shape[1], ssao.shape[0])
    n = len(materials) if materials is not None else 0
    rc = load("shaders_am" if alpha_masked else "shaders").vqref_forward_psmain(C.byref(inter), materials if n else None, n, _ref(s), C.byref(per_frame), C.byref(per_view),
                                     _ref(env), _ref(shadow), out.ctypes.data)
    assert rc == 0
    return out


def prepass_normals(ip, materials, alpha_masked=False):
    """DepthPrePass.hlsl:PSMain per pixel of the interpolant planes: float32 [H,W,4] = float4(SurfaceN * 0.5 + 0.5, 1); 0 where nothing is drawn.
    alpha_masked: the ENABLE_ALPHA_MASK permutation (libvqref_shaders_am.so)."""
    ip = [np.ascontiguousarray(p, np.float32) for p in ip]
    h, w = ip[0].shape[:2]
    out = np.empty((h, w, 4), np.float32)
    inter = abi.Interpolants(ip[0].ctypes.data, ip[1].ctypes.data, ip[2].ctypes.data, w, h, w)
    n = len(materials) if materials is not None else 0
    assert load("shaders_am" if alpha_masked else "shaders").vqref_prepass_normals(C.byref(inter), materials if n else None, n, out.ctypes.data) == 0
    return out


def forward_psmain_mrt(ip, sv_curr, sv_prev, materials, per_frame, per_view, ssao=None, env=None, shadow=None):
    """PSMain in the OUTPUT_ALBEDO + OUTPUT_MOTION_VECTORS permutation (libvqref_shaders_mrt.so): (color [H,W,4], albedo_metallic [H,W,4], motion_vectors [H,W,2]),
    float32; pixels without geometry hold the targets' clear value 0."""
    ip = [np.ascontiguousarray(p, np.float32) for p in ip]
    sv_curr, sv_prev = np.ascontiguousarray(sv_curr, np.float32), np.ascontiguousarray(sv_prev, np.float32)
    h, w = ip[0].shape[:2]
    out, alb, mv = np.empty((h, w, 4), np.float32), np.empty((h, w, 4), np.float32), np.empty((h, w, 2), np.float32)
    inter = abi.Interpolants(ip[0].ctypes.data, ip[1].ctypes.data, ip[2].ctypes.data, w, h, w)
    s = None
    if ssao is not None:
        ssao = np.ascontiguousarray(ssao, np.uint8)
        s = abi.SSAO(ssao.ctypes.data, ssao.shape[1], ssao.shape[0])
    n = len(materials) if materials is not None else 0
    rc = load("shaders_mrt").vqref_forward_psmain_mrt(C.byref(inter), materials if n else None, n, _ref(s), C.byref(per_frame), C.byref(per_view), _ref(env), _ref(shadow),
                                                      out.ctypes.data, sv_curr.ctypes.data, sv_prev.ctypes.data, w, alb.ctypes.data, mv.ctypes.data)
    assert rc == 0, rc
    return out, alb, mv


def conv_diffuse(chain, w0, h0, n_mips, res, t0=0, t1=-1):
    """PSMain_DiffuseIrradiance (step 0.010, the shader's default) for texels [t0, t1): float32 [6,res,res,4]"""
    chain = np.ascontiguousarray(chain, np.float32)
    out = np.zeros((6, res, res, 4), np.float32)
    assert load().vqref_conv_diffuse(chain.ctypes.data, w0, h0, n_mips, res, out.ctypes.data, t0, t1) == 0
    return out


def conv_specular_mip(chain, w0, h0, n_mips, res, roughness, mip):
    chain = np.ascontiguousarray(chain, np.float32)
    out = np.zeros((6, res, res, 4), np.float32)
    assert load().vqref_conv_specular(chain.ctypes.data, w0, h0, n_mips, res, roughness, float(w0), float(h0), mip, out.ctypes.data) == 0
    return out


def conv_specular_texels(chain, w0, h0, n_mips, res, roughness, mip, faces, xs, ys):
    """PSMain_SpecularIrradiance of mip `mip` (res^2 faces) at the texels (faces[k], xs[k], ys[k]): float32 [n, 4]"""
    chain = np.ascontiguousarray(chain, np.float32)
    faces, xs, ys = (np.ascontiguousarray(a, np.int32) for a in (faces, xs, ys))
    out = np.zeros((len(xs), 4), np.float32)
    assert load().vqref_conv_specular_texels(chain.ctypes.data, w0, h0, n_mips, res, roughness, float(w0), float(h0), mip,
                                             faces.ctypes.data, xs.ctypes.data, ys.ctypes.data, len(xs), out.ctypes.data) == 0
    return out


def conv_specular_taps(chain, w0, h0, n_mips, res, roughness, mip, face, x, y):
    """(taps float32 [n, 3] = (uv.x, uv.y, lod) of every SampleLevel call of the texel as the reference's code formed them, rgba float32 [4])"""
    chain = np.ascontiguousarray(chain, np.float32)
    taps = np.zeros((512, 3), np.float32)
    out = np.zeros(4, np.float32)
    n = load().vqref_conv_specular_taps(chain.ctypes.data, w0, h0, n_mips, res, roughness, float(w0), float(h0), mip, face, x, y, taps.ctypes.data, 512, out.ctypes.data)
    assert 0 <= n <= 512, n
    return taps[:n], out


def brdf_lut_texels(xs, ys):
    """CSMain_BRDFIntegration (1024^2 image, 2048 samples) at the given texels: float32 [n,2]"""
    xs, ys = np.ascontiguousarray(xs, np.int32), np.ascontiguousarray(ys, np.int32)
    out = np.zeros((len(xs), 2), np.float32)
    assert load().vqref_brdf_lut_texels(xs.ctypes.data, ys.ctypes.data, len(xs), out.ctypes.data) == 0
    return out


def _img32(img):
    img = np.ascontiguousarray(img, np.float32)
    assert img.ndim == 3 and img.shape[2] == 4
    return img


def blur_pass(img, direction):
    """CSMain_X (0) / CSMain_Y (1) of GaussianBlur.hlsl on RGBA32F values; returns the float4 written to the UAV (before storage rounding)"""
    img = _img32(img)
    out = np.empty_like(img)
    assert load().vqref_blur_pass(img.ctypes.data, img.shape[1], img.shape[0], direction, out.ctypes.data) == 0
    return out


def tonemap(img, params):
    img = _img32(img)
    out = np.empty_like(img)
    assert load().vqref_tonemap(img.ctypes.data, img.shape[1], img.shape[0], C.byref(params), out.ctypes.data) == 0
    return out


def skydome(equirect0, params, width, height):
    eq = _img32(equirect0)
    out = np.empty((height, width, 4), np.float32)
    assert load().vqref_skydome(eq.ctypes.data, eq.shape[1], eq.shape[0], C.byref(params), width, height, out.ctypes.data) == 0
    return out


def visualize(img, params):
    img = _img32(img)
    out = np.empty_like(img)
    assert load().vqref_visualize(img.ctypes.data, img.shape[1], img.shape[0], C.byref(params), out.ctypes.data) == 0
    return out


def apply_reflections(refl, scene):
    refl, scene = _img32(refl), _img32(scene).copy()
    assert load().vqref_apply_reflections(refl.ctypes.data, scene.ctypes.data, scene.shape[1], scene.shape[0]) == 0
    return scene


def apply_reflections_bv(refl, bv, scene):
    """ApplyReflections.hlsl:CSMain compiled with COMPOSITE_BOUNDING_VOLUMES: returns the composited scene (float32 values)"""
    refl, bv, scene = _img32(refl), _img32(bv), _img32(scene).copy()
    assert load().vqref_apply_reflections_bv(refl.ctypes.data, bv.ctypes.data, scene.ctypes.data, scene.shape[1], scene.shape[0]) == 0
    return scene


def ssr_environment_fallback(scene, depth, normals01, cb, env):
    """scene: [H,W,4] values (alpha = roughness); depth: [H,W]; normals01: [H,W,4] the UNORM-decoded [0,1] values; env: abi.EnvMap over host arrays."""
    scene, normals01 = _img32(scene), _img32(normals01)
    depth = np.ascontiguousarray(depth, np.float32)
    out = np.zeros_like(scene)
    assert load().vqref_ssr_environment_fallback(scene.ctypes.data, depth.ctypes.data, normals01.ctypes.data, scene.shape[1], scene.shape[0],
                                                 C.addressof(cb), C.addressof(env), out.ctypes.data) == 0
    return out


def fsr_easu(img, out_w, out_h, con):
    """FSR_EASU_CSMain over ceil(out/16)^2 workgroups of 64 lanes; returns RGB float32 [out_h, out_w, 3]"""
    img = _img32(img)
    con = np.ascontiguousarray(con, np.uint32)
    out = np.zeros((out_h, out_w, 4), np.float32)
    assert load().vqref_fsr_easu(img.ctypes.data, img.shape[1], img.shape[0], con.ctypes.data, out.ctypes.data, out_w, out_h) == 0
    return out[..., :3]


def fsr_rcas(img, con):
    img = _img32(img)
    con = np.ascontiguousarray(con, np.uint32)
    out = np.zeros_like(img)
    assert load().vqref_fsr_rcas(img.ctypes.data, img.shape[1], img.shape[0], con.ctypes.data, out.ctypes.data) == 0
    return out[..., :3]


def mip_chain(level0):
    """VQ_DXGI_UTILS::MipImage applied level by level (as TextureManager::GenerateMips does) while both dimensions are >= 2.
    level0: float32 [H,W,4] (16-byte MIN filter) or uint8 [H,W,4] (4-byte box filter). Returns the list of levels 1.. as [h,w,4]."""
    lib = load("mip")
    cur = np.ascontiguousarray(level0)
    bpp = 16 if cur.dtype == np.float32 else 4
    levels = []
    while cur.shape[0] >= 2 and cur.shape[1] >= 2:
        h, w = cur.shape[0] // 2, cur.shape[1] // 2
        dst = np.zeros((h, w, 4), cur.dtype)
        lib.vqref_mip_image(cur.ctypes.data, dst.ctypes.data, cur.shape[1], cur.shape[0], bpp)
        levels.append(dst)
        cur = dst
    return levels
