"""Thin ctypes binding over the C ABI (include/vqhip.h, vqengine_amd/lib/libvqhip.so).

torch is used only as plumbing: device memory (tensor.data_ptr()) and the current HIP stream.
There is NO fallback: if the library is missing, or no gfx950 device is visible, this raises."""
import ctypes as C
import os

import torch

from . import abi
from .abi import (FMT_RGBA32F, FMT_RGBA16F, FMT_RGBA8_UNORM, FMT_RG16F, FMT_RG32F, CONV_SEQUENTIAL)

# $VQHIP_LIBRARY_PATH: another build of the SAME library (the sanitizer build of `make -C vqengine_amd/csrc asan`); never a different implementation
_LIB_PATH = os.environ.get("VQHIP_LIBRARY_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvqhip.so")
_lib = None


class VQHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vqhip error {code}: {msg}")
        self.code = code


def lib_path():
    return _LIB_PATH


def load_library():
    """dlopen libvqhip.so and declare every entry point of include/vqhip.h. Loud failure if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise FileNotFoundError(f"{_LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for this path.")
    lib = C.CDLL(_LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    sig = {
        "vqhip_abi_version": (i32, []),
        "vqhip_create": (i32, [i32, C.POINTER(vp)]),
        "vqhip_destroy": (None, [vp]),
        "vqhip_last_error": (C.c_char_p, [vp]),
        "vqhip_forward_lighting": (i32, [vp, vp, C.POINTER(abi.GBuffer), C.POINTER(abi.PerFrameData), C.POINTER(abi.PerViewLightingData),
                                         vp, i32, C.POINTER(abi.EnvMap), C.POINTER(abi.ShadowMaps), vp, i32, i32]),
        "vqhip_forward_lighting_mrt": (i32, [vp, vp, C.POINTER(abi.GBuffer), C.POINTER(abi.PerFrameData), C.POINTER(abi.PerViewLightingData),
                                             vp, i32, C.POINTER(abi.EnvMap), C.POINTER(abi.ShadowMaps), vp, i32, i32, C.POINTER(abi.PsmainTargets)]),
        "vqhip_gaussian_blur": (i32, [vp, vp, vp, vp, vp, C.POINTER(abi.BlurParams), i32]),
        "vqhip_gaussian_blur_x": (i32, [vp, vp, vp, vp, C.POINTER(abi.BlurParams), i32]),
        "vqhip_gaussian_blur_y": (i32, [vp, vp, vp, vp, vp, vp, i32, C.POINTER(abi.BlurParams), i32]),
        "vqhip_tonemap": (i32, [vp, vp, vp, vp, i32, i32, C.POINTER(abi.TonemapperParams), i32, i32]),
        "vqhip_gaussian_blur_y_tonemap": (i32, [vp, vp, vp, vp, vp, vp, i32, C.POINTER(abi.BlurParams), C.POINTER(abi.TonemapperParams), i32, i32]),
        "vqhip_post_process": (i32, [vp, vp, vp, vp, i32, i32, C.POINTER(abi.TonemapperParams), i32, i32, i32]),
        "vqhip_post_process_tile": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, C.POINTER(abi.TonemapperParams), i32, i32]),
        "vqhip_brdf_lut": (i32, [vp, vp, vp, i32, i32, i32]),
        "vqhip_mip_level_count": (i32, [i32, i32]),
        "vqhip_mip_chain_bytes": (sz, [i32, i32, i32]),
        "vqhip_mip_level_offset_bytes": (sz, [i32, i32, i32]),
        "vqhip_mip_chain_min_rgba32f": (i32, [vp, vp, vp, i32, i32, i32]),
        "vqhip_specular_mip_count": (i32, [i32]),
        "vqhip_cube_bytes": (sz, [i32, i32, i32]),
        "vqhip_conv_diffuse": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, i32]),
        "vqhip_conv_specular": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32]),
        "vqhip_envmap_prefilter": (i32, [vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, C.POINTER(abi.EnvMapOut)]),
        "vqhip_max_materials": (i32, []),
        "vqhip_gbuffer_from_materials": (i32, [vp, vp, C.POINTER(abi.Interpolants), C.POINTER(abi.MaterialDesc), i32, f32,
                                               C.POINTER(abi.SSAO), C.POINTER(abi.GBuffer)]),
        "vqhip_forward_lighting_from_materials": (i32, [vp, vp, C.POINTER(abi.Interpolants), C.POINTER(abi.MaterialDesc), i32, C.POINTER(abi.SSAO),
                                                        C.POINTER(abi.PerFrameData), C.POINTER(abi.PerViewLightingData), vp, i32,
                                                        C.POINTER(abi.EnvMap), C.POINTER(abi.ShadowMaps), vp, i32, i32]),
        "vqhip_forward_lighting_from_materials_mrt": (i32, [vp, vp, C.POINTER(abi.Interpolants), C.POINTER(abi.MaterialDesc), i32, C.POINTER(abi.SSAO),
                                                            C.POINTER(abi.PerFrameData), C.POINTER(abi.PerViewLightingData), vp, i32,
                                                            C.POINTER(abi.EnvMap), C.POINTER(abi.ShadowMaps), vp, i32, i32, C.POINTER(abi.PsmainTargets)]),
        "vqhip_scene_normals_from_materials": (i32, [vp, vp, C.POINTER(abi.Interpolants), C.POINTER(abi.MaterialDesc), i32, vp, i32, i32]),
        "vqhip_mip_chain_bytes_rgba8": (sz, [i32, i32, i32]),
        "vqhip_mip_chain_box_rgba8": (i32, [vp, vp, vp, i32, i32, i32]),
        "vqhip_set_fresnel_pow": (i32, [vp, i32]),
        "vqhip_set_arithmetic": (i32, [vp, i32]),
        "vqhip_set_option": (i32, [vp, C.c_char_p, C.c_char_p]),
        "vqhip_unlit_composite": (i32, [vp, vp, C.POINTER(abi.Interpolants), vp, i32, vp, i32, i32, i32, i32]),
        "vqhip_skydome": (i32, [vp, vp, vp, i32, i32, C.POINTER(abi.SkydomeParams), C.POINTER(abi.Interpolants), vp, i32, i32, i32, i32]),
        "vqhip_hdr_parse_header": (i32, [C.c_char_p, sz, C.POINTER(i32), C.POINTER(i32), C.POINTER(sz)]),
        "vqhip_hdr_decode_rgba32f": (i32, [vp, vp, C.c_char_p, sz, vp, i32, i32]),
        "vqhip_hdr_downsize_rgba32f": (i32, [vp, vp, vp, i32, i32, vp, i32, i32]),
        "vqhip_fsr_easu_con": (None, [C.POINTER(C.c_uint32), f32, f32, f32, f32, f32, f32]),
        "vqhip_fsr_rcas_con": (None, [C.POINTER(C.c_uint32), f32]),
        "vqhip_fsr_easu": (i32, [vp, vp, vp, i32, i32, i32, C.POINTER(C.c_uint32), vp, i32, i32, i32]),
        "vqhip_fsr_rcas": (i32, [vp, vp, vp, vp, i32, i32, C.POINTER(C.c_uint32), i32, i32]),
        "vqhip_visualize": (i32, [vp, vp, vp, vp, i32, i32, C.POINTER(abi.VizParams), i32, i32]),
        "vqhip_apply_reflections": (i32, [vp, vp, vp, vp, i32, i32, i32]),
        "vqhip_composite_reflections": (i32, [vp, vp, vp, vp, vp, i32, i32, i32]),
        "vqhip_ssr_environment_fallback": (i32, [vp, vp, vp, i32, i32, vp, i32, vp, i32, i32, i32, i32, C.POINTER(abi.SSSRConstants), C.POINTER(abi.EnvMap),
                                                 vp, i32, i32, vp]),
        "vqhip_rowtile": (i32, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]),
        "vqhip_comm_unique_id": (i32, [vp]),
        "vqhip_comm_create": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "vqhip_comm_adopt": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "vqhip_comm_destroy": (None, [vp]),
        "vqhip_comm_query": (i32, [vp, C.POINTER(abi.CommInfo)]),
        "vqhip_comm_abort": (i32, [vp]),
        "vqhip_comm_loopback": (i32, [vp, vp, vp, vp, sz]),
        "vqhip_exchange_blur_halos": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
        "vqhip_composite_tiles": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.vqhip_abi_version() != abi.ABI_VERSION:
        raise RuntimeError("libvqhip.so ABI version mismatch")
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "vqhip_abi_version", "vqhip_create", "vqhip_destroy", "vqhip_last_error", "vqhip_forward_lighting", "vqhip_forward_lighting_mrt", "vqhip_forward_lighting_from_materials_mrt", "vqhip_scene_normals_from_materials",
    "vqhip_gaussian_blur", "vqhip_gaussian_blur_x", "vqhip_gaussian_blur_y", "vqhip_gaussian_blur_y_tonemap", "vqhip_tonemap", "vqhip_post_process", "vqhip_post_process_tile", "vqhip_brdf_lut",
    "vqhip_mip_level_count", "vqhip_mip_chain_bytes", "vqhip_mip_level_offset_bytes", "vqhip_mip_chain_min_rgba32f",
    "vqhip_specular_mip_count", "vqhip_cube_bytes", "vqhip_conv_diffuse", "vqhip_conv_specular", "vqhip_envmap_prefilter",
    "vqhip_max_materials", "vqhip_gbuffer_from_materials", "vqhip_forward_lighting_from_materials", "vqhip_mip_chain_bytes_rgba8", "vqhip_mip_chain_box_rgba8",
    "vqhip_skydome", "vqhip_unlit_composite", "vqhip_set_fresnel_pow", "vqhip_set_arithmetic", "vqhip_set_option", "vqhip_hdr_parse_header", "vqhip_hdr_decode_rgba32f", "vqhip_hdr_downsize_rgba32f",
    "vqhip_fsr_easu_con", "vqhip_fsr_rcas_con", "vqhip_fsr_easu", "vqhip_fsr_rcas", "vqhip_visualize", "vqhip_apply_reflections", "vqhip_composite_reflections", "vqhip_ssr_environment_fallback",
    "vqhip_rowtile", "vqhip_comm_unique_id", "vqhip_comm_create", "vqhip_comm_adopt", "vqhip_comm_destroy", "vqhip_comm_query", "vqhip_comm_abort", "vqhip_comm_loopback", "vqhip_exchange_blur_halos",
    "vqhip_composite_tiles",
]


def fsr_easu_con(in_w, in_h, out_w, out_h, container_w=None, container_h=None):
    """FFSR1_EASU::UpdateEASUConstantBlock (PostProcess.cpp:47-79): the 16-dword EASU cbuffer. Host-only."""
    con = (C.c_uint32 * 16)()
    load_library().vqhip_fsr_easu_con(con, in_w, in_h, container_w or in_w, container_h or in_h, out_w, out_h)
    return con


def fsr_rcas_con(sharpness_stops=0.2):
    """FFSR1_RCAS::UpdateRCASConstantBlock (PostProcess.cpp:39-45; default RCASSharpnessStops 0.2, PostProcess.h:129)."""
    con = (C.c_uint32 * 4)()
    load_library().vqhip_fsr_rcas_con(con, sharpness_stops)
    return con


def hdr_parse_header(data):
    """(width, height, data_offset) of a Radiance .hdr file held in `data` (bytes). Host-only; raises VQHipError when malformed."""
    lib = load_library()
    w, h, off = C.c_int(), C.c_int(), C.c_size_t()
    rc = lib.vqhip_hdr_parse_header(data, len(data), C.byref(w), C.byref(h), C.byref(off))
    if rc != 0:
        raise VQHipError(rc, (lib.vqhip_last_error(None) or b"").decode())
    return w.value, h.value, off.value

HALO_ROWS = 10            # VQHIP_HALO_ROWS
ALL_RANKS = -1            # VQHIP_ALL_RANKS
COMM_ID_BYTES = 128       # VQHIP_COMM_ID_BYTES


def _global_error(rc):
    raise VQHipError(rc, (load_library().vqhip_last_error(None) or b"").decode())


def rowtile(frame_height, world, rank):
    """(row0, rows) of rank `rank` (vqhip_rowtile)."""
    r0, n = C.c_int(), C.c_int()
    rc = load_library().vqhip_rowtile(frame_height, world, rank, C.byref(r0), C.byref(n))
    if rc != 0:
        _global_error(rc)
    return r0.value, n.value


def comm_unique_id():
    """bytes(128): ncclGetUniqueId through the C ABI; made on rank 0 and handed to the other ranks out of band."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = load_library().vqhip_comm_unique_id(buf)
    if rc != 0:
        _global_error(rc)
    return buf.raw


def _addr(x):
    """device pointer of a torch tensor, or host pointer of a numpy array (the mock-RCCL tests); None -> NULL"""
    if x is None:
        return C.c_void_p(None)
    return C.c_void_p(x.data_ptr()) if hasattr(x, "data_ptr") else C.c_void_p(x.ctypes.data)


class Comm:
    """vqhip_comm: the RCCL communicator of the row-tiled multi-GPU mode (include/vqhip.h). Collective constructor."""

    def __init__(self, unique_id, world, rank):
        self.lib = load_library()
        self.world, self.rank = world, rank
        h = C.c_void_p()
        rc = self.lib.vqhip_comm_create(unique_id, world, rank, C.byref(h))
        if rc != 0:
            _global_error(rc)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vqhip_comm_destroy(self._h)
            self._h = None

    def abort(self):
        """ncclCommAbort + free (vqhip_comm_abort): the way out of an exchange that does not complete. The object is dead afterwards."""
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            rc = self.lib.vqhip_comm_abort(h)
            if rc != 0:
                _global_error(rc)

    def query(self):
        """What the communicator reports about itself (vqhip_comm_query): RCCL's own rank count / rank, its version, the library's path."""
        info = abi.CommInfo()
        rc = self.lib.vqhip_comm_query(self._h, C.byref(info))
        if rc != 0:
            _global_error(rc)
        return {"world": info.world, "rank": info.rank, "nranks_seen": info.nranks_seen, "rank_seen": info.rank_seen,
                "version": info.rccl_version, "library_path": info.library_path.decode(errors="replace")}

    def loopback(self, src, dst, stream=None):
        """One grouped ncclSend + ncclRecv from this rank to itself (vqhip_comm_loopback): src -> dst, same byte size, enqueued on `stream`."""
        n = src.numel() * src.element_size() if hasattr(src, "numel") else src.nbytes
        rc = self.lib.vqhip_comm_loopback(self._h, stream, _addr(src), _addr(dst), n)
        if rc != 0:
            _global_error(rc)

    def exchange_blur_halos(self, x_tile, fmt, halo_top, halo_bottom, stream=None):
        """x_tile [rows, W, 4]; halo_top / halo_bottom: preallocated [10, W, 4] buffers (None at the frame's edges)."""
        rows, w = x_tile.shape[0], x_tile.shape[1]
        rc = self.lib.vqhip_exchange_blur_halos(self._h, stream, _addr(x_tile), w, rows, w, fmt, _addr(halo_top), _addr(halo_bottom))
        if rc != 0:
            _global_error(rc)

    def composite_tiles(self, tile, fmt, frame_height, root, frame, stream=None):
        """tile [rows, W, C] -> frame [frame_height, W, C] on `root` (or every rank: ALL_RANKS)."""
        rc = self.lib.vqhip_composite_tiles(self._h, stream, _addr(tile), tile.shape[1], frame_height, fmt, root, _addr(frame))
        if rc != 0:
            _global_error(rc)


_TORCH_DTYPE = {FMT_RGBA32F: (torch.float32, 4), FMT_RGBA16F: (torch.float16, 4), FMT_RGBA8_UNORM: (torch.uint8, 4),
                FMT_RG16F: (torch.float16, 2), FMT_RG32F: (torch.float32, 2)}


def empty_image(h, w, fmt, device):
    dt, ch = _TORCH_DTYPE[fmt]
    return torch.empty((h, w, ch), dtype=dt, device=device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _check_img(t, fmt, name, hw=None):
    """contiguous cuda image of `fmt`; hw = (rows, width) it must have (the C ABI takes raw pointers: a smaller `out` would be an
    out-of-bounds device write)."""
    dt, ch = _TORCH_DTYPE[fmt]
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dt and t.shape[-1] == ch):
        raise ValueError(f"{name}: expected contiguous cuda tensor [...,{ch}] of {dt}, got {tuple(t.shape)} {t.dtype} {t.device}")
    if hw is not None and tuple(t.shape[:2]) != tuple(hw):
        raise ValueError(f"{name}: expected {hw[0]} x {hw[1]} pixels, got {tuple(t.shape[:2])}")


def _halo_rows(halo_top, halo_bottom, fmt, width):
    """Both halos of a tile have the same row count (>= 10) and the tile's width and format."""
    rows = 0
    for hh in (halo_top, halo_bottom):
        if hh is not None:
            _check_img(hh, fmt, "halo", (hh.shape[0], width))
            if rows and hh.shape[0] != rows:
                raise ValueError(f"halo_top and halo_bottom must have the same number of rows, got {rows} and {hh.shape[0]}")
            rows = hh.shape[0]
    return rows


class Context:
    """One vqhip_ctx bound to one GPU. Calls enqueue on torch's current stream for that device."""

    def __init__(self, device_ordinal=0):
        self.lib = load_library()
        self.device = torch.device("cuda", device_ordinal)
        h = C.c_void_p()
        rc = self.lib.vqhip_create(device_ordinal, C.byref(h))
        if rc != 0:
            raise VQHipError(rc, (self.lib.vqhip_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vqhip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return C.c_void_p(s.cuda_stream)

    def _ck(self, rc):
        if rc != 0:
            raise VQHipError(rc, (self.lib.vqhip_last_error(self._h) or b"").decode())

    # ---- forward lighting (RenderSceneColor, SceneRendering.cpp:1619) --------------------------------------
    def _psmain_targets(self, h, w, albedo_fmt, motion_fmt, sv_curr, sv_prev, stream=None):
        """vqhip_psmain_targets + the tensors it points at: (struct, albedo_metallic | None, motion_vectors | None). The clears run on `stream` — the stream the
        kernel is launched on: the kernel skips uncovered pixels, so a clear that is not ordered in front of it on the same stream could wipe written pixels."""
        with torch.cuda.stream(stream if stream is not None else torch.cuda.current_stream(self.device)):
            return self._psmain_targets_on_current_stream(h, w, albedo_fmt, motion_fmt, sv_curr, sv_prev)

    def _psmain_targets_on_current_stream(self, h, w, albedo_fmt, motion_fmt, sv_curr, sv_prev):
        t = abi.PsmainTargets()
        albedo = motion = None
        if albedo_fmt is not None:
            albedo = empty_image(h, w, albedo_fmt, self.device).zero_()          # the clear value: uncovered pixels of the one-kernel PSMain keep it
            t.albedo_metallic, t.albedo_fmt, t.albedo_pitch_px = albedo.data_ptr(), albedo_fmt, w
        if motion_fmt is not None:
            if sv_curr is None or sv_prev is None:
                raise ValueError("motion vectors need sv_curr and sv_prev (float32 cuda [H,W,4])")
            _check_img(sv_curr, FMT_RGBA32F, "sv_curr", (h, w)); _check_img(sv_prev, FMT_RGBA32F, "sv_prev", (h, w))
            motion = empty_image(h, w, motion_fmt, self.device).zero_()
            t.motion_vectors, t.motion_fmt, t.motion_pitch_px = motion.data_ptr(), motion_fmt, w
            t.svPositionCurr, t.svPositionPrev, t.sv_pitch_px = sv_curr.data_ptr(), sv_prev.data_ptr(), w
        return t, albedo, motion

    def forward_lighting_mrt(self, gb, per_frame, per_view, albedo_fmt=FMT_RGBA16F, motion_fmt=None, sv_curr=None, sv_prev=None, out=None,
                             out_fmt=FMT_RGBA16F, extra_point=None, env=None, shadow=None, stream=None):
        """forward_lighting + the draw's other render targets (ForwardLighting.hlsl:382-389) from the same kernel: returns (out, albedo_metallic, motion_vectors);
        albedo_fmt / motion_fmt None = that target not bound (-> None)."""
        h, w = gb[0].shape[0], gb[0].shape[1]
        t, albedo, motion = self._psmain_targets(h, w, albedo_fmt, motion_fmt, sv_curr, sv_prev, stream=stream)
        out = self.forward_lighting(gb, per_frame, per_view, out=out, out_fmt=out_fmt, extra_point=extra_point, env=env, shadow=shadow, stream=stream, _targets=t)
        return out, albedo, motion

    def forward_lighting(self, gb, per_frame, per_view, out=None, out_fmt=FMT_RGBA16F, extra_point=None, env=None, shadow=None, stream=None, _targets=None):
        """gb: tuple of 4 float32 cuda tensors [H,W,4] (gb0..gb3). env: abi.EnvMap or None. Returns out tensor."""
        g0, g1, g2, g3 = gb
        h, w = g0.shape[0], g0.shape[1]
        for i, g in enumerate(gb):
            _check_img(g, FMT_RGBA32F, f"gb{i}", (h, w))             # all four planes share one shape
        if out is None:
            out = empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        gbuf = abi.GBuffer(g0.data_ptr(), g1.data_ptr(), g2.data_ptr(), g3.data_ptr(), w, h, w)
        n_extra = 0
        extra_ptr = C.c_void_p(None)
        if extra_point is not None and len(extra_point):
            n_extra = len(extra_point)
            extra_ptr = C.cast(extra_point, C.c_void_p)
        args = (self._h, self._stream(stream), C.byref(gbuf), C.byref(per_frame), C.byref(per_view), extra_ptr, n_extra, C.byref(env) if env is not None else None,
                C.byref(shadow) if shadow is not None else None, _ptr(out), out.shape[1], out_fmt)
        rc = self.lib.vqhip_forward_lighting(*args) if _targets is None else self.lib.vqhip_forward_lighting_mrt(*args, C.byref(_targets))
        self._ck(rc)
        return out

    # ---- post-process (RenderPostProcess, SceneRendering.cpp:2507) ------------------------------------------
    def gaussian_blur(self, src, fmt, tmp=None, out=None, stream=None):
        _check_img(src, fmt, "src")
        h, w = src.shape[0], src.shape[1]
        tmp = tmp if tmp is not None else torch.empty_like(src)
        out = out if out is not None else torch.empty_like(src)
        _check_img(tmp, fmt, "tmp", (h, w)); _check_img(out, fmt, "out", (h, w))
        p = abi.BlurParams(w, h)
        self._ck(self.lib.vqhip_gaussian_blur(self._h, self._stream(stream), _ptr(src), _ptr(tmp), _ptr(out), C.byref(p), fmt))
        return out

    def gaussian_blur_x(self, src, fmt, out=None, stream=None):
        _check_img(src, fmt, "src")
        out = out if out is not None else torch.empty_like(src)
        _check_img(out, fmt, "out", src.shape[:2])
        p = abi.BlurParams(src.shape[1], src.shape[0])
        self._ck(self.lib.vqhip_gaussian_blur_x(self._h, self._stream(stream), _ptr(src), _ptr(out), C.byref(p), fmt))
        return out

    def gaussian_blur_y(self, src, fmt, out=None, halo_top=None, halo_bottom=None, stream=None):
        _check_img(src, fmt, "src")
        out = out if out is not None else torch.empty_like(src)
        _check_img(out, fmt, "out", src.shape[:2])
        rows = _halo_rows(halo_top, halo_bottom, fmt, src.shape[1])
        p = abi.BlurParams(src.shape[1], src.shape[0])
        self._ck(self.lib.vqhip_gaussian_blur_y(self._h, self._stream(stream), _ptr(src), _ptr(out), _ptr(halo_top), _ptr(halo_bottom), rows, C.byref(p), fmt))
        return out

    def gaussian_blur_y_tonemap(self, src, fmt, out_fmt=FMT_RGBA8_UNORM, params=None, out=None, halo_top=None, halo_bottom=None, stream=None):
        """Fused CSMain_Y + Tonemapper (same bits as gaussian_blur_y followed by tonemap)."""
        _check_img(src, fmt, "src")
        h, w = src.shape[0], src.shape[1]
        out = out if out is not None else empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        rows = _halo_rows(halo_top, halo_bottom, fmt, w)
        params = params if params is not None else abi.TonemapperParams.default()
        p = abi.BlurParams(w, h)
        self._ck(self.lib.vqhip_gaussian_blur_y_tonemap(self._h, self._stream(stream), _ptr(src), _ptr(out), _ptr(halo_top), _ptr(halo_bottom), rows,
                                                         C.byref(p), C.byref(params), fmt, out_fmt))
        return out

    def post_process_tile(self, src, in_fmt, out_fmt=FMT_RGBA8_UNORM, params=None, out=None, halo_top=None, halo_bottom=None, stream=None):
        """The blur + tonemapper chain over one row tile; halo_top / halo_bottom are the neighbouring tiles' SCENE-COLOUR rows (>= 10, in in_fmt)."""
        _check_img(src, in_fmt, "src")
        h, w = src.shape[0], src.shape[1]
        out = out if out is not None else empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        rows = _halo_rows(halo_top, halo_bottom, in_fmt, w)
        params = params if params is not None else abi.TonemapperParams.default()
        self._ck(self.lib.vqhip_post_process_tile(self._h, self._stream(stream), _ptr(src), _ptr(out), _ptr(halo_top), _ptr(halo_bottom), rows, w, h,
                                                  C.byref(params), in_fmt, out_fmt))
        return out

    def post_process(self, src, in_fmt, out_fmt=FMT_RGBA8_UNORM, params=None, blur=True, out=None, stream=None):
        """RenderPostProcess's blur + tonemapper as one call (one kernel for RGBA16F -> RGBA8 and a per-channel curve); same bits as
        gaussian_blur_x -> gaussian_blur_y -> tonemap."""
        _check_img(src, in_fmt, "src")
        h, w = src.shape[0], src.shape[1]
        out = out if out is not None else empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        params = params if params is not None else abi.TonemapperParams.default()
        self._ck(self.lib.vqhip_post_process(self._h, self._stream(stream), _ptr(src), _ptr(out), w, h, C.byref(params), int(bool(blur)), in_fmt, out_fmt))
        return out

    def tonemap(self, src, in_fmt, out_fmt=FMT_RGBA8_UNORM, params=None, out=None, stream=None):
        _check_img(src, in_fmt, "src")
        h, w = src.shape[0], src.shape[1]
        out = out if out is not None else empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        params = params if params is not None else abi.TonemapperParams.default()
        self._ck(self.lib.vqhip_tonemap(self._h, self._stream(stream), _ptr(src), _ptr(out), w, h, C.byref(params), in_fmt, out_fmt))
        return out

    # ---- load-time IBL (ComputeBRDFIntegrationLUT Renderer.cpp:871, PreFilterEnvironmentMap EnvironmentMapRendering.cpp:139)
    def brdf_lut(self, size=1024, samples=2048, fmt=FMT_RG16F, stream=None):
        out = empty_image(size, size, fmt, self.device)
        self._ck(self.lib.vqhip_brdf_lut(self._h, self._stream(stream), _ptr(out), size, samples, fmt))
        return out

    def mip_chain(self, level0, stream=None):
        """level0: float32 cuda [H,W,4]. Returns (flat float32 chain [px,4], n_mips) incl. level 0 (min-filter mips)."""
        _check_img(level0, FMT_RGBA32F, "level0")
        h, w = level0.shape[0], level0.shape[1]
        n = abi.mip_level_count(w, h)
        chain = torch.empty((abi.mip_chain_px(w, h, n), 4), dtype=torch.float32, device=self.device)
        chain[: w * h].copy_(level0.reshape(-1, 4))
        self._ck(self.lib.vqhip_mip_chain_min_rgba32f(self._h, self._stream(stream), _ptr(chain), w, h, n))
        return chain, n

    # ---- G-buffer producer (ForwardLighting.hlsl:PSMain :226-287) --------------------------
    def mip_chain_rgba8(self, level0, stream=None):
        """level0: uint8 cuda [H,W,4], power-of-two dims. Returns (flat uint8 chain [px,4], n_mips), box-filtered mips."""
        _check_img(level0, FMT_RGBA8_UNORM, "level0")
        h, w = level0.shape[0], level0.shape[1]
        n = abi.mip_level_count(w, h)
        chain = torch.empty((abi.mip_chain_px(w, h, n), 4), dtype=torch.uint8, device=self.device)
        chain[: w * h].copy_(level0.reshape(-1, 4))
        self._ck(self.lib.vqhip_mip_chain_box_rgba8(self._h, self._stream(stream), _ptr(chain), w, h, n))
        return chain, n

    def gbuffer_from_materials(self, ip, materials, ambient, ssao=None, out=None, stream=None):
        """ip: 3 float32 cuda tensors [H,W,4] (vqhip_interpolants planes); materials: ctypes array of abi.MaterialDesc whose
        texture pointers are device pointers; ssao: uint8 cuda [H,W] or None. Returns the 4 G-buffer planes."""
        for i, t in enumerate(ip):
            _check_img(t, FMT_RGBA32F, f"ip{i}")
        h, w = ip[0].shape[0], ip[0].shape[1]
        if out is None:
            out = tuple(torch.empty((h, w, 4), dtype=torch.float32, device=self.device) for _ in range(4))
        for i, t in enumerate(out):
            _check_img(t, FMT_RGBA32F, f"gb{i}")
        inter = abi.Interpolants(ip[0].data_ptr(), ip[1].data_ptr(), ip[2].data_ptr(), w, h, w)
        gbuf = abi.GBuffer(out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), w, h, w)
        s = None
        if ssao is not None:
            if not (ssao.is_cuda and ssao.is_contiguous() and ssao.dtype == torch.uint8 and ssao.dim() == 2):
                raise ValueError("ssao: expected contiguous cuda uint8 [H,W]")
            s = abi.SSAO(ssao.data_ptr(), ssao.shape[1], ssao.shape[0])
        n = len(materials) if materials is not None else 0
        self._ck(self.lib.vqhip_gbuffer_from_materials(self._h, self._stream(stream), C.byref(inter), materials if n else None, n,
                                                       float(ambient), C.byref(s) if s is not None else None, C.byref(gbuf)))
        return out

    def scene_normals_from_materials(self, ip, materials, out_fmt=abi.FMT_R10G10B10A2_UNORM, out=None, stream=None):
        """The Z pre-pass's colour target (DepthPrePass.hlsl:PSMain): Tex_SceneNormals as int32 [H,W] (R10G10B10A2_UNORM bits; torch has no uint32) or
        float32 [H,W,4]. ip / materials as for gbuffer_from_materials (ip2 is not modified)."""
        for i, t in enumerate(ip):
            _check_img(t, FMT_RGBA32F, f"ip{i}")
        h, w = ip[0].shape[0], ip[0].shape[1]
        if out is None:
            out = (torch.empty((h, w), dtype=torch.int32, device=self.device) if out_fmt == abi.FMT_R10G10B10A2_UNORM
                   else empty_image(h, w, FMT_RGBA32F, self.device))
        inter = abi.Interpolants(ip[0].data_ptr(), ip[1].data_ptr(), ip[2].data_ptr(), w, h, w)
        n = len(materials) if materials is not None else 0
        self._ck(self.lib.vqhip_scene_normals_from_materials(self._h, self._stream(stream), C.byref(inter), materials if n else None, n, _ptr(out), out_fmt, w))
        return out

    def forward_lighting_from_materials_mrt(self, ip, materials, per_frame, per_view, albedo_fmt=FMT_RGBA16F, motion_fmt=None, sv_curr=None, sv_prev=None,
                                            **kw):
        """forward_lighting_from_materials + the draw's other render targets: returns (out, albedo_metallic, motion_vectors)"""
        h, w = ip[0].shape[0], ip[0].shape[1]
        t, albedo, motion = self._psmain_targets(h, w, albedo_fmt, motion_fmt, sv_curr, sv_prev, stream=kw.get("stream"))
        return self.forward_lighting_from_materials(ip, materials, per_frame, per_view, _targets=t, **kw), albedo, motion

    def forward_lighting_from_materials(self, ip, materials, per_frame, per_view, ssao=None, out=None, out_fmt=FMT_RGBA16F, extra_point=None,
                                        env=None, shadow=None, stream=None, _targets=None):
        """PSMain in one kernel: gbuffer_from_materials(ip, materials, per_frame.fAmbientLightingFactor, ssao) + forward_lighting, the G-buffer
        record never leaving registers. Same bits as the two calls."""
        for i, t in enumerate(ip):
            _check_img(t, FMT_RGBA32F, f"ip{i}")
        h, w = ip[0].shape[0], ip[0].shape[1]
        if out is None:
            out = empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        inter = abi.Interpolants(ip[0].data_ptr(), ip[1].data_ptr(), ip[2].data_ptr(), w, h, w)
        s = None
        if ssao is not None:
            if not (ssao.is_cuda and ssao.is_contiguous() and ssao.dtype == torch.uint8 and ssao.dim() == 2):
                raise ValueError("ssao: expected contiguous cuda uint8 [H,W]")
            s = abi.SSAO(ssao.data_ptr(), ssao.shape[1], ssao.shape[0])
        n = len(materials) if materials is not None else 0
        n_extra, extra_ptr = 0, C.c_void_p(None)
        if extra_point is not None and len(extra_point):
            n_extra, extra_ptr = len(extra_point), C.cast(extra_point, C.c_void_p)
        args = (self._h, self._stream(stream), C.byref(inter), materials if n else None, n, C.byref(s) if s is not None else None,
                C.byref(per_frame), C.byref(per_view), extra_ptr, n_extra, C.byref(env) if env is not None else None,
                C.byref(shadow) if shadow is not None else None, _ptr(out), out.shape[1], out_fmt)
        self._ck(self.lib.vqhip_forward_lighting_from_materials(*args) if _targets is None
                 else self.lib.vqhip_forward_lighting_from_materials_mrt(*args, C.byref(_targets)))
        return out

    # ---- FSR 1.0 (SceneRendering.cpp:2695-2784) -----------------------------------------------
    def fsr_easu(self, src, in_fmt, out_w, out_h, out_fmt=None, con=None, out=None, stream=None):
        _check_img(src, in_fmt, "src")
        out_fmt = in_fmt if out_fmt is None else out_fmt
        h, w = src.shape[0], src.shape[1]
        con = fsr_easu_con(w, h, out_w, out_h) if con is None else con
        if out is None:
            out = empty_image(out_h, out_w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (out_h, out_w))
        self._ck(self.lib.vqhip_fsr_easu(self._h, self._stream(stream), _ptr(src), w, h, in_fmt, con, _ptr(out), out_w, out_h, out_fmt))
        return out

    def fsr_rcas(self, src, in_fmt, out_fmt=None, con=None, out=None, stream=None):
        _check_img(src, in_fmt, "src")
        out_fmt = in_fmt if out_fmt is None else out_fmt
        h, w = src.shape[0], src.shape[1]
        con = fsr_rcas_con() if con is None else con
        if out is None:
            out = empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        self._ck(self.lib.vqhip_fsr_rcas(self._h, self._stream(stream), _ptr(src), _ptr(out), w, h, con, in_fmt, out_fmt))
        return out

    def apply_reflections(self, reflection, scene_color, fmt, stream=None):
        """ApplyReflections.hlsl:CSMain: scene_color.rgb += reflection.rgb in place (alpha kept)."""
        _check_img(reflection, fmt, "reflection")
        _check_img(scene_color, fmt, "scene_color")
        h, w = scene_color.shape[0], scene_color.shape[1]
        self._ck(self.lib.vqhip_apply_reflections(self._h, self._stream(stream), _ptr(reflection), _ptr(scene_color), w, h, fmt))
        return scene_color

    def composite_reflections(self, reflection, scene_color, fmt, bounding_volumes=None, stream=None):
        """VQRenderer::CompositeReflections: apply_reflections, or — with the light-bounds image — the COMPOSITE_BOUNDING_VOLUMES permutation
        (rgb = BV.rgb * BV.a + (scene + reflection) * (1 - BV.a), alpha = BV.a). In place on scene_color."""
        _check_img(reflection, fmt, "reflection", scene_color.shape[:2])
        _check_img(scene_color, fmt, "scene_color")
        if bounding_volumes is not None:
            _check_img(bounding_volumes, fmt, "bounding_volumes", scene_color.shape[:2])
        h, w = scene_color.shape[0], scene_color.shape[1]
        self._ck(self.lib.vqhip_composite_reflections(self._h, self._stream(stream), _ptr(reflection), _ptr(bounding_volumes), _ptr(scene_color), w, h, fmt))
        return scene_color

    def ssr_environment_fallback(self, scene_color, scene_fmt, depth, normals, normal_fmt, cb, env, out_fmt=None, extract_roughness=False, out=None, stream=None):
        """ClassifyReflectionTiles.hlsl: the environment-map fallback of SSR's tile classification (SampleEnvironmentMap under the condition of
        ClassifyTiles :146-152). scene_color: [H,W,4] image whose alpha is the roughness; depth: float32 [H,W]; normals: uint32 [H,W]
        (R10G10B10A2_UNORM) or float32 [H,W,4]; cb: abi.SSSRConstants; env: abi.EnvMap. Returns the radiance image (and the R8 roughness)."""
        _check_img(scene_color, scene_fmt, "scene_color")
        h, w = scene_color.shape[0], scene_color.shape[1]
        assert depth.dtype == torch.float32 and tuple(depth.shape) == (h, w) and depth.is_contiguous()
        if normal_fmt == abi.FMT_R10G10B10A2_UNORM:
            assert normals.dtype == torch.int32 and tuple(normals.shape) == (h, w) and normals.is_contiguous()
        else:
            _check_img(normals, normal_fmt, "normals", (h, w))
        out_fmt = FMT_RGBA16F if out_fmt is None else out_fmt
        if out is None:
            out = empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out", (h, w))
        rough = torch.empty((h, w), dtype=torch.uint8, device=self.device) if extract_roughness else None
        self._ck(self.lib.vqhip_ssr_environment_fallback(self._h, self._stream(stream), _ptr(scene_color), scene_fmt, 0, _ptr(depth), 0, _ptr(normals), normal_fmt, 0,
                                                         w, h, C.byref(cb), C.byref(env), _ptr(out), out_fmt, 0, _ptr(rough) if rough is not None else None))
        return (out, rough) if extract_roughness else out

    def visualize(self, src, in_fmt, params, out_fmt=None, out=None, stream=None):
        """Visualization.hlsl:CSMain (debug draw modes). params: abi.VizParams. src in the format of the target the mode shows: a colour image, the int32 [H,W]
        words of scene_normals_from_materials (in_fmt R10G10B10A2_UNORM) or the RG16F / RG32F motion vectors of forward_lighting_mrt; out_fmt defaults to in_fmt
        for colour inputs, RGBA16F otherwise."""
        if in_fmt == abi.FMT_R10G10B10A2_UNORM:
            if not (src.is_cuda and src.is_contiguous() and src.dtype == torch.int32 and src.dim() == 2):
                raise ValueError("src: expected contiguous cuda int32 [H,W] (R10G10B10A2_UNORM words)")
        else:
            _check_img(src, in_fmt, "src")
        if out_fmt is None:
            out_fmt = in_fmt if in_fmt in (FMT_RGBA32F, FMT_RGBA16F, FMT_RGBA8_UNORM) else FMT_RGBA16F
        h, w = src.shape[0], src.shape[1]
        if out is None:
            out = empty_image(h, w, out_fmt, self.device)
        _check_img(out, out_fmt, "out")
        self._ck(self.lib.vqhip_visualize(self._h, self._stream(stream), _ptr(src), _ptr(out), w, h, C.byref(params), in_fmt, out_fmt))
        return out

    # ---- HDRI ingest (Image::LoadFromFile -> stbi_loadf, TextureManager.cpp:566) ----------------
    def load_hdr(self, data, stream=None):
        """data: bytes of a Radiance .hdr file. Returns the decoded float32 cuda image [H,W,4] (alpha 1) == level 0 of the chain."""
        w, h, _ = hdr_parse_header(data)
        out = torch.empty((h, w, 4), dtype=torch.float32, device=self.device)
        self._ck(self.lib.vqhip_hdr_decode_rgba32f(self._h, self._stream(stream), data, len(data), _ptr(out), w, h))
        return out

    def hdr_downsize(self, img, out_w, out_h, stream=None):
        """Image::CreateResizedImage as the HDRI fallback uses it (EnvironmentMap.cpp:167): integer ratios only (k x k mean), else VQHipError(-3)."""
        _check_img(img, FMT_RGBA32F, "img")
        out = torch.empty((out_h, out_w, 4), dtype=torch.float32, device=self.device)
        self._ck(self.lib.vqhip_hdr_downsize_rgba32f(self._h, self._stream(stream), _ptr(img), img.shape[1], img.shape[0], _ptr(out), out_w, out_h))
        return out

    # ---- skydome (Skydome.hlsl:39-56, SceneRendering.cpp:1822-1850) -------------------------
    def skydome(self, equirect_level0, params, color, fmt, coverage_ip=None, stream=None):
        """equirect_level0: float32 cuda [h0,w0,4]; params: abi.SkydomeParams; color: scene-colour tensor [H,W,4] written in place
        where coverage_ip (the 3 interpolant planes, or None = everywhere) has material index < 0."""
        _check_img(equirect_level0, FMT_RGBA32F, "equirect_level0")
        _check_img(color, fmt, "color")
        h, w = color.shape[0], color.shape[1]
        cov = None
        if coverage_ip is not None:
            for i, t in enumerate(coverage_ip):
                _check_img(t, FMT_RGBA32F, f"ip{i}")
            cov = abi.Interpolants(coverage_ip[0].data_ptr(), coverage_ip[1].data_ptr(), coverage_ip[2].data_ptr(), w, h, w)
        self._ck(self.lib.vqhip_skydome(self._h, self._stream(stream), _ptr(equirect_level0), equirect_level0.shape[1], equirect_level0.shape[0],
                                        C.byref(params), C.byref(cov) if cov is not None else None, _ptr(color), w, h, w, fmt))
        return color

    def set_fresnel_pow(self, exp2_log2):
        """False (default): pow(1 - cos, 5) as the product x*((x*x)*(x*x)); True: exp2(5*log2 x), the engine's own DXC lowering."""
        self._ck(self.lib.vqhip_set_fresnel_pow(self._h, 1 if exp2_log2 else 0))

    def set_arithmetic(self, dxc):
        """False (default): the literal reading of dot / normalize / length / reflect; True: the DXC reading (FMA-chain dot, v * correctly rounded rsqrt)."""
        self._ck(self.lib.vqhip_set_arithmetic(self._h, abi.ARITH_DXC if dxc else abi.ARITH_LITERAL))

    def set_option(self, key, value):
        """vqhip_set_option: a tuning / A-B option of this context (include/vqhip.h lists them); value None restores the default."""
        self._ck(self.lib.vqhip_set_option(self._h, key.encode(), None if value is None else str(value).encode()))

    # the environment variables of rounds 1-3 map onto options (scripts that sweep forms): VQHIP_LUT_FORM -> "lut_form", ...
    ENV_OPTIONS = {"VQHIP_LUT_FORM": "lut_form", "VQHIP_DIFFUSE_FORM": "diffuse_form", "VQHIP_DIFFUSE_SEQ_FORM": "diffuse_seq_form",
                   "VQHIP_BLUR_Y_WGS": "blur_y_wgs", "VQHIP_SHADE_WG": "shade_wg", "VQHIP_PSMAIN_WAVES": "psmain_waves"}

    def set_option_env(self, env_name, value):
        v = None if value in (None, "", "default", "0") else value
        self.set_option(self.ENV_OPTIONS[env_name], v)

    def unlit_composite(self, coverage_ip, colors, color, fmt, stream=None):
        """Light gizmo meshes (Unlit.hlsl:PSMain, SceneRendering.cpp:1787-1819): pixels whose ip2.w index is -(2+k) get colors[k]
        (sequence of 4-tuples); `color` is written in place."""
        _check_img(color, fmt, "color")
        h, w = color.shape[0], color.shape[1]
        for i, t in enumerate(coverage_ip):
            _check_img(t, FMT_RGBA32F, f"ip{i}")
        cov = abi.Interpolants(coverage_ip[0].data_ptr(), coverage_ip[1].data_ptr(), coverage_ip[2].data_ptr(), w, h, w)
        arr = (abi.float4 * max(len(colors), 1))(*[abi.float4(*[float(v) for v in c]) for c in colors])
        self._ck(self.lib.vqhip_unlit_composite(self._h, self._stream(stream), C.byref(cov), C.cast(arr, C.c_void_p), len(colors), _ptr(color), w, h, w, fmt))
        return color

    def conv_diffuse(self, chain, w0, h0, n_mips, res=64, step=0.010, order=CONV_SEQUENTIAL, fmt=FMT_RGBA16F, stream=None):
        dt, ch = _TORCH_DTYPE[fmt]
        out = torch.empty((6, res, res, ch), dtype=dt, device=self.device)
        self._ck(self.lib.vqhip_conv_diffuse(self._h, self._stream(stream), _ptr(chain), w0, h0, n_mips, res, step, order, _ptr(out), fmt))
        return out

    def conv_specular(self, chain, w0, h0, n_mips, res0=128, order=CONV_SEQUENTIAL, fmt=FMT_RGBA16F, stream=None):
        dt, ch = _TORCH_DTYPE[fmt]
        mips = abi.specular_mip_count(res0)
        out = torch.empty((abi.cube_px(res0, mips), ch), dtype=dt, device=self.device)
        self._ck(self.lib.vqhip_conv_specular(self._h, self._stream(stream), _ptr(chain), w0, h0, n_mips, res0, order, _ptr(out), fmt))
        return out, mips

    def envmap_prefilter(self, chain, w0, h0, n_mips, diffuse_res=64, diffuse_step=0.010, spec_res0=128, order=CONV_SEQUENTIAL, stream=None):
        """Returns dict(diffuse_unblurred, diffuse_blurred, specular, spec_mips) of float16 tensors (reference formats)."""
        mips = abi.specular_mip_count(spec_res0)
        d0 = torch.empty((6, diffuse_res, diffuse_res, 4), dtype=torch.float16, device=self.device)
        d1 = torch.empty_like(d0)
        tmp = torch.empty((diffuse_res, diffuse_res, 4), dtype=torch.float16, device=self.device)
        sp = torch.empty((abi.cube_px(spec_res0, mips), 4), dtype=torch.float16, device=self.device)
        o = abi.EnvMapOut(d0.data_ptr(), d1.data_ptr(), tmp.data_ptr(), sp.data_ptr())
        self._ck(self.lib.vqhip_envmap_prefilter(self._h, self._stream(stream), _ptr(chain), w0, h0, n_mips, diffuse_res, diffuse_step,
                                                 spec_res0, order, C.byref(o)))
        return {"diffuse_unblurred": d0, "diffuse_blurred": d1, "specular": sp, "spec_mips": mips, "_tmp": tmp}


def make_envmap(diffuse_cube, spec_cube, spec_res0, spec_mips, lut):
    """abi.EnvMap over device tensors (keeps no references — the caller owns the tensors)."""
    return abi.EnvMap(diffuse_cube.data_ptr(), diffuse_cube.shape[1], spec_cube.data_ptr(), spec_res0, spec_mips,
                      lut.data_ptr(), lut.shape[0])
