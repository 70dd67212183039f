"""CPU-side checks of the drop-in boundary: struct layouts == VQ_SHADER_DATA, the C-ABI library
loads and exports every symbol include/vqhip.h declares, and the product has no CPU fallback."""
import ctypes as C
import os
import re
import subprocess

import pytest

from vqengine_amd import abi, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vqhip.h")


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"VQHIP_API\s+[\w\s\*]+?\b(vqhip_\w+)\s*\(", src)))


def test_header_compiles_as_c_and_cpp(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "vqhip.h"\nint main(void){return (int)sizeof(VQ_PerFrameData) - 7120;}\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c), "-o", str(tmp_path / "c.o")])
    cpp = tmp_path / "t.cpp"
    cpp.write_text(c.read_text())
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(cpp), "-o", str(tmp_path / "cpp.o")])


def test_struct_layouts_match_reference_cbuffers():
    # sizes / offsets derived from Shaders/LightingConstantBufferData.h:50-186; abi.py asserts the
    # offsets at import, here the totals once more + the C header's own static_asserts were compiled above
    assert C.sizeof(abi.PointLight) == 48 and C.sizeof(abi.SpotLight) == 64 and C.sizeof(abi.DirectionalLight) == 40
    assert C.sizeof(abi.SceneLighting) == 7088 and C.sizeof(abi.PerFrameData) == 7120
    assert C.sizeof(abi.PerViewLightingData) == 320 and C.sizeof(abi.MaterialData) == 80
    assert abi.NUM_LIGHTS__POINT == 100 and abi.NUM_LIGHTS__SPOT == 20


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 19, syms
    assert sorted(capi.EXPORTED_SYMBOLS) == syms, "capi.EXPORTED_SYMBOLS out of sync with include/vqhip.h"
    lib = capi.load_library()                       # loud failure if the .so is missing
    for s in syms:
        assert hasattr(lib, s), f"libvqhip.so does not export {s}"
    assert lib.vqhip_abi_version() == abi.ABI_VERSION == 3


def test_size_helpers_match_reference_mip_rules():
    lib = capi.load_library()
    # Image::CalculateMipLevelCount == floor(log2(max(w,h)))+1 ; specular cube drops the 1x1 level (EnvironmentMapRendering.cpp:63)
    assert lib.vqhip_mip_level_count(2048, 2048) == 12 and lib.vqhip_mip_level_count(2048, 1024) == 12 and lib.vqhip_mip_level_count(1, 1) == 1
    assert lib.vqhip_specular_mip_count(128) == 7 and lib.vqhip_specular_mip_count(256) == 8 and lib.vqhip_specular_mip_count(512) == 9
    assert lib.vqhip_mip_chain_bytes(4, 2, 3) == (8 + 2 + 1) * 16
    assert lib.vqhip_mip_level_offset_bytes(2048, 2048, 3) == (2048 ** 2 + 1024 ** 2 + 512 ** 2) * 16
    assert lib.vqhip_cube_bytes(128, 7, abi.FMT_RGBA16F) == 6 * sum((128 >> m) ** 2 for m in range(7)) * 8
    assert abi.mip_level_count(2048, 2048) == 12 and abi.specular_mip_count(128) == 7


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product must fail loudly (VQHIP_ERR_NO_DEVICE), not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    lib = capi.load_library()
    h = C.c_void_p()
    rc = lib.vqhip_create(0, C.byref(h))
    assert rc == abi.VQHIP_ERR_NO_DEVICE and not h.value
    assert b"no HIP device" in lib.vqhip_last_error(None) or b"gfx950" in lib.vqhip_last_error(None)
    with pytest.raises(capi.VQHipError):
        capi.Context(0)


def test_product_does_not_reference_the_oracle():
    """Nothing under vqengine_amd/ may import, include or link oracle/ (it is the checker, not the product)."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "vqengine_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"vqo_|libvqoracle|oracle_lib|#include\s+\"[^\"]*oracle/", txt):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
    out = subprocess.run(["ldd", capi.lib_path()], capture_output=True, text=True).stdout
    assert "vqoracle" not in out


def test_context_guard_is_declared_for_every_ctx_entry_point():
    """Every entry point that takes a vqhip_ctx marks the context busy for the duration of the call (CtxGuard, capi.hip): a second thread
    entering the same context is refused instead of corrupting the constant ring. Source-level check (the behaviour needs a GPU:
    tests/test_gpu_round3.py::test_context_refuses_a_second_thread)."""
    import re
    src = open(os.path.join(ROOT, "vqengine_amd", "csrc", "capi.hip")).read()
    entries = re.findall(r"\n(?:int|size_t) (vqhip_[a-z0-9_]+)\(vqhip_ctx\* ctx", src)
    assert len(entries) >= 24
    for name in entries:
        body = src[src.index(f" {name}(vqhip_ctx* ctx"):]
        body = body[:body.index("\n}\n")]
        assert "CTX_GUARD(ctx" in body, name
