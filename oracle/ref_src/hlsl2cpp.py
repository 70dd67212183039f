#!/usr/bin/env python3
"""hlsl2cpp.py — rewrite the reference's HLSL sources, read WHERE THEY LIE under /root/reference/Shaders, into text
that g++ accepts together with hlsl_shim.h. Output goes to oracle/_ref/gen/ only (git-ignored, never committed):
nothing of the reference is copied into the repository, the generated files exist on the machine that holds the
reference for as long as it takes to build oracle/_ref/libvqref_shaders.so.

The rewrites are purely syntactic — no expression is reordered, no constant touched:
  1. `: register(x)` bindings and `: SEMANTIC` annotations are dropped; `cbuffer X { T v; }` becomes `T v;`
  2. `[numthreads(..)]`, `[unroll]`, `[loop]`, `[branch]`, `[flatten]` attributes are dropped
  3. unsuffixed floating literals get an `f` suffix (HLSL literals are float, C++ ones double)
  4. `in` parameter qualifiers are dropped, `out` / `inout` parameters become references — also where a macro hides them
     (ffx_a.h: `#define inAF2 in AF2`, `#define outAF2 out AF2`)
  5. `(Type)0` zero-initialisation casts become `Type{}`
  6. swizzles of scalars (`0.5f.xx`, `(1.0f - r).xxx`) become vector constructors
  7. per-file patches listed in PATCHES below (each one says why)
  8. `float3 c = 0.0f;` (scalar splat initialisation) becomes `float3 c = float3(0.0f);`, `Texture2D<float4>` becomes `Texture2D`,
     `Texture2D<float>` becomes `Texture2DF`; `register(t0, space1)` is dropped like `register(t0)`
Usage: hlsl2cpp.py <reference Shaders dir> <output dir> file.hlsl [...]"""
import os
import re
import sys

SEMANTICS = r"SV_\w+|POSITION\d*|NORMAL\d*|TANGENT\d*|TEXCOORD\d*|COLOR\d*"
ZERO_CAST_TYPES = r"PSOutput|PSInput|BRDF_Surface|ShadowTestPCFData|float[234]|float[34]x[34]"

# (file, literal old, literal new, reason)
PATCHES = [
    ("ForwardLighting.hlsl", '#include "Tessellation.hlsl"', "", "hull/domain stages are not on the path (and are not C++-expressible)"),
    ("ForwardLighting.hlsl", "Surface.roughness.r)", "Surface.roughness)", "swizzle of a scalar member: `.r` of a float is the float"),
    ("ScreenSpaceReflections/ClassifyReflectionTiles.hlsl", '#include "../AMDFidelityFX/DNSR/ffx_denoiser_reflections_common.h"', "",
     "only FFX_DNSR_Reflections_RemapLane8x8 of that header is used, by CSMain (cut below); the header's min16float helpers are not C++-expressible"),
    ("ScreenSpaceReflections/ClassifyReflectionTiles.hlsl", "(dispatch_thread_id + 0.5) *", "(to_float(dispatch_thread_id) + 0.5) *",
     "HLSL promotes uint2 + float to float2; C++ would pick an integer overload"),
    ("LightingConstantBufferData.h", "#define NUM_LIGHTS__POINT 100", "#ifndef NUM_LIGHTS__POINT\n#define NUM_LIGHTS__POINT 100\n#endif",
     "the engine's light cap stays 100 in libvqref_shaders.so; BASELINE cfg5 (256 point lights) exceeds it, so a SECOND build "
     "(libvqref_shaders_l256.so, -DNUM_LIGHTS__POINT=256) raises the cap the way the header's own comment describes — nothing else differs"),
]
# (file, start marker, end marker): text from the first marker up to (not including) the second is dropped (end marker None: to the end of the file)
CUTS = [
    ("ScreenSpaceReflections/Common.hlsl", "uint PackFloat16(", "// Transforms origin to uv space", "half packing / ray-coordinate packing helpers: min16float types, not on the path"),
    ("ScreenSpaceReflections/ClassifyReflectionTiles.hlsl", "void IncrementRayCounter(", "bool IsReflectiveSurface(", "ray / tile list appends: atomics on UAV buffers (FidelityFX SSSR, out of scope)"),
    ("ScreenSpaceReflections/ClassifyReflectionTiles.hlsl", "bool IsBaseRay(", "float3 SampleEnvironmentMap(", "ray selection + groupshared counter (out of scope)"),
    ("ScreenSpaceReflections/ClassifyReflectionTiles.hlsl", "void ClassifyTiles(", None,
     "ClassifyTiles / CSMain: wave intrinsics (WaveReadLaneAt, WavePrefixCountBits ...); the three lines of ClassifyTiles that call SampleEnvironmentMap "
     "(:146-152) are restated by the harness ref_ssr.cpp, like the rasteriser around PSMain"),
    ("ForwardLighting.hlsl", "PSInput TransformVertex(", "PSOutput PSMain(", "vertex stage: outside the path, uses float4x3 casts of the world matrices"),
    ("DepthPrePass.hlsl", "PSInput TransformVertex(", "float4 PSMain(PSInput In)", "vertex stage + the Tessellation.hlsl include: outside the path (as for ForwardLighting.hlsl)"),
    ("CubemapConvolution.hlsl", "GSOut VSMain_PerFace(", "float4 PSMain_DiffuseIrradiance(", "vertex / geometry stages (TriangleStream): the cube rasterisation is the harness's"),
]


def translate(name, src):
    for f, old, new, _ in PATCHES:
        if f == name:
            assert old in src, (name, old)
            src = src.replace(old, new)
    for f, a, b, _ in CUTS:
        if f == name:
            i, j = src.index(a), (src.index(b) if b is not None else len(src))
            assert i < j
            src = src[:i] + src[j:]
    src = re.sub(r"#pragma once", "", src)
    # 1. bindings, semantics, cbuffers
    src = re.sub(r":\s*register\s*\(\s*\w+\s*(?:,\s*\w+\s*)?\)", "", src)
    src = re.sub(r"cbuffer[ \t]+\w+\s*\{([^}]*)\}[ \t]*;?", lambda m: m.group(1), src)
    src = re.sub(r":\s*(?:%s)\b" % SEMANTICS, "", src)
    # 2. attributes
    src = re.sub(r"^\s*\[\s*(?:numthreads|unroll|loop|branch|flatten)[^\]]*\]", "", src, flags=re.M)
    # 3. literals (not inside preprocessor lines, which only hold integers and names here)
    lit = re.compile(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])")
    out = []
    for line in src.split("\n"):
        if line.lstrip().startswith("#") and not line.lstrip().startswith("#define"):
            out.append(line)
        else:
            out.append(lit.sub(lambda m: m.group(1) + "f", line))
    src = "\n".join(out)
    # 6. scalar swizzles (before 4/5 so the parenthesised form is still intact)
    src = re.sub(r"(?<![\w.])(\d+\.?\d*f?)\.(x{2,4}|r{2,4})\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), src)
    src = re.sub(r"(?<![\w])\(([^()]*)\)\.(x{2,4}|r{2,4})\b", lambda m: "float%d(%s)" % (len(m.group(2)), m.group(1)), src)
    # 4. parameter qualifiers
    src = re.sub(r"\b(?:inout|out)\s+((?:const\s+)?[\w<>]+)\s+(\w+)", r"\1& \2", src)
    src = re.sub(r"\bconst\s+in\b", "const", src)
    src = re.sub(r"\bin\s+const\b", "const", src)
    src = re.sub(r"([(,]\s*)in\s+(?=[\w<>]+\s+\w+)", r"\1", src)
    # 8. scalar -> vector initialisation (`float3 c = 0.0f;` splats in HLSL), typed SRVs
    src = re.sub(r"\b(float[234])\s+(\w+)\s*=\s*([-+]?\d[\w.]*)\s*;", r"\1 \2 = \1(\3);", src)
    src = re.sub(r"\bTexture2D\s*<\s*float4\s*>", "Texture2D", src)
    src = re.sub(r"\bTexture2D\s*<\s*float\s*>", "Texture2DF", src)          # single-channel SRV: Load / operator[] return a float
    # 4b. the same qualifiers hidden in macros (ffx_a.h: `#define inAF2 in AF2`, `#define outAF2 out AF2`, `#define inoutAF2 inout AF2`)
    src = re.sub(r"^(\s*#define\s+\w+)\s+in\s+(\w+)\s*$", r"\1 \2", src, flags=re.M)
    src = re.sub(r"^(\s*#define\s+\w+)\s+(?:inout|out)\s+(\w+)\s*$", r"\1 \2&", src, flags=re.M)
    # 5. zero casts
    src = re.sub(r"\(\s*(%s)\s*\)\s*0\b(?!\.)" % ZERO_CAST_TYPES, r"\1{}", src)
    return src


def main():
    shader_dir, out_dir, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    os.makedirs(out_dir, exist_ok=True)
    for f in files:
        with open(os.path.join(shader_dir, f), encoding="latin-1") as fh:
            src = fh.read()
        os.makedirs(os.path.dirname(os.path.join(out_dir, f)), exist_ok=True)
        with open(os.path.join(out_dir, f), "w", encoding="latin-1") as fh:
            fh.write("// GENERATED by oracle/ref_src/hlsl2cpp.py from the reference's Shaders/%s - never commit\n" % f)
            fh.write(translate(f, src))


if __name__ == "__main__":
    main()
