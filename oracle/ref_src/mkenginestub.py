#!/usr/bin/env python3
"""mkenginestub.py — generates, at build time and into oracle/_ref/gen/enginestub/ only, the ONE engine header that the reference's
Source/Renderer/Rendering/RenderPass/RenderPass.h includes (RenderPass.h:20: "Renderer/Pipeline/PipelineStateObjects.h", which pulls the
D3D12 headers): a stand-in that declares the single name RenderPass.h takes from it — the element type of CollectPSOCreationParameters()'s
return value (RenderPass.h:58) — as an empty struct. The name is read from the reference's own headers (and the script fails if either file stops
mentioning it), so nothing here restates reference logic. With it, include/vqhip_passes.hpp compiles against the reference's REAL RenderPass.h
and links the reference's REAL RenderPass.cpp (tests/test_engine_header.py).   Usage: mkenginestub.py <reference Source dir> <out dir>"""
import os
import re
import sys

src_root, out = sys.argv[1], sys.argv[2]
rp = open(os.path.join(src_root, "Renderer/Rendering/RenderPass/RenderPass.h"), encoding="latin-1").read()
incs = re.findall(r'#include\s+"([^"]+)"', rp)
assert incs == ["Renderer/Pipeline/PipelineStateObjects.h"], f"RenderPass.h now includes {incs}: extend this script"
m = re.search(r"std::vector<(\w+)>\s+CollectPSOCreationParameters", rp)
assert m, "RenderPass.h no longer declares CollectPSOCreationParameters()"
name = m.group(1)
pso = open(os.path.join(src_root, incs[0]), encoding="latin-1").read()
assert re.search(r"\bstruct\s+%s\b" % name, pso), f"{incs[0]} does not define struct {name}"
dst = os.path.join(out, incs[0])
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, "w") as f:
    f.write("// GENERATED stub (oracle/ref_src/mkenginestub.py) - never commit\n#pragma once\nstruct %s {};\n" % name)
print("stub:", dst, "declares struct", name)
