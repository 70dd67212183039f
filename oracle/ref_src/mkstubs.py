#!/usr/bin/env python3
"""mkstubs.py — generates, at build time and into oracle/_ref/gen/winstub/ only, the minimal stand-ins that let the reference's
Source/Renderer/Resources/DXGIUtils.cpp (home of VQ_DXGI_UTILS::MipImage) compile with g++:
  dxgiformat.h        an `enum DXGI_FORMAT` holding exactly the DXGI_FORMAT_* names the file mentions (values are irrelevant to
                      MipImage, they only have to be distinct), `HRESULT` and the DXGI_ERROR_* names of its error-string switch
  Engine/GPUMarker.h  SCOPED_CPU_MARKER(x) as a no-op (the real header pulls <Windows.h> and the PIX runtime)
Nothing here restates reference logic; the names are read from the reference file itself.  Usage: mkstubs.py <DXGIUtils.cpp> <out dir>"""
import os
import re
import sys

src = open(sys.argv[1], encoding="latin-1").read()
out = sys.argv[2]
os.makedirs(os.path.join(out, "Engine"), exist_ok=True)
fmts = sorted(set(re.findall(r"\bDXGI_FORMAT_\w+", src)))
errs = sorted(set(re.findall(r"\bDXGI_ERROR_\w+", src)))
with open(os.path.join(out, "dxgiformat.h"), "w") as f:
    f.write("// GENERATED stub (oracle/ref_src/mkstubs.py) - never commit\n#pragma once\n#include <cstdint>\n#include <cstddef>\n#include <cstring>\n")
    f.write("typedef long HRESULT;\nenum DXGI_FORMAT {\n" + ",\n".join("  %s = %d" % (n, i) for i, n in enumerate(fmts)) + "\n};\n")
    f.write("enum {\n" + ",\n".join("  %s = %d" % (n, -1000 - i) for i, n in enumerate(errs)) + "\n};\n")
    f.write("#ifndef E_FAIL\nenum { S_OK = 0, E_FAIL = -1, E_INVALIDARG = -2, E_OUTOFMEMORY = -3, E_NOTIMPL = -4, S_FALSE = 1 };\n#endif\n")
with open(os.path.join(out, "Engine", "GPUMarker.h"), "w") as f:
    f.write("// GENERATED stub (oracle/ref_src/mkstubs.py) - never commit\n#pragma once\n#define SCOPED_CPU_MARKER(x)\n#define SCOPED_CPU_MARKER_C(x, c)\n#define SCOPED_CPU_MARKER_F(...)\n")
