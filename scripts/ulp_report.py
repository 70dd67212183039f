#!/usr/bin/env python3
"""Oracle (CPU) vs tests/golden/ref_outputs.npz in STORAGE-FORMAT ulps (fp16 for RGBA16F/RG16F targets, 8-bit for UNORM8):
the measurement the tolerances of tests/ref_cases.py are set from.   Usage: python scripts/ulp_report.py [case-substring]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import ref_cases  # noqa: E402


def key16(x):
    with np.errstate(over="ignore"):
        u = np.asarray(x, np.float32).astype(np.float16).view(np.uint16).astype(np.int32)
    return np.where(u & 0x8000, -(u & 0x7fff), u)


def three_way():
    """--reading dxc: product arithmetic (the oracle; the HIP kernels are bit-identical to it) vs the LITERAL reading of the reference's
    HLSL vs its DXC reading (hlsl_shim.h VQ_SHIM_DXC) on the bands of the BASELINE frames, in RGBA16F ulps of the scene colour."""
    lit = np.load(os.path.join(ROOT, "tests", "golden", "ref_outputs.npz"))
    dxc = np.load(os.path.join(ROOT, "tests", "golden", "ref_outputs_dxc.npz"))
    print("| band | pair | channels | max ulps | differing | > 1 ulp | > 4 ulps |\n|---|---|---|---|---|---|---|")
    for tag, (build, _, oracle_scene) in ref_cases.DXC_SCENES.items():
        o = oracle_scene(build())
        for a, b, name in ((o, lit[tag + "/scene"], "product vs literal"), (o, dxc[tag + "/scene"], "product vs dxc"), (lit[tag + "/scene"], dxc[tag + "/scene"], "literal vs dxc")):
            fin = np.isfinite(np.asarray(a, np.float32).astype(np.float16)) & np.isfinite(np.asarray(b, np.float32).astype(np.float16))
            d = np.abs(key16(a) - key16(b))[fin]
            print(f"| {tag} | {name} | {d.size} | {int(d.max())} | {np.mean(d > 0) * 100:.4f} % | {np.mean(d > 1) * 100:.4f} % | {np.mean(d > 4) * 100:.4f} % |")


def main():
    if "--reading" in sys.argv:
        assert sys.argv[sys.argv.index("--reading") + 1] == "dxc"
        return three_way()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_outputs.npz"))
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for c in ref_cases.CASES:
        if pat not in c.name or c.tol == "exact":
            continue
        inp = c.build()
        got, ref = np.asarray(c.oracle(inp)), fx[c.name]
        if got.dtype == np.uint8:
            d = np.abs(got.astype(np.int32) - (ref if ref.dtype == np.uint8 else ref_cases.to_unorm8(ref)).astype(np.int32))
        else:
            d = np.abs(key16(got) - key16(ref))
        print(f"{c.name:36s} n={d.size:7d} max={int(d.max()):4d} frac>0={np.mean(d > 0):.5f} frac>1={np.mean(d > 1):.6f} count>1={int((d > 1).sum())}")


if __name__ == "__main__":
    main()
