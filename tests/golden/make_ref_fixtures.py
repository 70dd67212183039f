#!/usr/bin/env python3
"""Generates tests/golden/ref_outputs.npz: the OUTPUTS of the reference's own sources (oracle/_ref, built by `make -C oracle ref`
from /root/reference) for every case of tests/ref_cases.py, plus a checksum of each case's (seed-built) inputs.
Runs only where /root/reference exists; the fixture travels to the GPU box, where tests/test_ref_fixtures.py compares the
oracle (CPU) and the HIP product (gpu) with it.   Usage: python tests/golden/make_ref_fixtures.py [--only case,case,...]
(--only: re-run just those cases and keep every other entry of the existing ref_outputs.npz as it is; the DXC file is left alone)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import ref_cases, ref_lib  # noqa: E402


def main():
    assert ref_lib.available("shaders") and ref_lib.available("fsr") and ref_lib.available("mip"), "build oracle/_ref first: make -C oracle ref"
    out = {}
    only = None
    path = os.path.join(ROOT, "tests", "golden", "ref_outputs.npz")
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
        assert only <= {c.name for c in ref_cases.CASES}, only
        with np.load(path) as z:
            out = {k: z[k] for k in z.files}
    for c in ref_cases.CASES:
        if only is not None and c.name not in only:
            continue
        inp = c.build()
        ref = np.asarray(c.ref(inp))
        if c.store == "f16":
            with np.errstate(over="ignore"):
                ref = ref.astype(np.float16)
        elif c.store == "u8":
            ref = ref_cases.to_unorm8(ref)
        out[c.name] = ref
        out[c.name + "/inputs"] = np.frombuffer(ref_cases.checksum(inp).encode(), np.uint8)
        print(f"{c.name:40s} {str(ref.shape):18s} {ref.dtype}")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
    if only is not None:
        return
    # the SECOND reading of the reference's intrinsics (hlsl_shim.h VQ_SHIM_DXC) on the BASELINE-shape bands: scene colour, RGBA16F
    assert ref_lib.available("shaders_dxc") and ref_lib.available("shaders_dxc_l256")
    dxc = {}
    for tag, (build, ref_dxc, _) in ref_cases.DXC_SCENES.items():
        inp = build()
        with np.errstate(over="ignore"):
            dxc[tag + "/scene"] = np.asarray(ref_dxc(inp)).astype(np.float16)
        dxc[tag + "/scene/inputs"] = np.frombuffer(ref_cases.checksum(inp).encode(), np.uint8)
        print(f"dxc reading {tag:28s} {dxc[tag + '/scene'].shape}")
    path = os.path.join(ROOT, "tests", "golden", "ref_outputs_dxc.npz")
    np.savez_compressed(path, **dxc)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
