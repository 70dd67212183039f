#!/usr/bin/env python3
"""Generates tests/golden/cfg4_env.npz: the load-time IBL outputs of BASELINE config 4 at FULL size, computed by the CPU oracle
(the 2048^2 synthetic equirect seed 0xE9 -> 12-level min-filter chain -> diffuse irradiance 6x64^2 at step 0.010 (99 382 taps per
texel) -> blur -> GGX-prefiltered specular 128^2 x 7 mips; BRDF LUT 1024^2 x 2048 samples) in the reference's storage formats
(RGBA16F / RG16F, stored as uint16 bit patterns).
Two users:
  * INPUT of the BASELINE-shape reference-source fixtures (tests/ref_cases.py band cases: cfg3 needs "IBL from cfg4's outputs");
    about 3 minutes of oracle time on 8 cores, far too long to rebuild inside every CPU test run;
  * golden for tests/test_gpu_parity.py::test_cfg4_env_matches_golden: the HIP product must reproduce every texel bit for bit.
Usage: python tests/golden/make_cfg4_env.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import oracle_lib as O  # noqa: E402
from vqengine_amd import abi, synth  # noqa: E402


def main():
    t = time.time()
    eq = synth.equirect(2048, 2048)
    chain, n = O.mip_chain(eq)
    pre = O.envmap_prefilter(chain, 2048, 2048, n, 64, 0.010, 128, abi.CONV_SEQUENTIAL)
    lut = O.brdf_lut(1024, 2048, abi.FMT_RG16F)
    path = os.path.join(ROOT, "tests", "golden", "cfg4_env.npz")
    np.savez_compressed(path, diffuse=pre["diffuse_blurred"].view(np.uint16), specular=pre["specular"].view(np.uint16),
                        lut=lut.view(np.uint16), spec_res0=np.int32(128), spec_mips=np.int32(pre["spec_mips"]))
    print(path, os.path.getsize(path), "bytes", f"{time.time() - t:.0f} s")


if __name__ == "__main__":
    main()
