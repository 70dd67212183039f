"""A slice of tests/fuzz/fuzz_ref.py in the CPU suite: the ORACLE against the reference's own HLSL (oracle/_ref, built where /root/reference is available) on the random frames
the GPU fuzzers draw — PSMain in both readings with and without casters, the post chain, FSR, the skydome, the reflections composite, the load-time IBL passes. Skipped where oracle/_ref is absent."""
import os
import sys

import pytest

from tests import ref_lib as R

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
pytestmark = pytest.mark.skipif(not (R.available("shaders") and R.available("shaders_dxc") and R.available("shaders_l256") and R.available("fsr")),
                                reason="oracle/_ref is not built here (needs /root/reference)")


@pytest.mark.parametrize("kind,first,count", [("shade", 9000001, 24), ("casters", 9000001, 24), ("post", 9000001, 40), ("wide", 9000001, 40), ("ibl", 9000001, 10), ("psmain", 9000001, 20)])
def test_oracle_against_the_reference_hlsl_on_random_frames(kind, first, count):
    import fuzz_ref
    compared = 0
    for seed in range(first, first + count):
        res = (fuzz_ref.run_post(seed) if kind == "post" else fuzz_ref.run_wide(seed) if kind == "wide" else fuzz_ref.run_ibl(seed) if kind == "ibl" else fuzz_ref.run_psmain(seed) if kind == "psmain"
               else fuzz_ref.run_shade(seed, kind == "casters"))
        if res is None:
            continue
        ch, above, strict, worst, cls, where, reading = res
        compared += ch
        assert strict == 0, f"{kind} seed {seed} ({reading}): {strict} channels above one unit where the two sides perform the same operations, first at {where}"
    assert compared > 1000
