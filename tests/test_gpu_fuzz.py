"""-m gpu: a slice of the randomised parity runs of tests/fuzz/fuzz_shade.py, fuzz_post.py, fuzz_casters.py, fuzz_ibl.py and fuzz_wide.py (each draws sizes, counts, formats, arithmetic readings, options and
special values per case from its seed and demands the HIP product's bits == the oracle's). The long runs are the scripts themselves (round 6: 9 862 shade cases, 10 087 post
cases, 16 195 caster cases, 9 567 load-time IBL cases on the GPU, profiles/r6w_fuzz.md); the seeds that ever failed are replayed in tests/test_gpu_casters.py."""
import os
import sys

import numpy as np
import pytest

from tests.test_gpu_parity import dev

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,first,count", [("fuzz_shade", 7000021, 60), ("fuzz_post", 7000021, 60), ("fuzz_casters", 7000021, 60), ("fuzz_ibl", 7000021, 60), ("fuzz_wide", 7000021, 60)])
def test_fuzz_slice(ctx, name, first, count):
    mod = __import__(name)
    bad = []
    for seed in range(first, first + count):
        n, idx, what = mod.run_case(ctx, seed, dev)[:3]
        if n:
            bad.append(f"{what}: {n} channels, first at {np.asarray(idx).tolist()[:1]}")
    assert not bad, "\n".join(bad)
