"""bench.py's bookkeeping that needs no GPU: the counter-derived constants it reports must have been measured on the kernel sources that are in
the tree (profiles/pmc_constants.json is stamped with their sha256 by scripts/pmc_refresh.sh), and the two BASELINE configs are what
BASELINE.json names."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_constants_were_measured_on_the_current_kernel_sources():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_constants.json")))
    assert d["kernel_sources_sha256"] == bench.kernel_source_hash(), \
        "shade.hip / vq_devmath.h / vq_sampling.h changed since profiles/pmc_constants.json was measured: rerun scripts/pmc_refresh.sh on the GPU box " \
        "(bench.py would print roofline.traffic = null and no valu_issue until then)"
    for key in ("cfg3/product", "cfg3/exp2_log2", "cfg5/product"):
        pmc, meta = bench.load_pmc_constants(*key.split("/"))
        assert pmc is not None and meta["stale"] is False, (key, meta)
        assert pmc["valu_instr_per_wave"] > 1000 and pmc["hbm_bytes_per_launch"] > 0


def test_configs_are_the_baseline_ones():
    c3, c5 = bench.CONFIGS["cfg3"], bench.CONFIGS["cfg5"]
    assert (c3["width"], c3["height"], c3["lights"], c3["env"], c3["scaling"]) == (3840, 2160, 64, True, "weak")
    assert (c5["width"], c5["height"], c5["lights"], c5["scaling"]) == (7680, 4320, 256, "strong")
    assert "4K,64 lights" in c3["metric"]
