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
    assert "vqengine_amd/csrc/vq_shade.h" in bench.PMC_SOURCES          # the per-pixel body of the shade kernel lives there since round 3
    for key in ("cfg3/product", "cfg3/exp2_log2", "cfg5/product", "cfg3_coherent/product"):
        pmc, meta = bench.load_pmc_constants(*key.split("/"))
        assert pmc is not None and meta["stale"] is False, (key, meta)
        assert pmc["valu_instr_per_wave"] > 1000 and pmc["hbm_bytes_per_launch"] > 0


def test_configs_are_the_baseline_ones():
    c2 = bench.CONFIGS["cfg2"]
    assert (c2["width"], c2["height"], c2["lights"], c2["env"], c2["seed"], c2["light_seed"]) == (1920, 1080, 16, False, 0xC0FFEE, 0x1600)
    c3, c5 = bench.CONFIGS["cfg3"], bench.CONFIGS["cfg5"]
    assert (c3["width"], c3["height"], c3["lights"], c3["env"], c3["scaling"]) == (3840, 2160, 64, True, "weak")
    assert (c5["width"], c5["height"], c5["lights"], c5["scaling"]) == (7680, 4320, 256, "strong")
    assert "4K,64 lights" in c3["metric"]


def test_every_invocation_reports_the_other_baseline_configs():
    """the default invocation times cfg5 strong scaling (every N) and cfg2 / cfg4 / the coherent frame / the tile curve (N = 1)
    next to the cfg3 headline; source-level check of the wiring (the values need a GPU: tests/test_gpu_bench_flow.py)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    wid = open(os.path.join(ROOT, "benchlib", "widened.py")).read()         # extras["widened"] = widened_report(...)
    assert "from benchlib.widened import widened_report" in src
    for key in ('extras["cfg5_strong"]', 'extras["tile_curve"]', 'extras["cfg2"]', 'extras["ibl_load"]', 'extras["coherent_scene"]', 'extras["widened"]', '"rccl": comms.info()'):
        assert key in src, key
    # every SURVEY 8f kernel rides in the driver-run line
    for key in ("gbuffer_producer_textured", "gbuffer_producer_textureless", "psmain_fused", "skydome_all_sky", "hdr_decode_2048", "fsr_easu_1440p_to_4k", "fsr_rcas_4k", "ssr_env_fallback_4k",
                "psmain_fused_mrt", "scene_normals_prepass"):
        assert f'res["{key}"]' in wid, key
    assert 'default="auto"' in src and "completes_within" in src          # the overlapped composite runs under a watchdog by default
