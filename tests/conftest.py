import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Order of the GPU suite (the driver runs `pytest -x -m gpu`): kernel-parity files first — reference fixtures, then product-vs-oracle — and the
# files that launch subprocesses (the C++ adaptor driver, bench.py under torchrun) last, so that no control-flow or environment problem of a
# launcher can keep a parity test from running. Within a rank the collection order (file name, then definition order) is kept.
_GPU_ORDER = ["test_ref_fixtures", "test_gpu_parity", "test_gpu_round3", "test_gpu_conv_forms", "test_fresnel_pow_modes", "test_gpu_arith_modes", "test_gpu_devmath",
              "test_gpu_gbuffer", "test_gpu_psmain_targets", "test_gpu_scene_normals", "test_gpu_ssr", "test_gpu_fsr", "test_gpu_hdri", "test_gpu_pitch", "test_gpu_frame",
              "test_unlit_composite"]
_GPU_LAST = ["test_gpu_passes", "test_gpu_mgpu", "test_gpu_bench_flow"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if item.get_closest_marker("gpu") is None:
            return 0
        if mod in _GPU_ORDER:
            return 1 + _GPU_ORDER.index(mod)
        if mod in _GPU_LAST:
            return 1000 + _GPU_LAST.index(mod)
        return 500
    items.sort(key=rank)                                     # stable: ties keep their collection order


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible — the product has no CPU fallback")
    from vqengine_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture
def set_opt(ctx):
    """set_opt(key, value): vqhip_set_option on the session's context for the duration of ONE test (the defaults come back afterwards)"""
    used = []

    def f(key, value):
        ctx.set_option(key, value)
        used.append(key)
    yield f
    for k in used:
        ctx.set_option(k, None)
