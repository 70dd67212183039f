"""scripts/compare_dump.py — the tool that diffs an engine-side (D3D12 / WARP) dump of Tex_SceneColor against this library in RGBA16F ulps
(docs/WARP_CALIBRATION.md): its self-test fabricates a dump from the oracle with one known 1-ulp difference and must find exactly that."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compare_dump_selftest():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "compare_dump.py"), "--selftest"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "selftest OK" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
