#!/usr/bin/env python3
"""Generates tests/golden/engine_default.npz (where /root/reference exists; ~4 min of CPU): the engine's default IBL configuration (tests/engine_default.py).
  spec_sample_ref     the REFERENCE'S HLSL (oracle/_ref, PSMain_SpecularIrradiance) at ~2 100 texels over all 9 mips of the 512^2 cube, RGBA16F
  band_ref            the reference's PSMain on a 3840 x 24 band lit by that environment with MaxEnvMapLODLevels = 9, RGBA16F
  sha_*               sha256 of the ORACLE'S full outputs (decoded image, 13-level chain, blurred diffuse cube, 512^2 x 9 specular cube): the GPU test demands the
                      product's whole cubes equal them bit for bit — every texel, not a sample
  lut comes from tests/golden/cfg4_env.npz (the 1024^2 x 2048 LUT does not depend on the environment)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import engine_default as E, oracle_lib as O, ref_cases, ref_lib as R  # noqa: E402
from vqengine_amd import abi  # noqa: E402


def main():
    t0 = time.time()
    data = E.hdr_file()
    img = O.hdr_decode(data)
    chain, n = O.mip_chain(img)
    assert n == 13 and img.shape == (E.H0, E.W0, 4)
    print(f"file {len(data)} bytes, chain {n} levels, {time.time() - t0:.0f} s", flush=True)
    tex = E.sample_texels()
    ref = np.zeros((len(tex), 4), np.float32)
    for m, faces, xs, ys, sel in E.split_texels(tex):
        rough = float(np.float32(m) / np.float32(E.SPEC_MIPS - 1))
        ref[sel] = R.conv_specular_texels(chain, E.W0, E.H0, n, E.SPEC_RES0 >> m, rough, m, faces, xs, ys)
    print(f"reference specular sample: {len(tex)} texels, {time.time() - t0:.0f} s", flush=True)
    pre = O.envmap_prefilter(chain, E.W0, E.H0, n, E.DIFF_RES, E.DIFF_STEP, E.SPEC_RES0, abi.CONV_SEQUENTIAL)
    assert pre["spec_mips"] == E.SPEC_MIPS
    print(f"oracle prefilter (diffuse 64^2 x 99 382 taps, specular 512^2 x 9): {time.time() - t0:.0f} s", flush=True)
    lut = ref_cases.cfg4_env()["lut"]
    env = {"diffuse": pre["diffuse_blurred"], "specular": pre["specular"], "spec_res0": E.SPEC_RES0, "spec_mips": E.SPEC_MIPS, "lut": lut}
    gb_raw, _, pf, extra, pv = E.band_inputs()
    band = R.forward_from_gbuffer(gb_raw, pf, pv, env=ref_cases.host_env(env), extra=extra)
    print(f"reference band: {time.time() - t0:.0f} s", flush=True)
    with np.errstate(over="ignore"):
        out = {"spec_sample_idx": tex, "spec_sample_ref": ref[:, :3].astype(np.float16), "band_ref": band[..., :3].astype(np.float16),
               "sha_file": np.frombuffer(E.sha(np.frombuffer(data, np.uint8)).encode(), np.uint8), "sha_image": np.frombuffer(E.sha(img).encode(), np.uint8),
               "sha_chain": np.frombuffer(E.sha(chain).encode(), np.uint8), "sha_diffuse": np.frombuffer(E.sha(pre["diffuse_blurred"]).encode(), np.uint8),
               "sha_specular": np.frombuffer(E.sha(pre["specular"]).encode(), np.uint8)}
    path = os.path.join(ROOT, "tests", "golden", "engine_default.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
