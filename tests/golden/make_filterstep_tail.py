#!/usr/bin/env python3
"""Generates tests/golden/specular_filterstep_tail.json (where /root/reference exists): every channel of the full-size cfg4 specular cube (128^2 x 7 mips, 393 192
channels) where the oracle (= the HIP product, bit for bit) is MORE than one RGBA16F ulp away from the reference's own HLSL — as DATA, with its cause:
for each such texel the equirect fetches whose fixed-point filter fractions (D3D's 8-bit fractions, oracle/vqo_sampling.h:fixed8) or LOD fractions differ between the
reference's evaluation (libm atan2f / asinf / log2f through oracle/ref_src/hlsl_shim.h) and the contract's (polynomials, oracle/vqo_math.h), with BOTH uv values.
tests/test_ref_fixtures.py / test_filterstep_tail.py demand that the set of deviating channels is exactly this list."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_lib as O, ref_cases, ref_lib as R  # noqa: E402
from vqengine_amd import abi  # noqa: E402

RES0, MIPS = 128, 7


def locate(t):
    base = 0
    for m in range(MIPS):
        r = RES0 >> m
        n = 6 * r * r
        if t < base + n:
            q = t - base
            return m, r, q // (r * r), q % r, (q // r) % r
        base += n
    raise ValueError(t)


def fixed8(v):
    return np.floor(v.astype(np.float64) * 256.0 + 0.5).astype(np.int64)


def main():
    i = ref_cases._cfg4_inputs()
    chain, n = i["chain"], i["n"]
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_outputs.npz"))
    ref = fx["cfg4_specular_128x7_full"]                                            # float16 [texels, 3]: the reference's HLSL
    got = O.conv_specular(chain, 2048, 2048, n, RES0, abi.CONV_SEQUENTIAL, abi.FMT_RGBA16F)[0][:, :3]
    d = ref_cases.ulp16_distance(got, ref)
    bad = np.argwhere(d > 1)
    out = {"case": "cfg4_specular_128x7_full", "channels_total": int(d.size), "channels_differing": int((d != 0).sum()), "channels_above_one_ulp": int(len(bad)), "entries": []}
    for t in sorted(set(int(b[0]) for b in bad)):
        m, r, f, x, y = locate(t)
        rough = float(np.float32(m) / np.float32(MIPS - 1))
        to, _ = O.conv_specular_taps(chain, 2048, 2048, n, RES0, t)
        tr, _ = R.conv_specular_taps(chain, 2048, 2048, n, r, rough, m, f, x, y)
        entry = {"texel": t, "mip": m, "face": int(f), "x": int(x), "y": int(y),
                 "channels": [{"channel": int(c), "oracle_half": int(got[t, c].view(np.uint16)), "reference_half": int(ref[t, c].view(np.uint16)), "ulps": int(d[t, c])}
                              for c in range(3) if d[t, c] > 1],
                 "taps_oracle": int(len(to)), "taps_reference": int(len(tr)), "taps_crossing_a_step_count": 0, "taps_crossing_a_step": []}
        if len(to) == len(tr):
            lod_o, lod_r = np.clip(to[:, 2], 0, n - 1), np.clip(tr[:, 2], 0, n - 1)
            lv = np.floor(np.minimum(lod_o, lod_r)).astype(np.int64)                # the finer of the two levels a tap blends
            W = np.maximum(2048 >> lv, 1).astype(np.float64)
            fu_o, fu_r = fixed8(to[:, 0].astype(np.float64) * W - 0.5), fixed8(tr[:, 0].astype(np.float64) * W - 0.5)
            fv_o, fv_r = fixed8(to[:, 1].astype(np.float64) * W - 0.5), fixed8(tr[:, 1].astype(np.float64) * W - 0.5)
            fl_o, fl_r = fixed8(lod_o), fixed8(lod_r)
            crossing = np.nonzero((fu_o != fu_r) | (fv_o != fv_r) | (fl_o != fl_r))[0]
            entry["taps_crossing_a_step_count"] = int(len(crossing))         # mip 0 (roughness 0): H = N for every sample, all 512 taps are the same fetch
            for k in crossing[:8]:
                entry["taps_crossing_a_step"].append({"tap": int(k), "uv_oracle": [float(to[k, 0]), float(to[k, 1])], "uv_reference": [float(tr[k, 0]), float(tr[k, 1])],
                                                      "lod_oracle": float(to[k, 2]), "lod_reference": float(tr[k, 2]),
                                                      "fraction_steps_u_v_lod": [int(fu_o[k] - fu_r[k]), int(fv_o[k] - fv_r[k]), int(fl_o[k] - fl_r[k])]})
        out["entries"].append(entry)
    path = os.path.join(ROOT, "tests", "golden", "specular_filterstep_tail.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, len(out["entries"]), "texels,", out["channels_above_one_ulp"], "channels; taps crossing a step per texel:", [e["taps_crossing_a_step_count"] for e in out["entries"]])


if __name__ == "__main__":
    main()
