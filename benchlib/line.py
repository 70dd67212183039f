"""Key order of the one JSON line and its flat `digest`."""
from .consts import HBM_PEAK_GBPS, SHADE_BYTES_PER_PX


def finish_line(out):
    """Key order of the one JSON line. The driver's record keeps `roofline`, `cpu_baseline` and `config` whole plus the LAST ~2 000 characters of the line: the long objects go
    first, and the line ends with the second-tier figures a reader needs to recompute the fractions — `valu_issue`, `cfg5_strong`, `stages` without their
    prose (the notes move to `notes`), then `digest`, a flat summary."""
    tail_keys = ("valu_issue", "cfg5_strong", "stages")
    notes = {}

    def strip(obj, path):
        if not isinstance(obj, dict):
            return obj
        res = {}
        for k, v in obj.items():
            if isinstance(v, str) and (k.endswith("note") or k in ("workload", "post_chain_co_runs_with")):
                notes[path + "." + k] = v
            else:
                res[k] = strip(v, path + "." + k)
        return res
    ordered = {k: v for k, v in out.items() if k not in tail_keys}
    tail = {k: strip(out[k], k) for k in tail_keys if k in out}
    ordered["notes"] = notes
    ordered.update(tail)
    g = lambda *ks: _dig(out, ks)                            # noqa: E731
    ordered["digest"] = {
        "shade_ms": g("roofline", "ms"), "shade_hbm_frac": g("roofline", "frac"), "shade_valu_frac_spec": g("valu", "frac"), "shade_valu_issue_frac": g("valu_issue", "frac"),
        "post_chain_alone_ms": g("stages", "isolated", "post_chain_ms"), "post_chain_hbm_frac_at_12_B_px": g("stages", "isolated", "post_chain_frac_of_hbm_peak"),
        "cfg2_shade_ms": g("cfg2", "shade_ms"), "cfg2_hbm_frac": g("cfg2", "hbm_frac"), "cfg2_valu_issue_frac": g("cfg2", "valu_issue", "frac"),
        "cfg1_shade_ms": g("cfg1", "shade_ms"), "cfg1_Mpix_s": g("cfg1", "shade_Mpix_s"), "cfg1_cpu_Mpix_s": g("cfg1", "cpu_baseline", "value"),
        "engine_max_shade_ms": g("engine_max", "shade_ms"), "engine_max_coherent_ms": g("engine_max", "coherent_content", "shade_ms"),
        "cfg5_strong_Mpix_s": g("cfg5_strong", "value"), "cfg5_strong_shade_ms": g("cfg5_strong", "shade_ms"),
        "conv_diffuse_ms": g("ibl_load", "conv_diffuse_ms"), "conv_specular_ms": g("ibl_load", "conv_specular_ms"), "brdf_lut_ms": g("ibl_load", "brdf_lut_warm_ms"),
        "engine_default_prefilter_ms": g("ibl_load", "engine_default", "prefilter_ms"), "engine_default_specular_ms": g("ibl_load", "engine_default", "conv_specular_ms"),
        "psmain_fused_ms": g("widened", "psmain_fused", "ms"), "sustained_Mpix_s": g("sustained", "value"), "cpu_baseline_Mpix_s": g("cpu_baseline", "value"),
        "bytes_per_px": {"shade": SHADE_BYTES_PER_PX, "post_chain": 12}, "hbm_peak_GBps": HBM_PEAK_GBPS, "px_4k": 3840 * 2160}
    return ordered


def _dig(d, keys):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d
