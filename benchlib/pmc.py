"""Counter-derived constants (profiles/pmc_constants.json): rocprofv3 PMC passes cannot run inside bench.py, so the line reads what scripts/pmc_refresh.sh measured and
refuses it when the kernel sources changed since."""
import hashlib
import json
import os

from .consts import ROOT

PMC_FILE = os.path.join(ROOT, "profiles", "pmc_constants.json")
PMC_SOURCES = ["vqengine_amd/csrc/shade.hip", "vqengine_amd/csrc/vq_shade.h", "vqengine_amd/csrc/vq_devmath.h", "vqengine_amd/csrc/vq_sampling.h", "vqengine_amd/csrc/Makefile"]


def kernel_source_hash():
    """sha256 over the shade kernel's sources and over the lines of the Makefile that decide its code generation (the HIPFLAGS assignments and their
    continuation lines) — a change of prerequisites or comments in the Makefile does not make the counters stale, a change of flags does."""
    h = hashlib.sha256()
    for f in PMC_SOURCES:
        data = open(os.path.join(ROOT, f), "rb").read()
        if f.endswith("Makefile"):
            keep, cont = [], False
            for ln in data.splitlines():
                if cont or (b"HIPFLAGS" in ln and not ln.lstrip().startswith(b"#") and not ln.startswith(b"\t")):
                    keep.append(ln.strip())
                    cont = ln.rstrip().endswith(b"\\")
                else:
                    cont = False
            data = b"\n".join(keep)
        h.update(data)
    return h.hexdigest()[:16]


def load_pmc_constants(config, fresnel_pow):
    """Counter-derived constants of the shade kernel (HBM bytes per launch, VALU instructions per wave) cannot be measured from inside
    bench.py: they are read from profiles/pmc_constants.json, which records the sha256 of the kernel sources they were measured on
    (scripts/pmc_refresh.sh). If the sources changed since, the constants are NOT used: the fields they feed are null and `stale` is set."""
    try:
        d = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None, {"stale": True, "why": "profiles/pmc_constants.json missing"}
    entry = d.get(f"{config}/{fresnel_pow}")
    meta = {"file": "profiles/pmc_constants.json", "kernel_sources_sha256": d.get("kernel_sources_sha256"), "measured_at_commit": d.get("measured_at_commit"),
            "profile": d.get("profile")}
    if entry is None:
        return None, dict(meta, stale=True, why=f"no entry for {config}/{fresnel_pow}")
    if d.get("kernel_sources_sha256") != kernel_source_hash():
        return None, dict(meta, stale=True, why="shade.hip / vq_shade.h / vq_devmath.h / vq_sampling.h / Makefile changed since the counters were collected", now=kernel_source_hash())
    return entry, dict(meta, stale=False)


def load_time_kernel_counters():
    """SQ_INSTS_VALU / TCP accesses per launch of the cfg4 load-time kernels (profiles/pmc_constants.json `load_time_kernels`), or None when conv.hip and its headers
    changed since they were counted: lets the line price those kernels against the ISSUE ceiling instead of a flop model that ignores what a tap really costs."""
    try:
        d = json.load(open(PMC_FILE)).get("load_time_kernels")
    except (OSError, ValueError):
        return None
    if not d:
        return None
    h = hashlib.sha256()
    for f in d["sources"]:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return d if h.hexdigest()[:16] == d["sources_sha256"] else None
