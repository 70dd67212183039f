"""Figures of the machine and of the workload that every part of the bench line prices against."""
import os

from vqengine_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md:35 (spec); 6290 measured copy ceiling
VALU_PEAK_TFLOPS = 157.3        # :40
SHADE_BYTES_PER_PX = 64 + 8     # 4 float4 G-buffer planes in + RGBA16F out
VALU_ISSUE_CEILING_TLIS = 66.7  # T lane-instructions/s: the FAST issue rate of the chip at steady-state clocks — plain fp32 add / mul / fma on registers, or v_pk_fma_f32 counted as two
                                # (scripts/ubench/valu_ceiling.hip is the packed form; scripts/ubench/mix_rate.hip, profiles/r5f_issue_classes.md: instructions with an SGPR source,
                                # conversions, compares, min / max and integer operations issue at 35 T, v_rcp / v_rsq at 17 T)
XGMI_LINK_GBPS = 153.0          # per direct GPU-GPU link, peak; the tile-curve model also quotes half of it
F16, R8 = abi.FMT_RGBA16F, abi.FMT_RGBA8_UNORM
