// fsr_con_ref.cpp — wrapper that compiles the REFERENCE's own CPU code for the FSR 1.0 constant blocks, from the sources
// where they lie (/root/reference/Shaders/AMDFidelityFX/FSR1.0/ffx_a.h + ffx_fsr1.h with A_CPU, exactly how the reference's
// Source/Engine/PostProcess/PostProcess.cpp:21-35 includes them), into oracle/_ref/libvqref_fsr.so.
// Test infrastructure: used by tests/golden/make_ref_fixtures.py (fixture generation, this container only) and by
// tests/test_ref_pinning.py to pin oracle/vqo_fsr.cpp's restatement. Nothing of the reference is copied into this repo;
// the .so is git-ignored and is never loaded by the product.
#define A_CPU 1
#include <cmath>
#include <cstdint>
#include "ffx_a.h"
#include "ffx_fsr1.h"

extern "C" {
// FFSR1_EASU::UpdateEASUConstantBlock, PostProcess.cpp:47-75: 16 dwords = con0..con3
void vqref_fsr_easu_con(uint32_t* con, float inVpX, float inVpY, float inSzX, float inSzY, float outX, float outY) {
    FsrEasuCon(con, con + 4, con + 8, con + 12, inVpX, inVpY, inSzX, inSzY, outX, outY);
}
// FFSR1_RCAS::UpdateRCASConstantBlock, PostProcess.cpp:39-45
void vqref_fsr_rcas_con(uint32_t* con, float sharpnessStops) { FsrRcasCon(con, sharpnessStops); }
// the CPU-side float -> half packing FsrRcasCon uses (ffx_a.h:482-550)
uint32_t vqref_half_bits(float f) { return AU1_AH1_AF1(f); }
}
