"""ISA lint of the hand-issued LDS reads in the 8-wave Winograd kernels (csrc/conv_wino.h) and the direct kernel
(csrc/conv_kernel.h).

The main loop issues its ds_reads through inline asm and waits with exact `s_waitcnt lgkmcnt(N)` counts, which the
compiler cannot see: a register copy or spill scheduled between a read and its wait would silently use stale
data.  This compiles the kernel for gfx950 (device code only, no GPU needed) and runs tools/check_async_lds.py
over the assembly: no instruction may touch a register of a still-outstanding ds_read."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "face-detection-and-tracking_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

SRC = """#include "conv_wino.h"
#include "conv_wino44.h"
#include "conv_n8.h"
namespace fdt { namespace {
template __global__ void conv_wino2_kernel<W_64x64W>(const ConvArgs);       // 3x3 pad 1, 8x32-pixel tile
template __global__ void conv_wino2_kernel<W_128x32R3>(const ConvArgs);     // 32-channel tile (detection heads)
template __global__ void conv_wino2_kernel<WD2_64x64W>(const ConvArgs);     // dilation 2
template __global__ void conv_wino4_kernel<W_64x64W>(const ConvArgs);       // quarter-split form
template __global__ void conv_wino4_kernel<WD2_64x64R3>(const ConvArgs);
// Winograd F(4x4,3x3) (conv_wino44.h): window reads, operand prefetch and V writes all hand-issued, seven role instances
template __global__ void conv_wino44_kernel<W44>(const ConvArgs);
template __global__ void conv_wino44_kernel<W44odd>(const ConvArgs);
template __global__ void conv_wino44_kernel<W44D2>(const ConvArgs);          // dilation 2: dword-pair window reads
template __global__ void conv_wino44b_kernel<W44>(const ConvArgs);           // twelve-wave form
// the direct kernel's pipelined main loop (conv_kernel.h): 1 + 1, 2 + 1 and 2 + 2 operand registers per step
template __global__ void conv_kernel<G_1x1_S1_K32, T_64x64>(const ConvArgs);
template __global__ void conv_kernel<G_1x1_S1, T_128x64W>(const ConvArgs);
template __global__ void conv_kernel<G_3x3_S2, T_128x128>(const ConvArgs);
template __global__ void conv_kernel<G_7x7_S2, T_128x64>(const ConvArgs);
// the vector-ALU head kernel (conv_n8.h): 12 window reads + weights one tap ahead; NO scratch traffic may appear in it
template __global__ void conv_n8_kernel<true>(const ConvArgs);
} }
"""


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_async_lds_reads_have_no_hazards(tmp_path):
    src = tmp_path / "wino_lint.hip"
    asm = tmp_path / "wino_lint.s"
    src.write_text(SRC)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                    "-I", CSRC, "-o", str(asm), str(src)], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    text = asm.read_text()
    # the hand-counted waits are really there (7 = 3 window rows + 4 weight pairs, 10 for the dilated window)
    assert "s_waitcnt lgkmcnt(7)" in text and "s_waitcnt lgkmcnt(10)" in text
    assert "s_waitcnt lgkmcnt(6)" in text and "s_waitcnt lgkmcnt(8)" in text      # quarter-split: 2 / 4 window reads + 4
    assert text.count("ds_read2_b64") > 0 and text.count("ds_read2st64_b32") > 0
    assert "s_waitcnt lgkmcnt(2)" in text and "s_waitcnt lgkmcnt(3)" in text and "s_waitcnt lgkmcnt(4)" in text   # direct kernel
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_lds.py"), str(asm)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert ": 0 hazards" in r.stdout
    # conv_wino44_kernel: 9 accumulator tiles + two operand sets + the window rows must fit two waves per SIMD without spills
    for nm in ("conv_wino44_kernelINS0_4W44TILb1ELi1EEEEEvNS_8ConvArgsE", "conv_wino44_kernelINS0_4W44TILb0ELi1EEEEEvNS_8ConvArgsE",
               "conv_wino44_kernelINS0_4W44TILb1ELi2EEEEEvNS_8ConvArgsE"):
        meta44 = text.split(".name:           _ZN3fdt12_GLOBAL__N_118" + nm, 1)[1][:1500]
        assert ".vgpr_spill_count: 0" in meta44 and int(meta44.split(".vgpr_count:")[1].split()[0]) <= 256
    # ... and its main loops carry no vector-ALU address arithmetic (the matrix pipe and the VALU do not co-execute on this chip):
    # every LDS-DMA of the Win % 4 == 0 forms is a buffer load with a scalar offset, never a global load behind 64-bit VALU adds
    body44 = text.split("conv_wino44_kernelINS0_4W44TILb1ELi1EEEEEvNS_8ConvArgsE:", 1)[1].split("s_endpgm", 1)[0]
    assert body44.count("buffer_load_dwordx4") >= 21 and "global_load_lds" not in body44
    # conv_n8_kernel<true>: a spill reloaded inside its loop would carry a vmcnt(0) that serialises the LDS-DMA pipeline
    # (measured: no load/compute overlap at all) -- the fast variant must stay spill-free at four waves per SIMD
    body = text.split("conv_n8_kernelILb1EEEvNS_8ConvArgsE:", 1)[1].split("s_endpgm", 1)[0]
    assert "scratch_" not in body and body.count("v_pk_fma_f32") == 288
    meta = text.split(".name:           _ZN3fdt12_GLOBAL__N_114conv_n8_kernelILb1EEEvNS_8ConvArgsE", 1)[1][:1200]
    assert ".vgpr_spill_count: 0" in meta and int(meta.split(".vgpr_count:")[1].split()[0]) <= 128


def test_lint_flags_a_use_before_the_wait(tmp_path):
    """The checker itself: a copy of a register whose ds_read is still outstanding must be reported."""
    bad = tmp_path / "bad.s"
    bad.write_text("_ZN3fdt17conv_wino2_kernelIfake:\n"
                   "\tds_read2_b64 v[10:13], v2 offset0:1 offset1:2\n"
                   "\tds_read2_b64 v[14:17], v2 offset0:3 offset1:4\n"
                   "\tv_mov_b32_e32 v20, v11\n"
                   "\ts_waitcnt lgkmcnt(1)\n"
                   "\tv_mov_b32_e32 v21, v12\n"          # fine: only the newest read is still in flight
                   "\tv_mov_b32_e32 v22, v15\n"          # hazard: v[14:17] not waited for
                   "\ts_endpgm\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_lds.py"), str(bad)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and ": 2 hazards" in r.stdout, r.stdout
