"""include/fdt.h is a C header and libfdt_hip.so a plain C-ABI library: a C99 translation unit that includes the
header compiles with -Wall -Werror -pedantic, links against the library with no Python or torch in the process,
and gets error codes (not crashes) for bad arguments.  No GPU is needed for what it calls."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "face-detection-and-tracking_amd", "csrc")

C_SRC = r"""
#include <stdio.h>
#include <string.h>
#include "fdt.h"

int main(void) {
  float priors[4 * 4];
  int rc, n = -1;
  printf("version %d\n", fdt_version());
  /* argument errors are reported through the return code + fdt_last_error(), never by crashing */
  rc = fdt_pairwise_iou(NULL, 3, NULL, 3, 7, NULL);
  if (rc != FDT_ERR_ARG) { printf("unexpected rc %d\n", rc); return 1; }
  if (strlen(fdt_last_error()) == 0) { printf("empty error text\n"); return 2; }
  if (fdt_model_create(99, 0) != NULL) { printf("bad arch accepted\n"); return 3; }
  if (fdt_tracker_create(0.4, 0.6, 5, 100000, 64) != NULL) { printf("oversized tracker accepted\n"); return 4; }
  rc = fdt_device_count(&n);          /* OK with a GPU, FDT_ERR_HIP without one: both are fine here */
  printf("device_count rc %d n %d\n", rc, n);
  (void)priors;
  printf("ok\n");
  return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_is_c99_and_library_links_from_plain_c(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libfdt_hip.so")):
        pytest.skip("libfdt_hip.so not built")
    src = tmp_path / "abi.c"
    exe = tmp_path / "abi"
    src.write_text(C_SRC)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", INC, str(src), "-o", str(exe),
                    "-L", LIBDIR, "-lfdt_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout
