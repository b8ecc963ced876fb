"""device_utils.cuh's LDS-free xor-lane exchanges (DPP quad_perm / row_shl + row_shr / row_ror, gfx950 v_permlane16/32_swap) against
__shfl_xor on one wave: every partner form, the combined sum / max forms and group_sum, bit for bit.  The check itself is a small
HIP program (tools/micro/lane_xor_check.hip) compiled here with hipcc: the helpers are device functions without a C-ABI entry of
their own, and the compiler defect they work around (the second result of the row-swap builtins, see lane_row_swap) only shows in
the values on a GPU."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_exchanges_match_shfl_xor(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(str(tmp_path), "lane_xor_check")
    b = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-I", os.path.join(ROOT, "llm-inference-engine_amd", "csrc"),
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "micro", "lane_xor_check.hip"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "MISMATCH" not in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]
    assert r.stdout.count(" ok ") >= 13, r.stdout
