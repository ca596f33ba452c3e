"""The production tile kernel keeps its pipeline's load targets in FIXED registers and waits for them by hand-counted `vmcnt`:
nothing the compiler emits may read or write such a register between its load and its wait (a compiler or flag change could make
it do so silently).  `profiles/lint_inflight.py` checks the generated gfx950 code of every k_tile_render instantiation; this test
runs it on the code hipcc generates from the tree's sources (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svgrasterize.py_amd", "csrc")


def test_no_hand_issued_load_target_is_touched_before_its_wait(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "svgr_hip_device.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-inline-asm", "-S",
                           "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "svgr_hip.hip")], stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "lint_inflight.py"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "lint_inflight: ok" in r.stdout and "k_tile_render" in r.stdout, r.stdout[-1000:]
