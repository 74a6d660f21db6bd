"""The C-ABI library loads and exports every symbol include/sph_hip.h declares; the ctypes
mirror of sph_params matches the C layout.  No compute calls (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from sph_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sph_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z_]+)\s*\(", src)))


def test_header_and_binding_list_the_same_entry_points():
    assert declared_functions() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = abi.load_library()
    for name in declared_functions():
        assert hasattr(lib, name), f"libsph_hip.so does not export {name}"
    assert b"gfx950" in lib.sph_version()


def test_struct_layouts_match_the_c_header(tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sph_hip.h"\n'
                    'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sph_params),'
                    'offsetof(sph_params, obstaclePos), offsetof(sph_params, gridResolutionX),'
                    'offsetof(sph_params, boundaryMatrix), offsetof(sph_params, boundaryHalfExtents),'
                    'sizeof(sph_phase), sizeof(sph_timings));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    P = abi.SphParams
    want = [C.sizeof(P), P.obstaclePos.offset, P.gridResolutionX.offset, P.boundaryMatrix.offset,
            P.boundaryHalfExtents.offset, C.sizeof(abi.SphPhase), C.sizeof(abi.SphTimings)]
    assert got == want


def test_kernel_ids_follow_pragma_kernel_order():
    assert abi.KERNEL_NAMES == ["CS_Clear", "CS_ClearGrid", "CS_BuildGrid", "CS_DensityPressure", "CS_ForceXSPH",
                                "CS_VvHalfStep", "CS_VvFullStep", "CS_BoundObs", "CS_VelocityDamping"]
    assert (abi.K_CLEAR, abi.K_BUILD_GRID, abi.K_VELOCITY_DAMPING) == (0, 2, 8)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(abi, "_LIB", None)
    monkeypatch.setenv("SPH_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        abi.load_library()


def test_no_product_code_touches_the_oracle():
    """The oracle is test infrastructure: nothing under sph_amd/ or host/ may mention it."""
    bad = []
    for top in ("sph_amd", "host"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\bimport\s+oracle|from\s+oracle|liboracle|ora_[a-z]+\(", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
