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


def _c_struct_fields(name):
    """[(field, array_len or 1)] of `typedef struct name {...}` in the header, declaration order."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct %s \{(.*?)\}" % name, src, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, rest = decl.split(None, 1)
        for item in rest.split(","):
            m = re.match(r"\s*(\w+)(?:\[(\d+)\])?\s*$", item)
            out.append((m.group(1), int(m.group(2) or 1), ctype))
    return out


def _cs_struct_fields(src, name):
    body = re.search(r"struct %s\s*\{(.*?)\n    \}" % name, src, flags=re.S).group(1)
    body = re.sub(r"//.*", "", body)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"public\s+(fixed\s+)?(\w+)\s+(.*)$", decl, flags=re.S)
        for item in m.group(3).split(","):
            mm = re.match(r"\s*(\w+)(?:\[(\d+)\])?\s*$", item)
            out.append((mm.group(1), int(mm.group(2) or 1), m.group(2)))
    return out


def test_csharp_binding_matches_header():
    """csharp/SphNative.cs cannot be compiled here (no .NET): check it textually against the header instead --
    every DllImport names a declared entry point, every non-slab entry point is bound, and the blittable structs
    have the header's fields in the header's order with the same array lengths and 4-byte scalar types."""
    src = open(os.path.join(ROOT, "csharp", "SphNative.cs")).read()
    bound = re.findall(r"public static extern \w+ (sph_[a-z_]+)\(", src)
    declared = declared_functions()
    assert len(bound) == len(set(bound))
    assert set(bound) <= set(declared), set(bound) - set(declared)
    unbound = {d for d in declared if d not in bound}
    assert all(d.startswith("sph_slab_") or d in ("sph_device_pointer", "sph_get_stream") for d in unbound), unbound
    ctype_of = {"uint32_t": "uint", "int32_t": "int", "float": "float"}
    for c_name, cs_name in (("sph_params", "SphParams"), ("sph_phase", "SphPhase")):
        want = [(f, n, ctype_of[t]) for f, n, t in _c_struct_fields(c_name)]
        got = _cs_struct_fields(src, cs_name)
        norm = lambda s: s.replace("_", "").lower()
        assert [(norm(f), n, t) for f, n, t in got] == [(norm(f), n, t) for f, n, t in want], cs_name
    assert len(_cs_struct_fields(src, "SphParticle")) == abi.PARTICLE_FLOATS
    t = _cs_struct_fields(src, "SphTimings")
    assert [x[2] for x in t] == ["float"] * 8 + ["int"] and len(_c_struct_fields("sph_timings")) == 9
    kern = re.search(r"enum SphKernel\s*\{(.*?)\}", src, flags=re.S).group(1)
    ids = dict((k.strip(), int(v)) for k, v in (kv.split("=") for kv in kern.replace("\n", " ").split(",")))
    assert [k for k, _ in sorted(ids.items(), key=lambda kv: kv[1])] == [n[3:] for n in abi.KERNEL_NAMES]


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/sph_hip.h must compile as C99 (no C++ or HIP types), and a C program that
    references every declared entry point must link against libsph_hip.so with gcc alone."""
    import subprocess
    header = os.path.join(ROOT, "include", "sph_hip.h")
    names = declared_functions()
    src = tmp_path / "uses_abi.c"
    src.write_text('#include "sph_hip.h"\n#include <stdio.h>\n'
                   "typedef void (*fn_t)(void);\nint main(void) {\n    fn_t fns[] = {" + ", ".join(f"(fn_t){n}" for n in names) + "};\n"
                   "    sph_params p; p.struct_size = (uint32_t)sizeof p;\n"
                   '    printf("%u %u %d %s\\n", (unsigned)(sizeof fns / sizeof fns[0]), p.struct_size, (int)sizeof(sph_slab_record), sph_version());\n'
                   "    return 0;\n}\n")
    exe = tmp_path / "uses_abi"
    lib_dir = os.path.join(ROOT, "sph_amd")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.dirname(header), str(src), "-o", str(exe),
                        "-L", lib_dir, "-l:libsph_hip.so", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    out = subprocess.run([str(exe)], capture_output=True, text=True)          # sph_version needs no GPU
    assert out.returncode == 0, out.stderr
    n, size, rec, _ = out.stdout.split(None, 3)
    assert int(n) == len(names) and int(size) == C.sizeof(abi.SphParams) and int(rec) == abi.SLAB_RECORD_BYTES
