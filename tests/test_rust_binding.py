"""The Rust side of the boundary exists as source only: no Rust toolchain in the build container.
What can be checked is checked here: integration/raytracer-hip-sys/src/lib.rs must mirror include/rt_hip.h field for
field - every struct (names, order and types of the fields), every constant, every extern "C" signature - and the
patch against the reference must name files and lines that exist in the reference snapshot it was made from."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_hip.h")
LIB_RS = os.path.join(ROOT, "integration", "raytracer-hip-sys", "src", "lib.rs")
PATCH = os.path.join(ROOT, "integration", "reference-hip.patch")

SCALARS = {"float": "f32", "double": "f64", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64",
           "uint8_t": "u8", "int": "c_int", "char": "c_char", "void": "c_void"}


def c_type_to_rust(c):
    """'const RtMeshTri*' -> '*const RtMeshTri', 'void*' -> '*mut c_void', 'uint32_t' -> 'u32', 'RtScene* const*' -> '*const *mut RtScene'"""
    c = " ".join(c.replace("*", " * ").split())
    if c.endswith("*"):
        inner = c[:-1].strip()
        if inner.endswith("const"):                      # T* const*: pointer to a const pointer
            return "*const " + c_type_to_rust(inner[:-len("const")])
        if inner.startswith("const ") and "*" not in inner:
            return "*const " + c_type_to_rust(inner[len("const "):])
        return "*mut " + c_type_to_rust(inner)
    assert not c.startswith("const "), c
    return SCALARS.get(c, c)


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse_header():
    text = strip_c_comments(open(HEADER).read())
    structs, consts, funcs = {}, {}, {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            names = [n.strip() for n in decl.split(",")]
            first = names[0]
            ctype, name0 = re.match(r"(.*?[\s\*])(\w+)$", first).groups()
            ctype = ctype.strip()
            for n in [name0] + names[1:]:
                fields.append((n, c_type_to_rust(ctype)))
        structs[m.group(1)] = fields
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    for m in re.finditer(r"#define\s+(RT_\w+)\s+\(?(-?\d+)\)?\s*$", body, flags=re.M):
        consts[m.group(1)] = int(m.group(2))
    for m in re.finditer(r"enum\s*\{(.*?)\}\s*;", body, flags=re.S):
        for item in m.group(1).split(","):
            item = item.strip()
            if item:
                name, value = [v.strip() for v in item.split("=")]
                consts[name] = int(value)
    flat = " ".join(re.sub(r"enum\s*\{.*?\}\s*;", " ", body, flags=re.S).split())
    for m in re.finditer(r"((?:const\s+)?\w+\s*\*?)\s*(rt_\w+)\s*\(([^)]*)\)\s*;", flat):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                ctype, pname = re.match(r"(.*?[\s\*])(\w+)$", a.strip()).groups()
                params.append((pname, c_type_to_rust(ctype)))
        funcs[name] = (None if ret == "void" else c_type_to_rust(ret), params)
    return structs, consts, funcs


def parse_lib_rs():
    text = re.sub(r"//[^\n]*", "", open(LIB_RS).read())
    structs, consts, funcs = {}, {}, {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = [(f.group(1), " ".join(f.group(2).split())) for f in re.finditer(r"pub (\w+):\s*([^,\n]+),", m.group(2))]
        structs[m.group(1)] = fields
    for m in re.finditer(r"pub const (RT_\w+):\s*\w+\s*=\s*(-?\d+);", text):
        consts[m.group(1)] = int(m.group(2))
    ext = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (rt_\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", ext):
        params = [(p.split(":")[0].strip(), " ".join(p.split(":", 1)[1].split())) for p in m.group(2).split(",") if p.strip()]
        funcs[m.group(1)] = (m.group(3).strip() if m.group(3) else None, params)
    return structs, consts, funcs


def test_structs_match_field_for_field():
    c_structs, _, _ = parse_header()
    r_structs, _, _ = parse_lib_rs()
    assert len(c_structs) >= 17
    for name, fields in c_structs.items():
        assert name in r_structs, f"{name} missing in lib.rs"
        assert r_structs[name] == fields, f"{name}: header {fields} != lib.rs {r_structs[name]}"
    assert set(r_structs) - set(c_structs) == {"RtScene"}          # the opaque handle: a typedef of an incomplete struct in C


def test_constants_match():
    _, c_consts, _ = parse_header()
    _, r_consts, _ = parse_lib_rs()
    assert len(c_consts) >= 30
    assert c_consts == r_consts


def test_extern_signatures_match():
    _, _, c_funcs = parse_header()
    _, _, r_funcs = parse_lib_rs()
    assert len(c_funcs) >= 20
    assert set(c_funcs) == set(r_funcs), set(c_funcs) ^ set(r_funcs)
    for name, (ret, params) in c_funcs.items():
        r_ret, r_params = r_funcs[name]
        assert r_ret == ret, f"{name}: return {ret} vs {r_ret}"
        assert [t for _, t in r_params] == [t for _, t in params], f"{name}: {params} vs {r_params}"
        assert [n for n, _ in r_params] == [n for n, _ in params], f"{name}: parameter names {params} vs {r_params}"


def test_header_symbols_are_exported_by_the_library():
    """... and the library the crate links really exports every one of them."""
    import ctypes
    from raytracer_amd import _ffi
    _, _, c_funcs = parse_header()
    lib = ctypes.CDLL(os.path.join(_ffi.REPO_ROOT, "raytracer_amd", "librt_hip.so")) if False else _ffi.load_hip()
    for name in c_funcs:
        assert hasattr(lib, name), name


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference snapshot only exists in the build container")
def test_patch_applies_to_the_reference_snapshot():
    """`patch --dry-run` against a scratch copy of the files the patch names (nothing is written to /root/reference)."""
    import shutil
    import subprocess
    import tempfile
    text = open(PATCH).read()
    files = sorted(set(re.findall(r"^--- a/(\S+)", text, flags=re.M)))
    assert "src/render.rs" in files and "raytracer-impl/src/implementation.rs" in files
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            src = os.path.join("/root/reference", f)
            if os.path.exists(src):
                os.makedirs(os.path.dirname(os.path.join(tmp, f)), exist_ok=True)
                shutil.copy(src, os.path.join(tmp, f))
        r = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tmp, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
