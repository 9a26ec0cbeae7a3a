"""bindings/csharp/Ycge.cs held to the C-ABI by machine (no .NET toolchain in the image: the C# is checked as text).

Every `[StructLayout(LayoutKind.Sequential)] struct` is parsed, laid out by the C rules .NET applies to blittable sequential structs
(natural alignment, fixed buffers inline), and compared - field names, order, types, offsets, total size - with its ctypes twin in
yetanotherconsolegameengine_amd/abi.py, which tests/test_host_cpu.py holds to include/ycge.h through the built library.  Every
`[DllImport]` is compared with abi._PROTOTYPES: the set of names (every export has a declaration and nothing else has), the parameter
count and the kind of each parameter and of the return value.  Enums are compared with the header's values.
"""
from __future__ import annotations

import ctypes as C
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from yetanotherconsolegameengine_amd import abi  # noqa: E402

CS = ROOT / "bindings" / "csharp"
SRC = (CS / "Ycge.cs").read_text()

PRIM = {"int": (4, 4, C.c_int32), "uint": (4, 4, C.c_uint32), "float": (4, 4, C.c_float), "double": (8, 8, C.c_double),
        "long": (8, 8, C.c_int64), "ulong": (8, 8, C.c_uint64), "IntPtr": (8, 8, "ptr"), "UIntPtr": (8, 8, C.c_size_t)}
TWINS = {"YVec3": abi.Vec3, "YMaterial": abi.Material, "YTexture": abi.Texture, "YPrim": abi.Prim, "YMesh": abi.Mesh,
         "YVoxelLookup": abi.VoxelLookup, "YGrid": abi.Grid, "YLight": abi.Light, "YScene": abi.Scene, "YConfig": abi.Config,
         "YFrameStats": abi.FrameStats, "YFlightInfo": abi.FlightInfo}


def parse_structs(src):
    """{name: [(field, type, count)]} for every sequential struct; methods and constructors are skipped (they contain parentheses)."""
    out = {}
    for m in re.finditer(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public (?:unsafe )?struct (\w+)[^{]*\{", src):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth:
            depth += {"{": 1, "}": -1}.get(src[j], 0)
            j += 1
        body = re.sub(r"//[^\n]*", "", src[i:j - 1])
        fields = []
        for stmt in re.finditer(r"public\s+(fixed\s+)?(\w+)\s+([^;(){}]+);", body):
            fixed, typ, names = stmt.groups()
            for n in names.split(","):
                n = n.strip()
                arr = re.fullmatch(r"(\w+)\[(\d+)\]", n)
                if fixed:
                    assert arr, (name, n)
                    fields.append((arr.group(1), typ, int(arr.group(2))))
                else:
                    assert re.fullmatch(r"\w+", n), (name, n)
                    fields.append((n, typ, 1))
        out[name] = fields
    return out


STRUCTS = parse_structs(SRC)


def layout(name):
    """[(field, offset, size, kind)], total size, alignment - C rules"""
    off, align, rows = 0, 1, []
    for fname, typ, count in STRUCTS[name]:
        if typ in PRIM:
            size, al, kind = PRIM[typ]
        else:
            sub_rows, size, al = layout(typ)
            kind = typ
        off = (off + al - 1) // al * al
        rows.append((fname, off, size * count, kind, count))
        off += size * count
        align = max(align, al)
    return rows, (off + align - 1) // align * align, align


def ctype_kind(t):
    if isinstance(t, type) and issubclass(t, C.Array):
        return ctype_kind(t._type_), t._length_
    if isinstance(t, type) and issubclass(t, C._Pointer) or t in (C.c_void_p, C.c_char_p):
        return "ptr", 1
    if isinstance(t, type) and issubclass(t, C.Structure):
        return next(k for k, v in TWINS.items() if v is t), 1
    return t, 1


@pytest.mark.parametrize("cs_name", sorted(TWINS))
def test_struct_matches_its_c_twin(cs_name):
    twin = TWINS[cs_name]
    assert cs_name in STRUCTS, f"{cs_name} is not declared in Ycge.cs"
    rows, size, _ = layout(cs_name)
    assert size == C.sizeof(twin), (cs_name, size, C.sizeof(twin))
    assert len(rows) == len(twin._fields_), (cs_name, [r[0] for r in rows], [f[0] for f in twin._fields_])
    for (fname, off, fsize, kind, count), (cname, ctype) in zip(rows, twin._fields_):
        assert fname.lower() == cname.replace("_", ""), (cs_name, fname, cname)          # AtrousCPhi <-> atrous_c_phi
        assert off == getattr(twin, cname).offset, (cs_name, fname, off, getattr(twin, cname).offset)
        assert fsize == getattr(twin, cname).size, (cs_name, fname)
        ck, cn = ctype_kind(ctype)
        if isinstance(ck, tuple):
            ck, cn = ck[0], cn
        same = kind == ck or (kind not in ("ptr",) and not isinstance(kind, str) and not isinstance(ck, str) and C.sizeof(kind) == C.sizeof(ck) and kind._type_ == ck._type_)
        assert same and count == cn, (cs_name, fname, kind, ck, count, cn)


def test_every_struct_in_the_file_has_a_twin():
    assert set(STRUCTS) == set(TWINS)


# ---- DllImports
def cs_param_kind(p):
    p = re.sub(r"\[\w+\]\s*", "", p.strip())
    typ = p.rsplit(" ", 1)[0].strip()
    if typ.startswith(("ref ", "out ")) or typ.endswith(("*", "[]")) or typ in ("IntPtr", "StringBuilder"):
        return "ptr"
    return {"int": "i32", "float": "f32", "long": "i64", "ulong": "u64", "UIntPtr": "u64", "nuint": "u64"}[typ]


def c_kind(t):
    if t is None:
        return "void"
    if (isinstance(t, type) and issubclass(t, C._Pointer)) or t in (C.c_void_p, C.c_char_p):
        return "ptr"
    return {C.c_int: "i32", C.c_int32: "i32", C.c_float: "f32", C.c_int64: "i64", C.c_long: "i64", C.c_uint64: "u64", C.c_size_t: "u64", C.c_ulong: "u64"}[t]


IMPORTS = {m.group(2): (m.group(1), [p for p in m.group(3).split(",") if p.strip()])
           for m in re.finditer(r"\[DllImport\(Lib\)\]\s*public static extern (\w+) (ycge_\w+)\(([^)]*)\);", SRC)}


def test_every_export_has_a_dllimport_and_nothing_else_has():
    assert set(IMPORTS) == set(abi.EXPORTED_SYMBOLS), (sorted(set(abi.EXPORTED_SYMBOLS) - set(IMPORTS)), sorted(set(IMPORTS) - set(abi.EXPORTED_SYMBOLS)))
    assert 'private const string Lib = "ycge_hip";' in SRC          # -> libycge_hip.so by the runtime's probing rules


@pytest.mark.parametrize("name", sorted(abi.EXPORTED_SYMBOLS))
def test_dllimport_signature(name):
    ret, params = IMPORTS[name]
    cres, cargs = abi._PROTOTYPES[name]
    assert len(params) == len(cargs), (name, params, cargs)
    assert {"int": "i32", "void": "void", "IntPtr": "ptr", "UIntPtr": "u64"}[ret] == c_kind(cres), (name, ret, cres)
    for p, a in zip(params, cargs):
        assert cs_param_kind(p) == c_kind(a), (name, p, a)
    # a `ref` / `out` / pointer to one of the structs must name the struct the C side takes
    for p, a in zip(params, cargs):
        m = re.search(r"\b(Y[A-Z]\w+)\b", p)
        if m and isinstance(a, type) and issubclass(a, C._Pointer) and issubclass(a._type_, C.Structure):
            assert TWINS[m.group(1)] is a._type_, (name, p, a)


def test_enums_and_constants_match_the_header():
    header = (ROOT / "include" / "ycge.h").read_text()

    def c_enum(prefix):
        return {m.group(1): int(m.group(2)) for m in re.finditer(r"\b" + prefix + r"(\w+)\s*=\s*(-?\d+)", header)}

    def cs_enum(name):
        body = re.search(r"public enum " + name + r"\s*\{([^}]*)\}", SRC).group(1)
        return {k.strip(): int(v) for k, v in (e.split("=") for e in body.split(",") if e.strip())}

    def norm(d):
        return {k.replace("_", "").lower(): v for k, v in d.items()}

    status = c_enum("YCGE_ERR_") | {"OK": 0}
    assert norm(cs_enum("YStatus")) == norm(status)
    assert norm(cs_enum("YMaterialKind")) == norm(c_enum("YCGE_MAT_"))
    assert norm(cs_enum("YPrimType")) == norm(c_enum("YCGE_PRIM_"))
    assert norm(cs_enum("YBuffer")) == norm(c_enum("YCGE_BUF_"))
    assert norm(cs_enum("YAccel")) == norm(c_enum("YCGE_ACCEL_"))
    assert norm(cs_enum("YExchange")) == norm(c_enum("YCGE_EXCHANGE_"))
    assert int(re.search(r"#define YCGE_ABI_VERSION (\d+)", header).group(1)) == abi.YCGE_ABI_VERSION == int(re.search(r"public const int AbiVersion = (\d+);", SRC).group(1))
    assert int(re.search(r"#define YCGE_MAX_DEVICES (\d+)", header).group(1)) == int(re.search(r"public const int MaxDevices = (\d+);", SRC).group(1))


def test_wrapper_and_flattener_only_call_what_is_declared():
    """every Ycge.ycge_* call in the wrapper, the flattener and the dump tool names a declared import, with the declared number of arguments"""
    files = [CS / "HipRaytraceWrapper.cs", CS / "SceneFlattener.cs", ROOT / "tools" / "ReferenceDump" / "Program.cs", ROOT / "tools" / "ReferenceDump" / "SceneFile.cs"]
    seen = set()
    for f in files:
        text = re.sub(r"//[^\n]*", "", f.read_text())
        for m in re.finditer(r"Ycge\.(ycge_\w+)\(", text):
            name = m.group(1)
            assert name in IMPORTS, (f.name, name)
            i, depth, args, cur = m.end(), 1, [], ""
            while depth:
                ch = text[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                if ch == "," and depth == 1:
                    args.append(cur); cur = ""
                else:
                    cur += ch
                i += 1
            if cur.strip():
                args.append(cur)
            assert len(args) == len(IMPORTS[name][1]), (f.name, name, args)
            seen.add(name)
    # the seam's calls are all there: create, upload, camera, frame, resize, destroy, the scene updates, the page-locked SDR buffer
    assert {"ycge_config_default", "ycge_create", "ycge_scene_upload", "ycge_set_camera", "ycge_render_frame", "ycge_resize", "ycge_destroy",
            "ycge_scene_update_objects", "ycge_scene_update_lights", "ycge_scene_update_texture", "ycge_alloc_host_buffer", "ycge_free_host_buffer"} <= seen
    wrapper = (CS / "HipRaytraceWrapper.cs").read_text()
    assert "public partial class RaytraceEntity" in wrapper and ": IConsoleRenderer" in wrapper
    for member in ("void SetCamera(Vec3", "void SetFov(float", "void TryFlipAndBlit(Framebuffer", "void Resize(Framebuffer"):      # RaytraceEntity.cs:12-18
        assert member in wrapper, member
    assert "ycge_pin_host_buffer" not in wrapper and "GCHandle" not in wrapper          # the SDR frame lives in the library's page-locked memory


def test_flattener_passes_each_primitive_the_parameters_the_header_documents():
    """SceneFlattener.cs: every `Prim(YPrimType.X, material, specular, reflectivity, ref, p...)` call passes as many floats as include/ycge.h
    documents for that primitive (`p = cx,cy,cz,radius` ...), the overriding primitives (plane, disk, rects, box: Surfaces.cs:65-66 ...) pass
    their own Specular / Reflectivity and the others zero - the same records yetanotherconsolegameengine_amd/scene.py builds."""
    header = (ROOT / "include" / "ycge.h").read_text()
    want = {}
    for m in re.finditer(r"YCGE_PRIM_(\w+) = \d+,?\s*/\* p = (.*?)(?:\s{2,}|\*/)", header):
        n = 0
        for tok in m.group(2).split(","):
            n += 3 if "xyz" in tok else 1
        want[m.group(1).replace("_", "").lower()] = n
    assert want == {"sphere": 4, "plane": 6, "disk": 7, "xyrect": 5, "xzrect": 5, "yzrect": 5, "box": 6, "cylindery": 7, "triangle": 9}, want
    text = re.sub(r"//[^\n]*", "", (CS / "SceneFlattener.cs").read_text())
    seen = {}
    for m in re.finditer(r"Prim\(YPrimType\.(\w+),", text):
        i, depth, args, cur = m.end(), 1, [], ""
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur.strip()); cur = ""
            else:
                cur += ch
            i += 1
        args.append(cur.strip())
        seen[m.group(1).lower()] = args
    for name, n in want.items():
        material, spec, refl, ref, *p = seen[name]
        assert len(p) == n, (name, p)
        assert ref == "-1"
        overrides = name in ("plane", "disk", "xyrect", "xzrect", "yzrect", "box")
        assert (spec, refl) != ("0", "0") if overrides else (spec, refl) == ("0", "0"), (name, spec, refl)
    assert seen["mesh"][3] == "idx" and seen["volumegrid"][3] == "idx" and len(seen["mesh"]) == 4 and len(seen["volumegrid"]) == 4
    # and the Python mirror passes the same counts (it is what the GPU tests flatten scenes with)
    py = (ROOT / "yetanotherconsolegameengine_amd" / "scene.py").read_text()
    for name, n in want.items():
        cs_name = {"xyrect": "XYRECT", "xzrect": "XZRECT", "yzrect": "YZRECT", "cylindery": "CYLINDER_Y"}.get(name, name.upper())
        m = re.search(r"prim\(abi\.PRIM_" + cs_name + r", mat_id\([^)]*\), \[(.*?)\]", py)
        assert m, name
        cnt = sum(3 if tok.strip().startswith("*") else 1 for tok in m.group(1).split(","))
        assert cnt == n, (name, m.group(1))
