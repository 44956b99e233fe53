"""CPU: the C# binding a maintainer would add held mechanically to include/rtow.h and abi.py.

No dotnet / Unity toolchain exists here, so the binding cannot be compiled; what can drift silently is checked instead: every export has exactly one
`DllImport` with the header's argument count, no entry point names a symbol the header lacks, every `[StructLayout(LayoutKind.Sequential)]` struct has the
size and field offsets of its ctypes mirror (which tests/test_abi.py holds to the C compiler's layout), the enum values are the header's, and the version
the binding says it checks is RTOW_API_VERSION.  Conventions follow the reference's own plugin binding (ThirdParty/nVidia OptiX Denoiser/OptixApi.cs:24-31,172-251)."""
import ctypes as C
import importlib
import os
import re

rt = importlib.import_module("raytracing-in-one-weekend_amd")
abi = rt.abi
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _binding_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    start = text.index("## 1. Binding")
    block = text[text.index("```csharp", start) + len("```csharp"):]
    return block[:block.index("```")]


def _header():
    return open(os.path.join(ROOT, "include", "rtow.h"), encoding="utf-8").read()


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def _split_args(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    return [a.strip() for a in arglist.split(",")]


def header_functions():
    src = _strip_comments(_header())
    out = {}
    for m in re.finditer(r"RTOW_API\s+[\w\s\*]+?\b(rtow\w+)\s*\(([^)]*)\)\s*;", src):
        out[m.group(1)] = _split_args(m.group(2))
    return out


def binding_functions():
    src = _strip_comments(_binding_source())
    out = {}
    for m in re.finditer(r'\[DllImport\(LibraryFilename,\s*EntryPoint\s*=\s*"(\w+)"[^\]]*\)\]\s*(?:public\s+)?static\s+extern\s+(?:unsafe\s+)?[\w\.]+\s+\w+\s*\(([^)]*)\)\s*;', src):
        assert m.group(1) not in out, "two DllImports for " + m.group(1)
        out[m.group(1)] = _split_args(m.group(2))
    return out


def test_every_export_is_bound_once_with_the_headers_argument_count():
    hdr, cs = header_functions(), binding_functions()
    assert sorted(hdr) == sorted(abi.EXPORTED_SYMBOLS), "abi.EXPORTED_SYMBOLS and include/rtow.h disagree"
    assert sorted(cs) == sorted(hdr), {"unbound": sorted(set(hdr) - set(cs)), "unknown entry points": sorted(set(cs) - set(hdr))}
    for name, args in hdr.items():
        assert len(cs[name]) == len(args), (name, cs[name], args)


def test_pointer_arguments_stay_pointers():
    """Argument by argument: where the header takes a pointer (or the opaque handle / a stream), the binding passes a pointer-sized thing (T*, ref / out T, IntPtr,
    string, RtowContext); where it takes a 32-bit scalar, a 32-bit scalar."""
    hdr, cs = header_functions(), binding_functions()
    for name, args in hdr.items():
        for c_arg, cs_arg in zip(args, cs[name]):
            c_pointer = "*" in c_arg or re.match(r"(const\s+)?RtowContext\b", c_arg) is not None
            cs_type = cs_arg.rsplit(" ", 1)[0]
            cs_pointer = "*" in cs_type or cs_type.startswith(("ref ", "out ")) or cs_type in ("IntPtr", "UIntPtr", "string", "RtowContext")
            if re.match(r"(const\s+)?size_t\b", c_arg):
                assert cs_type == "UIntPtr", (name, c_arg, cs_arg)
            elif c_pointer:
                assert cs_pointer and cs_type != "UIntPtr", (name, c_arg, cs_arg)
            else:
                assert not cs_pointer, (name, c_arg, cs_arg)
                assert cs_type in ("int", "uint", "float", "RtowGatherMask", "RtowMemcpyKind", "RtowResult"), (name, c_arg, cs_arg)


# ---- struct layouts: C# sequential layout with the types the binding uses ----
SCALARS = {"int": (4, 4), "uint": (4, 4), "float": (4, 4), "long": (8, 8), "ulong": (8, 8), "IntPtr": (8, 8), "RtowLogCallback": (8, 8), "RtowContextFlags": (4, 4),
           "float2": (8, 4), "float3": (12, 4), "float4": (16, 4), "int2": (8, 4), "uint2": (8, 4)}       # Unity.Mathematics vectors: fields of 4-byte scalars
MIRRORS = {"RtowTexture": "Texture", "RtowImage": "Image", "RtowMaterial": "Material", "RtowEntity": "Entity", "RtowSceneDesc": "SceneDesc", "RtowView": "View",
           "RtowEnvironment": "Environment", "RtowBlueNoiseDesc": "BlueNoiseDesc", "RtowStbNoiseDesc": "StbNoiseDesc", "RtowCubemapDesc": "CubemapDesc",
           "RtowSampleParams": "SampleParams", "RtowAccumBuffers": "AccumBuffers", "RtowContextOptions": "ContextOptions", "RtowSceneInfo": "SceneInfo",
           "RtowCommId": "CommId", "RtowMetrics": "Metrics", "RtowCombineParams": "CombineParams", "RtowHybridPlan": "HybridPlan"}


def binding_structs():
    src = _strip_comments(_binding_source())
    structs = {}
    for m in re.finditer(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public\s+(?:unsafe\s+)?struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            count = 1
            arr = re.match(r"\[MarshalAs\(UnmanagedType\.ByValArray,\s*SizeConst\s*=\s*(\d+)\)\]\s*(.*)", decl)
            if arr:
                count, decl = int(arr.group(1)), arr.group(2)
            fixed = re.match(r"public\s+fixed\s+(\w+)\s+(\w+)\[(\d+)\]", decl)
            if fixed:
                fields.append((fixed.group(2), fixed.group(1), int(fixed.group(3))))
                continue
            f = re.match(r"public\s+([\w\.]+\s*\*?(?:\[\])?)\s+(.+)", decl)
            assert f, (m.group(1), decl)
            ctype = f.group(1).replace(" ", "").replace("[]", "")
            for name in f.group(2).split(","):
                fields.append((name.strip(), ctype, count))
        structs[m.group(1)] = fields
    return structs


def _layout(structs, name, cache):
    if name in cache:
        return cache[name]
    off, align, offsets = 0, 1, []
    for fname, ctype, count in structs[name]:
        if ctype.endswith("*"):
            size, a = 8, 8
        elif ctype == "byte":
            size, a = 1, 1
        elif ctype in SCALARS:
            size, a = SCALARS[ctype]
        else:
            assert ctype in structs, (name, fname, ctype)
            size, a, _ = _layout(structs, ctype, cache)
        off = (off + a - 1) // a * a
        offsets.append((fname, off))
        off += size * count
        align = max(align, a)
    size = (off + align - 1) // align * align
    cache[name] = (size, align, offsets)
    return cache[name]


def test_struct_layouts_equal_the_ctypes_mirror():
    structs = binding_structs()
    assert sorted(structs) == sorted(MIRRORS), {"in the binding only": sorted(set(structs) - set(MIRRORS)), "missing from the binding": sorted(set(MIRRORS) - set(structs))}
    cache = {}
    for cs_name, py_name in MIRRORS.items():
        mirror = getattr(abi, py_name)
        size, _, offsets = _layout(structs, cs_name, cache)
        assert size == C.sizeof(mirror), (cs_name, size, C.sizeof(mirror))
        assert len(offsets) == len(mirror._fields_), (cs_name, [n for n, _ in offsets], [f[0] for f in mirror._fields_])
        for (fname, off), field in zip(offsets, mirror._fields_):
            assert fname.lower() == field[0].lower(), (cs_name, fname, field[0])                     # same order, same names (C# capitalises them)
            assert off == getattr(mirror, field[0]).offset, (cs_name, fname, off, getattr(mirror, field[0]).offset)


def _c_enum(name):
    body = re.search(r"typedef\s+enum\s+" + name + r"\s*\{(.*?)\}", _strip_comments(_header()), flags=re.S).group(1)
    out, nxt = {}, 0
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = (x.strip() for x in item.split("=", 1))
            v = v.replace("u", "")
            nxt = (1 << int(v.split("<<")[1])) if "<<" in v else int(v, 0)
        else:
            k = item
        out[k] = nxt
        nxt += 1
    return out


def _cs_enum(name):
    body = re.search(r"enum\s+" + name + r"\b[^{]*\{(.*?)\}", _strip_comments(_binding_source()), flags=re.S).group(1)
    return {k.strip(): int(v.strip(), 0) for k, v in (item.split("=") for item in body.split(",") if item.strip())}


def _camel(c_name, prefix):
    return "".join(w.capitalize() for w in c_name[len(prefix):].lower().split("_"))


def test_enum_values_are_the_headers():
    res = _cs_enum("RtowResult")
    for k, v in _c_enum("RtowResult").items():
        assert res[_camel(k, "RTOW_")] == v, k
    flags = _cs_enum("RtowContextFlags")
    c_flags = _c_enum("RtowContextFlags")
    for k, v in c_flags.items():
        if _camel(k, "RTOW_CONTEXT_") in flags:
            assert flags[_camel(k, "RTOW_CONTEXT_")] == v, k
    for k, v in flags.items():
        assert k == "None" or v in c_flags.values(), k
    mask = _cs_enum("RtowGatherMask")
    for k, v in _c_enum("RtowGatherMask").items():
        if _camel(k, "RTOW_GATHER_") in mask:
            assert mask[_camel(k, "RTOW_GATHER_")] == v, k


def test_version_the_binding_checks_is_the_headers():
    line = [l for l in _binding_source().splitlines() if '"rtowGetApiVersion"' in l]
    assert len(line) == 1
    said = re.search(r"==\s*(\d+)\s+for this header", line[0])
    assert said, "the GetApiVersion line must state the version it expects (`== N for this header`)"
    hdr = int(re.search(r"#define\s+RTOW_API_VERSION\s+(\d+)", _header()).group(1))
    assert hdr == abi.RTOW_API_VERSION
    assert int(said.group(1)) == hdr, "INTEGRATION.md binds version %s, include/rtow.h is version %d" % (said.group(1), hdr)
