"""The Rust binding a rusty-marcher maintainer adds (rusty-marcher_amd/rust/gpu.rs) cannot be
compiled here (no rustc / cargo in the image).  What can be checked without a compiler is that
it says the same thing as the C header it binds: every function of
include/rusty_marcher_amd.h is declared in its `extern "C"` block with the same name, the
same number of arguments and the same class of every argument and of the result (pointer /
f64 / 32- and 64-bit integers / usize / struct by value), and every `#[repr(C)]` struct has the
header struct's fields in the header's order with matching classes (the seam it serves:
engine/src/main.rs:331-333)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rusty_marcher_amd.h")
RUST = os.path.join(ROOT, "rusty-marcher_amd", "rust", "gpu.rs")

STRUCT_NAMES = {"rm_vec3": "RmVec3", "rm_reflectance": "RmReflectance", "rm_scene_desc": "RmSceneDesc",
                "rm_params": "RmParams", "rm_timing": "RmTiming", "rm_frame_times": "RmFrameTimes"}
BY_VALUE = {"rm_vec3": "struct:RmVec3"}


def c_class(t):
    t = re.sub(r"\bconst\b", "", t).strip()
    if "*" in t:
        return "ptr"
    t = re.sub(r"\s+", " ", t)
    known = {"double": "f64", "int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize",
             "uint8_t": "u8", "void": "void", "rm_status": "i32"}
    if t in known:
        return known[t]
    assert t.startswith("rm_"), t                      # a struct by value: rm_vec3 -> struct:RmVec3
    return "struct:" + "".join(w.capitalize() for w in t.split("_"))


def rust_class(t):
    t = t.strip()
    if t.startswith("*"):
        return "ptr"
    known = {"f64": "f64", "c_int": "i32", "i32": "i32", "u32": "u32", "u64": "u64", "usize": "usize", "u8": "u8"}
    if t in known:
        return known[t]
    assert t.startswith("Rm"), t
    return "struct:" + t


def split_args(s):
    s = s.strip()
    if s in ("", "void"):
        return []
    return [a.strip() for a in s.split(",")]


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*+)?)\s*(rm_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        classes = []
        for a in split_args(re.sub(r"\s+", " ", args)):
            a = re.sub(r"\s*/\*.*?\*/", "", a)
            typ = a.rsplit(" ", 1)[0] if not a.endswith("*") else a
            if "*" in a:
                typ = a[:a.rindex("*") + 1]
            classes.append(c_class(typ))
        out[name] = (c_class(ret), classes)
    return out


def rust_functions():
    text = re.sub(r"//[^\n]*", "", open(RUST).read())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"fn\s+(rm_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        classes = [rust_class(a.split(":", 1)[1]) for a in split_args(re.sub(r"\s+", " ", args))]
        out[name] = (rust_class(ret) if ret else "void", classes)
    return out


def header_structs():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(rm_[a-z0-9_]+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = re.sub(r"\s+", " ", decl).strip()
            if not decl:
                continue
            typ, names = decl.rsplit(" ", 1)[0], decl
            # "rm_vec3 position, color" / "double fov, half_fov" / "const rm_sphere *spheres"
            first, rest = decl.split(",")[0], decl.split(",")[1:]
            ptr = "*" in first
            base = first[:first.rindex("*") + 1] if ptr else first.rsplit(" ", 1)[0]
            names = [first[first.rindex("*") + 1:].strip() if ptr else first.rsplit(" ", 1)[1]] + [r.strip() for r in rest]
            for n in names:
                fields.append((n.lstrip("*").strip(), "ptr" if ptr or n.strip().startswith("*") else c_class(base)))
        out[m.group(1)] = fields
    return out


def rust_structs():
    text = re.sub(r"//[^\n]*", "", open(RUST).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)*pub\s+struct\s+(Rm[A-Za-z0-9]+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = f.strip()
            if not f:
                continue
            name, typ = f.replace("pub ", "").split(":", 1)
            fields.append((name.strip(), rust_class(typ)))
        out[m.group(1)] = fields
    return out


def test_every_header_function_is_bound_with_the_same_shape():
    c, r = header_functions(), rust_functions()
    assert len(c) >= 45 and "rm_render" in c and "rm_frame_submit_f64" in c
    assert sorted(c) == sorted(r), "declared in one and not the other: %s" % sorted(set(c) ^ set(r))
    for name in c:
        assert c[name] == r[name], "%s: header %s, gpu.rs %s" % (name, c[name], r[name])
    assert c["rm_scene_add_sphere"] == ("i32", ["ptr", "struct:RmVec3", "f64", "ptr"])
    assert c["rm_create_renderer"] == ("void", ["f64", "f64", "f64", "ptr"])


def test_repr_c_structs_have_the_header_layout():
    c, r = header_structs(), rust_structs()
    for cname, rname in STRUCT_NAMES.items():
        assert cname in c and rname in r, (cname, rname)
        assert c[cname] == r[rname], "%s vs %s:\n%s\n%s" % (cname, rname, c[cname], r[rname])
    assert [n for n, _ in c["rm_params"]][:5] == ["fov", "half_fov", "height", "width", "ratio"]
    assert len(c["rm_scene_desc"]) == 13 and c["rm_reflectance"][4] == ("is_glass_like", "i32")


def test_status_codes_used_by_the_shim_exist():
    text = open(RUST).read()
    assert "if status != 0" in text and "panic!" in text            # non-zero status -> panic, like the reference
    assert "rm_render_rows" in text and "rm_scene_upload" in text
    # the staging frame belongs to the library: the shim neither allocates nor frees page-locked memory per frame
    assert "rm_host_free(self" not in text
