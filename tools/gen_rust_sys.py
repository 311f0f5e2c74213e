#!/usr/bin/env python3
"""Generate bindings/rust/gs3d_sys.rs — the complete Rust `extern "C"` view of include/gs3d.h
(opaque handles, #[repr(C)] structs, status / enum constants, one declaration per function).
Mechanical: a maintainer of the reference crate drops the file in as `src/hip/sys.rs`
(INTEGRATION.md §1).  Rust is not installed in the build image, so the output is checked only for
being in sync with the header (tests/test_host_api.py) — it has not been compiled.
Run:  python tools/gen_rust_sys.py [--check]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gs3d.h")
OUT = os.path.join(ROOT, "bindings", "rust", "gs3d_sys.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "uint16_t": "u16",
           "uint8_t": "u8", "int8_t": "i8", "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char",
           "void": "c_void", "gs_status": "gs_status", "int": "i32"}
ENUMS = {"gs_sh_config", "gs_cov3d_config", "gs_display_mode", "gs_kernel_id"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def rust_type(ctype, array=None):
    """ctype like 'const gs_gaussian *', 'uint32_t', 'gs_buffer *const *const *'"""
    toks = ctype.replace("*", " * ").split()
    base, ptrs, const_base, pending_const = None, [], False, False
    for t in toks:
        if t == "const":
            if base is None:
                const_base = True
            elif ptrs:
                ptrs[-1] = True          # const applies to the pointer just seen
            else:
                const_base = True
        elif t == "*":
            ptrs.append(False)
        elif t in ("struct", "enum"):
            continue
        else:
            base = t
    r = SCALARS.get(base, "u32" if base in ENUMS else base)
    # innermost pointer's pointee constness is const_base; each further level uses the previous pointer's const
    consts = [const_base] + ptrs[:-1]
    for c in consts[:len(ptrs)]:
        r = ("*const " if c else "*mut ") + r
    if array is not None:            # array parameter decays to a pointer
        r = ("*const " if const_base else "*mut ") + r
    return r


def parse(text):
    text = strip_comments(text)
    enums, structs, opaque, funcs = [], [], [], []
    for m in re.finditer(r"(?:typedef\s+)?enum\s*\{(.*?)\}\s*(\w*)\s*;", text, flags=re.S):
        items = [i.strip() for i in m.group(1).split(",") if i.strip()]
        enums.append((m.group(2), [tuple(x.strip() for x in i.split("=")) for i in items]))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            f = " ".join(f.split())
            if not f:
                continue
            head, names = f.rsplit(" ", 1)[0], None
            # 'float pos[3], rot[4]' style does not occur; one or several comma-separated names may
            mm = re.match(r"(.+?)\s*((?:\**\w+(?:\[\w+\])*\s*,\s*)*\**\w+(?:\[\w+\])*)$", f)
            ctype, names = mm.group(1), mm.group(2)
            for n in names.split(","):
                n = n.strip()
                stars = n.count("*")
                n = n.replace("*", "")
                am = re.match(r"(\w+)((?:\[\w+\])*)", n)
                dims = re.findall(r"\[(\w+)\]", am.group(2))
                t = rust_type(ctype + " *" * stars)
                for d in reversed(dims):
                    t = "[%s; %s]" % (t, d)
                fields.append((am.group(1), t))
        structs.append((m.group(3), fields))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.append(m.group(2))
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    body = re.sub(r"(?:typedef\s+)?enum\s*\{.*?\}\s*\w*\s*;", " ", body, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(gs_\w+)\s*\(([^()]*)\)\s*;", body):
        ret = " ".join(m.group(1).split())
        if ret.startswith("typedef") or not ret:
            continue
        params = []
        plist = " ".join(m.group(3).split())
        if plist and plist != "void":
            for prm in plist.split(","):
                prm = prm.strip()
                am = re.match(r"(.+?)(\w+)(\[\w*\])?$", prm)
                params.append((am.group(2), rust_type(am.group(1), am.group(3))))
        funcs.append((m.group(2), params, None if ret == "void" else rust_type(ret)))
    return enums, structs, opaque, funcs


def generate():
    enums, structs, opaque, funcs = parse(open(HEADER).read())
    o = ["// gs3d_sys.rs — GENERATED from include/gs3d.h by tools/gen_rust_sys.py; do not edit.",
         "// Raw FFI of libgs3d_hip.so for the reference crate (INTEGRATION.md).  Not compiled in the build image.",
         "#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]",
         "use std::os::raw::{c_char, c_void};", "", "pub type gs_status = i32;", ""]
    for name, items in enums:
        if name:
            o.append("// enum %s (passed as u32)" % name)
        val = -1
        for it in items:
            if len(it) == 2:
                val = int(it[1], 0)
            else:
                val += 1
            ty = "gs_status" if it[0].startswith(("GS_OK", "GS_ERR")) else "u32"
            o.append("pub const %s: %s = %d;" % (it[0], ty, val))
        o.append("")
    for n in opaque:
        o.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % n)
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct %s {" % name)
        for fn, ft in fields:
            o.append("    pub %s: %s," % (fn, ft))
        o.append("}\n")
    o.append('#[link(name = "gs3d_hip")]\nextern "C" {')
    for name, params, ret in funcs:
        ps = ", ".join("%s: %s" % (("r#" + n) if n in ("in", "type", "fn", "ref", "box", "mod", "use") else n, t)
                       for n, t in params)
        o.append("    pub fn %s(%s)%s;" % (name, ps, (" -> " + ret) if ret else ""))
    o.append("}\n")
    return "\n".join(o), [f[0] for f in funcs]


def main():
    text, names = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("bindings/rust/gs3d_sys.rs is stale: run python tools/gen_rust_sys.py")
            sys.exit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print("wrote %s: %d functions" % (OUT, len(names)))


if __name__ == "__main__":
    main()
