"""Generates rust/zksaas-hip-sys/src/lib.rs -- the raw `extern "C"` declarations of EVERY entry point, constant and
struct of include/zksaas.h -- from the header itself.

usage: python tools/gen_rust_ffi.py            # rewrite the file
       python tools/gen_rust_ffi.py --check    # exit 1 if the committed file differs from what the header yields

There is no Rust toolchain in the build image, so the output cannot be compiled here; tests/test_rust_ffi.py keeps it
honest instead: it regenerates the file (drift = failure) and re-derives, independently of this generator's type map,
name / arity / pointer-or-integer class of every argument from the header and from the Rust text and compares them.
The safe layer above it (rust/zksaas-hip/src/*.rs) mirrors the reference's generic signatures
(dist-primitives/src/dfft/mod.rs:99-175, dmsm/mod.rs:59-102, dpp/mod.rs:15-87, utils/deg_red.rs:80-126,
secret-sharing/src/pss.rs:37-221, mpc-net/src/lib.rs:19-24)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zksaas.h")
OUT = os.path.join(ROOT, "rust", "zksaas-hip-sys", "src", "lib.rs")

SCALARS = {"int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "long long": "c_longlong",
           "long": "c_long", "double": "f64", "char": "c_char", "void": "c_void"}
STRUCTS = {"zk_ctx": "ZkCtx", "zk_net": "ZkNet", "zk_crs_share": "ZkCrsShare", "zk_groth16_masks": "ZkGroth16Masks"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: " " * 0 + "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def rust_type(ctype):
    """C type (no declarator name) -> Rust type.  Pointers are peeled right to left; `const` binds to what is on its left
    (or to the base type when it leads)."""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split(" ")
    if toks[0] == "const":                      # leading const qualifies the base type
        base_const, toks = True, toks[1:]
    else:
        base_const = False
    base = []
    while toks and toks[0] not in ("*", "const"):
        base.append(toks.pop(0))
    base = " ".join(base)
    if toks and toks[0] == "const":             # `T const`
        base_const, toks = True, toks[1:]
    if base in SCALARS:
        cur = SCALARS[base]
    elif base in STRUCTS:
        cur = STRUCTS[base]
    else:
        raise ValueError("unknown C type %r in %r" % (base, ctype))
    cur_const = base_const
    while toks:
        tok = toks.pop(0)
        if tok != "*":
            raise ValueError("cannot parse %r" % ctype)
        cur = ("*const " if cur_const else "*mut ") + cur
        cur_const = False
        if toks and toks[0] == "const":
            cur_const = True
            toks.pop(0)
    return cur


def parse_args(argtext):
    args = []
    argtext = " ".join(argtext.split())
    if argtext in ("", "void"):
        return args
    for a in argtext.split(","):
        a = a.strip()
        m = re.match(r"^(.*?)(\w+)\s*(\[(\d*)\])?$", a)
        ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
        if arr:                                  # T name[k] decays to a pointer to T
            ctype += "*"
        args.append((name, ctype, rust_type(ctype), m.group(4) if arr else None))
    return args


def parse_header(path=HEADER):
    raw = open(path).read()
    text = strip_comments(raw)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    consts = []
    for m in re.finditer(r"^\s*#define\s+(ZK_\w+)\s+(\d+)\s*$", strip_comments(raw), flags=re.M):
        consts.append((m.group(1), int(m.group(2)), "usize"))
    enums = []
    for m in re.finditer(r"enum\s+(\w+)\s*\{([^}]*)\}\s*;", text):
        vals = []
        for item in m.group(2).split(","):
            if "=" in item:
                k, v = item.split("=")
                vals.append((k.strip(), int(v.strip())))
        enums.append((m.group(1), vals))
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # `const void* a` / `const void *a, *b` / `size_t a, b` / `const void* a[6]`
            dm = re.match(r"^((?:const\s+)?(?:long long|\w+))\s*(.*)$", decl)
            base, rest = dm.group(1), dm.group(2)
            for d in rest.split(","):
                d = d.strip()
                stars = d.count("*")
                fm = re.match(r"^\**\s*(\w+)\s*(\[(\d+)\])?$", d.replace(" ", ""))
                rt = rust_type(base + "*" * stars)
                if fm.group(3):
                    rt = "[%s; %s]" % (rt, fm.group(3))
                fields.append((fm.group(1), rt))
        structs.append((m.group(1), fields))
    text_nostruct = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    text_nostruct = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", "", text_nostruct)
    text_nostruct = re.sub(r"enum\s+\w+\s*\{[^}]*\}\s*;", "", text_nostruct)
    text_nostruct = text_nostruct.replace('extern "C" {', "").replace("}", "")
    funcs = []
    for m in re.finditer(r"([\w\s\*]+?)\b(zk_\w+)\s*\(([^()]*)\)\s*;", text_nostruct):
        ret = " ".join(m.group(1).split())
        funcs.append((m.group(2), ret, None if ret == "void" else rust_type(ret), parse_args(m.group(3))))
    # the reference citation in the comment that trails a prototype on its own line(s), for the doc lines (comments that
    # precede a group of prototypes are not attributed: a wrong citation is worse than none)
    cites = {}
    for name, _, _, _ in funcs:
        pos = re.search(r"^(?:int|void|size_t|const char\*)\s+%s\(" % name, raw, flags=re.M).start()
        semi = raw.find(";", pos)
        eol = raw.find("\n", semi)
        tail = raw[pos:eol]
        if "/*" in tail and "*/" not in tail[tail.index("/*"):]:      # the trailing comment runs on to the next lines
            tail = raw[pos:raw.find("*/", semi)]
        c = re.findall(r"[\w\-/]+\.rs:\d+(?:-\d+)?", tail)
        cites[name] = ", ".join(c) if c else None
    return consts, enums, structs, funcs, cites


def emit(consts, enums, structs, funcs, cites):
    o = []
    o.append("//! Raw FFI of libzksaas_hip.so: every entry point, constant and struct of `include/zksaas.h`.")
    o.append("//! GENERATED by tools/gen_rust_ffi.py from the header -- do not edit; `python tools/gen_rust_ffi.py` rewrites it")
    o.append("//! and tests/test_rust_ffi.py fails when the two drift apart.  Data layout at the boundary is arkworks' in-memory")
    o.append("//! representation (Montgomery limbs, little endian), see the header's first comment.")
    o.append("#![allow(non_camel_case_types, non_snake_case, clippy::too_many_arguments)]")
    o.append("use core::ffi::{c_char, c_int, c_long, c_longlong, c_void};")
    o.append("")
    for name, val, ty in consts:
        o.append("pub const %s: %s = %d;" % (name, ty, val))
    for ename, vals in enums:
        o.append("")
        o.append("// enum %s" % ename)
        for k, v in vals:
            o.append("pub const %s: c_int = %d;" % (k, v))
    o.append("")
    for opaque in ("zk_ctx", "zk_net"):
        o.append("/// opaque `%s`" % opaque)
        o.append("#[repr(C)]")
        o.append("pub struct %s {" % STRUCTS[opaque])
        o.append("    _private: [u8; 0],")
        o.append("}")
    for sname, fields in structs:
        o.append("")
        o.append("/// `%s`" % sname)
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append("pub struct %s {" % STRUCTS[sname])
        for fname, rt in fields:
            o.append("    pub %s: %s," % (fname, rt))
        o.append("}")
    o.append("")
    o.append('#[link(name = "zksaas_hip")]')
    o.append('extern "C" {')
    for name, ret, rret, args in funcs:
        if cites.get(name):
            o.append("    /// replaces / serves %s" % cites[name])
        parts = []
        for an, ct, rt, arr in args:
            an = {"type": "type_", "in": "in_", "ref": "ref_", "box": "box_", "move": "move_", "loop": "loop_",
                  "fn": "fn_", "mod": "mod_", "self": "self_", "use": "use_", "match": "match_"}.get(an, an)
            parts.append("%s: %s%s" % (an, rt, " /* [%s] */" % arr if arr else ""))
        line = "    pub fn %s(%s)%s;" % (name, ", ".join(parts), "" if rret is None else " -> " + rret)
        if len(line) <= 118:
            o.append(line)
        else:
            o.append("    pub fn %s(" % name)
            for p in parts:
                o.append("        %s," % p)
            o.append("    )%s;" % ("" if rret is None else " -> " + rret))
    o.append("}")
    o.append("")
    return "\n".join(o)


def generate():
    return emit(*parse_header())


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            sys.stderr.write("rust/zksaas-hip-sys/src/lib.rs is out of date: run python tools/gen_rust_ffi.py\n")
            sys.exit(1)
        sys.exit(0)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print("wrote %s: %d functions" % (os.path.relpath(OUT, ROOT), text.count("pub fn ")))
