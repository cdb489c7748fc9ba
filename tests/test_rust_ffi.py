"""The Rust side of the boundary cannot be compiled here (no toolchain): these checks keep it honest mechanically.
  * rust/zksaas-hip-sys/src/lib.rs is what tools/gen_rust_ffi.py generates from include/zksaas.h today (no drift);
  * an INDEPENDENT pass (own header parser, own Rust parser, own type classes) agrees on name, arity and the
    pointer / 32-bit / 64-bit / double class of every argument and return value of every entry point;
  * the generated declarations cover exactly the zk_* symbols the built library exports;
  * every `sys::zk_*(…)` call in the hand-written shim (rust/zksaas-hip/src/*.rs) names a declared function and passes the
    number of arguments the header declares; the reference signatures the shim mirrors are all present."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SYS_RS = os.path.join(ROOT, "rust", "zksaas-hip-sys", "src", "lib.rs")
HEADER = os.path.join(ROOT, "include", "zksaas.h")


def test_generated_file_is_current():
    import gen_rust_ffi
    assert open(SYS_RS).read() == gen_rust_ffi.generate(), "run python tools/gen_rust_ffi.py"


def _c_class(ctype):
    t = " ".join(ctype.split())
    if "*" in t or "[" in t:
        return "ptr"
    t = t.replace("const ", "")
    return {"int": "i32", "uint32_t": "u32", "size_t": "w64", "uint64_t": "w64", "long long": "w64", "long": "w64",
            "double": "f64", "void": "void"}[t]


def _header_protos():
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    out = {}
    for m in re.finditer(r"\b(int|void|size_t|const char\s*\*)\s*(zk_\w+)\s*\(([^()]*)\)\s*;", text):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        classes = []
        for a in args:
            name = re.search(r"(\w+)\s*(\[\d*\])?$", a)
            decl = a[:name.start(1)] + (name.group(2) or "")
            classes.append((name.group(1), _c_class(decl)))
        out[m.group(2)] = (_c_class(m.group(1)), classes)
    return out


def _split_top(s):
    parts, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


def _rs_class(t):
    t = re.sub(r"/\*.*?\*/", "", t).strip()
    if t.startswith("*"):
        return "ptr"
    return {"c_int": "i32", "u32": "u32", "usize": "w64", "u64": "w64", "c_longlong": "w64", "c_long": "w64",
            "f64": "f64"}[t]


def _rust_decls():
    text = open(SYS_RS).read()
    body = text[text.index('extern "C" {'):]
    body = re.sub(r"///[^\n]*", "", body)
    out = {}
    for m in re.finditer(r"pub fn (zk_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", body, flags=re.S):
        args = []
        for a in _split_top(m.group(2)):
            name, ty = a.split(":", 1)
            args.append((name.strip().rstrip("_"), _rs_class(ty)))
        out[m.group(1)] = ("void" if m.group(3) is None else _rs_class(m.group(3)), args)
    return out


def test_every_entry_point_agrees_with_the_header():
    c, r = _header_protos(), _rust_decls()
    assert len(c) == len(r) >= 107
    assert set(c) == set(r)
    for name in c:
        assert c[name][0] == r[name][0], name
        assert [k for _, k in c[name][1]] == [k for _, k in r[name][1]], name
        assert [n.rstrip("_") for n, _ in c[name][1]] == [n for n, _ in r[name][1]], name


def test_structs_and_constants():
    text = open(SYS_RS).read()
    hdr = open(HEADER).read()
    for cname, rname in (("zk_crs_share", "ZkCrsShare"), ("zk_groth16_masks", "ZkGroth16Masks")):
        cbody = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        cbody = re.sub(r"/\*.*?\*/", "", cbody, flags=re.S)
        cfields = []
        for decl in cbody.split(";"):
            for d in re.sub(r"^\s*(const\s+)?\w+", "", decl.strip(), count=1).split(","):
                nm = re.search(r"(\w+)\s*(\[(\d+)\])?\s*$", d)
                if nm:
                    cfields.append((nm.group(1), nm.group(3)))
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % rname, text, flags=re.S).group(1)
        rfields = [(m.group(1), (re.search(r";\s*(\d+)\]", m.group(2)) or [None, None])[1])
                   for m in re.finditer(r"pub (\w+): ([^\n]+),", rbody)]
        assert cfields == rfields, cname
        assert "#[repr(C)]" in text[:text.index("pub struct %s" % rname)][-80:]
    for m in re.finditer(r"(ZK_\w+)\s*=\s*(\d+)", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)):
        assert re.search(r"pub const %s: c_int = %s;" % (m.group(1), m.group(2)), text), m.group(1)
    assert "pub const ZK_NET_ID_BYTES: usize = 512;" in text


def test_declarations_match_the_built_library():
    so = os.path.join(ROOT, "zk-saas_amd", "libzksaas_hip.so")
    if not os.path.exists(so):
        import pytest
        pytest.skip("library not built")
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in syms.splitlines() if ln.split() and ln.split()[-1].startswith("zk_")}
    assert exported == set(_rust_decls())


def test_shim_calls_match_the_declarations():
    decls = _rust_decls()
    files = sorted(glob.glob(os.path.join(ROOT, "rust", "zksaas-hip", "src", "*.rs")))
    assert {os.path.basename(f) for f in files} >= {"lib.rs", "error.rs", "net.rs", "pss.rs", "dfft.rs", "dmsm.rs", "dpp.rs",
                                                   "deg_red.rs", "groth16.rs"}
    used = set()
    for f in files:
        src = open(f).read()
        src = re.sub(r"//[^\n]*", "", src)
        for m in re.finditer(r"sys::(zk_\w+)\s*\(", src):
            name = m.group(1)
            assert name in decls, "%s: %s is not exported" % (os.path.basename(f), name)
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            nargs = len(_split_top(src[m.end():i - 1]))
            assert nargs == len(decls[name][1]), "%s: %s called with %d arguments, declared with %d" % (
                os.path.basename(f), name, nargs, len(decls[name][1]))
            used.add(name)
    # the reference surface the north star names is all routed
    assert used >= {"zk_ctx_create", "zk_pss_pack", "zk_pss_det_pack", "zk_pss_unpack", "zk_pss_unpack2", "zk_dist_d_fft",
                    "zk_dist_d_ifft", "zk_dist_d_msm", "zk_dist_deg_red", "zk_dist_deg_red_points", "zk_dist_d_pp", "zk_d_pp",
                    "zk_dist_circom_h", "zk_dist_groth16_prove", "zk_net_create", "zk_net_sync", "zk_last_error",
                    "zk_fft_mask_sample", "zk_msm_mask_sample", "zk_degred_mask_sample"}
    sigs = {"dfft.rs": ["pub async fn d_fft<", "pub async fn d_ifft<"], "dmsm.rs": ["pub async fn d_msm<"],
            "dpp.rs": ["pub async fn d_pp<"], "deg_red.rs": ["pub async fn deg_red<"],
            "groth16.rs": ["pub async fn circom_h<", "pub async fn dsha256<"]}
    for fname, needles in sigs.items():
        src = open(os.path.join(ROOT, "rust", "zksaas-hip", "src", fname)).read()
        for nd in needles:
            assert nd in src and "Result<" in src[src.index(nd):src.index(nd) + 1200] and "MpcNetError" in src, (fname, nd)
