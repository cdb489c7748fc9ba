"""The Rust side of the boundary cannot be compiled here (no toolchain): these checks keep it honest mechanically.
  * rust/zksaas-hip-sys/src/lib.rs is what tools/gen_rust_ffi.py generates from include/zksaas.h today (no drift);
  * an INDEPENDENT pass (own header parser, own Rust parser, own type classes) agrees on name, arity and the
    pointer / 32-bit / 64-bit / double class of every argument and return value of every entry point;
  * the generated declarations cover exactly the zk_* symbols the built library exports;
  * every `sys::zk_*(…)` call in the hand-written shims (rust/zksaas-hip, rust/zksaas-hip-groth16) names a declared function
    and passes the number of arguments the header declares;
  * the three manifests plus the dependency the patches add to the reference form an ACYCLIC graph
    (groth16 -> dist-primitives -> zksaas-hip -> {zksaas-hip-sys, mpc-net, secret-sharing});
  * (build container only) every shim `pub async fn` carries the reference's generic bounds, argument list and return
    type token for token -- a mask struct of dist-primitives arriving as its two fields is the one documented
    substitution -- and rust/patches/*.diff apply to the reference, touch nothing under groth16/ and add at most ten
    lines per hunk."""
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


SHIM_DIRS = [os.path.join(ROOT, "rust", "zksaas-hip", "src"), os.path.join(ROOT, "rust", "zksaas-hip-groth16", "src")]
REFERENCE = "/root/reference"


def test_shim_calls_match_the_declarations():
    decls = _rust_decls()
    files = sorted(f for d in SHIM_DIRS for f in glob.glob(os.path.join(d, "*.rs")))
    assert {os.path.relpath(f, os.path.join(ROOT, "rust")) for f in files} >= {
        "zksaas-hip/src/" + n for n in ("lib.rs", "error.rs", "net.rs", "pss.rs", "dfft.rs", "dmsm.rs", "dpp.rs", "deg_red.rs")
    } | {"zksaas-hip-groth16/src/lib.rs"}
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
                    "zk_dist_circom_h", "zk_dist_libsnark_h", "zk_dist_groth16_prove", "zk_net_create", "zk_net_sync",
                    "zk_net_enter", "zk_net_gather", "zk_net_scatter", "zk_net_bcast_host", "zk_last_error",
                    "zk_fft_mask_sample", "zk_msm_mask_sample", "zk_degred_mask_sample"}


# ---------------------------------------------------------------- the crates can be wired in: no dependency cycle
def _manifest_deps(path):
    """[dependencies] names of a Cargo.toml (path + registry; optional ones included)"""
    text = open(path).read()
    m = re.search(r"^\[dependencies\]\s*$(.*?)(?=^\[|\Z)", text, flags=re.S | re.M)
    deps = set()
    for ln in (m.group(1) if m else "").splitlines():
        ln = ln.split("#")[0].strip()
        if "=" in ln:
            deps.add(ln.split("=")[0].strip())
    return re.search(r'^name\s*=\s*"([^"]+)"', text, flags=re.M).group(1), deps


# what the reference's four crates depend on inside the workspace (dist-primitives/Cargo.toml:16-17, groth16/Cargo.toml:23-25,
# secret-sharing / mpc-net: nothing in the workspace); re-read from the manifests when the reference is mounted
REF_EDGES = {"mpc-net": set(), "secret-sharing": set(), "dist-primitives": {"secret-sharing", "mpc-net"},
             "groth16": {"secret-sharing", "mpc-net", "dist-primitives"}}


def _graph():
    ours = {}
    for crate in ("zksaas-hip-sys", "zksaas-hip", "zksaas-hip-groth16"):
        name, deps = _manifest_deps(os.path.join(ROOT, "rust", crate, "Cargo.toml"))
        assert name == crate
        ours[name] = deps
    ref = {k: set(v) for k, v in REF_EDGES.items()}
    if os.path.isdir(REFERENCE):
        for crate in ref:
            name, deps = _manifest_deps(os.path.join(REFERENCE, crate, "Cargo.toml"))
            assert deps & set(ref) == ref[crate], crate          # the recorded edges are the reference's
    # the edges the patches add (rust/patches/dist-primitives.diff: `+zksaas-hip = {...}` under [dependencies])
    patch = open(os.path.join(ROOT, "rust", "patches", "dist-primitives.diff")).read()
    added = set(re.findall(r"^\+([\w-]+)\s*=\s*\{", patch, flags=re.M))
    assert added == {"zksaas-hip"}
    ref["dist-primitives"] |= added
    assert not re.search(r"^\+[\w-]+\s*=\s*\{", open(os.path.join(ROOT, "rust", "patches", "mpc-net.diff")).read(), flags=re.M)
    nodes = set(ours) | set(ref)
    return {n: ({**ours, **ref}[n] & nodes) for n in nodes}


def test_dependency_graph_of_crates_and_patches_is_acyclic():
    g = _graph()
    # what makes the recipe wire-able: dist-primitives calls DOWN into zksaas-hip, which knows nothing above mpc-net / secret-sharing
    assert "zksaas-hip" in g["dist-primitives"]
    assert g["zksaas-hip"] == {"zksaas-hip-sys", "mpc-net", "secret-sharing"}
    assert g["zksaas-hip-sys"] == set()
    assert g["zksaas-hip-groth16"] >= {"groth16", "dist-primitives", "zksaas-hip"}
    state = {}

    def visit(n, path):
        if state.get(n) == 2:
            return
        assert state.get(n) != 1, "dependency cycle: " + " -> ".join(path + [n])
        state[n] = 1
        for d in sorted(g[n]):
            visit(d, path + [n])
        state[n] = 2
    for n in sorted(g):
        visit(n, [])
    # and the README tells the same story as the manifests
    readme = open(os.path.join(ROOT, "rust", "README.md")).read()
    assert "groth16 -> dist-primitives -> zksaas-hip" in readme and "patches/dist-primitives.diff" in readme


# ---------------------------------------------------------------- signatures: the reference's, token for token
def _tokens(s):
    s = re.sub(r"//[^\n]*", "", s)
    return re.findall(r"[A-Za-z_][A-Za-z_0-9]*|'[a-z]+|::|->|[<>\[\](){}&*,:;+=]|\d+", s)


def _signature(src, name, public=True):
    m = re.search(r"(pub )?async fn %s\s*<" % name, src)
    assert m, name
    i, depth = m.end(), 1
    while depth:                                   # generics
        depth += {"<": 1, ">": -1}.get(src[i], 0)
        if src[i:i + 2] == "->":
            depth += 1                             # the '>' of an arrow is not a bracket
        i += 1
    generics = src[m.end():i - 1]
    assert src[i:].lstrip()[0] == "("
    j = src.index("(", i) + 1
    k, depth = j, 1
    while depth:
        depth += {"(": 1, ")": -1}.get(src[k], 0)
        k += 1
    args = src[j:k - 1]
    rest = src[k:]
    ret = rest[:re.search(r"\bwhere\b|\{", rest).start()]
    return _tokens(generics), [_tokens(a) for a in _split_top(args) if a.strip()], _tokens(ret)


def _strip_mut(arg):
    return [t for t in arg if t != "mut"]


# a mask struct of dist-primitives arrives in zksaas-hip as its two fields (the crate cannot name a type of a crate that depends on it)
MASK_FIELDS = {
    ("fft_mask", "& FftMask < F >"): ["in_mask : & [ F ]", "out_mask : & [ F ]"],
    ("msm_mask", "& MsmMask < G >"): ["in_mask : & G", "out_mask : & G"],
    ("degred_mask", "& DegRedMask < F , F >"): ["in_mask : & [ F ]", "out_mask : & [ F ]"],
    ("degred_mask", "& DegRedMask < F , T >"): ["in_mask : & [ T ]", "out_mask : & [ T ]"],
}
CORE = [("dfft.rs", "d_fft", "dist-primitives/src/dfft/mod.rs"), ("dfft.rs", "d_ifft", "dist-primitives/src/dfft/mod.rs"),
        ("dmsm.rs", "d_msm", "dist-primitives/src/dmsm/mod.rs"), ("dpp.rs", "d_pp", "dist-primitives/src/dpp/mod.rs"),
        ("deg_red.rs", "deg_red", "dist-primitives/src/utils/deg_red.rs")]


def _expected_core_signature(ref_sig, name):
    generics, args, ret = ref_sig
    out = []
    for a in args:
        a = _strip_mut(a)
        key = (a[0], " ".join(a[2:]))
        if key in MASK_FIELDS:
            out += [f.split() for f in MASK_FIELDS[key]]
        else:
            out.append(a)
    if name == "deg_red":                          # the one bound the patch adds to the reference's own deg_red as well
        at = generics.index("UniformRand")
        generics = generics[:at + 1] + ["+", "'static"] + generics[at + 1:]
    return [t for t in generics if True], out, ret


def _drop_trailing_comma(toks):
    return toks[:-1] if toks and toks[-1] == "," else toks


def test_shim_signatures_are_the_references():
    import pytest
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference is mounted in the build container only")
    for fname, name, ref_file in CORE:
        ours = _signature(open(os.path.join(SHIM_DIRS[0], fname)).read(), name)
        want = _expected_core_signature(_signature(open(os.path.join(REFERENCE, ref_file)).read(), name), name)
        assert _drop_trailing_comma(ours[0]) == _drop_trailing_comma(want[0]), (name, "generic parameters and bounds")
        assert [_strip_mut(a) for a in ours[1]] == want[1], (name, "argument list")
        assert ours[2] == want[2], (name, "return type")
        assert "MpcSerNet" in ours[0], name
    # the fused forms name groth16's own types: exact
    g16 = open(os.path.join(SHIM_DIRS[1], "lib.rs")).read()
    ref = open(os.path.join(REFERENCE, "groth16/src/ext_wit.rs")).read()
    for name in ("circom_h", "libsnark_h"):
        ours, want = _signature(g16, name), _signature(ref, name)
        assert _drop_trailing_comma(ours[0]) == _drop_trailing_comma(want[0]), name
        assert ours[1] == want[1], name
        assert ours[2] == want[2], name
    # dsha256 (private to the reference's example, returns a bare tuple after unwrapping): same argument list, Result around it
    ours = _signature(g16, "dsha256")
    want = _signature(open(os.path.join(REFERENCE, "groth16/examples/sha256.rs")).read(), "dsha256")
    assert ours[1] == want[1]
    assert ours[0][:3] == want[0][:3] == ["E", ",", "Net"]
    assert ours[2] == _tokens("-> Result<") + want[2][1:] + _tokens(", MpcNetError>")


def test_patches_apply_to_the_reference_and_stay_small():
    import pytest
    import shutil
    import tempfile
    d = os.path.join(ROOT, "rust", "patches")
    mp, dp = open(os.path.join(d, "mpc-net.diff")).read(), open(os.path.join(d, "dist-primitives.diff")).read()
    # nothing under groth16/ is touched, and no hunk adds more than ten lines
    for text in (mp, dp):
        assert not re.search(r"^\+\+\+ b/groth16/", text, flags=re.M)
        for hunk in re.split(r"^@@.*?@@.*$", text, flags=re.M)[1:]:
            assert sum(1 for ln in hunk.splitlines() if ln.startswith("+") and not ln.startswith("+++")) <= 10
    assert set(re.findall(r"^\+\+\+ b/(\S+)", mp + dp, flags=re.M)) == {
        "mpc-net/src/lib.rs", "dist-primitives/Cargo.toml", "dist-primitives/src/dfft/mod.rs", "dist-primitives/src/dmsm/mod.rs",
        "dist-primitives/src/dpp/mod.rs", "dist-primitives/src/utils/deg_red.rs"}
    # every hand-off calls a function the shim defines, with as many arguments as it takes
    for fname, name, _ in CORE:
        sig = _signature(open(os.path.join(SHIM_DIRS[0], fname)).read(), name)
        m = re.search(r"zksaas_hip::%s::%s\((.*?)\)\.await" % (fname[:-3], name), dp)
        assert m, name
        assert len(_split_top(m.group(1))) == len(sig[1]), name
    assert "fn hip_backend(&self) -> Option<&(dyn core::any::Any + Send + Sync)>" in mp
    assert "fn hip_backend(&self) -> Option<&(dyn Any + Send + Sync)>" in open(os.path.join(SHIM_DIRS[0], "net.rs")).read()
    if not os.path.isdir(REFERENCE) or not shutil.which("patch"):
        pytest.skip("the reference is mounted in the build container only")
    import gen_rust_patches  # noqa: F401  (the committed diffs are what the generator writes today)
    with tempfile.TemporaryDirectory() as tmp:
        for crate in ("mpc-net", "dist-primitives", "groth16"):
            shutil.copytree(os.path.join(REFERENCE, crate), os.path.join(tmp, crate))
        before = {f: open(f).read() for f in glob.glob(os.path.join(tmp, "groth16", "**", "*.rs"), recursive=True)}
        for f in ("mpc-net.diff", "dist-primitives.diff"):
            r = subprocess.run(["patch", "-p1", "-s", "-i", os.path.join(d, f)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
        assert before == {f: open(f).read() for f in before}, "groth16/ must stay untouched"
        patched = open(os.path.join(tmp, "dist-primitives", "src", "dfft", "mod.rs")).read()
        assert patched.count("net.hip_backend().is_some()") == 2
