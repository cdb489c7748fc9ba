"""Writes rust/patches/*.diff: the changes a maintainer applies to the reference workspace so that `groth16/` runs on the
GPU with `Net = HipNet` and NO line of `groth16/` changed.

    python tools/gen_rust_patches.py [REFERENCE_ROOT]        (default /root/reference; build container only)

Two crates are touched:
  mpc-net          one defaulted method on `MpcNet` (`hip_backend`, returns None) -- how a generic `Net: MpcSerNet` reaches the
                   device net without any bound changing;
  dist-primitives  an optional dependency on `zksaas-hip` (feature `hip`) and, at the head of `d_fft`, `d_ifft`, `d_msm`,
                   `d_pp`, `deg_red`, a guarded hand-off of the round (<= 10 lines per function; `deg_red`'s `T` gains `'static`).
The diffs are produced by editing a copy of the reference files in memory and running difflib over (original, edited) with two
lines of context; nothing of the reference is stored beyond those context lines.  tests/test_rust_ffi.py re-applies them to a
scratch copy when the reference is mounted and checks the line budget per function.
"""
import difflib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HAND_OFF = {
    # file -> list of (anchor line that ends the signature of the function, lines inserted right after the opening brace)
    "dist-primitives/src/dfft/mod.rs": [
        ("d_fft", "    sid: MultiplexedStreamID,\n) -> Result<Vec<F>, MpcNetError> {\n    debug_assert_eq!(\n        pcoeff_share.len() * pp.l,",
         "    #[cfg(feature = \"hip\")]\n"
         "    if net.hip_backend().is_some() {\n"
         "        // zksaas-hip: fft1 + the king's closure on the GPU, gather / scatter over RCCL\n"
         "        return zksaas_hip::dfft::d_fft(pcoeff_share, &fft_mask.in_mask, &fft_mask.out_mask, rearrange, dom, pp, net, sid).await;\n"
         "    }\n"),
        ("d_ifft", "    sid: MultiplexedStreamID,\n) -> Result<Vec<F>, MpcNetError> {\n    debug_assert_eq!(\n        peval_share.len() * pp.l,",
         "    #[cfg(feature = \"hip\")]\n"
         "    if net.hip_backend().is_some() {\n"
         "        return zksaas_hip::dfft::d_ifft(peval_share, &fft_mask.in_mask, &fft_mask.out_mask, rearrange, dom, g, pp, net, sid).await;\n"
         "    }\n"),
    ],
    "dist-primitives/src/dmsm/mod.rs": [
        ("d_msm", ") -> Result<G, MpcNetError> {\n",
         "    #[cfg(feature = \"hip\")]\n"
         "    if net.hip_backend().is_some() {\n"
         "        return zksaas_hip::dmsm::d_msm(bases, scalars, &msm_mask.in_mask, &msm_mask.out_mask, pp, net, sid).await;\n"
         "    }\n"),
    ],
    "dist-primitives/src/dpp/mod.rs": [
        ("d_pp", ") -> Result<Vec<F>, MpcNetError> {\n",
         "    #[cfg(feature = \"hip\")]\n"
         "    if net.hip_backend().is_some() {\n"
         "        return zksaas_hip::dpp::d_pp(num, den, &degred_mask.in_mask, &degred_mask.out_mask, pp, net, sid).await;\n"
         "    }\n"),
    ],
    "dist-primitives/src/utils/deg_red.rs": [
        ("deg_red", ") -> Result<Vec<T>, MpcNetError> {\n",
         "    #[cfg(feature = \"hip\")]\n"
         "    if net.hip_backend().is_some() {\n"
         "        return zksaas_hip::deg_red::deg_red(x_share, &degred_mask.in_mask, &degred_mask.out_mask, pp, net, sid).await;\n"
         "    }\n"),
    ],
}


def edit_dist_primitives(path, text):
    for name, anchor, ins in HAND_OFF.get(path, []):
        at = text.index("pub async fn %s<" % name)
        pos = text.index(anchor, at)
        brace = text.index("{\n", pos) + 2
        text = text[:brace] + ins + text[brace:]
    if path.endswith("deg_red.rs"):
        # TypeId dispatch between T = F and T = a curve group needs T: 'static (every instantiation in the workspace is)
        at = text.index("pub async fn deg_red<")
        old = "    T: DomainCoeff<F> + CanonicalSerialize + CanonicalDeserialize + UniformRand,\n    Net: MpcSerNet,"
        pos = text.index(old, at)
        new = old.replace("UniformRand,\n", "UniformRand + 'static,\n")
        text = text[:pos] + new + text[pos + len(old):]
    if path.endswith("dist-primitives/Cargo.toml"):
        text = text.replace('mpc-net ={ version = "0.1.0", path = "../mpc-net" }\n',
                            'mpc-net ={ version = "0.1.0", path = "../mpc-net" }\n'
                            '# MI355X back end (rust/zksaas-hip of the zk-saas_amd repository); adjust the path to where it is checked out\n'
                            'zksaas-hip = { version = "0.6.0", path = "../../zk-saas_amd/rust/zksaas-hip", optional = true }\n', 1)
        text = text.replace("[dependencies]\n", '[features]\nhip = ["zksaas-hip"]\n\n[dependencies]\n', 1)
    return text


def edit_mpc_net(path, text):
    anchor = "    /// Is the network layer initalized?\n    fn is_init(&self) -> bool;\n"
    ins = ("    /// The GPU back end behind this net, if any (`zksaas_hip::HipNet` returns itself).  dist-primitives asks before a\n"
           "    /// round and hands it to the device; every other net keeps the CPU path.\n"
           "    fn hip_backend(&self) -> Option<&(dyn core::any::Any + Send + Sync)> {\n"
           "        None\n"
           "    }\n")
    pos = text.index(anchor) + len(anchor)
    return text[:pos] + ins + text[pos:]


def diff(ref, path, edit):
    with open(os.path.join(ref, path)) as f:
        a = f.read()
    b = edit(path, a)
    assert a != b, path
    lines = []
    for ln in difflib.unified_diff(a.splitlines(True), b.splitlines(True), "a/" + path, "b/" + path, n=2):
        lines.append(ln if ln.endswith("\n") else ln + "\n\\ No newline at end of file\n")
    return "".join(lines)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(ROOT, "rust", "patches")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "mpc-net.diff"), "w") as f:
        f.write(diff(ref, "mpc-net/src/lib.rs", edit_mpc_net))
    with open(os.path.join(out, "dist-primitives.diff"), "w") as f:
        for p in ["dist-primitives/Cargo.toml"] + sorted(HAND_OFF):
            f.write(diff(ref, p, edit_dist_primitives))
    print("wrote", out)


if __name__ == "__main__":
    main()
