#!/usr/bin/env python3
"""Regenerates sha256_witness_a1_b2.json by running the REFERENCE's own witness calculator
(/root/reference/fixtures/sha256/sha256_js/sha256.wasm + witness_calculator.js) under node with a = 1, b = 2
(the inputs of groth16/examples/sha256.rs:164-165).  Only data is stored: the witness size, the field prime, the
non-binary wires and a bitmap of the binary ones.  Needs /root/reference and node; not run on the GPU box."""
import json
import os
import shutil
import struct
import subprocess
import tempfile

REF = "/root/reference/fixtures/sha256/sha256_js"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tmp = tempfile.mkdtemp()
    for f in ("generate_witness.js", "witness_calculator.js"):
        shutil.copy(os.path.join(REF, f), tmp)
    with open(os.path.join(tmp, "input.json"), "w") as fh:
        json.dump({"a": "1", "b": "2"}, fh)
    subprocess.run(["node", "generate_witness.js", os.path.join(REF, "sha256.wasm"), "input.json", "out.wtns"], cwd=tmp,
                   check=True)
    d = open(os.path.join(tmp, "out.wtns"), "rb").read()
    assert d[:4] == b"wtns"
    _, nsec = struct.unpack("<II", d[4:12])
    off, secs = 12, {}
    for _ in range(nsec):
        sid, sz = struct.unpack("<IQ", d[off:off + 12])
        off += 12
        secs[sid] = d[off:off + sz]
        off += sz
    n8 = struct.unpack("<I", secs[1][:4])[0]
    prime = int.from_bytes(secs[1][4:4 + n8], "little")
    nw = struct.unpack("<I", secs[1][4 + n8:8 + n8])[0]
    w = [int.from_bytes(secs[2][i * n8:(i + 1) * n8], "little") for i in range(nw)]
    out = {
        "source": "fixtures/sha256/sha256_js/sha256.wasm run under node with a=1,b=2",
        "prime": str(prime), "witness_size": nw, "w0": str(w[0]), "public_output_w1": str(w[1]), "a_w2": str(w[2]),
        "b_w3": str(w[3]),
        "non_binary_wires": [[i, str(x)] for i, x in enumerate(w) if x not in (0, 1)][:50],
        "binary_wire_bitmap_hex": "".join("%02x" % sum(((w[i + j] == 1) << j) for j in range(8) if i + j < nw)
                                          for i in range(0, nw, 8)),
    }
    with open(os.path.join(HERE, "sha256_witness_a1_b2.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
