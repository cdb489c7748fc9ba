#!/usr/bin/env python3
"""Regenerates sha256_a1_b2.wtns.gz: the `.wtns` FILE written by the REFERENCE's own witness calculator
(/root/reference/fixtures/sha256/sha256_js/{generate_witness.js,witness_calculator.js,sha256.wasm}, layout in
witness_calculator.js:208-272) under node with a = 1, b = 2 (groth16/examples/sha256.rs:164-165), gzip-compressed
(mtime 0, so the bytes are reproducible).  Data only; needs /root/reference and node; never run on the GPU box."""
import gzip
import json
import os
import shutil
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
    data = open(os.path.join(tmp, "out.wtns"), "rb").read()
    with open(os.path.join(HERE, "sha256_a1_b2.wtns.gz"), "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, compresslevel=9) as gz:
            gz.write(data)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
