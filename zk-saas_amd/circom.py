"""Circom front end: iden3 `.r1cs` / `.wtns` files and the device-side R1CS -> QAP evaluation.

The reference gets this from the third-party `ark-circom` crate (`CircomConfig::new(wasm, r1cs)`,
groth16/examples/sha256.rs:160-177; not vendored under /root/reference).  What is mirrored here:

* `read_r1cs` / `write_r1cs`: iden3 binary R1CS format v1 (sections 1 = header, 2 = constraints, 3 = wire->label
  map).  ark-circom keeps circom's wire order (wire 0 = constant one, then public outputs, public inputs, private
  inputs) and reports `num_instance_variables = 1 + nPubOut + nPubIn`, which is what `R1CS` carries.
* `read_wtns` / `write_wtns`: the layout written by the reference's own witness calculator
  (fixtures/sha256/sha256_js/witness_calculator.js:208-272): magic "wtns", version 2, two sections
  (1 = n8, prime, witness size; 2 = little-endian field elements).
* `DeviceR1cs.qap(w_d)`: groth16/src/qap.rs:42-89 on the device through `zk_r1cs_qap`:
  a_i = <A_i, w>, b_i = <B_i, w>, c_i = a_i * b_i, a[nc .. nc+ni] = w[.. ni], zero padded to the domain size.

Nothing here falls back to a CPU computation of the QAP vectors: `DeviceR1cs` needs the HIP library.
"""
import struct

import numpy as np

from .api import DeviceBuffer
from .sha256_circuit import R1CS

R1CS_MAGIC = b"r1cs"
WTNS_MAGIC = b"wtns"


def _sections(data, magic):
    if data[:4] != magic:
        raise ValueError("bad magic %r (expected %r)" % (data[:4], magic))
    version, nsec = struct.unpack_from("<II", data, 4)
    off, secs = 12, {}
    for _ in range(nsec):
        if off + 12 > len(data):
            raise ValueError("truncated section table")
        sid, size = struct.unpack_from("<IQ", data, off)
        off += 12
        if off + size > len(data):
            raise ValueError("truncated section %d" % sid)
        secs.setdefault(sid, data[off:off + size])
        off += size
    return version, secs


def _load(src):
    if isinstance(src, (bytes, bytearray, memoryview)):
        return bytes(src)
    with open(src, "rb") as fh:
        return fh.read()


def read_r1cs(src):
    """Returns (R1CS, prime, info) with info = dict(n_pub_out, n_pub_in, n_prv_in, n_labels, wire_to_label)."""
    data = _load(src)
    version, secs = _sections(data, R1CS_MAGIC)
    if version != 1:
        raise ValueError("unsupported r1cs version %d" % version)
    if 1 not in secs or 2 not in secs:
        raise ValueError("r1cs file lacks the header or the constraint section")
    hdr = secs[1]
    n8 = struct.unpack_from("<I", hdr, 0)[0]
    prime = int.from_bytes(hdr[4:4 + n8], "little")
    n_wires, n_pub_out, n_pub_in, n_prv_in = struct.unpack_from("<IIII", hdr, 4 + n8)
    n_labels = struct.unpack_from("<Q", hdr, 20 + n8)[0]
    n_constraints = struct.unpack_from("<I", hdr, 28 + n8)[0]
    body, off = secs[2], 0
    rows = ([], [], [])
    for _ in range(n_constraints):
        for m in range(3):
            nterms = struct.unpack_from("<I", body, off)[0]
            off += 4
            lc = []
            for _ in range(nterms):
                wire = struct.unpack_from("<I", body, off)[0]
                coeff = int.from_bytes(body[off + 4:off + 4 + n8], "little")
                off += 4 + n8
                if wire >= n_wires:
                    raise ValueError("wire index %d out of range" % wire)
                lc.append((coeff % prime, wire))
            rows[m].append(lc)
    if off != len(body):
        raise ValueError("trailing bytes in the constraint section")
    ni = 1 + n_pub_out + n_pub_in
    w2l = None
    if 3 in secs:
        w2l = list(struct.unpack_from("<%dQ" % n_wires, secs[3], 0))
    r1cs = R1CS(ni, n_wires - ni, rows[0], rows[1], rows[2])
    return r1cs, prime, {"n_pub_out": n_pub_out, "n_pub_in": n_pub_in, "n_prv_in": n_prv_in, "n_labels": n_labels,
                         "wire_to_label": w2l}


def write_r1cs(r1cs, prime, n_pub_out=None, n_pub_in=0, path=None):
    """Serialises `r1cs` in the iden3 format (instance variables = constant one + outputs + inputs)."""
    n8 = (prime.bit_length() + 63) // 64 * 8
    ni = r1cs.num_instance_variables
    if n_pub_out is None:
        n_pub_out = ni - 1 - n_pub_in
    if 1 + n_pub_out + n_pub_in != ni:
        raise ValueError("public wire counts do not add up to num_instance_variables")
    n_wires = r1cs.num_variables
    hdr = struct.pack("<I", n8) + prime.to_bytes(n8, "little")
    hdr += struct.pack("<IIII", n_wires, n_pub_out, n_pub_in, n_wires - ni)
    hdr += struct.pack("<Q", n_wires) + struct.pack("<I", r1cs.num_constraints)
    body = bytearray()
    for i in range(r1cs.num_constraints):
        for rows in (r1cs.a, r1cs.b, r1cs.c):
            lc = rows[i]
            body += struct.pack("<I", len(lc))
            for coeff, wire in lc:
                body += struct.pack("<I", wire) + (coeff % prime).to_bytes(n8, "little")
    w2l = struct.pack("<%dQ" % n_wires, *range(n_wires))
    out = bytearray(R1CS_MAGIC + struct.pack("<II", 1, 3))
    for sid, sec in ((1, hdr), (2, bytes(body)), (3, w2l)):
        out += struct.pack("<IQ", sid, len(sec)) + sec
    if path:
        with open(path, "wb") as fh:
            fh.write(out)
    return bytes(out)


def read_wtns(src):
    """Returns (witness as a list of ints, prime)."""
    data = _load(src)
    version, secs = _sections(data, WTNS_MAGIC)
    if version != 2:
        raise ValueError("unsupported wtns version %d" % version)
    if 1 not in secs or 2 not in secs:
        raise ValueError("wtns file lacks a section")
    n8 = struct.unpack_from("<I", secs[1], 0)[0]
    prime = int.from_bytes(secs[1][4:4 + n8], "little")
    size = struct.unpack_from("<I", secs[1], 4 + n8)[0]
    if len(secs[2]) != size * n8:
        raise ValueError("witness section has %d bytes, expected %d" % (len(secs[2]), size * n8))
    raw = secs[2]
    return [int.from_bytes(raw[i * n8:(i + 1) * n8], "little") for i in range(size)], prime


def write_wtns(w, prime, path=None):
    n8 = (prime.bit_length() + 63) // 64 * 8
    s1 = struct.pack("<I", n8) + prime.to_bytes(n8, "little") + struct.pack("<I", len(w))
    s2 = b"".join((x % prime).to_bytes(n8, "little") for x in w)
    out = WTNS_MAGIC + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, len(s1)) + s1 + struct.pack("<IQ", 2, len(s2)) + s2
    if path:
        with open(path, "wb") as fh:
            fh.write(out)
    return out


def wtns_to_limbs(src, nl):
    """The witness section as a uint64 array [size][nl] of canonical little-endian limbs (no Python big ints)."""
    data = _load(src)
    _, secs = _sections(data, WTNS_MAGIC)
    n8 = struct.unpack_from("<I", secs[1], 0)[0]
    if n8 != nl * 8:
        raise ValueError("field size mismatch: file has %d bytes per element" % n8)
    return np.frombuffer(secs[2], dtype="<u8").reshape(-1, nl).copy()


class DeviceR1cs:
    """CSR copies of the A and B matrices on the device (C is not needed: the circom reduction sets c = a * b,
    qap.rs:66-70) and the QAP evaluation kernel."""

    def __init__(self, pp, r1cs):
        self.pp = pp
        self.num_constraints = r1cs.num_constraints
        self.num_instance_variables = r1cs.num_instance_variables
        self.num_variables = r1cs.num_variables
        nc, ni = r1cs.num_constraints, r1cs.num_instance_variables
        self.log_m = max(0, (nc + ni - 1).bit_length())
        self._mats = []
        for rows in (r1cs.a, r1cs.b):
            ptr = np.zeros(nc + 1, dtype=np.uint32)
            cols, vals = [], []
            for i, lc in enumerate(rows):
                for coeff, wire in lc:
                    cols.append(wire)
                    vals.append(coeff)
                ptr[i + 1] = len(cols)
            cols_d = pp.upload_u32(np.asarray(cols, dtype=np.uint32))
            ptr_d = pp.upload_u32(ptr)
            vals_d = pp.upload_fr(vals) if vals else pp.alloc_fr(1)
            self._mats.append((ptr_d, cols_d, vals_d))

    @classmethod
    def from_csr(cls, pp, num_constraints, num_instance_variables, num_variables, a_csr, b_csr):
        """a_csr / b_csr = (row_ptr uint32 [nc+1], cols uint32, vals uint64 [nnz][limbs] in Montgomery form):
        for matrices that are produced as arrays (synthetic circuits at scale) rather than Python lists."""
        self = cls.__new__(cls)
        self.pp = pp
        self.num_constraints, self.num_instance_variables = num_constraints, num_instance_variables
        self.num_variables = num_variables
        self.log_m = max(0, (num_constraints + num_instance_variables - 1).bit_length())
        self._mats = []
        for ptr, cols, vals in (a_csr, b_csr):
            if len(ptr) != num_constraints + 1 or int(ptr[-1]) != len(cols):
                raise ValueError("inconsistent CSR arrays")
            self._mats.append((pp.upload_u32(ptr), pp.upload_u32(cols),
                               DeviceBuffer.from_numpy(pp, np.ascontiguousarray(vals, dtype=np.uint64))
                               if len(cols) else pp.alloc_fr(1)))
        return self

    def qap(self, w_d, stream=None):
        """qap.rs:42-89: returns device vectors (a, b, c) of length m = 2^log_m in natural order."""
        pp = self.pp
        m = 1 << self.log_m
        a, b, c = pp.alloc_fr(m), pp.alloc_fr(m), pp.alloc_fr(m)
        (pa, ca, va), (pb, cb, vb) = self._mats
        pp._check(pp.lib.zk_r1cs_qap(pp.h, pa.ptr, ca.ptr, va.ptr, pb.ptr, cb.ptr, vb.ptr, w_d.ptr, self.num_variables,
                                     self.num_constraints, self.num_instance_variables, self.log_m, a.ptr, b.ptr, c.ptr,
                                     stream))
        return a, b, c
