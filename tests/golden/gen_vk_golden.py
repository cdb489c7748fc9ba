"""Copies the NUMBERS of the reference's snarkjs verification key (fixtures/verification_key.json: BN254 points and the
golden pairing value vk_alphabeta_12 = e(vk_alpha_1, vk_beta_2)) into tests/golden/verification_key_bn254.json.
Run in the build container (the reference checkout does not exist on the GPU box):
    python tests/golden/gen_vk_golden.py
"""
import json
import os

SRC = "/root/reference/fixtures/verification_key.json"
HERE = os.path.dirname(os.path.abspath(__file__))

d = json.load(open(SRC))
out = {
    "source": "tangle-network/zk-SaaS fixtures/verification_key.json (snarkjs groth16 vk of the sha256 circuit)",
    "protocol": d["protocol"], "curve": d["curve"], "nPublic": d["nPublic"],
    "vk_alpha_1": d["vk_alpha_1"], "vk_beta_2": d["vk_beta_2"], "vk_gamma_2": d["vk_gamma_2"],
    "vk_delta_2": d["vk_delta_2"], "vk_alphabeta_12": d["vk_alphabeta_12"], "IC": d["IC"],
}
with open(os.path.join(HERE, "verification_key_bn254.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote verification_key_bn254.json")
