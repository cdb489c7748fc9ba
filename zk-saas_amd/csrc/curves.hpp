// Curve configurations (one Engine<Cfg> instantiation each).
#pragma once
#include "params.hpp"
namespace zk {
struct CfgBn254 {   // ark-bn254: groth16/ runs over this curve (groth16/examples/sha256.rs:1)
  using FrP = Bn254Fr;
  using FqP = Bn254Fq;
  static constexpr bool HAS_G2 = true;
  // G1: y^2 = x^3 + 3; G2 on the D-type twist y^2 = x^3 + 3 / (9 + u); ark-ec's default point encoding
  static constexpr int B1 = 3, XI0 = 9, XI1 = 1;
  static constexpr bool TWIST_MUL = false, ZCASH = false, SQRT_3MOD4 = true;
};
struct CfgBls381 {  // not a dependency of the reference; BASELINE config 5
  using FrP = Bls381Fr;
  using FqP = Bls381Fq;
  static constexpr bool HAS_G2 = true;
  // G1: y^2 = x^3 + 4; G2 on the M-type twist y^2 = x^3 + 4 (1 + u); zcash / IETF point encoding (ark-bls12-381)
  static constexpr int B1 = 4, XI0 = 1, XI1 = 1;
  static constexpr bool TWIST_MUL = true, ZCASH = true, SQRT_3MOD4 = true;
};
struct CfgBls377 {  // ark-bls12-377: what secret-sharing/ and dist-primitives/ tests use
  using FrP = Bls377Fr;
  using FqP = Bls377Fq;
  static constexpr bool HAS_G2 = false;   // Fq2 non-residue is -5; G2 is not on the reference's hot path
  static constexpr int B1 = 1, XI0 = 0, XI1 = 1;
  static constexpr bool TWIST_MUL = false, ZCASH = false, SQRT_3MOD4 = false;   // q = 1 mod 4
};
}  // namespace zk
