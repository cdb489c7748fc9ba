// Curve configurations (one Engine<Cfg> instantiation each).
#pragma once
#include "params.hpp"
namespace zk {
struct CfgBn254 {   // ark-bn254: groth16/ runs over this curve (groth16/examples/sha256.rs:1)
  using FrP = Bn254Fr;
  using FqP = Bn254Fq;
  static constexpr bool HAS_G2 = true;
};
struct CfgBls381 {  // not a dependency of the reference; BASELINE config 5
  using FrP = Bls381Fr;
  using FqP = Bls381Fq;
  static constexpr bool HAS_G2 = true;
};
struct CfgBls377 {  // ark-bls12-377: what secret-sharing/ and dist-primitives/ tests use
  using FrP = Bls377Fr;
  using FqP = Bls377Fq;
  static constexpr bool HAS_G2 = false;   // Fq2 non-residue is -5; G2 is not on the reference's hot path
};
}  // namespace zk
