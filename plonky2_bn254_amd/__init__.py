"""MI355X-native prover for the BN254 scalar-multiplication STARKs of plonky2_bn254.

The product is `libbn254stark.so` (hand-written HIP kernels for gfx950 behind the C ABI declared in
include/bn254_stark.h).  This package is the thin host-side mirror used by tests and bench.py:
`Context.prove_g1(...)` corresponds to the body of the reference's
`G1StarkProofGenerator::run_once` (src/generators/g1/stark_proof.rs:136-179, lines 143-163).
"""
import os as _os

# one hardware queue per proof in flight (see csrc/capi.hip); must be in the environment before the HIP runtime starts
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .lib import Context, Proof, load_library, LibraryMissing, VerifyError, default_params, prove_batch_multi, verify_host  # noqa: F401
