"""Golden vectors for whole proofs: SHA-256 digests of the CPU oracle's proof transcripts on fixed synthetic inputs
(SURVEY.md section 8(c) strategy (ii)).  The reference holds no fixture of its own (no seeded test), so these pin the oracle's
output: `tests/test_gpu_golden.py` compares the GPU proofs against them without running the oracle.

usage: python tools/gen_proof_golden.py > tests/golden/proof_digests.json   (about ten minutes on 8 cores)
       python tools/gen_proof_golden.py --only g1_tall19 > one_case.json     (a single case; g1_tall19 = BASELINE configs[1] as
       ONE proof of 2^19 rows, the shape Bn254Hook::constrain produces for 1024 calls: minutes on a many-core host, ~25 GB)"""
import hashlib
import json
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tools import synth
from tests import oracle_lib


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def main():
    lib = oracle_lib.load()
    cases = [("g1", 0, synth.g1_inputs, 2, 0xA1), ("g1_full", 0, synth.g1_inputs, 128, 0xA2), ("g2", 1, synth.g2_inputs, 2, 0xA3),
             ("fq_exp", 2, synth.fq_inputs, 3, 0xA4), ("g1_tall19", 0, synth.g1_inputs, 1024, 0xA5)]
    if len(sys.argv) > 2 and sys.argv[1] == "--only":
        cases = [c for c in cases if c[0] == sys.argv[2]]
    else:
        cases = [c for c in cases if c[0] != "g1_tall19"]      # the tall case is generated on its own
    out = {}
    for name, kind, gen, n, seed in cases:
        ins = gen(n, seed=seed)
        off = ins[2] if len(ins) > 2 else None
        proof, outs, _, degree_bits = oracle_lib.prove(lib, kind, ins[0], ins[1], off)
        rc, msg = oracle_lib.verify(lib, kind, proof, degree_bits, ins[0], ins[1], off)
        assert rc == 0, msg
        out[name] = {"kind": kind, "n": n, "seed": seed, "degree_bits": int(degree_bits), "n_words": int(proof.size),
                     "sha256_proof_words": digest(proof), "sha256_outputs": digest(outs),
                     "sha256_trace_cap": digest(proof[:64]), "sha256_aux_cap": digest(proof[64:128]),
                     "sha256_quotient_cap": digest(proof[128:192]), "pow_witness": int(proof[-13]),
                     "first_words": [int(w) for w in proof[:4]]}
        print(name, "done", file=sys.stderr, flush=True)
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
