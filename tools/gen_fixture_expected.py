"""Writes tests/golden/fixture/<kind>_inputs.txt and <kind>_expected.txt: the committed half of the fork cross-check that
needs only cargo (rust/README.md, tools/diff_fixture.py).  The expectation is made by the CPU oracle; tests/test_gpu_fixture.py
checks on every GPU run that the GPU prover produces exactly these words, tests/test_oracle_golden.py that the oracle still does.
usage: python tools/gen_fixture_expected.py [g1 g2 fq]      (a few minutes on 8 cores)"""
import os
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from tests import oracle_lib
from tools import compare_fixture as cf
from tools import export_fixture_inputs as ex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "golden", "fixture")
CASES = {"g1": 3, "g2": 2, "fq": 5}
SEED = 0xF1C5
WIDTH_AUX = {"g1": (781, 456), "g2": (1295, 906), "fq": (427, 134)}


def section_offsets(kind, n_words, degree_bits):
    """(head length, final_poly offset, final_poly words): the word layout of include/bn254_stark.h for standard_fast_config."""
    w, a = WIDTH_AUX[kind]
    n_layers, d = 0, degree_bits
    while d > 5 and d + 1 - 4 >= 4:
        n_layers, d = n_layers + 1, d - 4
    head = 3 * 64 + 4 * (w + a) + 4 + 8 + 64 * n_layers
    n_final = 2 * (1 << d)
    return head, n_words - 13 - n_final, n_final


def expected_text(kind, n, trace, words, degree_bits):
    head, fp_off, n_final = section_offsets(kind, words.size, degree_bits)
    hx = lambda ws: "\n".join("%016x" % int(v) for v in ws)
    parts = ["kind %s inputs %d seed %x" % (kind, n, SEED), "trace %d %d" % trace.shape, hx(cf.column_digests(trace)),
             "words %d" % words.size,
             "section head 0 %d" % head, hx(words[:head]),
             "section final_poly %d %d" % (fp_off, n_final), hx(words[fp_off:fp_off + n_final]),
             "section pow_witness %d 1" % (words.size - 13), hx(words[-13:-12]),
             "section init_challenger_state %d 12" % (words.size - 12), hx(words[-12:])]
    return "\n".join(parts) + "\n"


def make(kind, lib):
    n = CASES[kind]
    s, x, o = ex.inputs(kind, n, SEED)
    k = ex.KINDS[kind]
    trace, _ = oracle_lib.generate_trace(lib, k, s, x, o)
    words, _, _, degree_bits = oracle_lib.prove(lib, k, s, x, o)
    rc, msg = oracle_lib.verify(lib, k, words, degree_bits, s, x, o)
    assert rc == 0, msg
    return s, x, o, expected_text(kind, n, trace, words, degree_bits), trace, words


if __name__ == "__main__":
    lib = oracle_lib.load()
    os.makedirs(DIR, exist_ok=True)
    for kind in (sys.argv[1:] or list(CASES)):
        s, x, o, text, _, _ = make(kind, lib)
        ex.write(os.path.join(DIR, kind + "_inputs.txt"), s, x, o)
        open(os.path.join(DIR, kind + "_expected.txt"), "w").write(text)
        print("wrote", kind, len(text), "bytes", flush=True)
