"""CPU tests: host-side logic and the C ABI surface (no compute calls without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

import plonky2_bn254_amd as pk
from tools import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "bn254_stark.h")).read()
    declared = sorted(set(re.findall(r"\b(bn254s_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    lib = ctypes.CDLL(pk.lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in bn254_stark.h but not exported"
    assert lib.bn254s_abi_version() == 1


def test_default_params_are_standard_fast_config():
    p = pk.default_params()
    assert (p.security_bits, p.num_challenges, p.rate_bits, p.cap_height, p.pow_bits, p.arity_bits, p.final_poly_bits,
            p.num_queries, p.min_rows_log2) == (100, 2, 1, 4, 16, 4, 5, 84, 16)
    assert p.struct_size == ctypes.sizeof(pk.lib.Params)


def test_context_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pk.Context(0)


def test_missing_library_is_an_error(monkeypatch):
    monkeypatch.setattr(pk.lib, "_lib", None)
    monkeypatch.setattr(pk.lib, "LIB_PATH", "/nonexistent/libbn254stark.so")
    with pytest.raises(pk.LibraryMissing):
        pk.load_library()


def test_synthetic_inputs_are_deterministic_and_on_curve():
    s1, x1, o1 = synth.g1_inputs(4)
    s2, x2, o2 = synth.g1_inputs(4)
    assert np.array_equal(s1, s2) and np.array_equal(x1, x2) and np.array_equal(o1, o2)
    for pt in list(x1) + list(o1):
        px, py = synth.words_to_int(pt[:4]), synth.words_to_int(pt[4:])
        assert (py * py - px * px * px - 3) % synth.P == 0
    assert synth.g1_mul(synth.R_ORDER - 1, synth.G1_GEN) == (1, synth.P - 2)


def test_oracle_proof_length_formula(oracle):
    # 3 caps, openings, 3 FRI caps, 84 query rounds, 16 final coefficients, pow witness, 12-word state
    W, A = 781, 456
    per_q = (W + A + 4) + 3 * 13 * 4 + (32 + 9 * 4) + (32 + 5 * 4) + (32 + 1 * 4)
    want = 3 * 64 + 2 * (2 * W + 2 * A) + 4 + 8 + 3 * 64 + 84 * per_q + 32 + 1 + 12
    assert oracle.orc_g1_proof_len(16) == want


def build_c_example(tmp_path):
    """examples/prove_g1.c with gcc -std=c99: the header is plain C and every entry point links without C++ or Python."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "prove_g1")
    libdir = os.path.join(root, "plonky2_bn254_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "prove_g1.c"), "-L" + libdir, "-lbn254stark", "-Wl,-rpath," + libdir, "-o", exe],
                   check=True)
    return exe


def test_c_example_builds_against_the_abi(tmp_path):
    import subprocess
    exe = build_c_example(tmp_path)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "bn254s_ctx_create" in r.stderr  # fails loudly without a GPU, no fallback


def test_ctl_values_rows_without_gpu():
    """bn254s_ctl_values is host-only: rows = 16-bit limbs of x | offset | scalar | timestamp and of the output | timestamp
    (scalar_mul_ctl.rs:57-80), checked against Python integers for G1 and the Fq-exp shape."""
    from plonky2_bn254_amd import lib as L
    lib = L.load_library()

    def limbs(words):
        v = synth.words_to_int(words)
        return [(v >> (16 * i)) & 0xFFFF for i in range(16)]

    s, x, o = synth.g1_inputs(3, seed=4)
    outs = np.arange(24, dtype=np.uint64).reshape(3, 8) * np.uint64(0x0123456789ABCDEF)
    ri, ro = np.zeros((3, 81), np.uint64), np.zeros((3, 33), np.uint64)
    assert lib.bn254s_ctl_values(0, L._ptr(s), L._ptr(x), L._ptr(o), L._ptr(outs), 3, L._ptr(ri), L._ptr(ro)) == 0
    for k in range(3):
        assert ri[k].tolist() == limbs(x[k, :4]) + limbs(x[k, 4:]) + limbs(o[k, :4]) + limbs(o[k, 4:]) + limbs(s[k]) + [k]
        assert ro[k].tolist() == limbs(outs[k, :4]) + limbs(outs[k, 4:]) + [k]
    fs, fx = synth.fq_inputs(2)
    fo = np.zeros((2, 4), np.uint64)
    fo[:, 0] = 1
    ri, ro = np.zeros((2, 33), np.uint64), np.zeros((2, 17), np.uint64)
    assert lib.bn254s_ctl_values(2, L._ptr(fs), L._ptr(fx), None, L._ptr(fo), 2, L._ptr(ri), L._ptr(ro)) == 0
    assert ri[1].tolist() == limbs(fx[1]) + limbs(fs[1]) + [1] and ro[1].tolist() == [1] + [0] * 15 + [1]
    assert lib.bn254s_ctl_values(0, L._ptr(s), L._ptr(x), None, L._ptr(outs), 3, L._ptr(ri), L._ptr(ro)) == -1


def test_fixture_tools_column_digest_and_io_round_trip(tmp_path):
    """tools/compare_fixture.py: the numpy column digest equals the fold rust/shim/dump_fixture.rs computes, and the dump /
    input text formats survive a round trip (the GPU part of the comparison is rehearsed in tests/test_gpu_fixture.py)."""
    from tools import compare_fixture as cf
    from tools import export_fixture_inputs as ex
    t = np.array([[1, 2, 3], [5, 0, 0xFFFFFFFFFFFFFFFF]], dtype=np.uint64)
    want = []
    for col in t:
        d = 0
        for v in col:
            d = (d * cf.DIGEST_K + int(v)) & 0xFFFFFFFFFFFFFFFF
        want.append(d)
    assert [int(v) for v in cf.column_digests(t)] == want
    words = np.arange(40, dtype=np.uint64) * np.uint64(0x0123456789ABCDEF)
    cf.write_dump(str(tmp_path / "d.txt"), t, words)
    nrows, dig, w = cf.parse_dump(str(tmp_path / "d.txt"))
    assert nrows == 3 and [int(v) for v in dig] == want and np.array_equal(w, words)
    for kind in ("g1", "fq"):
        s, x, o = ex.inputs(kind, 2, 0xF1C5)
        ex.write(str(tmp_path / "in.txt"), s, x, o)
        s2, x2, o2 = ex.read(str(tmp_path / "in.txt"), kind)
        assert np.array_equal(s, s2) and np.array_equal(x, x2) and (o is None) == (o2 is None)
