"""ctypes loader for the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None
VP = C.c_void_p


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(PATH)
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_gl_mul.restype = C.c_uint64
    lib.orc_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
    lib.orc_gl_inv.restype = C.c_uint64
    lib.orc_gl_inv.argtypes = [C.c_uint64]
    lib.orc_poseidon_permute.argtypes = [VP]
    lib.orc_hash_or_noop.argtypes = [VP, C.c_size_t, VP]
    lib.orc_two_to_one.argtypes = [VP, VP, VP]
    lib.orc_merkle_cap.argtypes = [VP, C.c_size_t, C.c_size_t, C.c_int, VP]
    lib.orc_challenger_observe_get.argtypes = [VP, C.c_size_t, VP, C.c_size_t]
    lib.orc_ntt.argtypes = [VP, C.c_size_t, C.c_int, C.c_uint64]
    lib.orc_commit_values.argtypes = [VP, C.c_size_t, C.c_size_t, VP, VP, VP]
    lib.orc_fq_mul.argtypes = [VP, VP, VP]
    lib.orc_fq_inv.argtypes = [VP, VP]
    lib.orc_g1_add.argtypes = [VP, VP, VP]
    lib.orc_generate_modulus_zero.argtypes = [VP, VP]
    lib.orc_g1_num_rows.restype = C.c_size_t
    lib.orc_g1_num_rows.argtypes = [C.c_size_t, C.c_int]
    lib.orc_g1_generate_trace.argtypes = [VP, VP, VP, C.c_size_t, C.c_int, VP, VP]
    lib.orc_g1_eval_constraints.argtypes = [VP, VP, VP, C.c_uint64, C.c_uint64, C.c_uint64, VP]
    lib.orc_g1_proof_len.restype = C.c_size_t
    lib.orc_g1_proof_len.argtypes = [C.c_int]
    lib.orc_g1_prove.argtypes = [VP, VP, VP, C.c_size_t, C.c_int, VP, C.c_size_t, VP, VP]
    lib.orc_g1_verify.argtypes = [VP, C.c_size_t, C.c_int, VP, VP, VP, C.c_size_t]
    lib.orc_stark_width.argtypes = [C.c_int]
    lib.orc_proof_len.restype = C.c_size_t
    lib.orc_proof_len.argtypes = [C.c_int, C.c_int]
    lib.orc_generate_trace.argtypes = [C.c_int, VP, VP, VP, C.c_size_t, C.c_int, VP, VP]
    lib.orc_prove.argtypes = [C.c_int, VP, VP, VP, C.c_size_t, C.c_int, VP, C.c_size_t, VP, VP]
    lib.orc_verify.argtypes = [C.c_int, VP, C.c_size_t, C.c_int, VP, VP, VP, C.c_size_t]
    lib.orc_eval_constraints.argtypes = [C.c_int, VP, VP, VP, C.c_uint64, C.c_uint64, C.c_uint64, VP]
    _lib = lib
    return lib


def ptr(a):
    return a.ctypes.data_as(VP) if a is not None else None


def commit_values(lib, values):
    C_, N = values.shape
    coeffs = np.zeros((C_, N), np.uint64)
    lde = np.zeros((C_, 2 * N), np.uint64)
    cap = np.zeros((16, 4), np.uint64)
    lib.orc_commit_values(ptr(values), C_, N, ptr(coeffs), ptr(lde), ptr(cap))
    return coeffs, lde, cap


def g1_prove(lib, s, x, o, min_rows_log2=16):
    n = s.shape[0]
    rows = lib.orc_g1_num_rows(n, min_rows_log2)
    degree_bits = rows.bit_length() - 1
    plen = lib.orc_g1_proof_len(degree_bits)
    proof = np.zeros(plen, np.uint64)
    outs = np.zeros((n, 8), np.uint64)
    tm = np.zeros(8)
    r = lib.orc_g1_prove(ptr(s), ptr(x), ptr(o), n, min_rows_log2, ptr(proof), plen, ptr(outs), tm.ctypes.data_as(VP))
    if r < 0:
        raise RuntimeError(lib.orc_last_error().decode())
    return proof[:r], outs, tm, degree_bits


def g1_verify(lib, proof, degree_bits, s, x, o):
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    r = lib.orc_g1_verify(ptr(proof), proof.shape[0], degree_bits, ptr(s), ptr(x), ptr(o), s.shape[0])
    return r, lib.orc_last_error().decode()


POINT_WORDS = {0: 8, 1: 16, 2: 4}


def generate_trace(lib, kind, s, x, o=None, min_rows_log2=16):
    n = s.shape[0]
    rows = lib.orc_g1_num_rows(n, min_rows_log2)
    tr = np.zeros((lib.orc_stark_width(kind), rows), np.uint64)
    outs = np.zeros((n, POINT_WORDS[kind]), np.uint64)
    rc = lib.orc_generate_trace(kind, ptr(s), ptr(x), ptr(o), n, min_rows_log2, ptr(tr), ptr(outs))
    if rc != 0:
        raise RuntimeError(lib.orc_last_error().decode())
    return tr, outs


def prove(lib, kind, s, x, o=None, min_rows_log2=16):
    n = s.shape[0]
    rows = lib.orc_g1_num_rows(n, min_rows_log2)
    degree_bits = rows.bit_length() - 1
    plen = lib.orc_proof_len(kind, degree_bits)
    proof = np.zeros(plen, np.uint64)
    outs = np.zeros((n, POINT_WORDS[kind]), np.uint64)
    tm = np.zeros(8)
    r = lib.orc_prove(kind, ptr(s), ptr(x), ptr(o), n, min_rows_log2, ptr(proof), plen, ptr(outs), tm.ctypes.data_as(VP))
    if r < 0:
        raise RuntimeError(lib.orc_last_error().decode())
    return proof[:r], outs, tm, degree_bits


def verify(lib, kind, proof, degree_bits, s, x, o=None):
    proof = np.ascontiguousarray(proof, dtype=np.uint64)
    r = lib.orc_verify(kind, ptr(proof), proof.shape[0], degree_bits, ptr(s), ptr(x), ptr(o), s.shape[0])
    return r, lib.orc_last_error().decode()
