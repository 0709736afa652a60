"""ctypes binding of libbn254stark.so (include/bn254_stark.h).  No CPU fallback: if the HIP library is
missing or a call fails, an exception is raised."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BN254S_LIB: another build of the same library (A/B measurements of compile-time knobs, tools/ubench/ab/); never a fallback
LIB_PATH = os.environ.get("BN254S_LIB") or os.path.join(_HERE, "libbn254stark.so")

U64P = C.POINTER(C.c_uint64)


class LibraryMissing(RuntimeError):
    pass


class VerifyError(RuntimeError):
    """bn254s_verify rejected the proof; the message is the reference verifier's error text."""


class Params(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "security_bits", "num_challenges", "rate_bits", "cap_height", "pow_bits",
        "arity_bits", "final_poly_bits", "num_queries", "min_rows_log2")]


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the proving path)")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.bn254s_abi_version.restype = C.c_int
    lib.bn254s_params_default.argtypes = [C.POINTER(Params)]
    lib.bn254s_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.bn254s_ctx_destroy.argtypes = [vp]
    lib.bn254s_ctx_trim.argtypes = [vp]
    lib.bn254s_last_error.argtypes = [vp]
    lib.bn254s_last_error.restype = C.c_char_p
    for name in ("bn254s_prove_g1", "bn254s_prove_g2"):
        getattr(lib, name).argtypes = [vp, C.POINTER(Params), vp, vp, vp, C.c_size_t, C.POINTER(vp)]
    lib.bn254s_prove_fq_exp.argtypes = [vp, C.POINTER(Params), vp, vp, C.c_size_t, C.POINTER(vp)]
    lib.bn254s_generate_trace.argtypes = [vp, C.c_int, vp, vp, vp, C.c_size_t, C.c_uint32, vp, vp]
    lib.bn254s_prove_g1_batch.argtypes = [vp, C.POINTER(Params), vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    lib.bn254s_prove_batch.argtypes = [vp, C.c_int, C.POINTER(Params), vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    lib.bn254s_prove_batch_begin.argtypes = [vp, C.c_int, C.POINTER(Params), vp, vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp),
                                             C.POINTER(vp)]
    lib.bn254s_prove_batch_end.argtypes = [vp]
    lib.bn254s_prove_batch_multi.argtypes = [C.POINTER(vp), C.c_size_t, C.c_int, C.POINTER(Params), vp, vp, vp, C.c_size_t,
                                             C.c_size_t, C.POINTER(vp)]
    lib.bn254s_proof_words.argtypes = [vp, C.POINTER(U64P), C.POINTER(C.c_size_t)]
    lib.bn254s_proof_outputs.argtypes = [vp, C.POINTER(U64P), C.POINTER(C.c_size_t)]
    lib.bn254s_proof_degree_bits.argtypes = [vp]
    lib.bn254s_proof_section.argtypes = [vp, C.c_int, C.POINTER(U64P), C.POINTER(C.c_size_t)]
    lib.bn254s_proof_stage_ms.argtypes = [vp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t)]
    lib.bn254s_stage_name.argtypes = [C.c_size_t]
    lib.bn254s_stage_name.restype = C.c_char_p
    lib.bn254s_proof_free.argtypes = [vp]
    lib.bn254s_proof_serialize.argtypes = [vp, vp, C.c_size_t]
    lib.bn254s_proof_serialize.restype = C.c_size_t
    lib.bn254s_verify.argtypes = [vp, C.c_int, C.POINTER(Params), C.c_uint32, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t]
    lib.bn254s_verify_host.argtypes = [C.c_int, C.POINTER(Params), C.c_uint32, vp, C.c_size_t, vp, vp, vp, vp, C.c_size_t, C.c_char_p,
                                       C.c_size_t]
    lib.bn254s_map_to_g2.argtypes = [vp, C.POINTER(Params), vp, vp, C.c_size_t, vp, vp, vp, C.POINTER(vp), C.POINTER(vp)]
    lib.bn254s_hash_to_fq2.argtypes = [vp, C.c_size_t, vp]
    lib.bn254s_hash_to_fq2_batch.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp]
    lib.bn254s_ctl_values.argtypes = [C.c_int, vp, vp, vp, vp, C.c_size_t, vp, vp]
    lib.bn254s_commit_values.argtypes = [vp, vp, C.c_size_t, vp, vp, vp]
    lib.bn254s_bench_ntt.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_float)]
    lib.bn254s_bench_ntt_clock.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.bn254s_bench_issue.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.bn254s_poseidon_permute.argtypes = [vp, vp, C.c_size_t]
    lib.bn254s_selftest_field.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.bn254s_selftest_fq_inv.argtypes = [vp, vp, C.c_size_t, vp]
    lib.bn254s_bench_copy.argtypes = [vp, C.c_size_t, C.c_int]
    lib.bn254s_bench_leafhash.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.bn254s_g1_generate_trace.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_uint32, vp, vp]
    _lib = lib
    return lib


def default_params() -> Params:
    p = Params()
    load_library().bn254s_params_default(C.byref(p))
    return p


def _ptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Proof:
    """Owns a bn254s_proof*; `words` is the canonical u64 layout documented in bn254_stark.h."""

    _stage_names = None

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle
        self._words = self._outputs = None
        self.degree_bits = lib.bn254s_proof_degree_bits(handle)
        ms, k = C.POINTER(C.c_float)(), C.c_size_t()
        lib.bn254s_proof_stage_ms(handle, C.byref(ms), C.byref(k))
        if Proof._stage_names is None or len(Proof._stage_names) != k.value:
            Proof._stage_names = [lib.bn254s_stage_name(i).decode() for i in range(k.value)]
        self.stage_ms = dict(zip(Proof._stage_names, ms[:k.value]))

    # the 1.1 MB of proof words are copied out of the library's buffer on first use (a throughput loop that only needs the caps
    # and the stage times does not pay for eight copies per step)
    def _fetch(self, fn, what):
        if not self._h:
            raise RuntimeError(f"Proof.{what}: the proof was closed before its {what} were read")
        data, n = U64P(), C.c_size_t()
        rc = fn(self._h, C.byref(data), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"bn254s_proof_{what} failed with {rc}")
        return data, n.value

    @property
    def words(self) -> np.ndarray:
        if self._words is None:
            data, n = self._fetch(self._lib.bn254s_proof_words, "words")
            self._words = np.ctypeslib.as_array(data, shape=(n,)).copy()
        return self._words

    @property
    def outputs(self) -> np.ndarray:
        if self._outputs is None:
            data, n = self._fetch(self._lib.bn254s_proof_outputs, "outputs")
            self._outputs = np.ctypeslib.as_array(data, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        return self._outputs

    def caps(self) -> np.ndarray:
        """The three Merkle caps (trace, auxiliary, quotient): the first 192 words of the proof."""
        if self._words is not None:
            return self._words[:192].copy()
        data, n = self._fetch(self._lib.bn254s_proof_words, "words")
        assert n >= 192
        return np.ctypeslib.as_array(data, shape=(192,)).copy()

    SECTIONS = ("trace_cap", "auxiliary_polys_cap", "quotient_polys_cap", "local_values", "next_values", "auxiliary_polys",
                "auxiliary_polys_next", "ctl_zs_first", "quotient_polys", "commit_phase_merkle_caps", "query_round_proofs",
                "final_poly", "pow_witness", "init_challenger_state")

    def section(self, name: str) -> np.ndarray:
        """One field of StarkProofWithMetadata (bn254s_proof_section): a copy of its words."""
        data, n = U64P(), C.c_size_t()
        rc = self._lib.bn254s_proof_section(self._h, self.SECTIONS.index(name), C.byref(data), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"bn254s_proof_section({name}) failed with {rc}")
        return np.ctypeslib.as_array(data, shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint64)

    def serialize(self) -> bytes:
        """Little-endian bytes of the word layout (bn254s_proof_serialize)."""
        need = self._lib.bn254s_proof_serialize(self._h, None, 0)
        buf = C.create_string_buffer(need)
        assert self._lib.bn254s_proof_serialize(self._h, buf, need) == need
        return buf.raw

    def close(self):
        """Frees the library's copy; `words` and `outputs` stay readable (they are copied out first)."""
        if self._h:
            try:
                self.words, self.outputs
            finally:
                self._lib.bn254s_proof_free(self._h)
                self._h = None

    def __del__(self):
        if getattr(self, "_h", None):  # (no copy on garbage collection: nobody can read it afterwards)
            self._lib.bn254s_proof_free(self._h)
            self._h = None


class Context:
    """One context per GPU (bn254s_ctx)."""

    def __init__(self, device: int = 0):
        self._h = None
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.bn254s_ctx_create(device, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"bn254s_ctx_create(device={device}) failed with {rc} (is a GPU visible?)")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            self._lib.bn254s_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def trim(self):
        """bn254s_ctx_trim: idle slots give their (grow-only) workspaces back to the driver."""
        self._check(self._lib.bn254s_ctx_trim(self._h), "bn254s_ctx_trim")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with {rc}: {self._lib.bn254s_last_error(self._h).decode()}")

    # ---- proving (mirrors run_once of the reference generators) ----
    def prove_g1(self, scalars, x, offset, params: Optional[Params] = None) -> Proof:
        params = params or default_params()
        n = scalars.shape[0]
        out = C.c_void_p()
        self._check(self._lib.bn254s_prove_g1(self._h, C.byref(params), _ptr(scalars), _ptr(x), _ptr(offset), n,
                                              C.byref(out)), "bn254s_prove_g1")
        return Proof(self._lib, out)

    def prove_g2(self, scalars, x, offset, params: Optional[Params] = None) -> Proof:
        """G2 scalar multiplications (points as 16 words x.c0, x.c1, y.c0, y.c1): run_once of G2StarkProofGenerator."""
        params = params or default_params()
        out = C.c_void_p()
        self._check(self._lib.bn254s_prove_g2(self._h, C.byref(params), _ptr(scalars), _ptr(x), _ptr(offset),
                                              scalars.shape[0], C.byref(out)), "bn254s_prove_g2")
        return Proof(self._lib, out)

    def prove_fq_exp(self, scalars, x, params: Optional[Params] = None) -> Proof:
        """Fq exponentiations x^s: run_once of FqStarkProofGenerator (src/generators/fq/stark_proof.rs:135-178)."""
        params = params or default_params()
        out = C.c_void_p()
        self._check(self._lib.bn254s_prove_fq_exp(self._h, C.byref(params), _ptr(scalars), _ptr(x), scalars.shape[0],
                                                  C.byref(out)), "bn254s_prove_fq_exp")
        return Proof(self._lib, out)

    def prove_g1_batch(self, scalars, x, offset, per_proof=128, params: Optional[Params] = None, keep=True):
        params = params or default_params()
        n = scalars.shape[0]
        k = (n + per_proof - 1) // per_proof
        outs = (C.c_void_p * k)()
        self._check(self._lib.bn254s_prove_g1_batch(self._h, C.byref(params), _ptr(scalars), _ptr(x), _ptr(offset), n,
                                                    per_proof, outs), "bn254s_prove_g1_batch")
        proofs = [Proof(self._lib, C.c_void_p(outs[i])) for i in range(k)]
        return proofs

    def prove_batch(self, kind, scalars, x, offset=None, per_proof=128, params: Optional[Params] = None):
        """kind 0 = G1, 1 = G2, 2 = Fq exp: n jobs cut into independent 128-instance proofs, pipelined on the GPU."""
        params = params or default_params()
        n = scalars.shape[0]
        k = (n + per_proof - 1) // per_proof
        outs = (C.c_void_p * k)()
        self._check(self._lib.bn254s_prove_batch(self._h, kind, C.byref(params), _ptr(scalars), _ptr(x), _ptr(offset), n,
                                                 per_proof, outs), "bn254s_prove_batch")
        return [Proof(self._lib, C.c_void_p(outs[i])) for i in range(k)]

    def prove_batch_begin(self, kind, scalars, x, offset=None, per_proof=128, params: Optional[Params] = None):
        """bn254s_prove_batch_begin: queues the batch and returns a handle; `handle.end()` waits and returns the proofs.  Batches
        run in the order they were begun, so beginning the next one before ending the current one keeps the GPU busy."""
        return BatchInFlight(self, kind, scalars, x, offset, per_proof, params or default_params())

    def verify(self, kind, words, degree_bits, scalars, x, offset, outputs, params: Optional[Params] = None):
        """Native `verify` (src/starks/common/verifier.rs:32-98 + CTL check): returns None or raises VerifyError(reason)."""
        params = params or default_params()
        words = np.ascontiguousarray(words, dtype=np.uint64)
        outputs = np.ascontiguousarray(outputs, dtype=np.uint64)
        rc = self._lib.bn254s_verify(self._h, kind, C.byref(params), degree_bits, _ptr(words), words.size, _ptr(scalars), _ptr(x),
                                     _ptr(offset), _ptr(outputs), scalars.shape[0])
        if rc == -8:
            raise VerifyError(self._lib.bn254s_last_error(self._h).decode())
        self._check(rc, "bn254s_verify")

    def map_to_g2(self, u, offsets, params: Optional[Params] = None):
        """u [n,8], offsets [n,16] -> (points [n,16], fq_jobs [2n,8], g2_jobs [n,20], fq proofs, g2 proofs): the config-5
        pipeline on the device (csrc/map_to_g2.hip)."""
        params = params or default_params()
        n = u.shape[0]
        pts = np.zeros((n, 16), np.uint64)
        fq_jobs = np.zeros((2 * n, 8), np.uint64)
        g2_jobs = np.zeros((n, 20), np.uint64)
        n_fq, n_g2 = (2 * n + 127) // 128, (n + 127) // 128
        pf, pg = (C.c_void_p * n_fq)(), (C.c_void_p * n_g2)()
        self._check(self._lib.bn254s_map_to_g2(self._h, C.byref(params), _ptr(u), _ptr(offsets), n, _ptr(pts), _ptr(fq_jobs),
                                               _ptr(g2_jobs), pf, pg), "bn254s_map_to_g2")
        return (pts, fq_jobs, g2_jobs, [Proof(self._lib, C.c_void_p(pf[i])) for i in range(n_fq)],
                [Proof(self._lib, C.c_void_p(pg[i])) for i in range(n_g2)])

    def hash_to_fq2_batch(self, inputs: np.ndarray) -> np.ndarray:
        """inputs [n, len] Goldilocks elements -> u [n, 8]: hash_to_fq2 (hash_to_g2.rs:76-87) of every row, on the device."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        n, ln = inputs.shape
        out = np.zeros((n, 8), np.uint64)
        self._check(self._lib.bn254s_hash_to_fq2_batch(self._h, _ptr(inputs) if ln else None, n, ln, _ptr(out)), "bn254s_hash_to_fq2_batch")
        return out

    def ctl_values(self, kind, scalars, x, offset, outputs):
        """(input rows [n, 81|145|33], output rows [n, 33|65|17]): the extra looking values of the two CTLs."""
        n = scalars.shape[0]
        pl = {0: 32, 1: 64, 2: 16}[kind]
        rows_in = np.zeros((n, (pl if kind == 2 else 2 * pl) + 17), np.uint64)
        rows_out = np.zeros((n, pl + 1), np.uint64)
        outputs = np.ascontiguousarray(outputs, dtype=np.uint64)
        self._check(self._lib.bn254s_ctl_values(kind, _ptr(scalars), _ptr(x), _ptr(offset), _ptr(outputs), n, _ptr(rows_in),
                                                _ptr(rows_out)), "bn254s_ctl_values")
        return rows_in, rows_out

    # ---- kernel-level entry points ----
    def commit_values(self, values: np.ndarray, want_coeffs=True, want_lde=True):
        ncols, n = values.shape
        assert n == 65536
        coeffs = np.zeros((ncols, n), np.uint64) if want_coeffs else None
        lde = np.zeros((ncols, 2 * n), np.uint64) if want_lde else None
        cap = np.zeros((16, 4), np.uint64)
        self._check(self._lib.bn254s_commit_values(self._h, _ptr(values), ncols, _ptr(coeffs), _ptr(lde), _ptr(cap)),
                    "bn254s_commit_values")
        return coeffs, lde, cap

    def bench_ntt(self, ncols: int, iters: int = 10) -> float:
        ms = C.c_float()
        self._check(self._lib.bn254s_bench_ntt(self._h, ncols, iters, C.byref(ms)), "bn254s_bench_ntt")
        return ms.value

    def bench_ntt_clock(self, ncols: int, iters: int = 10):
        """(ms per stage, mean shader MHz, slowest-interval MHz) of the NTT/LDE stage alone on the GPU."""
        ms, mhz, mn = C.c_float(), C.c_float(), C.c_float()
        self._check(self._lib.bn254s_bench_ntt_clock(self._h, ncols, iters, C.byref(ms), C.byref(mhz), C.byref(mn)), "bn254s_bench_ntt_clock")
        return ms.value, mhz.value, mn.value

    def bench_issue(self):
        """(ns per wave-instruction and SIMD of the half-rate vector class at 8 waves per SIMD, shader MHz meanwhile)."""
        ns, mhz = C.c_float(), C.c_float()
        self._check(self._lib.bn254s_bench_issue(self._h, C.byref(ns), C.byref(mhz)), "bn254s_bench_issue")
        return ns.value, mhz.value

    def bench_copy(self, words: int, iters: int = 3):
        self._check(self._lib.bn254s_bench_copy(self._h, words, iters), "bn254s_bench_copy")

    def bench_leafhash(self, ncols: int, log_leaves: int = 17, iters: int = 5) -> float:
        ms = C.c_float()
        self._check(self._lib.bn254s_bench_leafhash(self._h, ncols, log_leaves, iters, C.byref(ms)), "bn254s_bench_leafhash")
        return ms.value

    def selftest_field(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        """Debug: the hand-written field sequences (csrc/gl_asm.h) on operand pairs; returns out[n][17]."""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros((a.shape[0], 17), np.uint64)
        self._check(self._lib.bn254s_selftest_field(self._h, _ptr(a), _ptr(b), a.shape[0], _ptr(out)), "bn254s_selftest_field")
        return out

    def selftest_fq_inv(self, x: np.ndarray) -> np.ndarray:
        """Debug: x[n][4] canonical words -> out[n][8] = x^-1 mod p by divsteps and by Fermat (bn254s_selftest_fq_inv)."""
        x = np.ascontiguousarray(x, dtype=np.uint64)
        out = np.zeros((x.shape[0], 8), np.uint64)
        self._check(self._lib.bn254s_selftest_fq_inv(self._h, _ptr(x), x.shape[0], _ptr(out)), "bn254s_selftest_fq_inv")
        return out

    def poseidon_permute(self, states: np.ndarray) -> np.ndarray:
        st = np.ascontiguousarray(states, dtype=np.uint64).copy()
        self._check(self._lib.bn254s_poseidon_permute(self._h, _ptr(st), st.shape[0]), "bn254s_poseidon_permute")
        return st

    def generate_trace(self, kind, scalars, x, offset=None, min_rows_log2=16):
        """kind 0 = G1, 1 = G2, 2 = Fq exp.  Returns (trace[W, rows], outputs[n, 8|16|4])."""
        width, pw = {0: (781, 8), 1: (1295, 16), 2: (427, 4)}[kind]
        n = scalars.shape[0]
        rows = max(1 << min_rows_log2, 512 * n)
        rows = 1 << (rows - 1).bit_length()
        trace = np.zeros((width, rows), np.uint64)
        outs = np.zeros((n, pw), np.uint64)
        self._check(self._lib.bn254s_generate_trace(self._h, kind, _ptr(scalars), _ptr(x), _ptr(offset), n, min_rows_log2,
                                                    _ptr(trace), _ptr(outs)), "bn254s_generate_trace")
        return trace, outs

    def g1_generate_trace(self, scalars, x, offset, min_rows_log2=16, width=781):
        n = scalars.shape[0]
        rows = max(1 << min_rows_log2, 512 * n)
        rows = 1 << (rows - 1).bit_length()
        trace = np.zeros((width, rows), np.uint64)
        outs = np.zeros((n, 8), np.uint64)
        self._check(self._lib.bn254s_g1_generate_trace(self._h, _ptr(scalars), _ptr(x), _ptr(offset), n, min_rows_log2,
                                                       _ptr(trace), _ptr(outs)), "bn254s_g1_generate_trace")
        return trace, outs


def verify_host(kind, words, degree_bits, scalars, x, offset, outputs, params: Optional[Params] = None):
    """bn254s_verify_host: the native verifier without a context or a GPU (the AIR is evaluated on the host over the quadratic
    extension).  Returns None or raises VerifyError(reason)."""
    lib = load_library()
    params = params or default_params()
    words = np.ascontiguousarray(words, dtype=np.uint64)
    outputs = np.ascontiguousarray(outputs, dtype=np.uint64)
    buf = C.create_string_buffer(256)
    rc = lib.bn254s_verify_host(kind, C.byref(params), degree_bits, _ptr(words), words.size, _ptr(scalars), _ptr(x), _ptr(offset),
                                _ptr(outputs), scalars.shape[0], buf, 256)
    if rc == -8:
        raise VerifyError(buf.value.decode())
    if rc != 0:
        raise RuntimeError(f"bn254s_verify_host failed with {rc}: {buf.value.decode()}")


class BatchInFlight:
    """A batch between bn254s_prove_batch_begin and bn254s_prove_batch_end (keeps the input arrays alive meanwhile)."""

    def __init__(self, ctx, kind, scalars, x, offset, per_proof, params):
        self._ctx = ctx
        self._keep = (scalars, x, offset, params)
        n = scalars.shape[0]
        self._k = (n + per_proof - 1) // per_proof
        self._outs = (C.c_void_p * self._k)()
        self._h = C.c_void_p()
        ctx._check(ctx._lib.bn254s_prove_batch_begin(ctx._h, kind, C.byref(params), _ptr(scalars), _ptr(x), _ptr(offset), n, per_proof,
                                                     self._outs, C.byref(self._h)), "bn254s_prove_batch_begin")

    def end(self):
        if self._h is None:
            raise RuntimeError("batch already ended")
        h, self._h = self._h, None
        self._ctx._check(self._ctx._lib.bn254s_prove_batch_end(h), "bn254s_prove_batch_end")
        self._keep = None
        return [Proof(self._ctx._lib, C.c_void_p(self._outs[i])) for i in range(self._k)]

    def __del__(self):
        if getattr(self, "_h", None) is not None:  # never leave a batch running into freed inputs
            try:
                self.end()
            except Exception:
                pass


def prove_batch_multi(contexts, kind, scalars, x, offset=None, per_proof=128, params: Optional[Params] = None):
    """bn254s_prove_batch_multi: proof i on contexts[i % len(contexts)] (one Context per GPU), one process."""
    params = params or default_params()
    lib = contexts[0]._lib
    n = scalars.shape[0]
    k = (n + per_proof - 1) // per_proof
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    outs = (C.c_void_p * k)()
    rc = lib.bn254s_prove_batch_multi(handles, len(contexts), kind, C.byref(params), _ptr(scalars), _ptr(x), _ptr(offset), n, per_proof,
                                      outs)
    if rc != 0:
        raise RuntimeError(f"bn254s_prove_batch_multi failed with {rc}: " + "; ".join(c._lib.bn254s_last_error(c._h).decode()
                                                                                     for c in contexts))
    return [Proof(lib, C.c_void_p(outs[i])) for i in range(k)]
