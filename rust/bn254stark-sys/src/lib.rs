//! FFI declarations of `include/bn254_stark.h` (ABI version 1) - one `extern "C"` item per C function, same order.
//! Source only: never compiled in this repository (rust/README.md).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int};

/// `bn254s_params`: StarkConfig::standard_fast_config() + min_rows (reference stark_proof.rs:152-154).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct Bn254sParams {
    pub struct_size: u32,
    pub security_bits: u32,
    pub num_challenges: u32,
    pub rate_bits: u32,
    pub cap_height: u32,
    pub pow_bits: u32,
    pub arity_bits: u32,
    pub final_poly_bits: u32,
    pub num_queries: u32,
    pub min_rows_log2: u32,
}
#[repr(C)]
pub struct Bn254sCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct Bn254sProof {
    _p: [u8; 0],
}
#[repr(C)]
pub struct Bn254sBatch {
    _p: [u8; 0],
}

pub const BN254S_OK: c_int = 0;
pub const BN254S_E_INVALID_POINT: c_int = -4;
pub const BN254S_E_VERIFY: c_int = -8;
/// `kind` argument of the generic entry points.
pub const KIND_G1: c_int = 0;
pub const KIND_G2: c_int = 1;
pub const KIND_FQ_EXP: c_int = 2;
/// `id` argument of `bn254s_proof_section`.
pub const SEC_TRACE_CAP: c_int = 0;
pub const SEC_AUX_CAP: c_int = 1;
pub const SEC_QUOTIENT_CAP: c_int = 2;
pub const SEC_LOCAL_VALUES: c_int = 3;
pub const SEC_NEXT_VALUES: c_int = 4;
pub const SEC_AUX_POLYS: c_int = 5;
pub const SEC_AUX_POLYS_NEXT: c_int = 6;
pub const SEC_CTL_ZS_FIRST: c_int = 7;
pub const SEC_QUOTIENT_POLYS: c_int = 8;
pub const SEC_FRI_CAPS: c_int = 9;
pub const SEC_QUERY_ROUNDS: c_int = 10;
pub const SEC_FINAL_POLY: c_int = 11;
pub const SEC_POW_WITNESS: c_int = 12;
pub const SEC_INIT_CHALLENGER_STATE: c_int = 13;

extern "C" {
    pub fn bn254s_params_default(p: *mut Bn254sParams);
    pub fn bn254s_abi_version() -> c_int;
    pub fn bn254s_ctx_create(device_id: c_int, out: *mut *mut Bn254sCtx) -> c_int;
    pub fn bn254s_ctx_destroy(ctx: *mut Bn254sCtx);
    pub fn bn254s_last_error(ctx: *const Bn254sCtx) -> *const c_char;
    pub fn bn254s_prove_g1(ctx: *mut Bn254sCtx, params: *const Bn254sParams, scalars: *const u64, x: *const u64,
                           offset: *const u64, n: usize, out: *mut *mut Bn254sProof) -> c_int;
    pub fn bn254s_prove_g2(ctx: *mut Bn254sCtx, params: *const Bn254sParams, scalars: *const u64, x: *const u64,
                           offset: *const u64, n: usize, out: *mut *mut Bn254sProof) -> c_int;
    pub fn bn254s_prove_fq_exp(ctx: *mut Bn254sCtx, params: *const Bn254sParams, scalars: *const u64, x: *const u64,
                               n: usize, out: *mut *mut Bn254sProof) -> c_int;
    /// idle slots give their (grow-only) device workspaces back to the driver
    pub fn bn254s_ctx_trim(ctx: *mut Bn254sCtx) -> c_int;
    /// hash_to_fq2 of n messages of `len` Goldilocks elements each, on the device: out[n][8]
    pub fn bn254s_hash_to_fq2_batch(ctx: *mut Bn254sCtx, inputs: *const u64, n: usize, len: usize, out: *mut u64) -> c_int;
    pub fn bn254s_prove_batch(ctx: *mut Bn254sCtx, kind: c_int, params: *const Bn254sParams, scalars: *const u64,
                              x: *const u64, offset: *const u64, n_total: usize, per_proof: usize,
                              proofs: *mut *mut Bn254sProof) -> c_int;
    /// queues the batch on the context's worker pool and returns; `bn254s_prove_batch_end` waits for it.  Inputs and
    /// `proofs_out` must outlive the handle.  Other proving calls made meanwhile queue behind it (never share a slot).
    pub fn bn254s_prove_batch_begin(ctx: *mut Bn254sCtx, kind: c_int, params: *const Bn254sParams, scalars: *const u64,
                                    x: *const u64, offset: *const u64, n_total: usize, per_proof: usize,
                                    proofs_out: *mut *mut Bn254sProof, handle: *mut *mut Bn254sBatch) -> c_int;
    pub fn bn254s_prove_batch_end(handle: *mut Bn254sBatch) -> c_int;
    /// one context per GPU, proof i on context i mod n_ctx
    pub fn bn254s_prove_batch_multi(ctxs: *mut *mut Bn254sCtx, n_ctx: usize, kind: c_int, params: *const Bn254sParams,
                                    scalars: *const u64, x: *const u64, offset: *const u64, n_total: usize,
                                    per_proof: usize, proofs: *mut *mut Bn254sProof) -> c_int;
    pub fn bn254s_proof_words(p: *const Bn254sProof, data: *mut *const u64, len: *mut usize) -> c_int;
    pub fn bn254s_proof_section(p: *const Bn254sProof, id: c_int, data: *mut *const u64, len: *mut usize) -> c_int;
    pub fn bn254s_proof_degree_bits(p: *const Bn254sProof) -> c_int;
    pub fn bn254s_proof_outputs(p: *const Bn254sProof, data: *mut *const u64, len: *mut usize) -> c_int;
    pub fn bn254s_proof_serialize(p: *const Bn254sProof, buf: *mut u8, cap: usize) -> usize;
    pub fn bn254s_proof_free(p: *mut Bn254sProof);
    /// what set_ctl_values_target consumes (scalar_mul_ctl.rs:57-80 and twins)
    pub fn bn254s_ctl_values(kind: c_int, scalars: *const u64, x: *const u64, offset: *const u64, outputs: *const u64,
                             n: usize, in_rows: *mut u64, out_rows: *mut u64) -> c_int;
    /// 0, or BN254S_E_VERIFY with the reference verifier's error text in bn254s_last_error
    pub fn bn254s_verify(ctx: *mut Bn254sCtx, kind: c_int, params: *const Bn254sParams, degree_bits: u32,
                         words: *const u64, n_words: usize, scalars: *const u64, x: *const u64, offset: *const u64,
                         outputs: *const u64, n: usize) -> c_int;
    pub fn bn254s_map_to_g2(ctx: *mut Bn254sCtx, params: *const Bn254sParams, u: *const u64, offsets: *const u64, n: usize,
                            out_points: *mut u64, fq_jobs: *mut u64, g2_jobs: *mut u64, fq_proofs: *mut *mut Bn254sProof,
                            g2_proofs: *mut *mut Bn254sProof) -> c_int;
    pub fn bn254s_hash_to_fq2(input: *const u64, len: usize, out: *mut u64) -> c_int;
}
