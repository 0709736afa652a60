//! Per-thread GPU context and the three proving calls of the patched generators.
//! Source only: never compiled in this repository (rust/README.md).  Drop into the reference crate as
//! `src/starks/common/gpu.rs`; add `bn254stark-sys = { path = ".../rust/bn254stark-sys" }` to Cargo.toml.
//!
//! plonky2 runs witness generators on the thread that calls `prove`; a context is created lazily per thread and lives as
//! long as the thread (one call at a time per context is the library's threading rule).  BN254S_DEVICE selects the GPU.
use std::{cell::RefCell, ffi::CStr};

use bn254stark_sys::*;

pub struct Gpu(pub *mut Bn254sCtx);
impl Drop for Gpu {
    fn drop(&mut self) {
        unsafe { bn254s_ctx_destroy(self.0) }
    }
}
thread_local! {
    pub static GPU: RefCell<Option<Gpu>> = RefCell::new(None);
}

fn with_ctx<R>(f: impl FnOnce(*mut Bn254sCtx) -> R) -> R {
    GPU.with(|cell| {
        let mut slot = cell.borrow_mut();
        if slot.is_none() {
            let device = std::env::var("BN254S_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut ctx = std::ptr::null_mut();
            let rc = unsafe { bn254s_ctx_create(device, &mut ctx) };
            assert_eq!(rc, 0, "bn254s_ctx_create({device}) failed with {rc}");
            *slot = Some(Gpu(ctx));
        }
        f(slot.as_ref().unwrap().0)
    })
}

/// What a generator gets back: the proof in the flat word layout, log2(rows), and the outputs s*x+offset / x^s.
pub struct GpuProof {
    pub words: Vec<u64>,
    pub degree_bits: usize,
    pub outputs: Vec<u64>,
}

/// kind: KIND_G1 / KIND_G2 / KIND_FQ_EXP; `offset` empty for Fq-exp.  Panics on any error, like the `.unwrap()` of the
/// reference (stark_proof.rs:163): invalid point (a = -b met), unsupported shape, HIP failure.
pub fn prove(kind: i32, scalars: &[u64], x: &[u64], offset: &[u64], n: usize) -> GpuProof {
    with_ctx(|ctx| unsafe {
        let mut params = std::mem::zeroed::<Bn254sParams>();
        bn254s_params_default(&mut params); // standard_fast_config, min_rows 2^16 (stark_proof.rs:152-154)
        let mut proof = std::ptr::null_mut();
        let rc = match kind {
            KIND_G1 => bn254s_prove_g1(ctx, &params, scalars.as_ptr(), x.as_ptr(), offset.as_ptr(), n, &mut proof),
            KIND_G2 => bn254s_prove_g2(ctx, &params, scalars.as_ptr(), x.as_ptr(), offset.as_ptr(), n, &mut proof),
            _ => bn254s_prove_fq_exp(ctx, &params, scalars.as_ptr(), x.as_ptr(), n, &mut proof),
        };
        if rc != 0 {
            panic!("bn254s_prove (kind {kind}): {rc} ({:?})", CStr::from_ptr(bn254s_last_error(ctx)));
        }
        let (mut p, mut len) = (std::ptr::null(), 0usize);
        bn254s_proof_words(proof, &mut p, &mut len);
        let words = std::slice::from_raw_parts(p, len).to_vec();
        bn254s_proof_outputs(proof, &mut p, &mut len);
        let outputs = std::slice::from_raw_parts(p, len).to_vec();
        let degree_bits = bn254s_proof_degree_bits(proof) as usize;
        bn254s_proof_free(proof);
        GpuProof { words, degree_bits, outputs }
    })
}
