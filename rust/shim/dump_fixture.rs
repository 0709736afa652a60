//! Source only: never compiled in this repository (rust/README.md).
//! Three tests for the reference crate (`src/starks/common/dump_fixture.rs`, `#[cfg(test)] mod dump_fixture;` next to
//! `words.rs`): each reads inputs written by tools/export_fixture_inputs.py, runs the reference's own `generate_trace` +
//! `prove` (g1/scalar_mul_stark.rs:55-69, g2/scalar_mul_stark.rs:55-69, fields/exp_stark.rs:53-67, common/prover.rs:18-72)
//! and writes (a) one digest per trace column and (b) the proof in the word layout of include/bn254_stark.h.
//! tools/compare_fixture.py compares both with this library's results for the same inputs: a mismatch in (a) is a
//! trace-generation difference, a mismatch only in (b) a transcript-convention difference.
//!
//! Input file: first line n, then n lines of hexadecimal u64 words: scalar (4), then
//!   G1: x.x x.y offset.x offset.y (4 each);  G2: x (16: x.c0 x.c1 y.c0 y.c1) offset (16);  Fq-exp: x (4).
//! Output file: line 1 "<n_columns> <n_rows>", then one column digest per line, then "<n_words>" and one word per line.
#![cfg(test)]
use std::{fs, io::Write};

use num::BigUint;
use plonky2::{
    field::{goldilocks_field::GoldilocksField, polynomial::PolynomialValues, types::PrimeField64},
    plonk::config::PoseidonGoldilocksConfig,
    util::timing::TimingTree,
};
use starky::config::StarkConfig;

use crate::starks::{
    common::{prover::prove, words::*},
    curves::{
        g1::{scalar_mul_ctl::g1_scalar_mul_ctl, scalar_mul_stark::{G1ScalarMulInput, G1ScalarMulStark}},
        g2::{scalar_mul_ctl::g2_scalar_mul_ctl, scalar_mul_stark::{G2ScalarMulInput, G2ScalarMulStark}},
    },
    fields::{exp_ctl::fq_exp_ctl, exp_stark::{FqExpInput, FqExpStark}},
    LIMB_BITS,
};

type F = GoldilocksField;
const D: usize = 2;
type C = PoseidonGoldilocksConfig;

fn read_rows(words_per_row: usize) -> Vec<Vec<u64>> {
    let text = fs::read_to_string(std::env::var("FIXTURE_IN").expect("FIXTURE_IN")).unwrap();
    let mut lines = text.lines();
    let n: usize = lines.next().unwrap().trim().parse().unwrap();
    (0..n)
        .map(|_| {
            let w: Vec<u64> =
                lines.next().unwrap().split_whitespace().map(|h| u64::from_str_radix(h, 16).unwrap()).collect();
            assert_eq!(w.len(), words_per_row);
            w
        })
        .collect()
}
fn scalar(w: &[u64]) -> BigUint {
    BigUint::from_bytes_le(&w.iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>())
}
/// Polynomial hash of a column over 2^64: d = sum_i v_i K^(N-1-i); tools/compare_fixture.py computes the same with numpy.
fn column_digest(col: &PolynomialValues<F>) -> u64 {
    const K: u64 = 0x100000001B3;
    col.values.iter().fold(0u64, |d, v| d.wrapping_mul(K).wrapping_add(v.to_canonical_u64()))
}
fn write_out(trace: &[PolynomialValues<F>], words: &[u64]) {
    let mut f = fs::File::create(std::env::var("FIXTURE_OUT").expect("FIXTURE_OUT")).unwrap();
    writeln!(f, "{} {}", trace.len(), trace[0].values.len()).unwrap();
    for col in trace {
        writeln!(f, "{:016x}", column_digest(col)).unwrap();
    }
    writeln!(f, "{}", words.len()).unwrap();
    for w in words {
        writeln!(f, "{:016x}", w).unwrap();
    }
}

#[test]
fn dump_g1_fixture() {
    let inputs = read_rows(20)
        .iter()
        .enumerate()
        .map(|(t, w)| (G1ScalarMulInput { s: scalar(&w[0..4]), x: g1_from_words(&w[4..12]), offset: g1_from_words(&w[12..20]) }, t))
        .collect::<Vec<_>>();
    let stark = G1ScalarMulStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let trace = stark.generate_trace(&inputs, 1 << LIMB_BITS);
    let proof = prove::<F, C, _, D>(&stark, &config, &trace, &g1_scalar_mul_ctl(), &[], &mut TimingTree::default()).unwrap();
    write_out(&trace, &stark_proof_to_words::<F, C, D>(&proof));
}

#[test]
fn dump_g2_fixture() {
    let inputs = read_rows(36)
        .iter()
        .enumerate()
        .map(|(t, w)| (G2ScalarMulInput { s: scalar(&w[0..4]), x: g2_from_words(&w[4..20]), offset: g2_from_words(&w[20..36]) }, t))
        .collect::<Vec<_>>();
    let stark = G2ScalarMulStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let trace = stark.generate_trace(&inputs, 1 << LIMB_BITS);
    let proof = prove::<F, C, _, D>(&stark, &config, &trace, &g2_scalar_mul_ctl(), &[], &mut TimingTree::default()).unwrap();
    write_out(&trace, &stark_proof_to_words::<F, C, D>(&proof));
}

#[test]
fn dump_fq_exp_fixture() {
    let inputs = read_rows(8)
        .iter()
        .enumerate()
        .map(|(t, w)| (FqExpInput { s: scalar(&w[0..4]), x: fq_from_words(&w[4..8]) }, t))
        .collect::<Vec<_>>();
    let stark = FqExpStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let trace = stark.generate_trace(&inputs, 1 << LIMB_BITS);
    let proof = prove::<F, C, _, D>(&stark, &config, &trace, &fq_exp_ctl(), &[], &mut TimingTree::default()).unwrap();
    write_out(&trace, &stark_proof_to_words::<F, C, D>(&proof));
}

/// The inverse walk on the reference's own proof: words -> proof -> words is the identity and the rebuilt proof verifies.
#[test]
fn words_round_trip() {
    use crate::starks::common::verifier::verify;
    use crate::starks::curves::g1::scalar_mul_ctl::g1_generate_ctl_values;
    let inputs = read_rows(20)
        .iter()
        .enumerate()
        .map(|(t, w)| (G1ScalarMulInput { s: scalar(&w[0..4]), x: g1_from_words(&w[4..12]), offset: g1_from_words(&w[12..20]) }, t))
        .collect::<Vec<_>>();
    let stark = G1ScalarMulStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let ctl = g1_scalar_mul_ctl::<F>();
    let trace = stark.generate_trace(&inputs, 1 << LIMB_BITS);
    let proof = prove::<F, C, _, D>(&stark, &config, &trace, &ctl, &[], &mut TimingTree::default()).unwrap();
    let words = stark_proof_to_words::<F, C, D>(&proof);
    let degree_bits = trace[0].values.len().trailing_zeros() as usize;
    let back = stark_proof_from_words::<F, C, D>(&words, SHAPE_G1, degree_bits, &config);
    assert_eq!(stark_proof_to_words::<F, C, D>(&back), words);
    verify(&stark, &config, &ctl, &back, &[], &g1_generate_ctl_values::<F>(&inputs)).unwrap();
}
