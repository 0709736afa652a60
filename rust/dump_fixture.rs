//! Source only: never compiled in this repository (see rust/README.md).
//! Dumps a reference proof of the G1 scalar-multiplication STARK in the flat word layout of include/bn254_stark.h.
//!
//! Input file (tools/export_fixture_inputs.py): first line n, then n lines of 20 hexadecimal u64 words:
//! scalar (4, little-endian), x.x (4), x.y (4), offset.x (4), offset.y (4); coordinates canonical.
#![cfg(test)]

use std::{fs, io::Write};

use ark_bn254::{Fq, G1Affine};
use ark_ff::{BigInteger256, PrimeField as _};
use num::BigUint;
use plonky2::{
    field::{extension::FieldExtension, goldilocks_field::GoldilocksField, types::PrimeField64},
    plonk::config::PoseidonGoldilocksConfig,
    util::timing::TimingTree,
};
use starky::config::StarkConfig;

use crate::starks::{
    common::prover::prove,
    curves::g1::{
        scalar_mul_ctl::g1_scalar_mul_ctl,
        scalar_mul_stark::{G1ScalarMulInput, G1ScalarMulStark},
    },
    LIMB_BITS,
};

type F = GoldilocksField;
const D: usize = 2;
type C = PoseidonGoldilocksConfig;

fn fq_from_words(w: &[u64]) -> Fq {
    Fq::from_bigint(BigInteger256::new([w[0], w[1], w[2], w[3]])).expect("coordinate not below p")
}

#[test]
fn dump_g1_fixture() {
    let path_in = std::env::var("FIXTURE_IN").expect("FIXTURE_IN");
    let path_out = std::env::var("FIXTURE_OUT").expect("FIXTURE_OUT");
    let text = fs::read_to_string(path_in).unwrap();
    let mut lines = text.lines();
    let n: usize = lines.next().unwrap().trim().parse().unwrap();
    let inputs = (0..n)
        .map(|timestamp| {
            let w: Vec<u64> = lines
                .next()
                .unwrap()
                .split_whitespace()
                .map(|h| u64::from_str_radix(h, 16).unwrap())
                .collect();
            assert_eq!(w.len(), 20);
            let s = BigUint::from_bytes_le(&w[0..4].iter().flat_map(|x| x.to_le_bytes()).collect::<Vec<u8>>());
            let input = G1ScalarMulInput {
                s,
                x: G1Affine::new_unchecked(fq_from_words(&w[4..8]), fq_from_words(&w[8..12])),
                offset: G1Affine::new_unchecked(fq_from_words(&w[12..16]), fq_from_words(&w[16..20])),
            };
            (input, timestamp)
        })
        .collect::<Vec<_>>();

    let stark = G1ScalarMulStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let trace = stark.generate_trace(&inputs, 1 << LIMB_BITS);
    let mut timing = TimingTree::default();
    let proof = prove::<F, C, _, D>(&stark, &config, &trace, &g1_scalar_mul_ctl(), &[], &mut timing).unwrap();

    // ---- flat layout of include/bn254_stark.h --------------------------------------------------------------------
    let mut words: Vec<u64> = Vec::new();
    let mut push_f = |words: &mut Vec<u64>, x: F| words.push(x.to_canonical_u64());
    let p = &proof.proof;
    for cap in [&p.trace_cap, p.auxiliary_polys_cap.as_ref().unwrap(), p.quotient_polys_cap.as_ref().unwrap()] {
        for h in &cap.0 {
            for e in h.elements {
                push_f(&mut words, e);
            }
        }
    }
    let ext = |words: &mut Vec<u64>, v: &[<F as plonky2::field::extension::Extendable<D>>::Extension]| {
        for x in v {
            let arr: [F; D] = x.to_basefield_array();
            for e in arr {
                words.push(e.to_canonical_u64());
            }
        }
    };
    let o = &p.openings;
    ext(&mut words, &o.local_values);
    ext(&mut words, &o.next_values);
    ext(&mut words, o.auxiliary_polys.as_ref().unwrap());
    ext(&mut words, o.auxiliary_polys_next.as_ref().unwrap());
    for x in o.ctl_zs_first.as_ref().unwrap() {
        push_f(&mut words, *x);
    }
    ext(&mut words, o.quotient_polys.as_ref().unwrap());
    let fri = &p.opening_proof;
    for cap in &fri.commit_phase_merkle_caps {
        for h in &cap.0 {
            for e in h.elements {
                push_f(&mut words, e);
            }
        }
    }
    for q in &fri.query_round_proofs {
        for (leaf, path) in &q.initial_trees_proof.evals_proofs {
            for e in leaf {
                push_f(&mut words, *e);
            }
            for h in &path.siblings {
                for e in h.elements {
                    push_f(&mut words, e);
                }
            }
        }
        for step in &q.steps {
            ext(&mut words, &step.evals);
            for h in &step.merkle_proof.siblings {
                for e in h.elements {
                    push_f(&mut words, e);
                }
            }
        }
    }
    ext(&mut words, &fri.final_poly.coeffs);
    push_f(&mut words, fri.pow_witness);
    for e in proof.init_challenger_state.as_ref() {
        push_f(&mut words, *e);
    }

    let mut f = fs::File::create(path_out).unwrap();
    writeln!(f, "{}", words.len()).unwrap();
    for w in words {
        writeln!(f, "{:016x}", w).unwrap();
    }
}
