//! Flat word layout of include/bn254_stark.h  <->  starky's `StarkProofWithMetadata` (reference common/prover.rs:66-71).
//! Source only: never compiled in this repository (rust/README.md).  Drop into the reference crate as
//! `src/starks/common/words.rs` (`pub(crate) mod words;` in `src/starks/common/mod.rs`).
//!
//! Every Goldilocks element is one canonical u64, every extension element two (c0, c1), every `HashOut` four, every
//! `MerkleProof` its siblings bottom-up.  `to_words` is what `dump_fixture.rs` writes; `from_words` is its inverse and
//! what the patched generators (run_once_*.rs) feed to the reference's own `verify` and `set_stark_proof_target`.
use ark_bn254::{Fq, Fq2, G1Affine, G2Affine};
use ark_ff::{BigInteger256, PrimeField as _};
use plonky2::{
    field::{
        extension::{Extendable, FieldExtension},
        polynomial::PolynomialCoeffs,
        types::{Field, PrimeField64},
    },
    fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep},
    hash::{
        hash_types::{HashOut, RichField},
        merkle_proofs::MerkleProof,
        merkle_tree::MerkleCap,
    },
    plonk::config::{GenericConfig, Hasher},
};
use starky::{
    config::StarkConfig,
    proof::{StarkOpeningSet, StarkProof, StarkProofWithMetadata},
};

/// Shape of one of the three STARKs: trace width and number of auxiliary polynomials (2 x (helpers + 1) + 4 CTL Z's).
#[derive(Clone, Copy)]
pub struct Shape {
    pub width: usize,
    pub aux: usize,
}
pub const SHAPE_G1: Shape = Shape { width: 781, aux: 456 }; // scalar_mul_view.rs:13-14, 225 helpers + Z, twice, + 4
pub const SHAPE_G2: Shape = Shape { width: 1295, aux: 906 }; // g2/scalar_mul_view.rs:13-14
pub const SHAPE_FQ: Shape = Shape { width: 427, aux: 134 }; // fields/exp_view.rs:12-13
const NUM_QUOTIENT_POLYS: usize = 4; // 2 challenges x quotient_degree_factor 2
const NUM_CTL_ZS: usize = 4;

struct Reader<'a> {
    w: &'a [u64],
    at: usize,
}
impl<'a> Reader<'a> {
    fn f<F: RichField>(&mut self) -> F {
        let x = F::from_canonical_u64(self.w[self.at]);
        self.at += 1;
        x
    }
    fn fs<F: RichField>(&mut self, n: usize) -> Vec<F> {
        (0..n).map(|_| self.f()).collect()
    }
    fn ext<F: RichField + Extendable<D>, const D: usize>(&mut self) -> F::Extension {
        let arr: [F; D] = core::array::from_fn(|_| self.f());
        F::Extension::from_basefield_array(arr)
    }
    fn exts<F: RichField + Extendable<D>, const D: usize>(&mut self, n: usize) -> Vec<F::Extension> {
        (0..n).map(|_| self.ext::<F, D>()).collect()
    }
    fn hash<F: RichField>(&mut self) -> HashOut<F> {
        HashOut { elements: core::array::from_fn(|_| self.f()) }
    }
}

/// `words` -> proof.  `degree_bits` = log2(rows) (bn254s_proof_degree_bits); the FRI shape follows from the config exactly
/// as the prover derives it (`config.fri_params(degree_bits)`).
pub fn stark_proof_from_words<F, C, const D: usize>(
    words: &[u64],
    shape: Shape,
    degree_bits: usize,
    config: &StarkConfig,
) -> StarkProofWithMetadata<F, C, D>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let fri_params = config.fri_params(degree_bits);
    let cap_len = 1usize << config.fri_config.cap_height;
    let lde_bits = degree_bits + config.fri_config.rate_bits;
    let mut r = Reader { w: words, at: 0 };
    let mut cap = |r: &mut Reader| MerkleCap::<F, C::Hasher>((0..cap_len).map(|_| r.hash()).collect());
    let trace_cap = cap(&mut r);
    let auxiliary_polys_cap = Some(cap(&mut r));
    let quotient_polys_cap = Some(cap(&mut r));
    let openings = StarkOpeningSet {
        local_values: r.exts::<F, D>(shape.width),
        next_values: r.exts::<F, D>(shape.width),
        auxiliary_polys: Some(r.exts::<F, D>(shape.aux)),
        auxiliary_polys_next: Some(r.exts::<F, D>(shape.aux)),
        ctl_zs_first: Some(r.fs(NUM_CTL_ZS)),
        quotient_polys: Some(r.exts::<F, D>(NUM_QUOTIENT_POLYS)),
    };
    let arities = &fri_params.reduction_arity_bits;
    let commit_phase_merkle_caps = arities.iter().map(|_| cap(&mut r)).collect::<Vec<_>>();
    let initial_path = lde_bits - config.fri_config.cap_height;
    let query_round_proofs = (0..config.fri_config.num_query_rounds)
        .map(|_| {
            let evals_proofs = [shape.width, shape.aux, NUM_QUOTIENT_POLYS]
                .iter()
                .map(|&width| {
                    let leaf = r.fs(width);
                    let siblings = (0..initial_path).map(|_| r.hash()).collect();
                    (leaf, MerkleProof::<F, C::Hasher> { siblings })
                })
                .collect();
            let mut bits = lde_bits;
            let steps = arities
                .iter()
                .map(|&a| {
                    bits -= a;
                    let evals = r.exts::<F, D>(1 << a);
                    let siblings = (0..bits - config.fri_config.cap_height).map(|_| r.hash()).collect();
                    FriQueryStep { evals, merkle_proof: MerkleProof::<F, C::Hasher> { siblings } }
                })
                .collect();
            FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps }
        })
        .collect();
    let final_len = 1usize << (degree_bits - arities.iter().sum::<usize>());
    let final_poly = PolynomialCoeffs::new(r.exts::<F, D>(final_len));
    let pow_witness = r.f();
    let state: Vec<F> = r.fs(12);
    assert_eq!(r.at, words.len(), "word count does not match the shape");
    StarkProofWithMetadata {
        proof: StarkProof {
            trace_cap,
            auxiliary_polys_cap,
            quotient_polys_cap,
            openings,
            opening_proof: FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness },
        },
        init_challenger_state: <C::Hasher as Hasher<F>>::Permutation::new(state.into_iter()),
    }
}

/// proof -> `words` (the inverse of the above; used by dump_fixture.rs).
pub fn stark_proof_to_words<F, C, const D: usize>(proof: &StarkProofWithMetadata<F, C, D>) -> Vec<u64>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F>,
    C::Hasher: Hasher<F, Hash = HashOut<F>>,
{
    let mut w: Vec<u64> = Vec::new();
    fn f<F: PrimeField64>(w: &mut Vec<u64>, x: F) {
        w.push(x.to_canonical_u64());
    }
    fn ext<F: RichField + Extendable<D>, const D: usize>(w: &mut Vec<u64>, v: &[F::Extension]) {
        for x in v {
            let arr: [F; D] = x.to_basefield_array();
            for e in arr {
                f(w, e);
            }
        }
    }
    fn hash<F: RichField>(w: &mut Vec<u64>, h: &HashOut<F>) {
        for e in h.elements {
            f(w, e);
        }
    }
    let p = &proof.proof;
    for cap in [&p.trace_cap, p.auxiliary_polys_cap.as_ref().unwrap(), p.quotient_polys_cap.as_ref().unwrap()] {
        cap.0.iter().for_each(|h| hash(&mut w, h));
    }
    let o = &p.openings;
    ext::<F, D>(&mut w, &o.local_values);
    ext::<F, D>(&mut w, &o.next_values);
    ext::<F, D>(&mut w, o.auxiliary_polys.as_ref().unwrap());
    ext::<F, D>(&mut w, o.auxiliary_polys_next.as_ref().unwrap());
    o.ctl_zs_first.as_ref().unwrap().iter().for_each(|x| f(&mut w, *x));
    ext::<F, D>(&mut w, o.quotient_polys.as_ref().unwrap());
    let fri = &p.opening_proof;
    for cap in &fri.commit_phase_merkle_caps {
        cap.0.iter().for_each(|h| hash(&mut w, h));
    }
    for q in &fri.query_round_proofs {
        for (leaf, path) in &q.initial_trees_proof.evals_proofs {
            leaf.iter().for_each(|x| f(&mut w, *x));
            path.siblings.iter().for_each(|h| hash(&mut w, h));
        }
        for step in &q.steps {
            ext::<F, D>(&mut w, &step.evals);
            step.merkle_proof.siblings.iter().for_each(|h| hash(&mut w, h));
        }
    }
    ext::<F, D>(&mut w, &fri.final_poly.coeffs);
    f(&mut w, fri.pow_witness);
    proof.init_challenger_state.as_ref().iter().for_each(|x| f(&mut w, *x));
    w
}

// ---- BN254 wire formats of include/bn254_stark.h: little-endian u64 words, canonical (non-Montgomery) -----------------
pub fn fq_from_words(w: &[u64]) -> Fq {
    Fq::from_bigint(BigInteger256::new([w[0], w[1], w[2], w[3]])).expect("coordinate not below p")
}
pub fn fq_to_words(x: &Fq, out: &mut Vec<u64>) {
    out.extend(x.into_bigint().0);
}
pub fn g1_from_words(w: &[u64]) -> G1Affine {
    G1Affine::new_unchecked(fq_from_words(&w[0..4]), fq_from_words(&w[4..8]))
}
pub fn g1_to_words(p: &G1Affine, out: &mut Vec<u64>) {
    fq_to_words(&p.x, out);
    fq_to_words(&p.y, out);
}
pub fn g2_from_words(w: &[u64]) -> G2Affine {
    G2Affine::new_unchecked(
        Fq2::new(fq_from_words(&w[0..4]), fq_from_words(&w[4..8])),
        Fq2::new(fq_from_words(&w[8..12]), fq_from_words(&w[12..16])),
    )
}
pub fn g2_to_words(p: &G2Affine, out: &mut Vec<u64>) {
    for c in [p.x.c0, p.x.c1, p.y.c0, p.y.c1] {
        fq_to_words(&c, out);
    }
}
pub fn scalar_to_words(s: &num::BigUint, out: &mut Vec<u64>) {
    let mut d = s.to_u64_digits();
    d.resize(4, 0);
    out.extend(d);
}
