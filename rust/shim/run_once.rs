//! The patched `run_once` bodies of the three generators.  Source only (rust/README.md).
//! Each replaces, in its generator, the lines that compute the outputs with ark, build the stark, call `generate_trace` and
//! `prove` (g1/stark_proof.rs:143-163, g2/stark_proof.rs:143-163, fq/stark_proof.rs:142-162); reading the inputs, the
//! reference's own `verify`, `set_stark_proof_target` and `set_ctl_values_target` stay as they are.
use crate::starks::common::{
    gpu::{prove, GpuProof},
    words::*,
};
use bn254stark_sys::{KIND_FQ_EXP, KIND_G1, KIND_G2};

// ---- src/generators/g1/stark_proof.rs ---------------------------------------------------------------------------------
fn run_once(&self, pw: &PartitionWitness<F>, out_buffer: &mut GeneratedValues<F>) {
    let inputs = self.inputs.iter().enumerate()
        .map(|(timestamp, input)| (input.get_witness(pw), timestamp)).collect::<Vec<_>>();          // unchanged (:137-142)
    let (mut scalars, mut xs, mut offs) = (Vec::new(), Vec::new(), Vec::new());
    for (input, _) in &inputs {
        scalar_to_words(&input.s, &mut scalars);
        g1_to_words(&input.x, &mut xs);
        g1_to_words(&input.offset, &mut offs);
    }
    let GpuProof { words, degree_bits, outputs } = prove(KIND_G1, &scalars, &xs, &offs, inputs.len());
    for (output_t, o) in self.outputs.iter().zip(outputs.chunks(8)) {                                 // was mul_bigint (:143-149)
        output_t.set_witness(out_buffer, &g1_from_words(o));
    }
    let extra_looking_values = g1_generate_ctl_values::<F>(&inputs);                                  // unchanged (:150)
    let stark = G1ScalarMulStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let cross_table_lookups = g1_scalar_mul_ctl::<F>();
    let stark_proof = stark_proof_from_words::<F, C, D>(&words, SHAPE_G1, degree_bits, &config);
    crate::starks::common::verifier::verify(&stark, &config, &cross_table_lookups, &stark_proof, &[],
                                            &extra_looking_values).unwrap();                         // unchanged (:164-172)
    set_stark_proof_target(out_buffer, &self.stark_proof, &stark_proof.proof, self.zero);             // unchanged (:173)
    set_ctl_values_target(out_buffer, &self.extra_looking_values, &extra_looking_values);             // unchanged (:174-178)
}

// ---- src/generators/g2/stark_proof.rs ---------------------------------------------------------------------------------
fn run_once(&self, pw: &PartitionWitness<F>, out_buffer: &mut GeneratedValues<F>) {
    let inputs = self.inputs.iter().enumerate()
        .map(|(timestamp, input)| (input.get_witness(pw), timestamp)).collect::<Vec<_>>();
    let (mut scalars, mut xs, mut offs) = (Vec::new(), Vec::new(), Vec::new());
    for (input, _) in &inputs {
        scalar_to_words(&input.s, &mut scalars);
        g2_to_words(&input.x, &mut xs);
        g2_to_words(&input.offset, &mut offs);
    }
    let GpuProof { words, degree_bits, outputs } = prove(KIND_G2, &scalars, &xs, &offs, inputs.len());
    for (output_t, o) in self.outputs.iter().zip(outputs.chunks(16)) {
        output_t.set_witness(out_buffer, &g2_from_words(o));
    }
    let extra_looking_values = g2_generate_ctl_values::<F>(&inputs);
    let stark = G2ScalarMulStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let cross_table_lookups = g2_scalar_mul_ctl::<F>();
    let stark_proof = stark_proof_from_words::<F, C, D>(&words, SHAPE_G2, degree_bits, &config);
    crate::starks::common::verifier::verify(&stark, &config, &cross_table_lookups, &stark_proof, &[],
                                            &extra_looking_values).unwrap();
    set_stark_proof_target(out_buffer, &self.stark_proof, &stark_proof.proof, self.zero);
    set_ctl_values_target(out_buffer, &self.extra_looking_values, &extra_looking_values);
}

// ---- src/generators/fq/stark_proof.rs ---------------------------------------------------------------------------------
fn run_once(&self, pw: &PartitionWitness<F>, out_buffer: &mut GeneratedValues<F>) {
    let inputs = self.inputs.iter().enumerate()
        .map(|(timestamp, input)| (input.get_witness(pw), timestamp)).collect::<Vec<_>>();
    let (mut scalars, mut xs) = (Vec::new(), Vec::new());
    for (input, _) in &inputs {
        scalar_to_words(&input.s, &mut scalars);
        fq_to_words(&input.x, &mut xs);
    }
    let GpuProof { words, degree_bits, outputs } = prove(KIND_FQ_EXP, &scalars, &xs, &[], inputs.len());
    for (output_t, o) in self.outputs.iter().zip(outputs.chunks(4)) {                                 // was x.pow(s) (:142-148)
        output_t.set_witness(out_buffer, &fq_from_words(o));
    }
    let extra_looking_values = fq_generate_ctl_values::<F>(&inputs);
    let stark = FqExpStark::<F, D>::new();
    let config = StarkConfig::standard_fast_config();
    let cross_table_lookups = fq_exp_ctl::<F>();
    let stark_proof = stark_proof_from_words::<F, C, D>(&words, SHAPE_FQ, degree_bits, &config);
    crate::starks::common::verifier::verify(&stark, &config, &cross_table_lookups, &stark_proof, &[],
                                            &extra_looking_values).unwrap();
    set_stark_proof_target(out_buffer, &self.stark_proof, &stark_proof.proof, self.zero);
    set_ctl_values_target(out_buffer, &self.extra_looking_values, &extra_looking_values);
}
