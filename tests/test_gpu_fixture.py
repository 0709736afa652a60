"""Rehearsal of the fork cross-check (rust/README.md) for the three STARKs: the dump that rust/shim/dump_fixture.rs would
write is produced by the CPU oracle instead (same file format), and tools/compare_fixture.py must find trace digests and proof
identical; a corrupted dump must be localised to the trace or to the transcript."""
import numpy as np
import pytest

from tests import oracle_lib
from tools import compare_fixture as cf
from tools import export_fixture_inputs as ex

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n", [("g1", 3), ("g2", 2), ("fq", 5)])
def test_compare_fixture_pipeline(gpu_ctx, oracle, tmp_path, kind, n):
    k = ex.KINDS[kind]
    s, x, o = ex.inputs(kind, n, 0xF1C5)
    fin, fdump = str(tmp_path / "in.txt"), str(tmp_path / "dump.txt")
    ex.write(fin, s, x, o)
    s2, x2, o2 = ex.read(fin, kind)
    assert np.array_equal(s, s2) and np.array_equal(x, x2) and (o is None or np.array_equal(o, o2))
    trace, _ = oracle_lib.generate_trace(oracle, k, s, x, o)
    words, _, _, _ = oracle_lib.prove(oracle, k, s, x, o)
    cf.write_dump(fdump, trace, words)
    log = []
    assert cf.compare(gpu_ctx, kind, fin, fdump, out=log.append), log
    assert any("column digests identical" in ln for ln in log) and any("query rounds: identical" in ln for ln in log)
    # per-field accessor (bn254s_proof_section): the fields tile the word layout in order
    pr = {0: gpu_ctx.prove_g1, 1: gpu_ctx.prove_g2}[k](s, x, o) if k != 2 else gpu_ctx.prove_fq_exp(s, x)
    parts = [pr.section(name) for name in pr.SECTIONS]
    assert np.array_equal(np.concatenate(parts), pr.words)
    assert [p.size for p in parts[:3]] == [64, 64, 64] and parts[7].size == 4 and parts[8].size == 8
    assert parts[12].size == 1 and parts[13].size == 12 and parts[3].size == 2 * trace.shape[0]
    # a witness difference is reported as a trace difference ...
    bad = trace.copy()
    bad[7, 100] ^= np.uint64(1)
    cf.write_dump(fdump, bad, words)
    log = []
    assert not cf.compare(gpu_ctx, kind, fin, fdump, out=log.append)
    assert any(ln.startswith("trace: DIFFERENT") and "[7]" in ln for ln in log)


@pytest.mark.parametrize("kind", ["g1", "g2", "fq"])
def test_gpu_proof_equals_committed_fixture_expectation(gpu_ctx, tmp_path, kind):
    """tests/golden/fixture/<kind>_{inputs,expected}.txt are what a machine with cargo and no GPU diffs the reference's own dump
    against (tools/diff_fixture.py, rust/README.md): the GPU prover must produce exactly those words and trace column digests."""
    import os
    from tools import diff_fixture as df
    d = os.path.join(os.path.dirname(__file__), "golden", "fixture")
    k = ex.KINDS[kind]
    s, x, o = ex.read(os.path.join(d, kind + "_inputs.txt"), kind)
    trace, _ = gpu_ctx.generate_trace(k, s, x, o)
    pr = {0: gpu_ctx.prove_g1, 1: gpu_ctx.prove_g2}[k](s, x, o) if k != 2 else gpu_ctx.prove_fq_exp(s, x)
    dump = str(tmp_path / "gpu_dump.txt")
    cf.write_dump(dump, trace, pr.words)       # the format rust/shim/dump_fixture.rs writes
    log = []
    assert df.diff(os.path.join(d, kind + "_expected.txt"), dump, out=log.append), log
    assert any(ln.startswith("pow_witness") and "(same)" in ln for ln in log), log
    exp = df.parse_expected(os.path.join(d, kind + "_expected.txt"))
    assert exp["sections"]["head"][1][:64] == [int(v) for v in pr.section("trace_cap")]
    assert exp["sections"]["final_poly"][1] == [int(v) for v in pr.section("final_poly")]
