"""Batches that mix what the other parity tests hold uniform - read lengths on either side of every dword and variant
border, k from 0 up, ragged / rescue-sized / too-short windows, forced variants, both strands - shuffled so that every
wave holds all of them (scripts/fuzz_parity.py builds them; the kernels' wave-uniform decisions are what this is after).
CPU: the oracle against the compiled reference on such batches.  GPU: every family of the script against the oracle."""
import os
import sys

import numpy as np
import pytest

import common

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import fuzz_parity as fz  # noqa: E402


def _run(ctx, oracle, families, seeds):
    failed = []

    def report(label, n, bad, tasks, got, want):
        if len(bad):
            failed.append("%s: %d of %d, first task %d" % (label, len(bad), n, bad[0]))

    gen = common.genome()
    for f in families:
        for s in seeds:
            fz.FAMILIES[f](ctx, oracle, np.random.default_rng(s), gen, report)
            assert ctx.take_flags() == 0
            assert not failed, "[%s:%d] %s" % (f, s, "; ".join(failed))


def test_oracle_equals_reference_on_mixed_batches(oracle, reference):
    ctx = fz.RefCtx(common.genome())
    _run(ctx, oracle, ["distance"], range(101, 113))


@pytest.mark.gpu
@pytest.mark.parametrize("family", list(fz.FAMILIES))
def test_gpu_equals_oracle_on_mixed_batches(oracle, family):
    from whisper_amd import api
    codes, packed = common.genome()
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        _run(ctx, oracle, [family], range(201, 207) if family == "distance" else range(201, 204))
    finally:
        ctx.destroy()


@pytest.mark.gpu
def test_last_symbol_from_read_with_ascii_reads(oracle):
    """WAMD_LAST_SYMBOL_FROM_READ on a batch uploaded as ASCII: preprocessRawSeq never writes last_symbol, the answer is
    a fresh object's (0, LevMyers64.cpp:35) - the M256 tie-break then differs from the read's own last symbol."""
    from whisper_amd import api
    codes, packed, (blob, off, lens), tasks = common.distance_workload(150, 30, 6000, seed=77, encoding=1, true_frac=0.9)
    rc, want = oracle.distance(packed, blob, off, lens, 1, tasks, -1)
    rc0, want0 = oracle.distance(packed, blob, off, lens, 1, tasks, 0)
    assert rc == 0 and rc0 == 0
    np.testing.assert_array_equal(want, want0)
    ctx = api.Context(packed, ref_size=codes.numel(), cur_offset=0)
    try:
        reads = ctx.upload_reads(blob, off, lens, 1)
        got = ctx.dynamic_programming(reads, tasks, -1)
        reads.free()
    finally:
        ctx.destroy()
    np.testing.assert_array_equal(got, want)
