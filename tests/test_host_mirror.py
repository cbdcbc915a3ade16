"""The C++ mirror of the reference's verifier objects (whisper_amd/host/gpu_verifier.hpp):
per-read preprocess(), per-candidate dynamicProgramming()/LevDiag() enqueue, one flush.
The replay program is built by `make -C whisper_amd/csrc` (plain g++ over the C-ABI)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "whisper_amd", "host", "replay_batch")


def test_replay_program_is_built():
    assert os.path.exists(BIN), "run `make -C whisper_amd/csrc` (or __graft_entry__.build())"


def _vec(a):
    a = np.ascontiguousarray(a)
    return struct.pack("<Q", a.size) + a.tobytes()


@pytest.mark.gpu
def test_replay_matches_oracle(tmp_path, oracle):
    m, k = 150, 12
    codes, packed, (blob, off, lens), dtasks = common.distance_workload(m, k, 3000, seed=61, rc_frac=0.0)
    fl, fr = common.stage_segments(m, 5)[3]
    # LevDiag tasks on the same reads: same window start p - 5
    ltasks = np.zeros(len(dtasks), common.synth.LEVDIAG_TASK)
    ltasks["read_idx"] = dtasks["read_idx"]
    ltasks["position_in_text"] = dtasks["ref_pos"] + k
    ltasks["fixed_left"], ltasks["fixed_right"] = fl, fr
    ltasks["max_mismatches"] = 5
    ltasks["dir_gen_flag"] = 1
    _, want_d = oracle.distance(packed, blob, off, lens, 0, dtasks, -1)
    _, want_l = oracle.levdiag(packed, codes.numel(), 0, blob, off, lens, ltasks)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<4I", codes.numel(), 0, m, len(lens)))
        f.write(_vec(packed))
        f.write(_vec(blob))
        f.write(_vec(dtasks))
        f.write(_vec(ltasks))
    subprocess.run([BIN, str(fin), str(fout)], check=True, timeout=120)
    raw = np.fromfile(fout, np.uint32)
    got_d = raw[: 2 * len(dtasks)].reshape(-1, 2)
    got_l = raw[2 * len(dtasks):]
    np.testing.assert_array_equal(got_d[:, 0], want_d["edit_distance"])
    np.testing.assert_array_equal(got_d[:, 1], want_d["pos"])
    np.testing.assert_array_equal(got_l, want_l)
    # the switch a mapping stage flips: accepted candidates unchanged, the others {k + 1, 0}
    subprocess.run([BIN, str(fin), str(fout), "--accepted-only"], check=True, timeout=120)
    raw = np.fromfile(fout, np.uint32)
    got_a = raw[: 2 * len(dtasks)].reshape(-1, 2)
    rej = want_d["edit_distance"] > k
    assert rej.any() and (~rej).any()
    np.testing.assert_array_equal(got_a[~rej, 0], want_d["edit_distance"][~rej])
    np.testing.assert_array_equal(got_a[~rej, 1], want_d["pos"][~rej])
    assert (got_a[rej, 0] == k + 1).all() and (got_a[rej, 1] == 0).all()
    np.testing.assert_array_equal(raw[2 * len(dtasks):], want_l)
