"""GPU: the fragment kernel replaces IEEE division by short exact forms (csrc/hip/fastmath.hpp).  This runs
the on-device proof-by-exhaustion: every binary32 input in the admitted range for 1/x and 0.9/x, dense
near-boundary samples for the integer-size mod certificate, packed-vs-scalar arithmetic."""
import pytest

import rust_doom_amd as rd

pytestmark = pytest.mark.gpu


def test_exact_forms_are_exact():
    r = rd.selftest_fastmath()
    assert r['inputs_swept'] > 3_300_000_000          # 2 signs x 201 binades x 2^23 mantissas
    assert r['rcp_mismatches'] == 0 and r['div09_mismatches'] == 0
    assert r['packed_mismatches'] == 0
    assert r['mod_samples'] > 100_000_000
    assert r['mod_violations'] == 0
    assert r['mod_floor_differs'] > 0                 # the sweep does reach the cases the certificate exists for
    assert r["mod_certified"] > 0.15 * r["mod_samples"]  # not vacuous (3 of 4 probes sit on a boundary by design)
