"""Full-size parity against bytes the REFERENCE computed: one column (two for bn254) of 2^18 .. 2^20
full-width 256-bit scalars on 1024 independent generators + the chain, committed by the reference's
own CPU backend on the CPU box (tests/golden/make_golden_fullsize.py -> fullsize_golden.npz: the
commitments and a SHA-256 of every input array).  The inputs are rebuilt here from their recipe --
mt19937{0} bytes, the reference's generate_random_element + add through oracle/_ref -- so nothing
the product computes enters the expected values or the inputs.
Reference tests of this shape: sxt/multiexp/test/multiexponentiation.cc:290-451,
cbindings/pedersen.t.cc:368-460.
"""
import os

import numpy as np
import pytest

from tests.golden import make_golden_fullsize as recipe

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "fullsize_golden.npz")


def _case(cid, columns, log2n):
    gold = np.load(FIXTURE)
    key = f"curve{cid}_{columns}x2^{log2n}"
    scalars, gens = recipe.inputs(cid, columns, log2n)
    assert np.array_equal(recipe.sha(scalars), gold[key + "_scalars_sha256"]), "scalars differ"
    assert np.array_equal(recipe.sha(gens), gold[key + "_generators_sha256"]), "generators differ"
    return scalars, gens, gold[key + "_commitments"]


def test_fixture_inputs_are_reproducible(oracle):
    """CPU: the smallest case's inputs rebuild to the committed hashes, and the reference still
    produces the committed commitment (13 s on one core)"""
    scalars, gens, want = _case(3, 1, 18)
    got = oracle.commit(3, [(scalars[0], False)], gens)
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("cid,columns,log2n", recipe.CASES)
def test_gpu_matches_reference_computed_bytes(gpu_backend, oracle, cid, columns, log2n):
    api = gpu_backend
    scalars, gens, want = _case(cid, columns, log2n)
    before = api.load().bzamd_kernel_launch_count()
    cols = [(scalars[c], False) for c in range(columns)]
    # through the drop-in entry point, host buffers (sxt_*_compute_pedersen_commitments_with_generators)
    got = api.compute_pedersen_commitments(cid, cols, generators=gens)
    assert api.load().bzamd_kernel_launch_count() > before
    assert np.array_equal(got, want), f"curve {cid}: differs from the reference-computed commitment"
