"""GPU: libhipbfv against BEHZ and the hybrid key switch carried out over the INTEGERS -- no oracle in the comparison.

tests/test_oracle_behz_exact.py and tests/test_oracle_keyswitch_exact.py (CPU suite) hold the integer restatements and show that
the oracle gives their bits.  Here the library's multiply and multiply + relinearize are compared with the same integer
models directly: the library's own auxiliary base (sized by the derived bound) never appears in the model, so
this is the claim "the product does not depend on which auxiliary primes are used" tested on the device.  (The oracle is
used only to generate keys and to bring them to coefficient form.)
"""
import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.test_oracle_behz_exact import _prod, behz_multiply_over_the_integers
from tests.test_oracle_keyswitch_exact import _key_coefficients, _switch_over_the_integers

pytestmark = pytest.mark.gpu


def _operands(q, n, rng):
    Q = _prod(q)
    a = np.stack([rng.integers(0, p, (2, 2, n), dtype=np.uint64) for p in q], axis=2)
    b = np.stack([rng.integers(0, p, (2, 2, n), dtype=np.uint64) for p in q], axis=2)
    for i, p in enumerate(q):
        a[1, :, i, :] = (Q // 2) % p  # the operands that drive the tensor to its extreme
        b[1, :, i, :] = (Q // 2) % p
    return a, b


@pytest.mark.parametrize("n,bits,tbits", [(4096, None, 17), (8192, None, 20), (8192, [54, 54, 54, 56], 20), (16384, None, 20)])
def test_multiply_equals_behz_over_the_integers(n, bits, tbits):
    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, tbits)
    q = [int(p) for p in primes[:-1]]
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    a, b = _operands(q, n, np.random.default_rng(n + tbits))
    m = to_host(ev.multiply(to_device(a), to_device(b)))
    for i in range(2):
        assert (m[i] == behz_multiply_over_the_integers(a[i], b[i], q, t)).all(), i


def test_multiply_relin_equals_the_integer_models():
    from sunscreen_amd import Context, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    n = 4096
    primes = O.bfv_default(n)
    t = O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    O.seed(41)
    sk, pk, rk, gk = o.keygen()
    q = [int(p) for p in primes[:-1]]
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    a, b = _operands(q, n, np.random.default_rng(5))
    got = to_host(ev.multiply_relin(to_device(a), to_device(b), rkd))
    keyc = _key_coefficients(o, rk)
    for i in range(2):
        ct3 = behz_multiply_over_the_integers(a[i], b[i], q, t)
        add0, add1 = _switch_over_the_integers(o, ct3[2], keyc)
        want = np.zeros((2, len(q), n), dtype=np.uint64)
        for j, qj in enumerate(q):
            want[0, j] = (ct3[0, j].astype(object) + add0[j].astype(object)) % qj
            want[1, j] = (ct3[1, j].astype(object) + add1[j].astype(object)) % qj
        assert (got[i] == want).all(), i
