"""The steps either side of the evaluator on the device: BatchEncoder, Decryptor, Encryptor.

BatchEncoder and Decryptor are deterministic: bit-exact against the oracle.  Encryptor is randomised (the reference
pins its sampler only statistically): ciphertexts must decrypt to the message under the ORACLE's decryptor with the
noise budget of a fresh SEAL encryption, and the sampled polynomials must have the right distribution.
Reference interfaces: seal_fhe/src/encoder.rs:50-215, seal_fhe/src/encryptor_decryptor.rs:140-260,596-690.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.bfv_helpers import make_vec, oracle_for, params

pytestmark = pytest.mark.gpu


def _setup(name, seed=21):
    from sunscreen_amd import Context, PublicKey, SecretKey
    from sunscreen_amd.batch import BatchEvaluator

    n, primes, t = params(name)
    o = oracle_for(name)
    O.seed(seed)
    sk, pk, rk, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    return o, sk, pk, rk, ctx, BatchEvaluator(ctx), SecretKey.from_array(ctx, sk), PublicKey.from_array(ctx, pk)


@pytest.mark.parametrize("name", ["default_8192_17", "default_4096_16", "seal_fhe_unit", "default_16384_17", "default_2048_14", "default_32768_17"])
def test_batch_encoder_matches_oracle(name):
    from sunscreen_amd import BFVEncoder, HipBfvError, Plaintext
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, ctx, ev, skd, pkd = _setup(name)
    n, t = o.n, o.t
    rng = np.random.default_rng(3)
    vals = rng.integers(0, t, (5, n), dtype=np.uint64)
    vals[0] = 0
    vals[1] = t - 1
    vals[2] = np.arange(n, dtype=np.uint64) % t
    enc = to_host(ev.encode(to_device(vals)))
    for i in range(5):
        assert (enc[i] == o.batch_encode(vals[i])).all(), (name, i)
    dec = to_host(ev.decode(to_device(enc)))
    assert (dec == vals).all()
    # arbitrary plaintexts (not produced by encode) decode like the oracle
    pl = rng.integers(0, t, (3, n), dtype=np.uint64)
    d2 = to_host(ev.decode(to_device(pl)))
    for i in range(3):
        assert (d2[i] == o.batch_decode(pl[i])).all()
    # signed view (Encode2 / Decode2): representatives in [-(t>>1), t>>1]
    half = t >> 1
    sv = rng.integers(-half, half + 1, (2, n)).astype(np.int64)
    senc = ev.encode(to_device(sv.view(np.uint64)), signed=True)
    assert (to_host(senc) == np.stack([o.batch_encode((v % t).astype(np.uint64)) for v in sv])).all()
    assert (to_host(ev.decode(senc, signed=True)).view(np.int64) == sv).all()
    # handle level, the seal_fhe unit-test vectors (bfv_evaluator.rs:284-302) and partial slot vectors
    be = BFVEncoder(ctx)
    assert be.get_slot_count() == n
    v = [int(x) for x in make_vec(n)]  # n/2 - i: the seal_fhe unit tests' vector (every t here exceeds n)
    p = be.encode_signed(v)
    assert be.decode_signed(p) == v
    assert be.decode_unsigned(p) == [x % t for x in v]
    short = be.encode_unsigned([1, 2, 3])
    assert be.decode_unsigned(short)[:4] == [1, 2, 3, 0]
    with pytest.raises(HipBfvError) as ei:
        be.encode_unsigned([t])
    assert ei.value.kind == "InvalidArgument"
    with pytest.raises(HipBfvError):
        be.encode_signed([half + 1])
    assert isinstance(p, Plaintext)


def test_batch_encoder_requires_a_batching_plain_modulus():
    from sunscreen_amd import BFVEncoder, Context, HipBfvError

    ctx = Context.from_raw(4096, O.bfv_default(4096), 262144)  # BASELINE configs[0]: t = 2^18, no batching
    with pytest.raises(HipBfvError):
        BFVEncoder(ctx)


@pytest.mark.parametrize("name", ["default_8192_17", "default_4096_16", "seal_fhe_unit", "default_16384_17", "simple_multiply", "default_32768_17"])
def test_decryptor_bit_exact(name):
    from sunscreen_amd import Ciphertext, Decryptor, Plaintext
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, ctx, ev, skd, pkd = _setup(name)
    n, t = o.n, o.t
    rng = np.random.default_rng(5)
    msgs = rng.integers(0, t, (4, n), dtype=np.uint64)
    msgs[0, 3:] = 0  # short plaintext: SEAL trims the result to its significant coefficients
    cts = np.stack([o.encrypt(pk, m) for m in msgs])
    got = to_host(ev.decrypt(to_device(cts), skd))
    for i in range(4):
        assert (got[i] == o.decrypt(cts[i], sk)).all() and (got[i] == msgs[i]).all(), (name, i)
    # size-3 ciphertexts (an unrelinearised product) and arbitrary residues (decryption of noise is still deterministic)
    prod = np.stack([o.multiply(cts[1], cts[2]), o.multiply(cts[0], cts[3])])
    got3 = to_host(ev.decrypt(to_device(prod), skd))
    for i in range(2):
        assert (got3[i] == o.decrypt(prod[i], sk)).all()
    junk = np.stack([np.stack([rng.integers(0, q, (2, n), dtype=np.uint64) for q in o.primes[: o.K]], axis=1) for _ in range(2)])
    gotj = to_host(ev.decrypt(to_device(junk), skd))
    for i in range(2):
        assert (gotj[i] == o.decrypt(junk[i], sk)).all()
    # handle level
    d = Decryptor(ctx, skd)
    p = d.decrypt(Ciphertext.from_array(ctx, cts[0]))
    assert isinstance(p, Plaintext) and p.len() == 3 and [p.get_coefficient(k) for k in range(3)] == [int(x) for x in msgs[0, :3]]
    # invariant noise budget (encryptor_decryptor.rs:640-660; seal_fhe/tests/assumptions.rs:138-194 relies on it):
    # fresh, after a multiplication (published cost of one multiply: 26-29 bits, Tables_of_things.md:7-14), exhausted
    for arr in (cts[1], prod[0], junk[0]):
        assert d.invariant_noise_budget(Ciphertext.from_array(ctx, arr)) == o.noise_budget(arr, sk), name
    assert d.invariant_noise_budget(Ciphertext.from_array(ctx, junk[0])) == 0


@pytest.mark.parametrize("name", ["default_8192_17", "default_4096_16", "seal_fhe_unit", "default_16384_17", "default_32768_17"])
def test_encryptor_produces_fresh_seal_ciphertexts(name):
    from sunscreen_amd import Ciphertext, Decryptor, Encryptor, Plaintext
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, ctx, ev, skd, pkd = _setup(name)
    n, t = o.n, o.t
    rng = np.random.default_rng(7)
    batch = 6
    msgs = rng.integers(0, t, (batch, n), dtype=np.uint64)
    cts = to_host(ev.encrypt(to_device(msgs), pkd, seed=1234))
    ref_budget = o.noise_budget(o.encrypt(pk, msgs[0]), sk)
    for i in range(batch):
        assert (o.decrypt(cts[i], sk) == msgs[i]).all(), (name, i)
        assert abs(o.noise_budget(cts[i], sk) - ref_budget) <= 2, (name, i, ref_budget)
        for k, q in enumerate(o.primes[: o.K]):
            assert int(cts[i][:, k].max()) < q
    # reproducible for equal (seed, op), independent otherwise
    again = to_host(ev.encrypt(to_device(msgs), pkd, seed=1234))
    assert (again == cts).all()
    shifted = to_host(ev.encrypt(to_device(msgs[:2]), pkd, seed=1234, first_op=1))
    assert (shifted[0] != cts[0]).any() and (o.decrypt(shifted[0], sk) == msgs[0]).all()
    other = to_host(ev.encrypt(to_device(msgs[:1]), pkd, seed=99))
    assert (other[0] != cts[0]).any()
    # a shared plaintext
    one = to_host(ev.encrypt(to_device(msgs[0]), pkd, seed=5))
    assert one.shape == (1, 2, o.K, n) and (o.decrypt(one[0], sk) == msgs[0]).all()
    # handle level: encrypt -> evaluator -> decrypt entirely through the SEAL-named C ABI
    enc, dec = Encryptor(ctx, pkd, seed=42), Decryptor(ctx, skd)
    pa, pb = Plaintext.from_coefficients([3]), Plaintext.from_coefficients([5, 1])
    ca, cb = enc.encrypt(pa), enc.encrypt(pb)
    assert (ca.to_array() != enc.encrypt(pa).to_array()).any()  # fresh randomness per call
    from sunscreen_amd import BFVEvaluator, RelinearizationKeys

    be = BFVEvaluator(ctx)
    prod = be.relinearize(be.multiply(ca, cb), RelinearizationKeys.from_array(ctx, rk))
    out = dec.decrypt(prod)
    assert [out.get_coefficient(k) for k in range(out.len())] == [15 % t, 3]
    assert isinstance(ca, Ciphertext)


def test_encryption_noise_statistics():
    """u is uniform ternary, e0 / e1 are rounded Gaussians with sigma 3.2 clipped at 19.  With a special prime the
    fresh noise is divided by q_sp and only rounding error survives, so the sampler is checked on a parameter set
    WITHOUT a special prime (n = 2048, one 54-bit prime: SEAL encrypts at the only level): with m = 0 the decryption
    phase c0 + c1*s = u*e_pk + e0 + e1*s is a sum of 2N+1 products of small terms with standard deviation
    sigma * sqrt(4N/3 + 1); a wrong ternary or Gaussian sampler moves it far outside a 10 % band."""
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, ctx, ev, skd, pkd = _setup("default_2048_14", seed=31)
    assert o.KK == 1
    n = o.n
    zero = np.zeros((16, n), dtype=np.uint64)
    cts = to_host(ev.encrypt(to_device(zero), pkd, seed=77))
    q0 = o.primes[0]
    devs = []
    for ct in cts:
        v = o.dot_with_secret(ct, sk)[0].astype(np.int64)  # phase mod q_0: the noise is far below q_0/2
        devs.append(np.where(v > q0 // 2, v - q0, v))
    v = np.concatenate(devs).astype(np.float64)
    expect = 3.2 * np.sqrt(4 * n / 3 + 1)
    assert abs(v.mean()) < 0.05 * expect
    assert 0.9 * expect < v.std() < 1.1 * expect, (v.std(), expect)
    # and with a special prime only the rounding error of the divide-and-round survives: sqrt((2N/3 + 1) / 12)
    o2, sk2, pk2, _, _, ev2, _, pkd2 = _setup("default_4096_16", seed=32)
    c2 = to_host(ev2.encrypt(to_device(np.zeros((8, o2.n), dtype=np.uint64)), pkd2, seed=78))
    q = o2.primes[0]
    w = np.concatenate([np.where((x := o2.dot_with_secret(c, sk2)[0].astype(np.int64)) > q // 2, x - q, x) for c in c2]).astype(np.float64)
    expect2 = np.sqrt((2 * o2.n / 3 + 1) / 12 + 1)
    assert 0.8 * expect2 < w.std() < 1.3 * expect2, (w.std(), expect2)


def test_key_handles_wire_format_roundtrip():
    """SecretKey / PublicKey cross the boundary in the SEAL 4.0 wire format (key_generator.rs:203-255,354-395): a
    secret key serialises as a Plaintext, a public key as a key-level NTT-form Ciphertext.  The golden secret key of
    seal_fhe/tests/data/secret_key.bin (tests/golden/seal_key_fixture.npz) must serialise to the fixture's words."""
    from sunscreen_amd import Context, HipBfvError, PublicKey, SecretKey

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seal_key_fixture.npz"))
    n = 8192
    primes = O.coeff_modulus_create(n, [50, 30, 30, 50, 50])
    t = O.plain_batching(n, 20)
    o = O.Oracle(n, primes, t)
    ctx = Context.from_raw(n, primes, t)
    tern = g["sk_ternary"].astype(np.int64)
    sk = np.stack([o.ntt(i, np.where(tern < 0, q + tern, tern).astype(np.uint64)) for i, q in enumerate(primes)])
    golden = [str(x) for x in g["sk_sha256"]]  # one digest per residue polynomial of the fixture
    assert [hashlib.sha256(row.tobytes()).hexdigest() for row in sk] == golden
    key = SecretKey.from_array(ctx, sk)
    raw = key.as_bytes(compression=0)
    assert raw[:8] == bytes([0x5E, 0xA1, 16, 4, 0, 0, 0, 0]) and len(raw) == int.from_bytes(raw[8:16], "little")
    payload = raw[-sk.nbytes :]  # DynArray payload = the fixture's words
    assert [hashlib.sha256(payload[i * n * 8 : (i + 1) * n * 8]).hexdigest() for i in range(len(primes))] == golden
    back = SecretKey.from_bytes(ctx, key.as_bytes())
    assert back.as_bytes(compression=0) == raw
    O.seed(4)
    _, pk, _, _ = o.keygen(relin=False)
    pkh = PublicKey.from_array(ctx, pk)
    blob = pkh.as_bytes()
    assert PublicKey.from_bytes(ctx, blob).as_bytes(compression=0) == pkh.as_bytes(compression=0)
    with pytest.raises(HipBfvError):  # a public key is not a secret key
        SecretKey.from_bytes(ctx, blob)
    other = Context.from_raw(4096, O.bfv_default(4096), O.plain_batching(4096, 16))
    with pytest.raises(HipBfvError):  # parms_id mismatch
        PublicKey.from_bytes(other, blob)
    bad = sk.copy()
    bad[1, 5] = primes[1]
    with pytest.raises(HipBfvError):
        SecretKey.from_array(ctx, bad)


# ---- KeyGenerator (seal_fhe/src/key_generator.rs:20-200): device-made keys judged by the oracle ----------------------
def _centred(x, q):
    x = x.astype(object)
    return np.where(x > q // 2, x - q, x)


def _dyadic(a, b, q):
    return np.array((a.astype(object) * b.astype(object)) % q, dtype=np.uint64)


@pytest.mark.parametrize("name", ["default_8192_17", "default_4096_16", "seal_fhe_unit", "default_16384_17"])
def test_key_generator_keys_are_valid_under_the_oracle(name):
    from sunscreen_amd import Context, KeyGenerator

    n, primes, t = params(name)
    o = oracle_for(name)
    ctx = Context.from_raw(n, primes, t)
    KK, K = len(primes), len(primes) - 1
    kg = KeyGenerator(ctx, seed=77)
    sk = kg.secret_key().to_array(ctx)
    assert sk.shape == (KK, n)
    # the secret is one ternary polynomial, present under every key prime
    s_coeff = [_centred(o.ntt(i, sk[i], inverse=True), primes[i]) for i in range(KK)]
    for i in range(1, KK):
        assert (s_coeff[i] == s_coeff[0]).all()
    assert set(np.unique(s_coeff[0]).tolist()) <= {-1, 0, 1}
    frac = np.mean(s_coeff[0] != 0)
    assert 0.6 < frac < 0.73, frac
    # public key: pk0 + pk1*s = -e with e a clipped rounded Gaussian (sigma 3.2, |e| <= 19), one e for all residues
    pk = kg.create_public_key().to_array(ctx)
    errs = []
    for i in range(KK):
        q = primes[i]
        v = (pk[0, i].astype(object) + _dyadic(pk[1, i], sk[i], q).astype(object)) % q
        errs.append(_centred(o.ntt(i, np.array(v, dtype=np.uint64), inverse=True), q))
    for i in range(1, KK):
        assert (errs[i] == errs[0]).all()
    e = errs[0].astype(np.float64)
    assert np.abs(e).max() <= 19
    assert 2.9 < e.std() < 3.5 and abs(e.mean()) < 0.25, (e.std(), e.mean())
    # pk1 is uniform: the top bit of the range is hit about half the time
    assert 0.45 < np.mean(pk[1, 0] > primes[0] // 2) < 0.55
    # the oracle encrypts under the device public key and decrypts with the device secret key
    O.seed(5)
    osk, opk, ork, _ = o.keygen()
    msg = np.arange(n, dtype=np.uint64) % t
    ct = o.encrypt(pk, msg)
    assert (o.decrypt(ct, sk) == msg).all()
    assert o.noise_budget(ct, sk) >= o.noise_budget(o.encrypt(opk, msg), osk) - 2
    if K < 1:
        return
    # relinearisation key: structure (digit z hides q_sp * s^2 on residue z only) and use
    rk = kg.create_relinearization_keys()
    rka = rk.to_array(ctx)
    assert rka.shape == (K, 2, KK, n)
    qsp = primes[-1]
    for z in range(K):
        for i in range(KK):
            q = primes[i]
            v = rka[z, 0, i].astype(object) + _dyadic(rka[z, 1, i], sk[i], q).astype(object)
            if i == z:
                v = v - (qsp % q) * _dyadic(sk[i], sk[i], q).astype(object)
            err = _centred(o.ntt(i, np.array(v % q, dtype=np.uint64), inverse=True), q)
            assert np.abs(err.astype(np.float64)).max() <= 19, (z, i)
    a = o.encrypt(pk, msg)
    prod3 = o.multiply(a, a)
    want = o.decrypt(prod3, sk)
    got = o.relinearize(prod3, rka)
    assert got.shape[0] == 2 and (o.decrypt(got, sk) == want).all()
    assert o.noise_budget(got, sk) >= o.noise_budget(prod3, sk) - 3
    # Galois keys: SEAL's default set (column swap + every power-of-two row rotation, both directions)
    logn = n.bit_length() - 1
    gk = kg.create_galois_keys()
    elts = sorted({2 * n - 1} | {pow(3, 1 << i, 2 * n) for i in range(logn - 1)} | {pow(3, -(1 << i), 2 * n) for i in range(logn - 1)})
    present = [e for e in range(1, 2 * n, 2) if gk.has_key((e - 1) // 2)] if n <= 4096 else [e for e in elts if gk.has_key((e - 1) // 2)]
    # 2(log n - 1) + 1 list entries in SEAL's get_elts_all; 3^(n/4) is its own inverse, so one fewer distinct keys
    assert present == elts and len(elts) == 2 * (logn - 1)
    if (t - 1) % (2 * n) == 0:
        slots = np.arange(n, dtype=np.uint64) * 3 % t
        cts = o.encrypt(pk, o.batch_encode(slots))
        for step in (1, -2, n // 4):
            elt = o.galois_elt_from_step(step)
            gka = {elt: gk.to_array(ctx, (elt - 1) // 2)}
            r = o.batch_decode(o.decrypt(o.rotate_rows(cts, step, gka), sk))
            half = n // 2
            want_rows = np.concatenate([np.roll(slots[:half], -step), np.roll(slots[half:], -step)])
            assert (r == want_rows).all(), step
        gka = {2 * n - 1: gk.to_array(ctx, n - 1)}
        r = o.batch_decode(o.decrypt(o.rotate_columns(cts, gka), sk))
        assert (r == np.concatenate([slots[n // 2 :], slots[: n // 2]])).all()


def test_key_generator_from_existing_secret_and_reproducibility():
    from sunscreen_amd import Context, GaloisKeys, KeyGenerator, PublicKey, RelinearizationKeys, SecretKey

    name = "default_8192_17"
    n, primes, t = params(name)
    o = oracle_for(name)
    O.seed(9)
    osk, opk, ork, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    kg = KeyGenerator.new_from_secret_key(ctx, SecretKey.from_array(ctx, osk))
    assert (kg.secret_key().to_array(ctx) == osk).all()
    pk = kg.create_public_key()
    msg = np.arange(n, dtype=np.uint64) * 7 % t
    ct = o.encrypt(pk.to_array(ctx), msg)
    assert (o.decrypt(ct, osk) == msg).all()  # the ORACLE's secret key opens it
    rk = kg.create_relinearization_keys()
    p3 = o.multiply(ct, ct)
    assert (o.decrypt(o.relinearize(p3, rk.to_array(ctx)), osk) == o.decrypt(p3, osk)).all()
    elt = o.galois_elt_from_step(3)
    gk = kg.create_galois_keys([elt])
    assert gk.has_key((elt - 1) // 2) and not gk.has_key(n - 1)
    rot = o.rotate_rows(ct, 3, {elt: gk.to_array(ctx, (elt - 1) // 2)})
    assert (o.batch_decode(o.decrypt(rot, osk))[: n // 2] == np.roll(o.batch_decode(msg)[: n // 2], -3)).all()
    # keys by rotation step (SEAL create_galois_keys(steps)): step 0 is the column rotation, too large a step is refused
    gs = kg.create_galois_keys(steps=[3, -2, 0])
    for st in (3, -2, 0):
        e = o.galois_elt_from_step(st) if st else 2 * n - 1
        assert gs.has_key((e - 1) // 2)
    assert not gs.has_key((o.galois_elt_from_step(1) - 1) // 2)
    e2 = o.galois_elt_from_step(-2)
    rot = o.rotate_rows(ct, -2, {e2: gs.to_array(ctx, (e2 - 1) // 2)})
    assert (o.batch_decode(o.decrypt(rot, osk))[: n // 2] == np.roll(o.batch_decode(msg)[: n // 2], 2)).all()
    with pytest.raises(Exception):
        kg.create_galois_keys(steps=[n // 2])
    # seeded generators repeat; successive keys from one generator differ; unseeded generators differ
    a, b = KeyGenerator(ctx, seed=5), KeyGenerator(ctx, seed=5)
    assert (a.secret_key().to_array(ctx) == b.secret_key().to_array(ctx)).all()
    pa, pb = a.create_public_key().to_array(ctx), b.create_public_key().to_array(ctx)
    assert (pa == pb).all()
    assert not (a.create_public_key().to_array(ctx) == pa).all()
    assert (a.create_relinearization_keys().to_array(ctx) == (b.create_public_key(), b.create_relinearization_keys())[1].to_array(ctx)).all()
    c, d = KeyGenerator(ctx), KeyGenerator(ctx)
    assert not (c.secret_key().to_array(ctx) == d.secret_key().to_array(ctx)).all()
    # wire format round trip of device-made keys (SEAL 4.0 serialisation)
    for cls, key in ((SecretKey, a.secret_key()), (PublicKey, a.create_public_key())):
        back = cls.from_bytes(ctx, key.as_bytes())
        assert (back.to_array(ctx) == key.to_array(ctx)).all()
    rk2 = RelinearizationKeys.from_bytes(ctx, rk.as_bytes())
    assert (rk2.to_array(ctx) == rk.to_array(ctx)).all()
    gk2 = GaloisKeys.from_bytes(ctx, gk.as_bytes())
    assert (gk2.to_array(ctx, (elt - 1) // 2) == gk.to_array(ctx, (elt - 1) // 2)).all()


def test_device_only_round_trip_keygen_encrypt_evaluate_decrypt():
    """The seal_fhe crate's own usage pattern (bfv_evaluator.rs tests): every object made by this library."""
    from sunscreen_amd import BFVEncoder, BFVEvaluator, Context, Decryptor, Encryptor, HipBfvError, KeyGenerator

    n, primes, t = params("seal_fhe_unit")
    ctx = Context.from_raw(n, primes, t)
    kg = KeyGenerator(ctx)
    enc = Encryptor(ctx, kg.create_public_key())
    dec = Decryptor(ctx, kg.secret_key())
    rk, gk = kg.create_relinearization_keys(), kg.create_galois_keys()
    be, ev = BFVEncoder(ctx), BFVEvaluator(ctx)
    v = [int(x) for x in make_vec(n)]
    a = enc.encrypt(be.encode_signed(v))
    fresh = dec.invariant_noise_budget(a)
    assert fresh > 20
    sq = ev.relinearize(ev.multiply(a, a), rk)
    assert be.decode_signed(dec.decrypt(sq)) == [((x * x + t // 2) % t) - t // 2 for x in v]
    assert 0 < dec.invariant_noise_budget(sq) < fresh
    # the f64 noise measure (encryptor_decryptor.rs:660-683) tells the same story: budget ~ -log2(2 * noise)
    import math

    na, nsq = dec.invariant_noise(a), dec.invariant_noise(sq)
    assert 0 < na < nsq < 0.5
    assert abs(-math.log2(2 * na) - fresh) <= 1 and abs(-math.log2(2 * nsq) - dec.invariant_noise_budget(sq)) <= 1
    r = be.decode_signed(dec.decrypt(ev.rotate_rows(a, -1, gk)))
    half = n // 2
    assert r[:half] == v[half - 1 : half] + v[: half - 1] and r[half:] == v[n - 1 :] + v[half : n - 1]
    c = be.decode_signed(dec.decrypt(ev.rotate_columns(a, gk)))
    assert c == v[half:] + v[:half]
    # a generator for a context without a special prime cannot make key-switching keys
    n2, primes2, t2 = params("default_2048_14")
    assert len(primes2) == 1
    with pytest.raises(HipBfvError) as ei:
        KeyGenerator(Context.from_raw(n2, primes2, t2)).create_relinearization_keys()
    assert ei.value.kind == "InternalError"  # COR_E_INVALIDOPERATION, as convert_seal_error maps it (error.rs:82-91)


# ---- the fork-only export path logproof consumes: PolynomialArray + encryption components -----------
@pytest.mark.parametrize("name", ["seal_fhe_unit", "default_4096_16", "default_16384_17"])
def test_polynomial_array_and_encryption_components(name):
    """seal_fhe/src/data_structures.rs:330-534 (sizes, round trips, clone, drop) plus what logproof relies on
    (bfv_statement.rs:159-160): the returned u, e, r satisfy the encryption equations EXACTLY over the data primes."""
    from sunscreen_amd import (BFVEncoder, Ciphertext, Context, Decryptor, Encryptor, KeyGenerator, Plaintext, PolynomialArray)

    n, primes, t = params(name)
    o = oracle_for(name)
    K = len(primes) - 1
    q = 1
    for p in primes[:K]:
        q *= p
    ctx = Context.from_raw(n, primes, t)
    kg = KeyGenerator(ctx, seed=3)
    pk, sk = kg.create_public_key(), kg.secret_key()
    enc = Encryptor.with_public_and_secret_key(ctx, pk, sk)
    dec = Decryptor(ctx, sk)
    be = BFVEncoder(ctx)
    data = [i % t for i in range(n)]
    plain = be.encode_unsigned(data)
    m = np.array([plain.get_coefficient(i) for i in range(n)], dtype=object)

    ct, u, e, r = enc.encrypt_return_components(plain)
    assert be.decode_unsigned(dec.decrypt(ct)) == data
    assert dec.invariant_noise_budget(ct) > 0
    pa_pk = PolynomialArray.new_from_public_key(ctx, pk)
    pa_sk = PolynomialArray.new_from_secret_key(ctx, sk)
    pa_ct = PolynomialArray.new_from_ciphertext(ctx, ct)
    for a, polys in ((pa_pk, 2), (pa_sk, 1), (pa_ct, 2), (u, 1), (e, 2)):
        assert a.is_reserved() and a.is_rns()
        assert (a.num_polynomials(), a.poly_modulus_degree(), a.coeff_modulus_size()) == (polys, n, K)
        assert a.as_u64s().size == polys * n * K
    assert r.len() == n
    assert not PolynomialArray().is_reserved()
    assert PolynomialArray.new_from_ciphertext(ctx, Ciphertext()).is_reserved()
    # poly_array_u64_and_bytes_match: the array of a ciphertext is the ciphertext's data
    assert (pa_ct.as_u64s() == ct.to_array().ravel()).all()
    assert [ct.get_data(i) for i in (0, 1, n, 2 * K * n - 1)] == [int(pa_ct.as_u64s()[i]) for i in (0, 1, n, 2 * K * n - 1)]
    # keys come out of NTT form, restricted to the data primes
    skn = sk.to_array(ctx)
    s_rns = pa_sk.as_rns_u64s().reshape(K, n)
    for i in range(K):
        assert (s_rns[i] == o.ntt(i, skn[i], inverse=True)).all()
    pkn = pk.to_array(ctx)
    p_rns = pa_pk.as_rns_u64s().reshape(2, K, n)
    assert (p_rns[1, K - 1] == o.ntt(K - 1, pkn[1, K - 1], inverse=True)).all()
    # multiprecision: [poly][coeff][limb], the CRT-composed value in [0, q); and back is the identity
    before = pa_ct.as_u64s().copy()
    mp = pa_ct.as_multiprecision_u64s().reshape(2, n, K)
    assert pa_ct.is_rns() and (pa_ct.as_u64s() == before).all()
    rns = before.reshape(2, K, n)
    for p_ in range(2):
        for x in (0, 1, n // 2, n - 1):
            v = sum(int(mp[p_, x, l]) << (64 * l) for l in range(K))
            assert v < q
            assert [v % primes[i] for i in range(K)] == [int(rns[p_, i, x]) for i in range(K)]
    pa_ct.to_multiprecision()
    assert pa_ct.is_multiprecision() and (pa_ct.as_u64s() == mp.ravel()).all()
    assert (pa_ct.as_rns_u64s() == before).all() and pa_ct.is_multiprecision()
    pa_ct.to_rns()
    assert (pa_ct.as_u64s() == before).all()
    # small polynomials compose to small values or q - small
    ump = u.as_multiprecision_u64s().reshape(n, K)
    uv = [sum(int(ump[x, l]) << (64 * l) for l in range(K)) for x in range(64)]
    assert set(uv) <= {0, 1, q - 1}

    def centred(a, i):
        return _centred(a, primes[i])

    u_rns, e_rns = u.as_rns_u64s().reshape(K, n), e.as_rns_u64s().reshape(2, K, n)
    assert set(np.unique(centred(u_rns[0], 0)).tolist()) <= {-1, 0, 1}
    assert np.abs(centred(e_rns[0, 0], 0).astype(np.float64)).max() <= 19
    for i in range(1, K):
        assert (centred(u_rns[i], i) == centred(u_rns[0], 0)).all() and (centred(e_rns[1, i], i) == centred(e_rns[1, 0], 0)).all()
    rv = np.array([r.get_coefficient(i) for i in range(n)], dtype=object)
    assert (rv == ((q % t) * m + (t + 1) // 2) // t).all()
    delta = q // t
    c = ct.to_array()

    def negacyclic(a_i, b_i, i):
        prod = _dyadic(o.ntt(i, a_i), o.ntt(i, b_i), primes[i])
        return o.ntt(i, prod, inverse=True).astype(object)

    for i in range(K):
        qi = primes[i]
        want0 = (delta % qi * m + rv + negacyclic(p_rns[0, i], u_rns[i], i) + e_rns[0, i].astype(object)) % qi
        want1 = (negacyclic(p_rns[1, i], u_rns[i], i) + e_rns[1, i].astype(object)) % qi
        assert (c[0, i].astype(object) == want0).all() and (c[1, i].astype(object) == want1).all(), i

    # secret-key mode: c0 = delta m + r - (c1 s + e)
    cs, es, rs = enc.encrypt_symmetric_return_components(plain)
    assert es.num_polynomials() == 1 and es.coeff_modulus_size() == K and rs.len() == n
    assert be.decode_unsigned(dec.decrypt(cs)) == data
    assert (np.array([rs.get_coefficient(i) for i in range(n)], dtype=object) == rv).all()
    csa, es_rns = cs.to_array(), es.as_rns_u64s().reshape(K, n)
    assert np.abs(centred(es_rns[0], 0).astype(np.float64)).max() <= 19
    for i in range(K):
        qi = primes[i]
        want0 = (delta % qi * m + rv - negacyclic(csa[1, i], s_rns[i], i) - es_rns[i].astype(object)) % qi
        assert (csa[0, i].astype(object) == want0).all(), i
    c2 = enc.encrypt_symmetric(plain)
    assert be.decode_unsigned(dec.decrypt(c2)) == data
    assert dec.invariant_noise_budget(c2) >= dec.invariant_noise_budget(enc.encrypt(plain)) - 1  # symmetric noise is no larger
    assert not (c2.to_array() == cs.to_array()).all()
    only_sk = Encryptor.with_secret_key(ctx, sk)
    assert be.decode_unsigned(dec.decrypt(only_sk.encrypt_symmetric(plain))) == data
    from sunscreen_amd import HipBfvError

    with pytest.raises(HipBfvError):
        only_sk.encrypt(plain)
    with pytest.raises(HipBfvError):
        Encryptor(ctx, pk).encrypt_symmetric(plain)

    # the "deterministic" feature: equal seeds, equal ciphertexts and components (encryptor_decryptor.rs:886-935)
    z = [0] * 8
    d1, d2 = enc.encrypt_deterministic(plain, z), enc.encrypt_deterministic(plain, z)
    assert (d1.to_array() == d2.to_array()).all() and be.decode_unsigned(dec.decrypt(d1)) == data
    assert not (enc.encrypt_deterministic(plain, [1] + [0] * 7).to_array() == d1.to_array()).all()
    k1, k2 = enc.encrypt_return_components(plain, z), enc.encrypt_return_components(plain, z)
    assert (k1[0].to_array() == k2[0].to_array()).all() and k1[1] == k2[1] and k1[2] == k2[2]
    s1, s2 = enc.encrypt_symmetric_return_components(plain, z), enc.encrypt_symmetric_return_components(plain, z)
    assert (s1[0].to_array() == s2[0].to_array()).all() and s1[1] == s2[1]
    # the fused encryption path (errors regenerated inside encrypt_finish_kernel) and the component-returning path
    # (errors materialised) are the same function of (seed, counter): identical bits
    fa = Encryptor(ctx, pk, seed=11).encrypt(plain)
    fb = Encryptor(ctx, pk, seed=11).encrypt_return_components(plain, disable_special_modulus=False)[0]
    assert (fa.to_array() == fb.to_array()).all()
    # clone / drop (data_structures.rs:497-533)
    uc = u.clone()
    assert uc == u and uc.get_handle().value != u.get_handle().value
    if K > 1:
        low = pa_pk.drop_modulus()
        assert (low.num_polynomials(), low.poly_modulus_degree(), low.coeff_modulus_size()) == (2, n, K - 1)
        assert (low.as_rns_u64s().reshape(2, K - 1, n) == p_rns[:, : K - 1]).all()
        lmp = low.as_multiprecision_u64s().reshape(2, n, K - 1)
        v = sum(int(lmp[1, 5, l]) << (64 * l) for l in range(K - 1))
        assert [v % primes[i] for i in range(K - 1)] == [int(p_rns[1, i, 5]) for i in range(K - 1)]
    # components of a lower level of the chain: the array follows the ciphertext's level
    from sunscreen_amd import BFVEvaluator

    if K > 1:
        lower = BFVEvaluator(ctx).mod_switch_to_next(ct)
        pl = PolynomialArray.new_from_ciphertext(ctx, lower)
        assert pl.coeff_modulus_size() == K - 1 and (pl.as_u64s() == lower.to_array().ravel()).all()
    assert isinstance(r, Plaintext)
