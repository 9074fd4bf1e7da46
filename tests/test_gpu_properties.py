"""Size-independent properties of the HIP path at BASELINE.json's full sizes (n=8192, 4096-item batches)
where the CPU oracle would take minutes, plus batching / chunking / threading invariants.

Properties used (all exact, mod q_i, bit for bit):
  * INTT(NTT(x)) == x and NTT(a+b) == NTT(a)+NTT(b)                              (config 2: 4096 polys x 3 primes)
  * sub(add(a,b), b) == a,  negate(negate(a)) == a
  * multiply(a,b) == multiply(b,a)   (BEHZ is symmetric in its operands)
  * a result never depends on the batch position, the chunking or the path (split vs whole-polynomial kernels)
  * spot items of the full batch equal the oracle; every output word is a canonical residue
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.bfv_helpers import oracle_for, params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _uniform(torch, shape_prefix, primes, n, gen, dev="cuda:0"):
    out = torch.empty(shape_prefix + (len(primes), n), dtype=torch.int64, device=dev)
    for i, q in enumerate(primes):
        out[..., i, :] = torch.randint(0, q, shape_prefix + (n,), generator=gen, device=dev, dtype=torch.int64)
    return out


def _mod_add(torch, a, b, primes):
    out = torch.empty_like(a)
    for i, q in enumerate(primes):
        s = a[..., i, :] + b[..., i, :]
        out[..., i, :] = torch.where(s >= q, s - q, s)
    return out


def test_config2_ntt_roundtrip_and_linearity_full_batch():
    import torch
    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_host

    n, primes, t = params("default_8192_17")
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(0x5EA10001)
    B, P = 4096, 3
    a = _uniform(torch, (B,), primes[:P], n, gen)
    b = _uniform(torch, (B,), primes[:P], n, gen)
    s = _mod_add(torch, a, b, primes[:P])
    fa = ev.ntt(a.reshape(B * P, n).clone(), P)
    fb = ev.ntt(b.reshape(B * P, n).clone(), P)
    fs = ev.ntt(s.reshape(B * P, n).clone(), P)
    assert torch.equal(fs.reshape(B, P, n), _mod_add(torch, fa.reshape(B, P, n), fb.reshape(B, P, n), primes[:P]))
    back = ev.ntt(fa.clone(), P, inverse=True)
    assert torch.equal(back.reshape(B, P, n), a)
    # 64-poly subset against the oracle (BASELINE.md section 3 parity gate)
    o = oracle_for("default_8192_17")
    got = to_host(fa.reshape(B, P, n)[:22])
    src = to_host(a[:22])
    for i in range(22):
        for p in range(P):
            assert (got[i, p] == o.ntt(p, src[i, p])).all()


def test_config3_mul_relin_full_batch_properties():
    import torch
    from sunscreen_amd import Context, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    name = "default_8192_17"
    n, primes, t = params(name)
    o = oracle_for(name)
    O.seed(2024)
    sk, pk, rk, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    K = ctx.K
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(7)
    B = 4096
    a = _uniform(torch, (B, 2), primes[:K], n, gen)
    b = _uniform(torch, (B, 2), primes[:K], n, gen)
    # real encryptions at both ends of the batch and duplicated items across a chunk boundary
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 200, (4, n)).astype(np.uint64)
    enc = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals])
    for pos, src in ((0, 0), (1, 1), (B - 1, 2), (B - 2, 3)):
        a[pos] = to_device(enc[src : src + 1])[0]
        b[pos] = to_device(enc[(src + 1) % 4 : (src + 1) % 4 + 1])[0]
    a[1500], b[1500] = a[0], b[0]
    out = ev.multiply_relin(a, b, rkd)
    torch.cuda.synchronize()
    assert torch.equal(out[1500], out[0])  # independent of batch position / chunk
    # commutativity of the tensor product
    out_t = ev.multiply_relin(b, a, rkd)
    assert torch.equal(out, out_t)
    # canonical residues everywhere
    for i, q in enumerate(primes[:K]):
        assert int(out[:, :, i, :].max()) < q and int(out[:, :, i, :].min()) >= 0
    # spot items against the oracle (bit-exact) and decrypt-correct
    h = to_host(out[[0, 1, B - 1, B - 2]])
    for j, (pos, src) in enumerate(((0, 0), (1, 1), (B - 1, 2), (B - 2, 3))):
        ref = o.relinearize(o.multiply(enc[src], enc[(src + 1) % 4]), rk)
        assert (h[j] == ref).all()
        assert (o.batch_decode(o.decrypt(h[j], sk)) == (vals[src] * vals[(src + 1) % 4]) % t).all()
    # chunking invariance: a different chunk size gives identical bits
    ev.set_chunk_ops(37)
    out_c = ev.multiply_relin(a[:300], b[:300], rkd)
    assert torch.equal(out_c, out[:300])
    # add / sub / negate identities on the full batch
    s = ev.add(a, b)
    assert torch.equal(ev.sub(s, b), a)
    assert torch.equal(ev.negate(ev.negate(a)), a)


def _extreme_rows(primes, K, n):
    """Operand pairs at the edges of the BEHZ bounds: every residue q_i - 1, all zero, and alternating."""
    import numpy as np

    hi = np.stack([np.full((2, n), q - 1, dtype=np.int64) for q in primes[:K]], axis=1)  # [2, K, n]
    zero = np.zeros_like(hi)
    alt = hi.copy()
    alt[:, :, ::2] = 0
    one = np.zeros_like(hi)
    one[:, :, 0] = 1
    return np.stack([hi, hi, alt, zero, one]), np.stack([hi, alt, alt, hi, hi])


def test_split_and_whole_polynomial_paths_agree():
    """The head/middle/tail kernels and the whole-polynomial kernels are two implementations of the same
    arithmetic, and the library's own FP64 auxiliary base must give the same bits as SEAL's 61-bit base:
    run the same inputs (random pairs plus operands at the edges of the BEHZ bounds) through every variant in
    separate processes, and check the edge cases against the oracle (which always uses SEAL's base)."""
    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.bfv_helpers import params
from tests.test_gpu_properties import _extreme_rows
from sunscreen_amd import Context, RelinearizationKeys, GaloisKeys
from sunscreen_amd.batch import BatchEvaluator
from oracle import bfv_oracle as O
n, primes, t = params("default_8192_17")
o = O.Oracle(n, primes, t); O.seed(9); sk, pk, rk, gk = o.keygen(galois_elts=[3])
ctx = Context.from_raw(n, primes, t); ev = BatchEvaluator(ctx)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(11)
a = torch.empty((9, 2, ctx.K, n), dtype=torch.int64, device="cuda:0"); b = torch.empty_like(a)
for i, q in enumerate(primes[:ctx.K]):
    a[:, :, i, :] = torch.randint(0, q, (9, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
    b[:, :, i, :] = torch.randint(0, q, (9, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
xa, xb = _extreme_rows(primes, ctx.K, n)
a = torch.cat([a, torch.from_numpy(xa).cuda()]); b = torch.cat([b, torch.from_numpy(xb).cuda()])
r = ev.multiply_relin(a, b, RelinearizationKeys.from_array(ctx, rk))
m = ev.multiply(a, b)
g = ev.apply_galois(a, 3, GaloisKeys.from_arrays(ctx, gk))
ev.set_transparent_check(False)  # the extreme rows include all-zero operands; x * x of those is what is compared
sq = ev.multiply(a, a)            # one operand twice: the squaring specialisation of the split kernels
sqr = ev.multiply_relin(b, b, RelinearizationKeys.from_array(ctx, rk))
pl = torch.randint(0, t, (a.shape[0], n), generator=gen, device="cuda:0", dtype=torch.int64)
pl[1, 1:] = 0; pl[1, 0] = t - 1   # a monomial in the upper half: SEAL's shortcut uses the coefficient without the centred lift
mp = ev.multiply_plain(a, pl)     # per-op plaintexts
mp1 = ev.multiply_plain(m, pl[0]) # one plaintext for every op, size-3 ciphertexts
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([x.cpu().numpy().ravel() for x in (r, m, g, sq, sqr, mp, mp1)]))
""" % ROOT
    import tempfile

    variants = (
        ("split", {"HIPBFV_NO_SMALL_BATCH": "1"}),  # every variant below inherits the pin from the `pipeline_selection` fixture too
        ("split_unfused_tail", {"HIPBFV_NO_FUSED_TAIL": "1"}),  # multiply then relinearize through a c0/c1/c2 buffer instead of mulrelin_tail
        ("split_unfused_head", {"HIPBFV_NO_FUSED_HEAD": "1"}),  # c2 through HBM between mul_tail and ks_head instead of mulrelin_head
        ("small_batch_selection", {"HIPBFV_NO_SMALL_BATCH": "0"}),  # the product default: these 14 ciphertexts take the whole-polynomial multiply
        ("split_unpacked", {"HIPBFV_NO_PACK": "1"}),  # 8-byte instead of 48-bit packed intermediates
        ("split_no_grid", {"HIPBFV_NO_GRID": "1"}),  # base-conversion sums reduced term by term instead of once (griddot.hpp)
        ("whole", {"HIPBFV_NO_SPLIT_MUL": "1", "HIPBFV_NO_SPLIT_KS": "1"}),
        ("seal_aux", {"HIPBFV_SEAL_AUX": "1"}),
        ("seal_aux_whole", {"HIPBFV_SEAL_AUX": "1", "HIPBFV_NO_SPLIT_MUL": "1", "HIPBFV_NO_SPLIT_KS": "1"}),
        ("int", {"HIPBFV_NO_F64": "1"}),
    )
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for tag, env in variants:
            path = os.path.join(td, tag + ".npy")
            subprocess.check_call([sys.executable, "-c", script, path], env=dict(os.environ, **env))
            outs.append(np.load(path))
    for (tag, _), out in zip(variants[1:], outs[1:]):
        assert (outs[0] == out).all(), tag
    # the edge-case pairs against the oracle
    from oracle import bfv_oracle as O
    from tests.bfv_helpers import params

    n, primes, t = params("default_8192_17")
    o = O.Oracle(n, primes, t)
    o.throw_on_transparent = False  # the extreme rows include all-zero polynomials: the bits are what is compared here
    O.seed(9)
    sk, pk, rk, gk = o.keygen(galois_elts=[3])
    K = len(primes) - 1
    xa, xb = _extreme_rows(primes, K, n)
    count = 9 + len(xa)
    r = outs[0][: count * 2 * K * n].reshape(count, 2, K, n)
    m = outs[0][count * 2 * K * n : count * 5 * K * n].reshape(count, 3, K, n)
    for i in range(len(xa)):
        om = o.multiply(xa[i].astype(np.uint64), xb[i].astype(np.uint64))
        assert (m[9 + i].astype(np.uint64) == om).all(), i
        assert (r[9 + i].astype(np.uint64) == o.relinearize(om, rk)).all(), i
    # the squares against the oracle's multiply(x, x): two random items and every extreme row
    g_words = count * 2 * K * n
    sq0 = count * 5 * K * n + g_words
    sq = outs[0][sq0 : sq0 + count * 3 * K * n].reshape(count, 3, K, n).astype(np.uint64)
    sqr0 = sq0 + count * 3 * K * n
    sqr = outs[0][sqr0 : sqr0 + count * 2 * K * n].reshape(count, 2, K, n).astype(np.uint64)
    assert outs[0].size == sqr0 + count * 2 * K * n + count * 2 * K * n + count * 3 * K * n  # ... + mp + mp1
    for i in range(len(xa)):
        assert (sq[9 + i] == o.multiply(xa[i].astype(np.uint64), xa[i].astype(np.uint64))).all(), i
        xs = xb[i].astype(np.uint64)
        assert (sqr[9 + i] == o.relinearize(o.multiply(xs, xs), rk)).all(), i


@pytest.mark.parametrize("name", ["default_4096_16", "default_8192_17", "default_16384_17", "seal_fhe_unit"])
def test_grid_base_conversion_sums_give_the_same_bits(name):
    """The multiply's q -> Bsk sums are formed exactly on an FP64 grid and reduced once (griddot.hpp) or, with
    HIPBFV_NO_GRID=1, reduced term by term: every degree's head / tail instantiation (4 and 8 primes) must produce the same
    product in both forms, on random operands and at the edges of the bounds, and the edge cases must equal the oracle."""
    import tempfile

    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.bfv_helpers import params
from tests.test_gpu_properties import _extreme_rows
from sunscreen_amd import Context
from sunscreen_amd.batch import BatchEvaluator
n, primes, t = params(sys.argv[2])
ctx = Context.from_raw(n, primes, t); ev = BatchEvaluator(ctx)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(23)
a = torch.empty((5, 2, ctx.K, n), dtype=torch.int64, device="cuda:0"); b = torch.empty_like(a)
for i, q in enumerate(primes[:ctx.K]):
    a[:, :, i, :] = torch.randint(0, q, (5, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
    b[:, :, i, :] = torch.randint(0, q, (5, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
xa, xb = _extreme_rows(primes, ctx.K, n)
a = torch.cat([a, torch.from_numpy(xa).cuda()]); b = torch.cat([b, torch.from_numpy(xb).cuda()])
m = ev.multiply(a, b)
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([m.cpu().numpy().ravel(), np.array([int(ctx.conv_grid)], dtype=np.int64)]))
""" % ROOT
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for tag, env in (("grid", {}), ("per_term", {"HIPBFV_NO_GRID": "1"})):
            path = os.path.join(td, tag + ".npy")
            subprocess.check_call([sys.executable, "-c", script, path, name], env=dict(os.environ, **env))
            outs.append(np.load(path))
    assert outs[1][-1] == 0, "HIPBFV_NO_GRID=1 must select the per-term form"
    switched_off = any(os.environ.get(k) == "1" for k in ("HIPBFV_NO_GRID", "HIPBFV_SEAL_AUX", "HIPBFV_NO_F64"))
    if name.startswith("default_") and not switched_off:  # (seal_fhe_unit has 50-bit primes: the plan may refuse them)
        assert outs[0][-1] == 1, "the default parameter sets are within the grid plan's bounds"
    assert (outs[0][:-1] == outs[1][:-1]).all()
    n, primes, t = params(name)
    o = oracle_for(name)
    K = len(primes) - 1
    xa, xb = _extreme_rows(primes, K, n)
    m = outs[0][:-1].reshape(5 + len(xa), 3, K, n)
    o = O.Oracle(n, primes, t)  # a private one: the all-zero operand gives a transparent product, compared bit for bit here
    o.throw_on_transparent = False
    for i in range(len(xa)):
        assert (m[5 + i].astype(np.uint64) == o.multiply(xa[i].astype(np.uint64), xb[i].astype(np.uint64))).all(), i


@pytest.mark.parametrize(
    "n,bits",
    [
        (4096, [36, 36, 36, 36, 36, 37]),          # K = 5: the 8-prime instantiation with small primes (a coarse grid suffices)
        (4096, [44] * 8 + [45]),                    # K = 8, n = 8192-sized primes
        (8192, [47] * 6 + [48]),                    # K = 6 close to the plan's limit
        (4096, [49] * 6 + [50]),                    # K = 6 with the widest FP64-policy primes
        (16384, [48, 48, 48, 49, 49, 49, 49, 49, 49]),  # the n = 16384 default sizes
    ],
)
def test_multiply_around_the_grid_plan_limit(n, bits):
    """The 8-prime head / tail instantiation (5..8 data primes) across prime sizes up to plan_grid_dot()'s bounds: the product
    equals the oracle's (SEAL's 61-bit base, integer arithmetic) on random operands and at the edges, whichever form the
    context's plan selected."""
    import torch

    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator

    primes = O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    o.throw_on_transparent = False  # _extreme_rows has an all-zero operand: the (transparent) product's bits are compared
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    K = len(primes) - 1
    rng = np.random.default_rng(n + sum(bits))
    a = np.stack([rng.integers(0, q, (2, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2).astype(np.int64)
    b = np.stack([rng.integers(0, q, (2, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2).astype(np.int64)
    xa, xb = _extreme_rows(primes, K, n)
    a, b = np.concatenate([a, xa]), np.concatenate([b, xb])
    m = ev.multiply(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy().astype(np.uint64)
    for i in range(len(a)):
        assert (m[i] == o.multiply(a[i].astype(np.uint64), b[i].astype(np.uint64))).all(), (i, ctx.aux_fp64, ctx.conv_grid)


@pytest.mark.parametrize(
    "n,bits",
    [
        (16384, [48, 48, 48, 49, 49, 49, 49, 49, 49]),  # the SEAL default sizes of n = 16384: rows 0-2 and the 10 auxiliary rows packed
        (8192, [49, 44, 48, 49, 47, 49, 48]),           # K = 6: packed and 8-byte data rows interleaved (rows 1, 2, 4 packed)
        (8192, [49, 49, 49, 49, 49, 49]),               # K = 5: no data row packed, every auxiliary row packed
    ],
)
def test_per_row_packing_of_the_multiply_intermediates(n, bits, monkeypatch):
    """r04: with data primes on both sides of 2^48 the split multiply packs its intermediates PER ROW (DevCtx::pack_mul == 2: the
    rows whose prime is below 2^48 travel as 6 bytes, the others as 8; the default since r06, HIPBFV_PACK_ROWS=0 restores 8-byte
    rows -- context.cpp).  The plain product, the fused multiply + relinearize and the squaring instantiations -- each reads
    and writes the mixed rows in its own kernels -- equal the oracle on random operands and at the edges of the BEHZ bounds, and the
    default context (8-byte rows throughout) gives the same bits."""
    import torch

    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator

    if any(os.environ.get(k) == "1" for k in ("HIPBFV_SEAL_AUX", "HIPBFV_NO_F64")):
        pytest.skip("packed rows exist on the FP64 pipe with the library's own auxiliary base only")
    primes = O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    o.throw_on_transparent = False  # _extreme_rows has an all-zero operand: the (transparent) product's bits are compared
    O.seed(31)
    sk, pk, rk, gk = o.keygen(galois_elts=[3, 2 * n - 1])
    K = len(primes) - 1
    rng = np.random.default_rng(n + sum(bits))
    a = np.stack([rng.integers(0, q, (3, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2).astype(np.int64)
    b = np.stack([rng.integers(0, q, (3, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2).astype(np.int64)
    xa, xb = _extreme_rows(primes, K, n)
    a, b = np.concatenate([a, xa]), np.concatenate([b, xb])
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    got = {}
    monkeypatch.delenv("HIPBFV_NO_PACK", raising=False)
    for tag, env in (("rows", None), ("bytes8", "0")):
        if env:
            monkeypatch.setenv("HIPBFV_PACK_ROWS", env)
        else:
            monkeypatch.delenv("HIPBFV_PACK_ROWS", raising=False)
        ctx = Context.from_raw(n, primes, t)  # the switch is read when the context is built
        ev = BatchEvaluator(ctx)
        ev.set_transparent_check(False)
        assert ctx.aux_fp64 and ctx.packed_mul_rows == (env is None) and (ctx.packed_mul == (env is None)), (tag, ctx.packed_mul, ctx.packed_mul_rows)
        # r06: the key switch's rows T / ACC per KEY prime too (DevCtx::pack_ks == 2) when some key primes are below 2^48 -- the second
        # set has the SPECIAL prime among them (its accumulator row packed), the third none
        some_below = any(p < (1 << 48) for p in primes)
        assert ctx.packed_ks_rows == (env is None and some_below) and ctx.packed_ks == ctx.packed_ks_rows, (tag, ctx.packed_ks, ctx.packed_ks_rows)
        rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
        prod = ev.multiply(da, db)
        # (the stand-alone relinearize and the rotations run ks_head / ks_tail, the fused forms mulrelin_head / mulrelin_tail)
        got[tag] = [x.cpu().numpy().astype(np.uint64) for x in (prod, ev.multiply_relin(da, db, rkd), ev.multiply(da, da), ev.multiply_relin(db, db, rkd),
                                                                ev.relinearize(prod, rkd), ev.rotate_rows(da, 1, gkd), ev.rotate_columns(db, gkd))]
    for x, y in zip(got["rows"], got["bytes8"]):
        assert (x == y).all()
    m, r, sq, sqr, rl, rr, rc = got["rows"]
    assert (rl == r).all()
    for i in (0, len(a) - 1):
        assert (rr[i] == o.rotate_rows(a[i].astype(np.uint64), 1, gk)).all(), i
        assert (rc[i] == o.rotate_columns(b[i].astype(np.uint64), gk)).all(), i
    for i in list(range(2)) + list(range(3, len(a))):  # two random items and every edge row
        ua, ub = a[i].astype(np.uint64), b[i].astype(np.uint64)
        om = o.multiply(ua, ub)
        assert (m[i] == om).all(), i
        assert (r[i] == o.relinearize(om, rk)).all(), i
        assert (sq[i] == o.multiply(ua, ua)).all(), i
        assert (sqr[i] == o.relinearize(o.multiply(ub, ub), rk)).all(), i


def test_per_row_packing_down_the_modulus_chain_at_n16384():
    """The level contexts of n = 16384 (sunscreen_runtime never switches levels itself -- seal_fhe::Evaluator::mod_switch_to_next
    is the caller's, seal_fhe/src/evaluator_base.rs -- but a program that does lands here): K = 8 -> 7 -> 6 -> 5 keep the per-row
    packed pipelines (three 48-bit primes beside 49-bit ones, their own row masks and residue lists), K = 4 leaves them (four-prime
    instantiations, 8-byte rows).  Batches of six (above the small-batch cut-over), every level against the oracle of that level:
    the fused multiply + relinearize, the stand-alone relinearize and a rotation."""
    import torch

    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator

    if any(os.environ.get(k) == "1" for k in ("HIPBFV_SEAL_AUX", "HIPBFV_NO_F64")):
        pytest.skip("packed rows exist on the FP64 pipe with the library's own auxiliary base only")
    n = 16384
    primes, t = O.bfv_default(n), O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    O.seed(77)
    sk, pk, rk, gk = o.keygen(galois_elts=[3])
    ctx = Context.from_raw(n, primes, t)
    rng = np.random.default_rng(5)
    vals = rng.integers(0, 30, (2, n)).astype(np.uint64)
    ra, rb = (o.encrypt(pk, o.batch_encode(v)) for v in vals)
    per_row = not (os.environ.get("HIPBFV_NO_PACK") or os.environ.get("HIPBFV_PACK_ROWS") == "0")
    for level in range(5):  # K = 8, 7, 6, 5, 4
        ev = BatchEvaluator(ctx)
        if per_row:
            assert ctx.packed_mul_rows == (o.K > 4) and ctx.packed_ks_rows == (o.K > 4), (level, o.K)
        rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
        da = torch.from_numpy(np.stack([ra, rb, ra, rb, ra, rb]).astype(np.int64)).cuda()
        db = torch.from_numpy(np.stack([rb, rb, ra, ra, rb, ra]).astype(np.int64)).cuda()
        prod = o.multiply(ra, rb)
        exp = o.relinearize(prod, rk)
        got = ev.multiply_relin(da, db, rkd).cpu().numpy().astype(np.uint64)
        assert (got[0] == exp).all() and (got[4] == exp).all(), level
        assert (got[1] == o.relinearize(o.multiply(rb, rb), rk)).all(), level
        rel = ev.relinearize(ev.multiply(da, db), rkd).cpu().numpy().astype(np.uint64)
        assert (rel == got).all(), level
        rot = ev.rotate_rows(da, 1, gkd).cpu().numpy().astype(np.uint64)
        assert (rot[0] == o.rotate_rows(ra, 1, gk)).all() and (rot[1] == o.rotate_rows(rb, 1, gk)).all(), level
        if level == 4:
            break
        # the next level: the oracle's view drops the last data prime everywhere (keys keep the special prime's row)
        sw = ev.mod_switch(torch.from_numpy(np.stack([ra, rb]).astype(np.int64)).cuda()).cpu().numpy().astype(np.uint64)
        ra, rb = o.mod_switch_to_next(ra), o.mod_switch_to_next(rb)
        assert (sw[0] == ra).all() and (sw[1] == rb).all(), level
        o2 = o.next_level()
        rows = list(range(o2.K)) + [o.KK - 1]
        rk = np.ascontiguousarray(rk[: o2.K][:, :, rows, :])
        gk = {e: np.ascontiguousarray(g[: o2.K][:, :, rows, :]) for e, g in gk.items()}
        o, ctx = o2, ctx.next_level()
        assert ctx.K == o.K


def test_concurrent_host_threads_on_one_evaluator():
    """sunscreen_runtime/src/run.rs:415-469 calls one evaluator from a rayon pool: handle-level calls must be
    thread-safe (one non-blocking HIP stream per host thread, no shared mutable scratch)."""
    from sunscreen_amd import BFVEvaluator, Ciphertext, Context, RelinearizationKeys

    name = "default_4096_16"
    n, primes, t = params(name)
    o = oracle_for(name)
    O.seed(31)
    sk, pk, rk, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    ev = BFVEvaluator(ctx)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    rng = np.random.default_rng(3)
    nthreads, iters = 8, 6
    vals = rng.integers(0, 30, (nthreads, 2, n)).astype(np.uint64)
    cts = [[o.encrypt(pk, o.batch_encode(vals[i, j])) for j in range(2)] for i in range(nthreads)]
    expected = [o.add(o.relinearize(o.multiply(c[0], c[1]), rk), c[0]) for c in cts]
    errors = []

    def worker(i):
        try:
            a, b = Ciphertext.from_array(ctx, cts[i][0]), Ciphertext.from_array(ctx, cts[i][1])
            for _ in range(iters):
                m = ev.multiply(a, b)
                ev.relinearize_inplace(m, rkd)
                r = ev.add(m, a)
                if not (r.to_array() == expected[i]).all():
                    errors.append((i, "mismatch"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    assert not errors, errors


def test_batches_beyond_the_grid_dimension_limit():
    """Launch grids carry the batch in a 16-bit dimension: batches above 65535 items (and a chunk size set above it)
    must be split correctly by every operation.  Checked against the same operation on slices around the chunk
    boundaries, which in turn are covered by the oracle tests."""
    import torch

    from sunscreen_amd import Context, GaloisKeys, PublicKey, RelinearizationKeys, SecretKey
    from sunscreen_amd.batch import BatchEvaluator

    n = 4096
    primes, t = O.bfv_default(n), O.plain_batching(n, 16)
    o = O.Oracle(n, primes, t)
    O.seed(77)
    elt = o.galois_elt_from_step(1)
    sk, pk, rk, gk = o.keygen(galois_elts=[elt])
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    ev.set_chunk_ops(1 << 20)
    rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
    skd, pkd = SecretKey.from_array(ctx, sk), PublicKey.from_array(ctx, pk)
    B = 66000
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(5)
    K = ctx.K
    a = torch.empty((B, 2, K, n), dtype=torch.int64, device="cuda:0")
    for i, q in enumerate(primes[:K]):
        a[:, :, i, :] = torch.randint(0, q, (B, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
    b = a.flip(0).contiguous()
    pl = torch.randint(1, t, (B, n), generator=gen, device="cuda:0", dtype=torch.int64)
    probes = [0, 1, 3275, 3276, 3277, 6552, 32767, 32768, 65534, 65535, 65536, B - 1]

    def check(full, fn):
        for i in probes:
            assert torch.equal(full[i : i + 1], fn(i)), i

    check(ev.add(a, b), lambda i: ev.add(a[i : i + 1].contiguous(), b[i : i + 1].contiguous()))
    check(ev.negate(a), lambda i: ev.negate(a[i : i + 1].contiguous()))
    check(ev.add_plain(a, pl), lambda i: ev.add_plain(a[i : i + 1].contiguous(), pl[i : i + 1].contiguous()))
    check(ev.multiply_plain(a, pl), lambda i: ev.multiply_plain(a[i : i + 1].contiguous(), pl[i : i + 1].contiguous()))
    m = ev.multiply(a, b)
    check(m, lambda i: ev.multiply(a[i : i + 1].contiguous(), b[i : i + 1].contiguous()))
    check(ev.relinearize(m, rkd), lambda i: ev.relinearize(m[i : i + 1].contiguous(), rkd))
    del m
    check(ev.multiply_relin(a, b, rkd), lambda i: ev.multiply_relin(a[i : i + 1].contiguous(), b[i : i + 1].contiguous(), rkd))
    check(ev.rotate_rows(a, 1, gkd), lambda i: ev.rotate_rows(a[i : i + 1].contiguous(), 1, gkd))
    check(ev.decrypt(a, skd), lambda i: ev.decrypt(a[i : i + 1].contiguous(), skd))
    enc = ev.encode(pl % t)
    check(enc, lambda i: ev.encode((pl[i : i + 1] % t).contiguous()))
    assert torch.equal(ev.decode(enc), pl % t)
    ct = ev.encrypt(enc, pkd, seed=9)
    check(ct, lambda i: ev.encrypt(enc[i : i + 1].contiguous(), pkd, seed=9, first_op=i))
    assert torch.equal(ev.decrypt(ct, skd), enc)


def test_lane_split_geometry_build_gives_the_same_bits_at_n16384():
    """The N = 16384 pipelines exist in two geometries (sunscreen_amd/csrc/nttshape.hpp, HIPBFV_GEOM14): the default (blocks of
    N/4) and the lane-split one (blocks of N/8; pairs of lanes share eight coefficients and trade two values per residue with
    v_permlane32_swap inside the head's and the tail's three stages), built as sunscreen_amd/lib/variants/libhipbfv_geom8.so
    by `make variants`.  Same inputs through both libraries in separate processes: multiply, the fused multiply+relinearize,
    relinearize and a rotation must agree word for word, and with the oracle."""
    lib = os.path.join(ROOT, "sunscreen_amd", "lib", "variants", "libhipbfv_geom8.so")
    assert os.path.exists(lib), "build the variant library first: make -C sunscreen_amd/csrc variants (build() does)"
    script = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.bfv_helpers import params
from sunscreen_amd import Context, RelinearizationKeys, GaloisKeys
from sunscreen_amd.batch import BatchEvaluator
from oracle import bfv_oracle as O
n, primes, t = params("default_16384_17")
o = O.Oracle(n, primes, t); O.seed(14); sk, pk, rk, gk = o.keygen(galois_elts=[3])
ctx = Context.from_raw(n, primes, t); ev = BatchEvaluator(ctx)
gen = torch.Generator(device="cuda:0"); gen.manual_seed(14)
a = torch.empty((5, 2, ctx.K, n), dtype=torch.int64, device="cuda:0"); b = torch.empty_like(a)
for i, q in enumerate(primes[:ctx.K]):
    a[:, :, i, :] = torch.randint(0, q, (5, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
    b[:, :, i, :] = torch.randint(0, q, (5, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
    a[4, :, i, :] = q - 1          # every residue at its maximum
    b[4, :, i, ::2] = 0
rkd = RelinearizationKeys.from_array(ctx, rk)
r = ev.multiply_relin(a, b, rkd)
m = ev.multiply(a, b)
r2 = ev.relinearize(m, rkd)
g = ev.apply_galois(a, 3, GaloisKeys.from_arrays(ctx, gk))
torch.cuda.synchronize()
assert torch.equal(r, r2)
np.save(sys.argv[1], np.concatenate([x.cpu().numpy().ravel() for x in (a, b, r, m, g)]))
""" % ROOT
    import tempfile

    outs = []
    with tempfile.TemporaryDirectory() as td:
        for tag, env in (("default", {}), ("geom8", {"HIPBFV_LIB": lib})):
            path = os.path.join(td, tag + ".npy")
            subprocess.check_call([sys.executable, "-c", script, path], env=dict(os.environ, **env))
            outs.append(np.load(path))
    assert (outs[0] == outs[1]).all()
    from tests.bfv_helpers import params

    n, primes, t = params("default_16384_17")
    K = len(primes) - 1
    o = O.Oracle(n, primes, t)
    o.throw_on_transparent = False
    O.seed(14)
    sk, pk, rk, gk = o.keygen(galois_elts=[3])
    w = 5 * 2 * K * n
    a = outs[1][:w].reshape(5, 2, K, n).astype(np.uint64)
    b = outs[1][w : 2 * w].reshape(5, 2, K, n).astype(np.uint64)
    r = outs[1][2 * w : 3 * w].reshape(5, 2, K, n).astype(np.uint64)
    for i in (0, 4):
        assert (r[i] == o.relinearize(o.multiply(a[i], b[i]), rk)).all(), i
