"""GPU batch executor (program graphs) vs the oracle interpreter, bit for bit, plus the semantic results
of the reference's examples (examples/chi_sq, examples/dot_prod, examples/pir; run.rs:595-881)."""
import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.bfv_helpers import oracle_for, params
from tests.oracle_program import run_program

pytestmark = pytest.mark.gpu


def _ctx(name, galois=None, seed=3):
    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator

    n, primes, t = params(name)
    o = oracle_for(name)
    O.seed(seed)
    sk, pk, rk, gk = o.keygen(galois_elts=galois)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    return o, sk, pk, rk, gk, ev, RelinearizationKeys.from_array(ctx, rk), (GaloisKeys.from_arrays(ctx, gk) if gk else None)


def _signed(o, v):
    v = v.astype(np.int64)
    return np.where(v > o.t // 2, v - o.t, v)


def test_chi_sq_graph_batched():
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.workloads import chi_sq_optimized

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_8192_17")
    prog = chi_sq_optimized()
    assert prog.num_outputs() == 4
    batch = 3
    rng = np.random.default_rng(1)
    vals = rng.integers(0, 7, (3, batch, o.n)).astype(np.uint64)
    vals[:, 0, :] = np.array([2, 7, 9], dtype=np.uint64)[:, None]  # the example's own inputs (main.rs:236-238)
    cts = [np.stack([o.encrypt(pk, o.batch_encode(vals[a, i])) for i in range(batch)]) for a in range(3)]
    outs = prog.run(ev, [to_device(c) for c in cts], rkd)
    outs = [to_host(t) for t in outs]
    for i in range(batch):
        ref = run_program(o, prog.nodes, prog.edges, [c[i] for c in cts], rk)
        n0, n1, n2 = (vals[a, i].astype(np.int64) for a in range(3))
        x, y = 2 * n0 + n1, 2 * n2 + n1
        expect = [(4 * n0 * n2 - n1 * n1) ** 2, 2 * x * x, x * y, 2 * y * y]
        for k in range(4):
            assert (outs[k][i] == ref[k]).all(), (i, k)
            assert (_signed(o, o.batch_decode(o.decrypt(outs[k][i], sk))) == expect[k]).all()
    d0 = [int(_signed(o, o.batch_decode(o.decrypt(outs[k][0], sk)))[0]) for k in range(4)]
    assert d0 == [529, 242, 275, 1250]  # examples/chi_sq/src/main.rs expected values for (2, 7, 9)


def test_dot_product_graph_rotations_and_json_roundtrip():
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram
    from sunscreen_amd.workloads import dot_product

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16", galois="all")
    lanes = o.n // 2
    prog = FheProgram.from_json(dot_product(lanes).to_json())  # through the serde JSON form
    batch = 2
    rng = np.random.default_rng(2)
    va = rng.integers(0, 4, (batch, o.n)).astype(np.uint64)
    vb = rng.integers(0, 4, (batch, o.n)).astype(np.uint64)
    ca = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in va])
    cb = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vb])
    (out,) = prog.run(ev, [to_device(ca), to_device(cb)], rkd, gkd)
    out = to_host(out)
    for i in range(batch):
        (ref,) = run_program(o, prog.nodes, prog.edges, [ca[i], cb[i]], rk, gk)
        assert (out[i] == ref).all()
        dot = int((va[i].astype(np.int64) * vb[i].astype(np.int64)).sum()) % o.t
        assert (o.batch_decode(o.decrypt(out[i], sk)) == dot).all()


def test_pir_row_graph_with_plaintext_arguments():
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.workloads import pir_row

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("simple_multiply")
    cols = 4
    prog = pir_row(cols)
    batch = 2
    rng = np.random.default_rng(4)

    def enc_scalar(v):
        p = np.zeros(o.n, dtype=np.uint64)
        p[0] = v % o.t
        return p

    sel = 2
    row_q = np.stack([o.encrypt(pk, enc_scalar(1)) for _ in range(batch)])
    col_q = [np.stack([o.encrypt(pk, enc_scalar(1 if j == sel else 0)) for _ in range(batch)]) for j in range(cols)]
    db_vals = rng.integers(1, 400, (cols, batch))
    db = [np.stack([enc_scalar(int(db_vals[j, i])) for i in range(batch)]) for j in range(cols)]
    inputs = [to_device(row_q)] + [to_device(c) for c in col_q] + [to_device(d) for d in db]
    (out,) = prog.run(ev, inputs, rkd)
    out = to_host(out)
    for i in range(batch):
        (ref,) = run_program(o, prog.nodes, prog.edges, [row_q[i]] + [c[i] for c in col_q] + [d[i] for d in db], rk)
        assert (out[i] == ref).all()
        assert int(o.decrypt(out[i], sk)[0]) == int(db_vals[sel, i])


def test_pir_matrix_vector_product_in_the_transform_domain():
    """examples/pir: col[i] = sum_j database[i][j] * col_query[j].  The batched primitive (one transform per query
    ciphertext, database transformed once, accumulation before the inverse transform) must give the bits of the
    reference's node-by-node multiply_plain / add sequence -- including SEAL's monomial rule for database entries that
    are a single coefficient -- and the whole lookup must decrypt to the selected entry."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.workloads import pir_lookup

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    n, t = o.n, o.t
    rows, cols = 5, 19  # 19 columns: exercises the lazy 128-bit accumulation across its 16-term reduction point
    rng = np.random.default_rng(12)
    db = rng.integers(1, t, (rows, cols, n), dtype=np.uint64)
    db[0, 0, 1:] = 0
    db[0, 0, 0] = t - 1          # monomial with an upper-half coefficient (the Signed encoding of -1)
    db[1, 2, :] = 0
    db[1, 2, 5] = 3              # monomial, small coefficient, x^5
    db[2, 3, 7:] = 0             # short plaintext
    colq = np.stack([o.encrypt(pk, o.batch_encode(rng.integers(0, 5, n).astype(np.uint64))) for _ in range(cols)])
    ctn = ev.ct_to_ntt(to_device(colq))
    dbn = ev.plain_to_ntt(to_device(db))
    got = to_host(ev.dot_plain_ntt(ctn, dbn))
    for i in range(rows):
        ref = o.multiply_plain(colq[0], db[i, 0])
        for j in range(1, cols):
            ref = o.add(ref, o.multiply_plain(colq[j], db[i, j]))
        assert (got[i] == ref).all(), i
    # the full lookup with one-hot queries (scalar encoding as in the example: value in coefficient 0)
    def scalar(v):
        p = np.zeros(n, dtype=np.uint64)
        p[0] = v % t
        return p

    vals = rng.integers(1, 1000, (rows, cols))
    dbs = np.stack([np.stack([scalar(int(vals[i, j])) for j in range(cols)]) for i in range(rows)])
    sel_r, sel_c = 3, 11
    cq = np.stack([o.encrypt(pk, scalar(1 if j == sel_c else 0)) for j in range(cols)])
    rq = np.stack([o.encrypt(pk, scalar(1 if i == sel_r else 0)) for i in range(rows)])
    out = to_host(pir_lookup(ev, to_device(cq), to_device(rq), ev.plain_to_ntt(to_device(dbs)), rkd))
    assert out.shape == (1, 2, o.K, n)
    assert int(o.decrypt(out[0], sk)[0]) == int(vals[sel_r, sel_c])
    # and bit for bit against the oracle evaluating the same expression
    acc = None
    for i in range(rows):
        col = o.multiply_plain(cq[0], dbs[i, 0])
        for j in range(1, cols):
            col = o.add(col, o.multiply_plain(cq[j], dbs[i, j]))
        term = o.relinearize(o.multiply(col, rq[i]), rk)
        acc = term if acc is None else o.add(acc, term)
    assert (out[0] == acc).all()


def test_plaintext_literal_nodes():
    """`a + b * 7`-style programs carry their constants as Literal::Plaintext(bytes) nodes
    (sunscreen/tests/fhe_program_tests.rs:283-310; bytes = bincode(InnerPlaintext) holding Params + the SEAL wire
    format, sunscreen/src/fhe/mod.rs:370-376).  Build one through the API and through JSON."""
    from sunscreen_amd import HipBfvError, Plaintext
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram, encode_plaintext_literal

    name = "default_4096_16"
    n, primes, t = params(name)
    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx(name)
    rng = np.random.default_rng(8)
    lit_vals = rng.integers(0, 50, o.n).astype(np.uint64)
    lit_coeffs = o.batch_encode(lit_vals)
    seal_bytes = Plaintext.from_coefficients([int(c) for c in lit_coeffs]).as_bytes()
    blob = encode_plaintext_literal(n, primes, t, seal_bytes)

    p = FheProgram()
    a = p.append_input_ciphertext(0)
    b = p.append_input_ciphertext(1)
    lit = p.append_plaintext_literal(blob)
    m = p.append_multiply_plaintext(b, lit)
    p.append_output_ciphertext(p.append_add(a, p.append_add_plaintext(m, lit)))
    q = FheProgram.from_json(p.to_json())  # serde_json writes the literal as an array of byte values
    assert q.nodes == p.nodes

    batch = 2
    va = rng.integers(0, 100, (batch, o.n)).astype(np.uint64)
    vb = rng.integers(0, 100, (batch, o.n)).astype(np.uint64)
    ca = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in va])
    cb = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vb])
    for prog in (p, q):
        (out,) = prog.run(ev, [to_device(ca), to_device(cb)], rkd)
        out = to_host(out)
        for i in range(batch):
            (ref,) = run_program(o, prog.nodes, prog.edges, [ca[i], cb[i]], rk, literals={lit: lit_coeffs})
            assert (out[i] == ref).all()
            assert (o.batch_decode(o.decrypt(out[i], sk)) == (va[i] + vb[i] * lit_vals + lit_vals) % o.t).all()

    # a literal built for other parameters is rejected when the program runs (the reference re-creates a context
    # from the literal's own Params, serialization.rs:62-140; here the evaluator's context is the only one)
    other = FheProgram()
    x = other.append_input_ciphertext(0)
    bad = other.append_plaintext_literal(encode_plaintext_literal(n, primes, t + 2, Plaintext.from_coefficients([1]).as_bytes()))
    other.append_output_ciphertext(other.append_add_plaintext(x, bad))
    with pytest.raises(HipBfvError):
        other.run(ev, [to_device(ca)], rkd)
    # malformed bytes are rejected when the node is added
    for junk in (b"", blob[:20], blob[:-3], b"\x01" + blob[1:]):
        with pytest.raises(HipBfvError):
            FheProgram().append_plaintext_literal(junk)


def test_program_errors():
    from sunscreen_amd import HipBfvError
    from sunscreen_amd.batch import to_device
    from sunscreen_amd.program import FheProgram

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    p = FheProgram()
    a = p.append_input_ciphertext(0)
    p.append_output_ciphertext(p.append_rotate_left(a, p.append_input_literal(1)))
    ct = to_device(np.stack([o.encrypt(pk, np.zeros(1, dtype=np.uint64))]))
    with pytest.raises(HipBfvError):  # MissingGaloisKeys (run.rs:188-190)
        p.run(ev, [ct], rkd, None)
    with pytest.raises(HipBfvError):
        FheProgram.from_json('{"graph": {"nodes": [{"operation": "Frobnicate"}], "edges": []}}')
    cyc = FheProgram()
    x = cyc.append_input_ciphertext(0)
    n1 = cyc._node("Negate")
    n2 = cyc._node("Negate")
    cyc._edge(n2, n1, "Unary")
    cyc._edge(n1, n2, "Unary")
    cyc.append_output_ciphertext(n2)
    with pytest.raises(HipBfvError):
        cyc.run(ev, [ct], rkd)


def test_transparent_results_fail_the_run_like_the_reference():
    """sunscreen/tests/features.rs:8-34: without the `transparent-ciphertexts` feature `runtime.run` of `a * 0` is an
    error (SEAL is built with SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT, seal_fhe/build.rs:46-66; run.rs:78-82 collapses it
    into a SealError).  The batch executor watches every node's results on the device and fails the run, naming the
    first offending input set; the asynchronous hipbfv_batch_* operations record the same thing for BatchEvaluator.check."""
    from sunscreen_amd import HipBfvError, Plaintext, _lib
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram, encode_plaintext_literal

    name = "default_4096_16"
    n, primes, t = params(name)
    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx(name)
    rng = np.random.default_rng(21)
    batch = 5
    vals = rng.integers(0, 50, (batch, o.n)).astype(np.uint64)
    ca = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals])
    da = to_device(ca)

    # a * 0 with the zero as a plaintext literal, exactly the features.rs program
    zero = encode_plaintext_literal(n, primes, t, Plaintext.from_coefficients([0]).as_bytes())
    p = FheProgram()
    a = p.append_input_ciphertext(0)
    p.append_output_ciphertext(p.append_multiply_plaintext(a, p.append_plaintext_literal(zero)))
    with pytest.raises(HipBfvError, match="transparent") as ei:
        p.run(ev, [da], rkd)
    assert ei.value.hresult & 0xFFFFFFFF == 0x80131509  # COR_E_INVALIDOPERATION (seal_fhe/src/lib.rs:28-34)

    # x - x in the middle of a graph whose OUTPUT is not transparent: the reference fails at the Sub node
    q = FheProgram()
    x = q.append_input_ciphertext(0)
    y = q.append_input_ciphertext(1)
    q.append_output_ciphertext(q.append_add(q.append_sub(x, x), y))
    with pytest.raises(HipBfvError, match="transparent"):
        q.run(ev, [da, da], rkd)
    with pytest.raises(RuntimeError, match="transparent"):
        run_program(o, q.nodes, q.edges, [ca[0], ca[0]], rk)

    # only ONE input set of the batch is degenerate (per-item plaintexts, item 3 is zero): the error names it
    r = FheProgram()
    x = r.append_input_ciphertext(0)
    r.append_output_ciphertext(r.append_multiply_plaintext(x, r.append_input_plaintext(1)))
    plains = rng.integers(1, t, (batch, o.n)).astype(np.uint64)
    (ok,) = r.run(ev, [da, to_device(plains)], rkd)
    assert (to_host(ok)[1] == o.multiply_plain(ca[1], plains[1])).all()
    plains[3] = 0
    with pytest.raises(HipBfvError, match="input set 3"):
        r.run(ev, [da, to_device(plains)], rkd)

    # the asynchronous batch operations: nothing is raised by the operation itself, check() reports and resets
    ev.check()
    ev.sub(da, da)
    ev.add(da, da)
    with pytest.raises(HipBfvError, match="transparent"):
        ev.check()
    ev.check()  # reset by the previous read
    ev.multiply_relin(da, da, rkd)
    ev.check()
    # the crate's `transparent-ciphertexts` feature = SEAL built without the throw: nothing is watched
    ev.set_transparent_check(False)
    ev.sub(da, da)
    ev.check()
    ev.set_transparent_check(True)
    assert _lib.load().hipbfv_set_throw_on_transparent(False) == 0
    try:
        (outz,) = q.run(ev, [da, da], rkd)
        assert (to_host(outz) == ca).all()  # (x - x) + y == y
    finally:
        assert _lib.load().hipbfv_set_throw_on_transparent(True) == 0


def _scalar(o, v):
    p = np.zeros(o.n, dtype=np.uint64)
    p[0] = v % o.t
    return p


def test_whole_pir_lookup_graph_through_the_scheduled_executor():
    """examples/pir/src/main.rs:16-45, the WHOLE `lookup` program as the compiler emits it (one input set, every database entry
    a plaintext argument), through hipbfv_Program_Run: the schedule is one transform-domain matrix product, one batched
    multiply+relinearize and one sum.  Database as coefficient-form plaintext arguments and as arguments transformed once
    (TransformedPlaintext): both equal the oracle's node-by-node evaluation bit for bit, and the hand-written pir_lookup."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram, TransformedPlaintext
    from sunscreen_amd.workloads import pir_lookup, pir_lookup_graph

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    rows, cols = 5, 19
    prog = FheProgram.from_json(pir_lookup_graph(rows, cols).to_json())
    assert prog.describe()[:3] == [f"plain_matrix members={rows} columns={cols}", f"mul_relin members={rows}", f"sum members=1 terms={rows} direct_outputs=1"]
    rng = np.random.default_rng(33)
    vals = rng.integers(1, 1000, (rows, cols))
    db = np.stack([np.stack([_scalar(o, int(vals[i, j])) for j in range(cols)]) for i in range(rows)])
    db[1, 2] = rng.integers(1, o.t, o.n, dtype=np.uint64)  # a general plaintext among the monomials (breaks the one-hot decode of row 1 only)
    sel_r, sel_c = 3, 11
    cq = np.stack([o.encrypt(pk, _scalar(o, 1 if j == sel_c else 0)) for j in range(cols)])
    rq = np.stack([o.encrypt(pk, _scalar(o, 1 if i == sel_r else 0)) for i in range(rows)])
    host_args = [cq[j] for j in range(cols)] + [rq[i] for i in range(rows)] + [db[i, j] for i in range(rows) for j in range(cols)]
    (ref,) = run_program(o, prog.nodes, prog.edges, host_args, rk)
    assert int(o.decrypt(ref, sk)[0]) == int(vals[sel_r, sel_c])
    dcq, drq, ddb = to_device(cq), to_device(rq), to_device(db)
    cts = [dcq[j : j + 1] for j in range(cols)] + [drq[i : i + 1] for i in range(rows)]
    (out1,) = prog.run(ev, cts + [ddb[i, j] for i in range(rows) for j in range(cols)], rkd)
    assert (to_host(out1)[0] == ref).all()
    dbn = ev.plain_to_ntt(ddb)
    run2 = prog.prepare(ev, cts + [TransformedPlaintext(dbn[i, j]) for i in range(rows) for j in range(cols)], rkd)
    for _ in range(2):  # bound once, run twice
        (out2,) = run2()
        assert (to_host(out2)[0] == ref).all()
    assert (to_host(pir_lookup(ev, dcq, drq, dbn, rkd))[0] == ref).all()
    # separately allocated arguments (nothing adjacent in memory: every operand list is staged through a pointer table)
    scattered = [t.clone() for t in cts] + [TransformedPlaintext(dbn[i, j].clone()) for i in range(rows) for j in range(cols)]
    (out3,) = prog.run(ev, scattered, rkd)
    assert (to_host(out3)[0] == ref).all()


def test_pir_lookup_with_the_product_on_a_side_stream(monkeypatch):
    """HIPBFV_PIR_OVERLAP=1 (opt-in: measured slower inside the executor, program_plan.cpp): the plaintext-matrix product of a
    lookup with >= 64 rows runs on a side stream in row chunks and the merged multiply + relinearize takes each chunk behind its
    event.  Same bits as the serial schedule and as the oracle's node-by-node evaluation; one input set, 3 and 4
    chunks (64 rows: 21 + 21 + 22 and 4 x 16)."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram, TransformedPlaintext
    from sunscreen_amd.workloads import pir_lookup_graph

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    rows, cols = 64, 3
    prog = FheProgram.from_json(pir_lookup_graph(rows, cols).to_json())
    rng = np.random.default_rng(64)
    vals = rng.integers(1, 1000, (rows, cols))
    db = np.stack([np.stack([_scalar(o, int(vals[i, j])) for j in range(cols)]) for i in range(rows)])
    sel_r, sel_c = 41, 2
    cq = np.stack([o.encrypt(pk, _scalar(o, 1 if j == sel_c else 0)) for j in range(cols)])
    rq = np.stack([o.encrypt(pk, _scalar(o, 1 if i == sel_r else 0)) for i in range(rows)])
    host_args = [cq[j] for j in range(cols)] + [rq[i] for i in range(rows)] + [db[i, j] for i in range(rows) for j in range(cols)]
    (ref,) = run_program(o, prog.nodes, prog.edges, host_args, rk)
    assert int(o.decrypt(ref, sk)[0]) == int(vals[sel_r, sel_c])
    dcq, drq = to_device(cq), to_device(rq)
    dbn = ev.plain_to_ntt(to_device(db))
    args = [dcq[j : j + 1] for j in range(cols)] + [drq[i : i + 1] for i in range(rows)] + [TransformedPlaintext(dbn[i, j]) for i in range(rows) for j in range(cols)]
    (serial,) = prog.run(ev, args, rkd)
    assert (to_host(serial)[0] == ref).all()
    monkeypatch.setenv("HIPBFV_PIR_OVERLAP", "1")
    for chunks in ("3", "4"):
        monkeypatch.setenv("HIPBFV_PIR_CHUNKS", chunks)
        for _ in range(2):
            (out,) = prog.run(ev, args, rkd)
            assert (to_host(out)[0] == ref).all(), chunks


def test_zero_plaintext_inside_a_transform_domain_sum_fails_like_multiply_plain():
    """sunscreen/tests/features.rs:8-34: a * 0 is an error.  Inside a sum of products that stays in the transform domain the single
    product never exists, so the zero plaintext itself must raise the failure -- and name the input set."""
    from sunscreen_amd import HipBfvError
    from sunscreen_amd.batch import to_device
    from sunscreen_amd.workloads import pir_row

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("simple_multiply")
    cols, batch = 3, 4
    prog = pir_row(cols)
    assert prog.describe()[0] == f"plain_matrix members=1 columns={cols}"
    rng = np.random.default_rng(5)
    row_q = np.stack([o.encrypt(pk, _scalar(o, 1)) for _ in range(batch)])
    col_q = [np.stack([o.encrypt(pk, _scalar(o, j + 1)) for _ in range(batch)]) for j in range(cols)]
    db = [rng.integers(1, o.t, (batch, o.n), dtype=np.uint64) for _ in range(cols)]
    ins = lambda: [to_device(row_q)] + [to_device(c) for c in col_q] + [to_device(d) for d in db]  # noqa: E731
    prog.run(ev, ins(), rkd)
    db[1][2] = 0
    with pytest.raises(HipBfvError, match="input set 2"):
        prog.run(ev, ins(), rkd)
    with pytest.raises(RuntimeError, match="transparent"):
        run_program(o, prog.nodes, prog.edges, [row_q[2]] + [c[2] for c in col_q] + [d[2] for d in db], rk)
    # The same arguments ALREADY TRANSFORMED (TransformedPlaintext, kind 2): the executor no longer sees coefficients, so the
    # producer is where a * 0 has to fail (it used to pass silently) -- and name the plaintext.
    good = ev.plain_to_ntt(to_device(db[0]))
    assert good.shape == (batch, ev.K, o.n)
    with pytest.raises(HipBfvError, match="batch item 2"):
        ev.plain_to_ntt(to_device(db[1]))
    ev.check()  # the status word was reset by the failure: later operations start clean


def test_add_and_sub_of_different_ciphertext_sizes():
    """run.rs:217-236 hands Add / Sub operands of any sizes to SEAL, which pads the shorter one: (a*b) + a is a size-3 ciphertext,
    a - (a*b) negates the product's third polynomial.  Relinearised afterwards so that the outputs are size 2."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    p = FheProgram()
    a, b = p.append_input_ciphertext(0), p.append_input_ciphertext(1)
    m = p.append_multiply(a, b)  # two users: stays an unfused size-3 product
    p.append_output_ciphertext(p.append_relinearize(p.append_add(m, a)))
    p.append_output_ciphertext(p.append_relinearize(p.append_sub(b, p.append_negate(m))))
    p.append_output_ciphertext(p.append_relinearize(p.append_add(p.append_sub(a, m), p.append_add(m, m))))
    rng = np.random.default_rng(6)
    for batch in (1, 3, 40):
        va = rng.integers(0, 30, (batch, o.n)).astype(np.uint64)
        vb = rng.integers(0, 30, (batch, o.n)).astype(np.uint64)
        ca = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in va])
        cb = np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vb])
        outs = [to_host(t) for t in p.run(ev, [to_device(ca), to_device(cb)], rkd)]
        for i in range(min(batch, 3)):
            refs = run_program(o, p.nodes, p.edges, [ca[i], cb[i]], rk)
            for k in range(3):
                assert (outs[k][i] == refs[k]).all(), (batch, i, k)
            prod = va[i].astype(np.int64) * vb[i]
            expect = [prod + va[i], vb[i] + prod, va[i] + prod]
            for k in range(3):
                assert (o.batch_decode(o.decrypt(outs[k][i], sk)) == expect[k] % o.t).all()


@pytest.mark.parametrize("batch", [1, 2, 40])
def test_scheduled_and_node_by_node_executors_agree(batch, monkeypatch):
    """The reference's examples through both executors -- ready nodes merged into batched launches (batch 1 and 2: the
    reference's own call shape is ONE input set) or every member on its own with Adds folded into key-switch tails (batch 40)
    against one node at a time (HIPBFV_PROGRAM_SERIAL=1) -- word for word, and item 0 against the oracle interpreter."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.workloads import chi_sq_optimized, dot_product

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16", galois="all")
    rng = np.random.default_rng(40 + batch)
    for prog, nin in ((chi_sq_optimized(), 3), (dot_product(o.n // 2), 2)):
        vals = rng.integers(0, 7, (nin, batch, o.n)).astype(np.uint64)
        cts = [np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals[a]]) for a in range(nin)]
        dev = [to_device(c) for c in cts]
        monkeypatch.delenv("HIPBFV_PROGRAM_SERIAL", raising=False)
        got = [to_host(t) for t in prog.run(ev, dev, rkd, gkd)]
        monkeypatch.setenv("HIPBFV_PROGRAM_SERIAL", "1")
        serial = [to_host(t) for t in prog.run(ev, dev, rkd, gkd)]
        monkeypatch.delenv("HIPBFV_PROGRAM_SERIAL", raising=False)
        ref = run_program(o, prog.nodes, prog.edges, [c[0] for c in cts], rk, gk)
        for k in range(len(got)):
            assert (got[k] == serial[k]).all(), (batch, k)
            assert (got[k][0] == ref[k]).all(), (batch, k)


def test_scheduled_executor_corner_cases_against_the_oracle_and_the_node_by_node_executor(monkeypatch):
    """Shapes the examples do not have: sums of plaintext products over DIFFERENT ciphertext lists (two matrix groups) with
    per-item, shared, literal and pre-transformed plaintexts mixed in one sum; two rotations by the same step (one merged
    launch at small batch) next to a different step and a NAF chain; Relinearize of a size-2 value (a copy) and of a product
    with two users (unfused); a node that is output twice; an input that is output unchanged; a value nobody uses."""
    from sunscreen_amd import Plaintext
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram, TransformedPlaintext, encode_plaintext_literal

    name = "default_4096_16"
    n, primes, t = params(name)
    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx(name, galois=[3, 9, 81, 2 * 4096 - 1])  # steps 1, 2, 4 and the column swap
    rng = np.random.default_rng(77)
    lit_coeffs = o.batch_encode(rng.integers(1, 50, n).astype(np.uint64))
    blob = encode_plaintext_literal(n, primes, t, Plaintext.from_coefficients([int(c) for c in lit_coeffs]).as_bytes())

    p = FheProgram()
    a, b, c = (p.append_input_ciphertext(i) for i in range(3))
    p_item, p_shared, p_ntt = p.append_input_plaintext(3), p.append_input_plaintext(4), p.append_input_plaintext(5)
    lit = p.append_plaintext_literal(blob)
    # group 1: sums over (a, b, c); group 2: a sum over (b, a)
    s1 = p.append_add(p.append_add(p.append_multiply_plaintext(a, p_item), p.append_multiply_plaintext(b, p_shared)), p.append_multiply_plaintext(c, lit))
    s2 = p.append_add(p.append_add(p.append_multiply_plaintext(a, p_ntt), p.append_multiply_plaintext(b, lit)), p.append_multiply_plaintext(c, p_item))
    s3 = p.append_add(p.append_multiply_plaintext(b, p_shared), p.append_multiply_plaintext(a, p_item))
    r1 = p.append_rotate_left(s1, p.append_input_literal(1))
    r2 = p.append_rotate_left(s2, p.append_input_literal(1))      # same step: one merged launch at small batch
    r3 = p.append_rotate_left(s3, p.append_input_literal(5))      # no key for step 5: the NAF chain 1 + 4 (rotate_internal)
    r4 = p.append_swap_rows(r1)
    m = p.append_multiply(r2, r3)                                 # two users: stays an unfused product
    rl = p.append_relinearize(m)
    big = p.append_relinearize(p.append_add(m, m))                # relinearise the size-3 sum
    cp = p.append_relinearize(a)                                  # size 2: a copy
    p.append_negate(c)                                            # a value nobody uses
    for node in (rl, big, cp, r4, r4, a, p.append_sub(rl, r4)):
        p.append_output_ciphertext(node)
    q = FheProgram.from_json(p.to_json())
    for batch in (1, 3, 40):
        cts = [np.stack([o.encrypt(pk, o.batch_encode(rng.integers(0, 20, n).astype(np.uint64))) for _ in range(batch)]) for _ in range(3)]
        per_item = rng.integers(1, t, (batch, n), dtype=np.uint64)
        shared = rng.integers(1, t, n, dtype=np.uint64)
        for_ntt = rng.integers(1, t, n, dtype=np.uint64)
        dev = [to_device(x) for x in cts] + [to_device(per_item), to_device(shared), TransformedPlaintext(ev.plain_to_ntt(to_device(for_ntt)))]
        monkeypatch.delenv("HIPBFV_PROGRAM_SERIAL", raising=False)
        got = [to_host(x) for x in q.run(ev, dev, rkd, gkd)]
        assert len(got) == 7
        for i in range(min(batch, 2)):
            ref = run_program(o, q.nodes, q.edges, [x[i] for x in cts] + [per_item[i], shared, for_ntt], rk, gk, literals={lit: lit_coeffs})
            for k in range(7):
                assert (got[k][i] == ref[k]).all(), (batch, i, k, q.describe())
        # the node-by-node executor takes coefficient-form plaintexts only: same program, the pre-transformed argument as coefficients
        monkeypatch.setenv("HIPBFV_PROGRAM_SERIAL", "1")
        serial = [to_host(x) for x in q.run(ev, dev[:5] + [to_device(for_ntt)], rkd, gkd)]
        monkeypatch.delenv("HIPBFV_PROGRAM_SERIAL", raising=False)
        for k in range(7):
            assert (serial[k] == got[k]).all(), (batch, k)


def test_one_program_object_run_from_several_threads():
    """sunscreen_runtime shares one compiled program between requests; the schedule is built once (under a lock) and every
    run keeps its tables in its own thread's arena."""
    import threading

    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.workloads import chi_sq_optimized

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    prog = chi_sq_optimized()
    rng = np.random.default_rng(5)
    sets = []
    for _ in range(4):
        vals = rng.integers(0, 7, (3, 2, o.n)).astype(np.uint64)
        sets.append([to_device(np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals[a]])) for a in range(3)])
    want = [[to_host(x) for x in prog.run(ev, s, rkd)] for s in sets]
    errors = []

    def work(i):
        try:
            import torch

            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(5):
                    got = [to_host(x) for x in prog.run(ev, sets[i], rkd)]
                    for k in range(4):
                        assert (got[k] == want[i][k]).all(), (i, k)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [x.start() for x in ths]
    [x.join() for x in ths]
    assert not errors, errors


@pytest.mark.parametrize("name", ["default_4096_16", "default_16384_17"])
def test_sums_around_products_ride_in_the_products_last_kernel(name, monkeypatch):
    """r06, Plan::LinFold / kernels.hpp MemberTail: mult * (x * y) +- z written by the fused multiply + relinearize's last kernel --
    the reference adds node by node (sunscreen_runtime/src/run.rs:130-176: Add / Sub each a SEAL call), modular sums give the same
    bits in any grouping.  Every fold shape (multipliers 2 ... 4, a subtracted addend, an addend made by ANOTHER launch of the same
    round, a product that is itself a program output beside a folded one, no addend) and the shapes that must NOT fold (five copies,
    a negated product), at batch 2 and batch 40 -- the ready products of a round are ONE launch over members x batch items, operands read
    and results written where they are (MemberHead / MemberTail tables) -- against the oracle interpreter, the node-by-node executor, the
    scheduled one with the folds switched off, and the one with a launch per product (HIPBFV_NO_MERGED_PRODUCTS=1)."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx(name)

    def mul(p, a, b):
        return p.append_relinearize(p.append_multiply(a, b))

    p = FheProgram()
    x, y, z = (p.append_input_ciphertext(i) for i in range(3))
    xy = mul(p, x, y)
    o1 = p.append_sub(p.append_add(p.append_add(xy, xy), xy), z)            # 3 x y - z
    zz = mul(p, z, z)
    o2 = p.append_add(zz, zz)                                              # 2 z^2 (a square; no addend)
    yz = mul(p, y, z)
    t = p.append_add(yz, yz)
    xx = mul(p, x, x)
    o3 = p.append_sub(p.append_add(t, t), xx)                              # 4 y z - x^2: the addend is made by the squares' launch
    o4 = mul(p, x, z)                                                      # a plain product beside the folded ones (merged: its own buffer)
    xz2 = mul(p, z, x)
    t2 = p.append_add(xz2, xz2)
    o5 = p.append_add(p.append_add(t2, t2), xz2)                           # five copies: an ordinary sum
    o6 = p.append_sub(x, mul(p, y, y))                                     # x - y^2: the product is negated -- no fold
    o7 = p.append_add(mul(p, y, x), x)                                     # x y + x (the old two-term fold's shape)
    for node in (o1, o2, o3, o4, o5, o6, o7, xx):
        p.append_output_ciphertext(node)
    desc = p.describe()
    assert sum(int(l.split("lin_foldable=")[1].split()[0]) for l in desc if "lin_foldable" in l) == 4, desc
    rng = np.random.default_rng(9)
    for batch in (2, 40):
        vals = rng.integers(0, 5, (3, batch, o.n)).astype(np.uint64)
        cts = [np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals[a]]) for a in range(3)]
        if batch > 2:  # every residue of one operand at its maximum / at zero: the epilogue's range (mult * (q - 1) + (q - 1))
            cts[2][1, :, :, :] = (np.array(o.primes[: o.K], dtype=np.uint64) - 1)[None, :, None]
        dev = [to_device(c) for c in cts]
        monkeypatch.delenv("HIPBFV_PROGRAM_SERIAL", raising=False)
        monkeypatch.delenv("HIPBFV_NO_MEMBER_TAILS", raising=False)
        got = [to_host(t_) for t_ in p.run(ev, dev, rkd, gkd)]
        monkeypatch.setenv("HIPBFV_NO_MEMBER_TAILS", "1")
        plain = [to_host(t_) for t_ in p.run(ev, dev, rkd, gkd)]
        monkeypatch.delenv("HIPBFV_NO_MEMBER_TAILS", raising=False)
        monkeypatch.setenv("HIPBFV_NO_MERGED_PRODUCTS", "1")  # batch 40: every product a launch of its own, its sum in its last kernel
        apart = [to_host(t_) for t_ in p.run(ev, dev, rkd, gkd)]
        monkeypatch.delenv("HIPBFV_NO_MERGED_PRODUCTS", raising=False)
        for k in range(len(got)):
            assert (got[k] == apart[k]).all(), (name, batch, k)
        monkeypatch.setenv("HIPBFV_PROGRAM_SERIAL", "1")
        serial = [to_host(t_) for t_ in p.run(ev, dev, rkd, gkd)]
        monkeypatch.delenv("HIPBFV_PROGRAM_SERIAL", raising=False)
        for k in range(len(got)):
            assert (got[k] == plain[k]).all() and (got[k] == serial[k]).all(), (name, batch, k)
        for i in (0, 1):
            ref = run_program(o, p.nodes, p.edges, [c[i] for c in cts], rk, gk)
            for k in range(len(got)):
                assert (got[k][i] == ref[k]).all(), (name, batch, i, k)
        a, b, c = (vals[j][0].astype(np.int64) for j in range(3))
        assert (o.batch_decode(o.decrypt(got[0][0], sk)) == (3 * a * b - c) % o.t).all()
        assert (o.batch_decode(o.decrypt(got[2][0], sk)) == (4 * b * c - a * a) % o.t).all()


def test_a_dozen_ready_products_in_one_launch_sequence(monkeypatch):
    """A round of twelve products (a dot product written out term by term, the shape a compiler emits without SIMD packing): one launch
    sequence over 12 x batch items through the operand / destination tables, the pairwise sums behind them folded where the pattern
    allows (x_i y_i + x_j y_j: one product folds, reading the other's result from ANOTHER group only -- here all are in one group, so
    none folds and the sum runs as one n-ary launch).  Against the oracle interpreter and the launch-per-product executor."""
    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.program import FheProgram

    o, sk, pk, rk, gk, ev, rkd, gkd = _ctx("default_4096_16")
    p = FheProgram()
    xs = [p.append_input_ciphertext(i) for i in range(12)]
    ys = [p.append_input_ciphertext(12 + i) for i in range(12)]
    prods = [p.append_relinearize(p.append_multiply(a, b)) for a, b in zip(xs, ys)]
    acc = prods[0]
    for t in prods[1:]:
        acc = p.append_add(acc, t)
    p.append_output_ciphertext(acc)
    p.append_output_ciphertext(p.append_add(prods[3], prods[3]))  # (prods[3] has two users: it exists on its own)
    desc = p.describe()
    assert desc[0] == "mul_relin members=12", desc
    rng = np.random.default_rng(12)
    for batch in (3, 40):
        vals = rng.integers(0, 9, (24, batch, o.n)).astype(np.uint64)
        cts = [np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals[a]]) for a in range(24)]
        dev = [to_device(c) for c in cts]
        monkeypatch.delenv("HIPBFV_NO_MERGED_PRODUCTS", raising=False)
        got = [to_host(t) for t in p.run(ev, dev, rkd, gkd)]
        monkeypatch.setenv("HIPBFV_NO_MERGED_PRODUCTS", "1")
        apart = [to_host(t) for t in p.run(ev, dev, rkd, gkd)]
        monkeypatch.delenv("HIPBFV_NO_MERGED_PRODUCTS", raising=False)
        for k in range(2):
            assert (got[k] == apart[k]).all(), (batch, k)
        ref = run_program(o, p.nodes, p.edges, [c[0] for c in cts], rk, gk)
        for k in range(2):
            assert (got[k][0] == ref[k]).all(), (batch, k)
        dot = (vals[:12, 0].astype(np.int64) * vals[12:, 0].astype(np.int64)).sum(axis=0)
        assert (o.batch_decode(o.decrypt(got[0][0], sk)) == dot % o.t).all()
