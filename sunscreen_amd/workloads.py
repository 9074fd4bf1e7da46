"""Program graphs of the reference's BFV examples used by BASELINE.json configs 4 and 5.

Each builder returns an `FheProgram` with the node structure the Sunscreen compiler emits for the
example (a Relinearize after every Multiply: sunscreen_backend/src/transforms/insert_relinearizations.rs:45-60).
"""
from __future__ import annotations

from .program import FheProgram


def _mul(p: FheProgram, a: int, b: int) -> int:
    return p.append_relinearize(p.append_multiply(a, b))


def chi_sq_optimized() -> FheProgram:
    """examples/chi_sq/src/main.rs:59-88 `chi_sq_optimized_impl`: 3 inputs -> (alpha, b1, b2, b3);
    6 mul+relin, 8 add, 1 sub."""
    p = FheProgram()
    n0, n1, n2 = (p.append_input_ciphertext(i) for i in range(3))
    x = p.append_add(p.append_add(n0, n0), n1)
    y = p.append_add(p.append_add(n2, n2), n1)
    n02 = _mul(p, n0, n2)
    n02 = p.append_add(n02, n02)
    n02 = p.append_add(n02, n02)
    n1sq = _mul(p, n1, n1)
    alpha = p.append_sub(n02, n1sq)
    alpha = _mul(p, alpha, alpha)
    b1 = _mul(p, x, x)
    b1 = p.append_add(b1, b1)
    b2 = _mul(p, x, y)
    b3 = _mul(p, y, y)
    b3 = p.append_add(b3, b3)
    for out in (alpha, b1, b2, b3):
        p.append_output_ciphertext(out)
    return p


def dot_product(lanes: int) -> FheProgram:
    """examples/dot_prod/src/main.rs:38-75: c = a*b; log2(lanes) rotate-and-add steps; + swap_rows."""
    p = FheProgram()
    a, b = p.append_input_ciphertext(0), p.append_input_ciphertext(1)
    c = _mul(p, a, b)
    shift = 1
    while shift < lanes:
        c = p.append_add(c, p.append_rotate_left(c, p.append_input_literal(shift)))
        shift *= 2
    c = p.append_add(c, p.append_swap_rows(c))
    p.append_output_ciphertext(c)
    return p


def pir_row(columns: int) -> FheProgram:
    """One row of examples/pir/src/main.rs:16-45: sum_j (col_query_j * db_j) where db_j are plaintext
    arguments, then multiplied by the row query.  Inputs: 0 = row query (ct), 1..columns = column
    queries (ct), columns+1..2*columns = database plaintexts."""
    p = FheProgram()
    row = p.append_input_ciphertext(0)
    acc = None
    for j in range(columns):
        cq = p.append_input_ciphertext(1 + j)
        db = p.append_input_plaintext(1 + columns + j)
        term = p.append_multiply_plaintext(cq, db)
        acc = term if acc is None else p.append_add(acc, term)
    p.append_output_ciphertext(_mul(p, acc, row))
    return p


def pir_lookup_graph(rows: int, columns: int) -> FheProgram:
    """The WHOLE `lookup` program of examples/pir/src/main.rs:16-45 for a rows x columns database, node for node as the
    compiler emits it: col[i] = db[i][0] * col_query[0]; col[i] = col[i] + db[i][j] * col_query[j]; sum = col[0] * row_query[0];
    sum = sum + col[i] * row_query[i] (a Relinearize after every Multiply).  Arguments, in the order of the fhe_program's
    signature: 0..columns-1 = col_query, columns..columns+rows-1 = row_query, then the database row-major
    (argument columns + rows + i * columns + j = database[i][j]).  One output."""
    p = FheProgram()
    cq = [p.append_input_ciphertext(j) for j in range(columns)]
    rq = [p.append_input_ciphertext(columns + i) for i in range(rows)]
    total = None
    for i in range(rows):
        col = None
        for j in range(columns):
            term = p.append_multiply_plaintext(cq[j], p.append_input_plaintext(columns + rows + i * columns + j))
            col = term if col is None else p.append_add(col, term)
        prod = _mul(p, col, rq[i])
        total = prod if total is None else p.append_add(total, prod)
    p.append_output_ciphertext(total)
    return p


def pir_lookup(ev, col_query, row_query, db_ntt, relin_keys):
    """examples/pir/src/main.rs:16-45 `lookup` for a sqrt(DB) x sqrt(DB) database, on the batch primitives instead of a
    node-by-node graph: col[i] = sum_j database[i][j] * col_query[j] as ONE transform-domain matrix-vector product
    (the column queries are transformed once, the pre-transformed database streams through once), then
    sum_i col[i] * row_query[i] as one batched multiply+relinearize and a pairwise reduction.

    col_query: int64[cols, 2, K, N], row_query: int64[rows, 2, K, N] (device), db_ntt: int64[rows, cols, K, N] from
    BatchEvaluator.plain_to_ntt.  Returns int64[1, 2, K, N].  Bit-identical to the reference's evaluation order
    except for the final summation order, which modular addition does not see."""
    col = ev.dot_plain_ntt(ev.ct_to_ntt(col_query), db_ntt)
    prod = ev.multiply_relin(col, row_query, relin_keys)
    while prod.shape[0] > 1:
        half = prod.shape[0] // 2
        head = ev.add(prod[:half].contiguous(), prod[half : 2 * half].contiguous())
        prod = head if prod.shape[0] == 2 * half else __import__("torch").cat([head, prod[2 * half :]])
    return prod


def pir_lookup_sharded(ev, col_query, row_query_local, db_ntt_local, relin_keys, root: int = 0):
    """examples/pir with the database sharded by ROW over the ranks (SURVEY 8d config 5a, 8e "Exception"): this rank holds
    rows [lo, hi) of the transform-domain database and the matching slice of the row query; `pir_lookup` over them leaves one
    partial ciphertext per GPU, and `dist.reduce_ciphertexts` sums them on `root` (gather + add: the only data-path exchange
    in scope).  Returns int64[1, 2, K, N] on `root`, None elsewhere.  Modular addition is exact and order-free, so the answer
    has the bits of the unsharded lookup."""
    from . import dist as D

    return D.reduce_ciphertexts(pir_lookup(ev, col_query, row_query_local, db_ntt_local, relin_keys), ev.add, root)
