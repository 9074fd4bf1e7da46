"""The graph executor's schedule (csrc/program_plan.cpp) is host logic: these tests read it through hipbfv_Program_Describe,
no GPU needed.  The reference's `traverse` (sunscreen_runtime/src/run.rs:372-472) runs every ready node concurrently; the
schedule turns "concurrently" into "one launch" -- the shapes below are what bench.py's program workloads rely on."""
import pytest

from sunscreen_amd.program import FheProgram
from sunscreen_amd.workloads import chi_sq_optimized, dot_product, pir_lookup_graph, pir_row


def _kinds(lines):
    return [l.split()[0] for l in lines]


def test_chi_sq_levels_six_products_in_three_launch_groups():
    """examples/chi_sq/src/main.rs:59-88: 6 mul+relin of which 3 are squares.  After the additions x = 2 n0 + n1, y = 2 n2 + n1
    five products are ready at once: squares {n1^2, x^2, y^2} and general {n0 n2, x y}; alpha^2 follows.  Node by node that
    is six launch sequences, scheduled it is three."""
    lines = chi_sq_optimized().describe()
    mr = [l for l in lines if l.startswith("mul_relin")]
    # r06: the sums AROUND the products ride in the products' last kernels (Plan::LinFold): 2 x^2 and 2 y^2 (program outputs), and
    # 4 n0 n2 - n1^2 -- whose addend n1^2 is made by the squares' launch, so that group runs first
    assert mr == ["mul_relin members=3 square lin_foldable=2", "mul_relin members=2 lin_foldable=1 direct_outputs=1",
                  "mul_relin members=1 square direct_outputs=1"], lines
    # the Add / Sub chains are n-ary sums: x = n0 + n0 + n1 and y = n2 + n2 + n1 in one launch before any product ...
    sums = [l for l in lines if l.startswith("sum")]
    assert lines[0] == "sum members=2 terms=6", lines
    # ... and ONE launch after the five products for whatever was not folded at run time: 2 x^2, 2 y^2 and p + p + p + p - n1^2 (the
    # doubled n0 n2 is used twice by one Add only: its terms are taken twice instead of materialising it)
    assert sums[1:] == ["sum members=3 terms=9 direct_outputs=2"], lines
    assert _kinds(lines).count("output") == 4


def test_linear_folds_of_products():
    """Plan::LinFold: mult copies (<= 4) of a fused product plus at most one other ciphertext, the product used by nothing else."""
    def prog(build):
        p = FheProgram()
        x, y, z = (p.append_input_ciphertext(i) for i in range(3))
        p.append_output_ciphertext(build(p, x, y, z))
        return p.describe()

    def mul(p, a, b):
        return p.append_relinearize(p.append_multiply(a, b))

    # 3 x y - z: foldable (the addend is a program input)
    lines = prog(lambda p, x, y, z: p.append_sub(p.append_add(p.append_add(mul(p, x, y), mul(p, x, y)), mul(p, x, y)), z))
    assert any("lin_foldable" in l for l in lines) is False, lines  # three DIFFERENT product nodes: three members, one sum
    def tripled(p, x, y, z):
        m = mul(p, x, y)
        return p.append_sub(p.append_add(p.append_add(m, m), m), z)
    lines = prog(tripled)
    assert lines[0] == "mul_relin members=1 lin_foldable=1", lines
    # five copies: beyond the kernel's multiplier range -- an ordinary sum
    def five(p, x, y, z):
        m = mul(p, x, y)
        t = p.append_add(m, m)
        return p.append_add(p.append_add(t, t), m)
    assert not any("lin_foldable" in l for l in prog(five))
    # the product has another user: it must exist on its own
    def shared(p, x, y, z):
        m = mul(p, x, y)
        p.append_output_ciphertext(m)
        return p.append_add(m, m)
    assert not any("lin_foldable" in l for l in prog(shared))
    # a negated product, or two other terms: not this pattern
    assert not any("lin_foldable" in l for l in prog(lambda p, x, y, z: p.append_sub(z, mul(p, x, y))))
    assert not any("lin_foldable" in l for l in prog(lambda p, x, y, z: p.append_add(p.append_add(mul(p, x, y), z), x)))
    # two products into one sum, a square and a general one (two launch groups): one folds, reading the other -- which runs first
    def two(p, x, y, z):
        return p.append_add(mul(p, x, y), mul(p, z, z))
    lines = prog(two)
    assert [l for l in lines if l.startswith("mul_relin")] == ["mul_relin members=1 square", "mul_relin members=1 lin_foldable=1"], lines
    # ... both in ONE group: the addend would be made by the same launch
    lines = prog(lambda p, x, y, z: p.append_add(mul(p, x, y), mul(p, x, z)))
    assert not any("lin_foldable" in l for l in lines), lines


def test_dot_product_rotations_keep_their_foldable_adds():
    lines = dot_product(8).describe()
    rot = [l for l in lines if l.startswith("rotate")]
    assert len(rot) == 4 and all("add_foldable=1" in l for l in rot), lines  # 1, 2, 4 and swap_rows: c = c + rotate(c)
    assert rot[-1].startswith("rotate members=1 swap_rows")


def test_pir_lookup_is_one_matrix_product_one_batched_multiply_and_one_sum():
    """examples/pir/src/main.rs:16-45 for a 6 x 5 database: 30 MultiplyPlaintext, 24 + 5 Add, 6 Multiply + Relinearize."""
    rows, cols = 6, 5
    p = pir_lookup_graph(rows, cols)
    assert len(p.nodes) == cols + rows + 2 * rows * cols + (cols - 1) * rows + 2 * rows + (rows - 1) + 1
    lines = p.describe()
    assert lines == [f"plain_matrix members={rows} columns={cols}", f"mul_relin members={rows}", f"sum members=1 terms={rows} direct_outputs=1", "output members=1"], lines


def test_pir_row_graph_is_one_matrix_row():
    lines = pir_row(4).describe()
    assert lines[0] == "plain_matrix members=1 columns=4" and lines[1].startswith("mul_relin members=1"), lines


def test_identically_cancelling_sums_stay_visible():
    """(x - x) + y: SEAL throws on the transparent intermediate (seal_fhe/build.rs:46-66); folding it into the outer sum would
    hide that, so it stays a result of its own."""
    p = FheProgram()
    x, y = p.append_input_ciphertext(0), p.append_input_ciphertext(1)
    z = p.append_add(p.append_sub(x, x), y)
    p.append_output_ciphertext(z)
    lines = p.describe()
    assert [l for l in lines if l.startswith("sum")] == ["sum members=1 terms=2", "sum members=1 terms=2 direct_outputs=1"], lines
    # whereas (x - y) + y folds: nothing cancels identically inside
    q = FheProgram()
    x, y = q.append_input_ciphertext(0), q.append_input_ciphertext(1)
    q.append_output_ciphertext(q.append_add(q.append_sub(x, y), y))
    assert [l for l in q.describe() if l.startswith("sum")] == ["sum members=1 terms=3 direct_outputs=1"]


def test_mixed_products_do_not_become_a_plain_matrix():
    """A sum with one term that is not a plaintext product is an ordinary sum over materialised products."""
    p = FheProgram()
    a, b = p.append_input_ciphertext(0), p.append_input_ciphertext(1)
    t0 = p.append_multiply_plaintext(a, p.append_input_plaintext(2))
    t1 = p.append_multiply_plaintext(b, p.append_input_plaintext(3))
    p.append_output_ciphertext(p.append_add(p.append_add(t0, t1), a))
    lines = p.describe()
    assert _kinds(lines) == ["plain_op", "plain_op", "sum", "output"], lines


def test_static_errors_surface_before_anything_runs():
    from sunscreen_amd.seal import HipBfvError

    p = FheProgram()
    a = p.append_input_ciphertext(0)
    pl = p.append_input_plaintext(1)
    p.append_output_ciphertext(p.append_multiply(a, pl))
    with pytest.raises(HipBfvError, match="right operand is not a ciphertext"):
        p.describe()
    q = FheProgram()
    a = q.append_input_ciphertext(0)
    q.append_output_ciphertext(q.append_multiply(a, a))  # a size-3 output
    with pytest.raises(HipBfvError, match="size-2"):
        q.describe()


def test_transparent_intermediates_are_recognised_by_value_not_by_node():
    """SEAL refuses a TRANSPARENT result (polynomials 1.. all zero), which need not be zero and need not come from one node used
    twice: (x*y - x*y) + z has two product nodes; (x - (x + p)) + z has c0 = -delta*p and c1 = 0.  Neither difference may be
    folded into the outer sum."""
    for build in ("dup_products", "plain_offset"):
        p = FheProgram()
        x, y, z = (p.append_input_ciphertext(i) for i in range(3))
        if build == "dup_products":
            m1 = p.append_relinearize(p.append_multiply(x, y))
            m2 = p.append_relinearize(p.append_multiply(y, x))  # Multiply is commutative: the same value
            d = p.append_sub(m1, m2)
        else:
            d = p.append_sub(x, p.append_add_plaintext(x, p.append_input_plaintext(3)))
        p.append_output_ciphertext(p.append_add(d, z))
        sums = [l for l in p.describe() if l.startswith("sum")]
        assert sums == ["sum members=1 terms=2", "sum members=1 terms=2 direct_outputs=1"], (build, p.describe())
