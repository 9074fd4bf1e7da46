"""Oracle-side interpreter for program graphs: the checker for the GPU batch executor (test infrastructure, like
the rest of oracle/: only tests/, smoke() and bench.py's parity gate / cpu_baseline leg may import it).

Walks the node list exactly like the reference's `run_program_unchecked` (sunscreen_runtime/src/run.rs:160-341),
one oracle call per node, for ONE input set."""
import numpy as np


def run_program(o, nodes, edges, inputs, rk=None, gk=None, literals=None):
    """literals: node index -> plaintext coefficient array, for Literal::Plaintext nodes (the checker is told the
    coefficients directly; decoding the bincode/SEAL bytes is the product's job)."""
    n = len(nodes)
    left, right = [None] * n, [None] * n
    for s, d, kind in edges:
        if kind == "Right":
            right[d] = s
        else:
            left[d] = s
    val = [None] * n
    outs = []
    for i, (op, arg) in enumerate(nodes):  # builders append operands before users: index order is topological
        L = val[left[i]] if left[i] is not None else None
        R = val[right[i]] if right[i] is not None else None
        if op in ("InputCiphertext", "InputPlaintext"):
            val[i] = inputs[arg]
        elif op == "Literal":
            val[i] = literals[i] if isinstance(arg, dict) else arg
        elif op == "Add":
            val[i] = o.add(L, R)
        elif op == "Sub":
            val[i] = o.sub(L, R)
        elif op == "Negate":
            val[i] = o.negate(L)
        elif op == "Multiply":
            val[i] = o.multiply(L, R)
        elif op == "Relinearize":
            val[i] = o.relinearize(L, rk) if L.shape[0] == 3 else L.copy()
        elif op == "AddPlaintext":
            val[i] = o.add_plain(L, R)
        elif op == "SubPlaintext":
            val[i] = o.sub_plain(L, R)
        elif op == "MultiplyPlaintext":
            val[i] = o.multiply_plain(L, R)
        elif op == "ShiftLeft":
            val[i] = o.rotate_rows(L, int(R), gk)
        elif op == "ShiftRight":
            val[i] = o.rotate_rows(L, -int(R), gk)
        elif op == "SwapRows":
            val[i] = o.rotate_columns(L, gk)
        elif op == "OutputCiphertext":
            outs.append(L)
        else:
            raise ValueError(op)
    return outs
