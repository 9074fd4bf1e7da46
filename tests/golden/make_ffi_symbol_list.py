"""Lists every FFI symbol the reference's `seal_fhe` crate binds (`bindgen::<name>(...)` in seal_fhe/src/*.rs), with the
number of arguments at its call sites, into tests/golden/seal_fhe_ffi_symbols.txt ("Name argc" per line).  Run in the
build container (needs /root/reference); the list travels, the reference does not.  tests/test_cabi_cpu.py checks that
libhipbfv declares and exports each of them with that arity."""
import glob
import os
import re

REF = "/root/reference/seal_fhe/src"


def call_arity(text: str, start: int) -> int:
    """text[start] is the '(' of a call; count top-level arguments."""
    depth, args, seen = 0, 0, False
    i = start
    while i < len(text):
        ch = text[i]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return args + (1 if seen else 0)
        elif ch == "," and depth == 1:
            args += 1
            seen = False
        elif depth >= 1 and not ch.isspace():
            seen = True
        i += 1
    raise ValueError("unbalanced call")


arity: dict[str, set[int]] = {}
for path in sorted(glob.glob(os.path.join(REF, "*.rs"))):
    text = open(path).read()
    for m in re.finditer(r"bindgen::([A-Z][A-Za-z_0-9]*)\s*\(", text):
        arity.setdefault(m.group(1), set()).add(call_arity(text, m.end() - 1))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seal_fhe_ffi_symbols.txt")
with open(out, "w") as f:
    for name in sorted(arity):
        assert len(arity[name]) == 1, (name, arity[name])
        f.write(f"{name} {next(iter(arity[name]))}\n")
print(len(arity), "symbols ->", out)
