"""The oracle-side graph interpreter lives with the oracle (oracle/program_interp.py); tests import it from here."""
from oracle.program_interp import run_program  # noqa: F401
