"""Run the reference CLI under the oracle interpreter: python3.9 run_pyseer.py <pyseer args>."""
import shim  # noqa: F401  (must precede statsmodels)
from pyseer.__main__ import main
main()
