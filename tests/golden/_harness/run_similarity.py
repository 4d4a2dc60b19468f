"""Run the reference's similarity script under the oracle interpreter: python3.9 run_similarity.py <similarity args>."""
import shim  # noqa: F401
from pyseer.similarity import main
main()
