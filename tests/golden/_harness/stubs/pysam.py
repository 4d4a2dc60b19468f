"""Inert stand-in so pyseer.input imports without pysam (k-mer/Rtab paths never use it)."""
class VariantFile(object):
    def __init__(self, *a, **k):
        raise RuntimeError("pysam is not available in the oracle harness")
