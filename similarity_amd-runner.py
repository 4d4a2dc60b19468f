#!/usr/bin/env python
"""Run the similarity (kinship) tool from the source tree: same command line as pyseer's `similarity`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pyseer_amd.similarity import main  # noqa: E402

if __name__ == '__main__':
    main()
