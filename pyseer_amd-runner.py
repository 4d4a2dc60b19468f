#!/usr/bin/env python
"""Run the association front-end from the source tree: same command line as pyseer (see `python pyseer_amd-runner.py -h`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pyseer_amd.__main__ import main  # noqa: E402

if __name__ == '__main__':
    main()
