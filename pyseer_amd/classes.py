"""Result containers; field order is the contract (pyseer/classes.py:3-22)."""
from collections import namedtuple

LMM = namedtuple('LMM', ['kmer', 'pattern', 'af', 'prep', 'pvalue', 'kbeta', 'bse', 'frac_h2',
                         'max_lineage', 'kstrains', 'nkstrains', 'notes', 'prefilter', 'filter'])

Seer = namedtuple('Seer', ['kmer', 'pattern', 'af', 'prep', 'pvalue', 'kbeta', 'bse', 'intercept', 'betas',
                           'max_lineage', 'kstrains', 'nkstrains', 'notes', 'prefilter', 'filter'])

# bit order of the engine's flags word (include/seerhip.h SH_NOTE_*)
NOTE_ORDER = ('af-filter', 'pre-filtering-failed', 'bad-chisq', 'high-bse', 'perfectly-separable-data',
              'matrix-inversion-error', 'firth-fail', 'missing-data-error', 'lrt-filtering-failed')
FLAG_PREFILTER = 1 << 16
FLAG_FILTER = 1 << 17
FLAG_FIRTH_SENSITIVE = 1 << 18          # include/seerhip.h SH_FLAG_FIRTH_SENSITIVE: the reference's own Firth answer on this row depends on the order of its sums


def notes_from_flags(flags):
    """flags word -> the reference's notes set."""
    return set(n for i, n in enumerate(NOTE_ORDER) if (int(flags) >> i) & 1)
