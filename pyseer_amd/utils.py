"""TSV row formatting (pyseer/utils.py:39-105): numbers as '%.2E', non-finite as empty, lineage label or NA, notes last."""
import numpy as np

from .classes import NOTE_ORDER


def _num(x):
    return '%.2E' % float(x) if (x is not None and np.isfinite(x)) else ''


def format_output(item, lineage_dict=None, model='seer', print_samples=False):
    """One output row for a Seer / LMM (/ Enet) tuple.  Notes are emitted in the engine's fixed bit order (the reference
    joins a Python set, whose order is not defined)."""
    fields = [str(item.kmer)]
    if model in ("enet", "rf"):
        fields += [_num(x) for x in (item.af, item.prep, item.pvalue, item.kbeta)]
    else:
        fields += [_num(x) for x in (item.af, item.prep, item.pvalue, item.kbeta, item.bse)]
        if model == 'lmm':
            fields.append(_num(item.frac_h2))
        else:
            fields.append(_num(item.intercept))
            betas = item.betas
            if betas is not None and not np.all(np.equal(betas, None)):
                fields += [_num(x) for x in betas]
    if lineage_dict is not None:
        ml = item.max_lineage
        fields.append(lineage_dict[ml] if (ml is not None and np.isfinite(ml)) else 'NA')
    if print_samples:
        fields += [','.join(item.kstrains), ','.join(item.nkstrains)]
    notes = [n for n in NOTE_ORDER if n in item.notes] + sorted(n for n in item.notes if n not in NOTE_ORDER)
    fields.append(','.join(notes))
    return '\t'.join(fields)
