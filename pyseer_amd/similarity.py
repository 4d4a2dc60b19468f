"""Sample similarity (kinship) matrix from variant presence/absence: mirrors the reference's `similarity` script
(pyseer/similarity.py) with the accumulation K = G G^T done by libseerhip (csrc/sim_kernels.hip: bit-plane AND + popcount).

Same command line, same stderr progress lines, same TSV on stdout."""
import sys

import numpy as np

from . import __version__

block_size = 1000                      # pyseer/similarity.py:16 (sets the granularity of the 'Matrix size' progress lines)
_FLUSH_ROWS = 1 << 16                  # packed rows handed to the device per call


def get_options(argv=None):
    import argparse

    description = 'Calculate a similarity matrix using variant presence/absence information'
    parser = argparse.ArgumentParser(description=description, prog='similarity')
    parser.add_argument("samples", help="List of sample names to use")
    variant_group = parser.add_mutually_exclusive_group(required=True)
    variant_group.add_argument('--kmers', default=None, help='Kmers file')
    variant_group.add_argument('--vcf', default=None, help='VCF file. Will filter any non \'PASS\' sites')
    variant_group.add_argument('--pres', default=None, help='Presence/absence .Rtab matrix as produced by roary and piggy')
    parser.add_argument('--min-af', type=float, default=0.01, help='Minimum AF [Default: 0.01]')
    parser.add_argument('--max-af', type=float, default=0.99, help='Maximum AF [Default: 0.99]')
    parser.add_argument('--max-missing', type=float, default=0.05, help='Maximum missing (vcf/Rtab) [Default: 0.05]')
    parser.add_argument('--uncompressed', action='store_true', default=False,
                        help='Uncompressed kmers file [Default: gzipped]')
    parser.add_argument('--gpu', type=int, default=0, help='Device index [Default: 0]')
    parser.add_argument('--python-reader', action='store_true', default=False,
                        help='Parse k-mer files with the Python reader instead of the native one')
    parser.add_argument('--version', action='version', version='%(prog)s ' + __version__)
    return parser.parse_args(argv)


class SimilarityAccumulator(object):
    """K = G G^T over packed presence rows, accumulated on the device; samples that carry a missing call in a kept variant
    come out as NaN rows/columns, as np.matmul over the reference's NaN-holding G gives (similarity.py:113)."""

    def __init__(self, n_samples, device=0):
        from .engine import Engine
        self.n = n_samples
        self.engine = Engine(n_samples, device=device)
        self.engine.sim_begin()
        self._pending = []
        self._rows = 0
        self._nan = np.zeros(n_samples, dtype=bool)

    def add_packed(self, bits):
        if bits.shape[0] == 0:
            return
        self._pending.append(bits)
        self._rows += bits.shape[0]
        if self._rows >= _FLUSH_ROWS:
            self.flush()

    def add_dense(self, k):
        """One variant as the reference's k vector (0/1, NaN = missing call)."""
        k = np.asarray(k, dtype=float)
        miss = np.isnan(k)
        self._nan |= miss
        from .packing import pack_variants
        self.add_packed(pack_variants(np.where(miss, 0.0, k)[None, :]))

    def flush(self):
        if self._pending:
            self.engine.sim_accumulate(np.ascontiguousarray(np.concatenate(self._pending, axis=0)))
        self._pending = []
        self._rows = 0

    def finish(self):
        self.flush()
        K = self.engine.sim_finish()
        if self._nan.any():
            K[self._nan, :] = np.nan
            K[:, self._nan] = np.nan
        return K


def similarity_matrix(p, var_type, infile, path, all_strains, sample_order, min_af, max_af, max_missing, uncompressed,
                      device=0, python_reader=False, progress=None):
    """The loop of similarity.py:99-113 over PackedBlocks."""
    from .input import iter_packed_blocks, iter_packed_blocks_native
    acc = SimilarityAccumulator(len(p), device=device)
    if var_type == "kmers" and not python_reader:
        blocks = iter_packed_blocks_native(p, path, min_af, max_af, block_size)
    else:
        blocks = iter_packed_blocks(p, var_type, infile, all_strains, sample_order, min_af, max_af, max_missing,
                                    uncompressed, block_size, want_patterns=False)
    nblocks = 0
    for blk in blocks:
        st_arr = np.asarray(blk.status)
        # blocks of the native reader carry every parsed row (the association driver lets the engine apply the AF window); here only the kept
        # rows enter K = G G^T, as the reference's load_var_block drops the filtered ones (pyseer/input.py:693)
        acc.add_packed(blk.bits if (st_arr == 0).all() or blk.bits.shape[0] != st_arr.shape[0] else np.ascontiguousarray(blk.bits[st_arr == 0]))
        for st, k in zip(blk.status, blk.ks):
            if st == 2:
                acc.add_dense(k)
        nblocks += 1
        if progress is not None:
            progress(nblocks * block_size)
    if nblocks == 0 and progress is not None:
        progress(block_size)
    return acc.finish()


def main(argv=None):
    import pandas as pd
    from .input import open_variant_file

    options = get_options(argv)
    sample_list = []
    with open(options.samples, 'r') as sample_file:
        for sample in sample_file:
            sample_list.append(sample.rstrip())
    p = pd.Series(np.zeros(len(sample_list)), index=sample_list)
    all_strains = set(p.index)
    sample_order = []
    infile = None
    path = None
    if options.kmers:
        var_type, path = "kmers", options.kmers
        if options.python_reader:
            infile, sample_order = open_variant_file("kmers", options.kmers, uncompressed=options.uncompressed)
    elif options.vcf:
        var_type = "vcf"
        infile, sample_order = open_variant_file("vcf", options.vcf)
    else:
        var_type = "Rtab"
        infile, sample_order = open_variant_file("Rtab", options.pres)
    sys.stderr.write("Reading in variants\n")
    K = similarity_matrix(p, var_type, infile, path, all_strains, sample_order, options.min_af, options.max_af,
                          options.max_missing, options.uncompressed, device=options.gpu,
                          python_reader=options.python_reader,
                          progress=lambda n: sys.stderr.write('Matrix size ' + str(n) + '\n'))
    sys.stderr.write("Calculating sample similarity\n")
    pd.DataFrame(K, index=p.index, columns=p.index).to_csv(sys.stdout, sep='\t')


if __name__ == "__main__":
    main()
