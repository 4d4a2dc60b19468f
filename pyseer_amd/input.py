"""Variant stream and loaders, mirroring pyseer/input.py's contracts for k-mer and Rtab input.

The generators keep the reference's tuple layouts (iter_variants: the 16-tuple that is the argument list of
fixed_effects_regression, input.py:608-620; load_var_block: (variants, variant_mat, eof), input.py:678-707) so existing
callers keep working; `iter_packed_blocks` is the feed of the GPU driver: the same parsing, but presence goes straight
into packed bit rows.  VCF input needs pysam (absent here) and is not supported.
"""
import binascii
import gzip
import hashlib
import os
import sys

import numpy as np
import pandas as pd

from . import classes as var_obj
from . import _route
from .packing import row_bytes_for


# ---------------------------------------------------------------------------------------------------------------
# loaders (once per run)
# ---------------------------------------------------------------------------------------------------------------
def _read_indexed_table(path, what, **kw):
    t = pd.read_csv(path, index_col=0, sep='\t', **kw)
    t.index = t.index.astype(str)
    if np.any(t.index.duplicated()):
        sys.stderr.write('%s file contains duplicated sample names\n' % what)
        sys.exit(1)
    return t


def load_phenotypes(infile, column):
    """pandas.Series of phenotypes indexed by sample (input.py:24-59): last column unless `column` is named."""
    p = pd.read_csv(infile, index_col=0, sep='\t')
    if p.shape[1] < 1:
        sys.stderr.write('Phenotype file must contain at least one phenotype column\n')
        sys.exit(1)
    p.index = p.index.astype(str)
    if np.any(p.index.duplicated()):
        sys.stderr.write('Phenotype file contains duplicated sample names\n')
        sys.exit(1)
    p = (p[p.columns[-1]] if column is None else p[column]).dropna()
    if not pd.api.types.is_numeric_dtype(p.values.dtype):
        sys.stderr.write('Phenotypes must be numeric\n')
        sys.exit(1)
    return p


def cmdscale(D):
    """Classical MDS (cmdscale.py:15-54): double-centre the squared distances, keep positive-eigenvalue axes."""
    D = np.asarray(D, dtype=float)
    n = D.shape[0]
    H = np.eye(n) - np.ones((n, n)) / n
    B = -H.dot(D ** 2).dot(H) / 2
    evals, evecs = np.linalg.eigh(B)
    order = np.argsort(evals)[::-1]
    evals, evecs = evals[order], evecs[:, order]
    pos = evals > 0
    return evecs[:, pos] * np.sqrt(evals[pos]), evals[pos]


def load_structure(infile, p, max_dimensions, mds_type="classic", n_cpus=1, seed=None):
    """Distance matrix -> MDS projection restricted to phenotyped samples, columns scaled by max-abs (input.py:62-137)."""
    m = _read_indexed_table(infile, 'Structure')
    sys.stderr.write("Structure matrix has dimension " + str(m.shape) + "\n")
    keep = p.index.intersection(m.index).intersection(m.columns)
    m = m.loc[keep, keep]
    if len(keep) == 0:
        sys.stderr.write('None of the phenotyped samples were found in population structure matrix\n')
        sys.exit(1)
    if mds_type == "classic":
        projection, _ = cmdscale(m.values)
    else:
        from sklearn import manifold
        metric = mds_type != "non-metric"
        if mds_type not in ("metric", "non-metric"):
            sys.stderr.write("Unsupported mds type chosen. Assuming metric\n")
        mds = manifold.MDS(n_components=max_dimensions, metric=metric, n_jobs=n_cpus, random_state=seed,
                           dissimilarity='precomputed')
        projection = mds.fit_transform(m.values)
    m = pd.DataFrame(projection, index=m.index)
    for i in range(m.shape[1]):
        m[i] = m[i] / max(abs(m[i]))
    return m


def load_covariates(infile, covariates, p):
    """Covariate table -> design columns: 'Nq' quantitative, 'N' categorical dummy-encoded (input.py:184-248)."""
    c = _read_indexed_table(infile, 'Covariate', header=0)
    if len(p.index.difference(c.index)) > 0:
        sys.stderr.write("All samples with a phenotype must be present in covariate file\n")
        sys.exit(1)
    c = c.loc[p.index.intersection(c.index)]
    if covariates is None:
        return pd.DataFrame([])
    cols = []
    for col in covariates:
        cnum = int(col.rstrip('q'))
        if cnum == 1 or cnum > c.shape[1] + 1:
            sys.stderr.write('Covariates columns values should be > 1 and less than or equal to total number of ' +
                             'columns (%d)\n' % (c.shape[1] + 1))
            return None
        series = c.iloc[:, cnum - 2]
        if col[-1] == 'q':
            cols.append(series)
        else:
            categories = set(series)
            categories.pop()                       # one level is the reference class (arbitrary, as in the reference)
            for i, categ in enumerate(categories):
                cols.append(pd.Series([1 if x == categ else 0 for x in series.values], index=c.index,
                                      name=c.columns[cnum - 2] + "_" + str(i)))
    return pd.concat(cols, axis=1) if cols else pd.DataFrame([])


def load_lineage(infile, p):
    """Cluster file (sample<ws>cluster) -> (n x clusters 0/1 matrix, sorted labels) (input.py:140-181)."""
    names, labels = [], []
    for line in open(infile):
        f = line.rstrip().split()
        names.append(f[0]); labels.append(f[1])
    lin = pd.Series(labels, index=pd.Index(names).astype(str))
    lin = lin.loc[~lin.index.duplicated()]
    if len(p.index.difference(lin.index)) > 0:
        sys.stderr.write("All samples with a phenotype must be present in lineage file\n")
        sys.exit(1)
    lin = lin.loc[p.index.intersection(lin.index)]
    lineages = sorted(set(lin.values))
    mat = np.stack([(lin.values == c).astype(int) for c in lineages], axis=1)
    return mat, lineages


def open_variant_file(var_type, var_file, burden_file=None, burden_regions=None, uncompressed=False):
    """(handle, sample_order) (input.py:268-299)."""
    if var_type == "kmers":
        return (open(var_file) if uncompressed else gzip.open(var_file, 'r')), []
    if var_type == "Rtab":
        fh = open(var_file)
        header = fh.readline().rstrip()
        return fh, [str(x) for x in header.split()[1:]]
    raise ValueError('Variants type not supported (VCF input needs pysam, which this build does not ship)')


# ---------------------------------------------------------------------------------------------------------------
# per-variant parsing
# ---------------------------------------------------------------------------------------------------------------
def hash_pattern(k):
    """base64(md5(raw bytes of the presence vector)) + newline; the dtype of k matters (input.py:710-723)."""
    return binascii.b2a_base64(hashlib.md5(k.view(np.uint8)).digest())


def read_variant(infile, p, var_type, burden, burden_regions, uncompressed, all_strains, sample_order,
                 keep_list=None, noparse=False):
    """(eof, k, var_name, kstrains, nkstrains, af, missing) for the next line (input.py:301-454)."""
    if var_type not in ('kmers', 'Rtab'):
        raise ValueError('Variants type not supported')
    line = infile.readline()
    if not line or noparse:
        return True, None, None, None, None, None, None
    d = {}
    if var_type == "kmers":
        if not uncompressed:
            line = line.decode()
        var_name = line.split()[0]
        strains = line.rstrip().split('|')[1].lstrip().split()
        if keep_list is not None and var_name not in keep_list:
            return False, None, None, None, None, None, None
        d = {str(x.split(':')[0]): 1 for x in strains}
    else:
        try:
            fields = line.rstrip().split('\t')
        except TypeError:
            fields = line.decode().rstrip().split('\t')
        var_name, calls = fields[0], fields[1:]
        if keep_list is not None and var_name not in keep_list:
            return False, None, None, None, None, None, None
        if len(calls) == 0:
            raise ValueError('No sample data found; is this a Rtab file?')
        if len(calls) != len(sample_order):
            raise ValueError('Unexpected mismatch between header and data row')
        for present, sample in zip(calls, sample_order):
            if present not in ('0', '1', '.', ''):
                raise ValueError('Rtab file not binary')
            if present == '1':
                d[sample] = 1
            elif present in ('.', ''):
                d[sample] = np.nan
    kstrains = sorted(set(d.keys()).intersection(all_strains))        # missing calls count as present here
    nkstrains = sorted(all_strains.difference(set(kstrains)))
    for x in nkstrains:
        d[x] = 0
    af = float(len(kstrains)) / len(all_strains)
    if len(kstrains) == 0:
        sys.stderr.write("No observations of " + var_name + " in selected samples\n")
    k = np.array([d[x] for x in p.index if x in d])
    missing = float(np.sum(np.isnan(k))) / len(all_strains)
    return False, k, var_name, kstrains, nkstrains, af, missing


def iter_variants(p, m, cov, var_type, burden, burden_regions, infile, all_strains, sample_order, lineage_effects,
                  lineage_clusters, min_af, max_af, max_missing, filter_pvalue, lrt_pvalue, null_fit, firth_null,
                  uncompressed, continuous):
    """Generator of fixed_effects_regression argument tuples (input.py:505-620); p slot is None for AF/missing-filtered."""
    while True:
        eof, k, var_name, kstrains, nkstrains, af, missing = read_variant(infile, p, var_type, burden, burden_regions,
                                                                          uncompressed, all_strains, sample_order)
        if eof:
            return
        pattern = hash_pattern(k) if k is not None else None
        keep = (k is not None) and (min_af <= af <= max_af) and not (missing > max_missing)
        yield (var_name, p.values if keep else None, k, m, cov.values, af, pattern, lineage_effects, lineage_clusters,
               filter_pvalue, lrt_pvalue, null_fit, firth_null, kstrains, nkstrains, continuous)


def load_var_block(var_type, p, burden, burden_regions, infile, all_strains, sample_order, min_af, max_af, max_missing,
                   uncompressed, block_size):
    """Generator of (variants, variant_mat, eof) blocks for the LMM (input.py:638-707)."""
    while True:
        variants = []
        variant_mat = np.zeros((len(p), block_size))
        eof = False
        for var_idx in range(block_size):
            eof, k, var_name, kstrains, nkstrains, af, missing = read_variant(infile, p, var_type, burden, burden_regions,
                                                                              uncompressed, all_strains, sample_order)
            if eof:
                break
            if k is None or af < min_af or af > max_af or missing > max_missing:
                pattern = None
            else:
                pattern = hash_pattern(k)
                variant_mat[:, var_idx] = k
            variants.append((var_obj.LMM(var_name, pattern, af, np.nan, np.nan, np.nan, np.nan, np.nan, np.nan, kstrains,
                                         nkstrains, set(), True, True), p.values, k))
        yield (variants, variant_mat, eof)
        if eof:
            break
    yield None, None, True


def iter_variants_lmm(variant_iter, lmm, h2, lineage, lineage_clusters, covariates, continuous, filter_pvalue, lrt_pvalue):
    """input.py:623-635"""
    for variants, variant_mat, eof in variant_iter:
        if len(variants) == 0:
            break
        yield (lmm, h2, variants, variant_mat, lineage, lineage_clusters, covariates, continuous, filter_pvalue, lrt_pvalue)
        if eof:
            break


# ---------------------------------------------------------------------------------------------------------------
# GPU feed
# ---------------------------------------------------------------------------------------------------------------
class PackedBlock(object):
    """One block of parsed variants: metadata lists + packed presence rows for those that reach the engine."""
    __slots__ = ("_names", "patterns", "afs", "kstrains", "nkstrains", "status", "bits", "row_of", "ks", "last_k",
                 "names_blob", "name_off")

    def __init__(self, n_samples, capacity):
        self._names, self.patterns, self.afs, self.kstrains, self.nkstrains = [], [], [], [], []
        self.status = []            # 0 = to engine, 1 = af/missing filtered, 2 = carries missing calls (NaN in k)
        self.ks = []                # the dense k only for status 2 (host-side handling of the error path)
        self.row_of = []            # row in self.bits, or -1
        self.names_blob = self.name_off = None     # concatenated names + offsets when the native reader supplied them
        self.last_k = None          # dense k of the LAST variant parsed into the block (see __main__: lmm.py:209-213)
        self.bits = np.zeros((capacity, row_bytes_for(n_samples)), dtype=np.uint8)

    @property
    def names(self):
        """Variant names as a list of str; decoded from the name blob only when somebody asks (the block sink does not)."""
        if self._names is None:
            b, o = self.names_blob, self.name_off
            self._names = [b[o[i]:o[i + 1]].decode() for i in range(len(o) - 1)]
        return self._names

    @names.setter
    def names(self, v):
        self._names = v

    def __len__(self):
        return len(self.status)


def iter_packed_blocks(p, var_type, infile, all_strains, sample_order, min_af, max_af, max_missing, uncompressed,
                       block_size, want_patterns=True):
    """Parse `block_size` variants at a time into a PackedBlock (same filters as iter_variants / load_var_block)."""
    n = len(p)
    while True:
        blk = PackedBlock(n, block_size)
        nrow = 0
        eof = False
        for _ in range(block_size):
            eof, k, name, ks, nks, af, missing = read_variant(infile, p, var_type, False, None, uncompressed, all_strains,
                                                              sample_order)
            if eof:
                break
            blk.names.append(name); blk.afs.append(af); blk.kstrains.append(ks); blk.nkstrains.append(nks)
            blk.last_k = k
            if k is None or not (min_af <= af <= max_af) or missing > max_missing:
                blk.patterns.append(hash_pattern(k) if k is not None else None)
                blk.status.append(1); blk.row_of.append(-1); blk.ks.append(None)
                continue
            blk.patterns.append(hash_pattern(k) if want_patterns else b'')
            if missing > 0:
                blk.status.append(2); blk.row_of.append(-1); blk.ks.append(k)
                continue
            packed = np.packbits(k.astype(bool), bitorder="little")
            blk.bits[nrow, :packed.shape[0]] = packed
            blk.status.append(0); blk.row_of.append(nrow); blk.ks.append(None)
            nrow += 1
        blk.bits = blk.bits[:nrow]
        if blk.names:
            yield blk
        if eof:
            return


class NativeKmerReader(object):
    """Blocks of packed k-mer presence rows straight from the text file through libseerhip's reader (csrc/reader.cpp):
    gzip inflate + tokenise + sample lookup + bit packing in C++ (OpenMP over the lines of a block).  No GPU needed."""

    def __init__(self, path, sample_names, block_size=3000, max_name=256):
        import ctypes as C
        from . import _abi
        self._C, self._abi = C, _abi
        self._lib = _abi.load()
        self.samples = [str(x) for x in sample_names]
        self.n = len(self.samples)
        arr = (C.c_char_p * self.n)(*[x.encode() for x in self.samples])
        h = self._lib.sh_reader_open(str(path).encode(), arr, self.n)
        if not h:
            raise IOError(self._lib.sh_reader_error().decode())
        self._h = C.c_void_p(h)
        self.block_size = int(block_size)
        self.row_bytes = row_bytes_for(self.n)
        self._names = C.create_string_buffer(self.block_size * max_name)
        self._off = np.zeros(self.block_size + 1, dtype=np.int64)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sh_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def raw_blocks(self):
        """(bits, counts, names_blob, name_off) per block; names stay one bytes blob + offsets (what the block sink consumes)."""
        C, abi = self._C, self._abi
        while True:
            bits = np.zeros((self.block_size, self.row_bytes), dtype=np.uint8)
            counts = np.zeros(self.block_size, dtype=np.int32)
            nv = self._lib.sh_reader_next(self._h, self.block_size, bits.ctypes.data_as(abi.c_u8p), self.row_bytes,
                                          counts.ctypes.data_as(C.POINTER(C.c_int32)), self._names, len(self._names),
                                          self._off.ctypes.data_as(C.POINTER(C.c_int64)))
            if nv == -2:                                   # long names (unitigs): nothing was consumed, retry with room for them
                self._names = C.create_string_buffer(int(self._lib.sh_reader_names_needed(self._h)) * 2 + 1024)
                continue
            if nv < 0:
                raise IOError(self._lib.sh_reader_error().decode())
            if nv == 0:
                return
            yield bits[:nv], counts[:nv], self._names.raw[:self._off[nv]], self._off[:nv + 1].copy()

    def __iter__(self):
        """(names, bits, counts) per block, names decoded to str."""
        for bits, counts, blob, off in self.raw_blocks():
            yield [blob[off[v]:off[v + 1]].decode() for v in range(counts.shape[0])], bits, counts


def prefetched(iterable, depth=2):
    """Run `iterable` in a background thread, `depth` items ahead.  The native reader spends its time inside ctypes calls (GIL
    released: inflate + OpenMP parse), so block k+1 is read and packed while block k is on the GPU / being formatted."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    end = object()

    def work():
        try:
            for item in iterable:
                q.put((item, None))
        except BaseException as ex:           # re-raised in the consumer
            q.put((None, ex))
        q.put((end, None))

    t = threading.Thread(target=work, daemon=True)
    t.start()
    while True:
        item, ex = q.get()
        if ex is not None:
            raise ex
        if item is end:
            return
        yield item


def strains_from_bits(row, samples_sorted_idx, samples):
    """(kstrains, nkstrains), both lexicographically sorted like read_variant's (input.py:439-440), from one packed row."""
    present = np.unpackbits(row, bitorder="little")[:len(samples)].astype(bool)
    ks = [samples[i] for i in samples_sorted_idx if present[i]]
    nks = [samples[i] for i in samples_sorted_idx if not present[i]]
    return ks, nks


_NO_STRAINS = ()           # shared placeholder when the sample lists are not wanted (read-only)


class _Repeat(object):
    """n copies of one read-only value, indexable and iterable like the list it stands for (a block's per-row lists of placeholders were
    four 262 144-element lists per block, built while holding the GIL)."""
    __slots__ = ("v", "n")

    def __init__(self, v, n):
        self.v, self.n = v, int(n)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self.v] * len(range(*i.indices(self.n)))
        if not -self.n <= i < self.n:
            raise IndexError(i)
        return self.v

    def __iter__(self):
        import itertools
        return itertools.repeat(self.v, self.n)


class RawBlock(object):
    """One block as the job stream takes it (engine.Job.submit): the parsed rows and nothing derived from them -- the AF window, the masks,
    the counters and the choice of the printed rows happen on the device.  dma: the rows may be read by the device where they lie
    (a registered window of the packed-cache mapping); release(): called once the block has been collected."""
    __slots__ = ("blob", "off", "counts", "bits", "dma", "release")

    def __init__(self, blob, off, counts, bits, dma=False, release=None):
        self.blob, self.off, self.counts, self.bits, self.dma, self.release = blob, off, counts, bits, dma, release

    def __len__(self):
        return int(self.counts.shape[0])


def _raw_block(blob, off, counts, bits, dma=False, release=None):
    off = np.require(off, dtype=np.int64, requirements=["A", "C"]); counts = np.require(counts, dtype=np.int32, requirements=["A", "C"])   # (views of a mapping may be misaligned)
    for i in np.nonzero(counts == 0)[0]:                    # (read_variant's message, pyseer/input.py:441-443)
        sys.stderr.write("No observations of " + bytes(blob[off[i]:off[i + 1]]).decode() + " in selected samples\n")
    return RawBlock(blob, off, counts, bits, dma, release)


def _block_from_raw(n, samples, order, names, blob, off, bits, counts, min_af, max_af, want_patterns, want_samples):
    """PackedBlock from one raw block of the native reader / the packed cache (all parsed variants, before the AF filter)."""
    blk = PackedBlock(n, 0)
    blk.names_blob, blk.name_off = blob, off
    afs = counts.astype(np.float64) / n
    keep = (afs >= min_af) & (afs <= max_af)
    blk.names = names                                    # None = decode lazily from the blob
    # arrays, not lists (round 3: 262 144-element .tolist() calls and their np.asarray() twins in the sink were a third of a block's
    # host time), and EVERY row goes to the engine in place: no compacted copy of the kept rows (170 MB per block at N = 5000).  The
    # driver sets the engine's own AF window (sh_set_af_filter: same inclusive bounds on count / n, input.py:608), so rows outside it
    # cost the GPU nothing; their statistics are masked by `status` in the sink either way.
    blk.afs = afs
    blk.status = np.where(keep, 0, 1).astype(np.int8)
    nv = int(counts.shape[0])
    blk.row_of = np.arange(nv, dtype=np.int64)
    blk.ks = _Repeat(None, nv)
    blk.bits = bits
    if want_samples:
        sp = [strains_from_bits(bits[i], order, samples) for i in range(nv)]
        blk.kstrains = [a for a, _ in sp]; blk.nkstrains = [b for _, b in sp]
    else:
        blk.kstrains = _Repeat(_NO_STRAINS, nv); blk.nkstrains = blk.kstrains
    if want_patterns:
        dense = np.unpackbits(bits, axis=1, bitorder="little")[:, :n].astype(np.int64)
        blk.patterns = [hash_pattern(dense[i]) for i in range(nv)]
    else:
        blk.patterns = _Repeat(b'', nv)
    for i in np.nonzero(counts == 0)[0]:
        sys.stderr.write("No observations of " + blob[off[i]:off[i + 1]].decode() + " in selected samples\n")
    blk.last_k = np.unpackbits(bits[nv - 1], bitorder="little")[:n].astype(np.int64)
    return blk


def iter_packed_blocks_native(p, path, min_af, max_af, block_size, want_patterns=False, want_samples=False, save_to=None, raw=False):
    """Same PackedBlock stream as iter_packed_blocks for k-mer files, fed by the native reader.  save_to: a PackedCacheWriter that
    receives every raw block (all parsed variants, before the AF filter).  raw: RawBlock objects for the job stream instead."""
    samples = [str(x) for x in p.index]
    order = sorted(range(len(samples)), key=lambda i: samples[i])
    n = len(samples)
    reader = NativeKmerReader(path, samples, block_size)
    def finished():                      # the whole of a block's host preparation runs on the reader's thread, ahead of the engine
        for bits, counts, blob, off in reader.raw_blocks():
            if save_to is not None:
                save_to.write_block(blob, off, counts, bits)
            if raw:
                yield _raw_block(blob, off, counts, bits)
            else:
                yield _block_from_raw(n, samples, order, None, blob, off, bits, counts, min_af, max_af, want_patterns, want_samples)

    for blk in prefetched(finished()):
        yield blk


def iter_packed_blocks_native_multi(p, paths, min_af, max_af, block_size, want_patterns=False, want_samples=False,
                                    ahead_bytes=2 << 30, raw=False):
    """Several k-mer files as ONE stream, in the order given (`--kmers a.gz b.gz ...`), with all of them read at once.

    A single gzip stream inflates serially (38 k k-mers/s at N = 5000 on the GPU host, against tens of millions per second of
    compute), so the only way to feed the engine faster from gzip is more streams.  Every file gets its own native reader on its
    own thread, started immediately; what a reader produces are packed bit rows (N / 8 bytes per k-mer, ~40x smaller than the text),
    queued per file.  The consumer drains the queues in file order, so the output is what the concatenated file would give; while it
    works through the first file the others are parsed into their queues, each bounded by `ahead_bytes` of packed rows."""
    import queue
    import threading
    samples = [str(x) for x in p.index]
    order = sorted(range(len(samples)), key=lambda i: samples[i])
    n = len(samples)
    end = object()

    class Feed(object):
        def __init__(self, path):
            self.q = queue.Queue()
            self.room = threading.Semaphore(0)
            self.held = 0
            self.lock = threading.Lock()
            self.reader, self.open_error = None, None
            try:
                self.reader = NativeKmerReader(path, samples, block_size)
            except BaseException as ex:       # re-raised in the consumer, in file order
                self.open_error = ex
            self.t = threading.Thread(target=self.work, daemon=True)
            self.t.start()

        def work(self):
            try:
                if self.open_error is not None:
                    raise self.open_error
                for raw in self.reader.raw_blocks():
                    nb = int(raw[0].nbytes) + int(raw[2].nbytes if hasattr(raw[2], "nbytes") else len(raw[2]))
                    while True:
                        with self.lock:
                            if self.held == 0 or self.held + nb <= ahead_bytes:
                                self.held += nb
                                break
                        self.room.acquire()
                    self.q.put((raw, nb, None))
            except BaseException as ex:       # re-raised in the consumer
                self.q.put((None, 0, ex))
            self.q.put((end, 0, None))

        def blocks(self):
            while True:
                raw, nb, ex = self.q.get()
                if ex is not None:
                    raise ex
                if raw is end:
                    return
                with self.lock:
                    self.held -= nb
                self.room.release()
                yield raw

    # one look-ahead budget and one pool of parser threads for all the files (round 2 gave every file 2 GB and a full pool: 16 GB and ~380
    # threads for 8 files under a 16-CPU quota)
    ahead_bytes = max(64 << 20, int(ahead_bytes) // max(1, len(paths)))
    from . import _abi
    _abi.load().sh_reader_set_concurrency(len(paths))
    try:
        feeds = [Feed(path) for path in paths]                       # every reader is opened here, under the shared-pool hint
    finally:
        _abi.load().sh_reader_set_concurrency(1)
    for f in feeds:
        for bits, counts, blob, off in f.blocks():
            if raw:
                yield _raw_block(blob, off, counts, bits)
            else:
                yield _block_from_raw(n, samples, order, None, blob, off, bits, counts, min_af, max_af, want_patterns, want_samples)


# ---------------------------------------------------------------------------------------------------------------
# Packed cache: the parsed k-mer file as bit rows, so that further runs over the same samples (another phenotype column,
# other covariates, LMM after fixed effects) skip gzip and text parsing and stream at the engine's rate.
#   header: b"SEERPK01", uint32 n_samples, uint32 row_bytes, uint64 len, sample names joined by "\n" (phenotype order)
#   blocks: uint64 nv, uint64 name_bytes, int64 name_off[nv+1], int32 counts[nv], names, uint8 bits[nv*row_bytes]; nv = 0 ends
# ---------------------------------------------------------------------------------------------------------------
_PK_MAGIC = b"SEERPK01"


class PackedCacheWriter(object):
    """Writes to `path + ".part"` and renames onto `path` in close(), after the terminator and an fsync: a run that is interrupted or fails
    leaves no `path` at all (round 2 wrote in place, and a truncated cache with a valid header was accepted by every later run until it
    died in the middle of its output).  `stamp` = (sidecar path, text): written only once the cache itself is in place."""

    def __init__(self, path, samples, stamp=None):
        self._path, self._tmp, self._stamp = path, path + ".part", stamp
        self._f = open(self._tmp, "wb")
        names = "\n".join(samples).encode()
        self._f.write(_PK_MAGIC + np.array([len(samples), row_bytes_for(len(samples))], dtype="<u4").tobytes()
                      + np.array([len(names)], dtype="<u8").tobytes() + names)

    def write_block(self, blob, off, counts, bits):
        nv = int(counts.shape[0])
        self._f.write(np.array([nv, len(blob)], dtype="<u8").tobytes())
        self._f.write(np.ascontiguousarray(off, dtype="<i8").tobytes())
        self._f.write(np.ascontiguousarray(counts, dtype="<i4").tobytes())
        self._f.write(blob)
        self._f.write(np.ascontiguousarray(bits, dtype=np.uint8).tobytes())

    def close(self):
        import os
        if self._f is not None:
            self._f.write(np.array([0, 0], dtype="<u8").tobytes())
            self._f.flush()
            os.fsync(self._f.fileno())
            self._f.close()
            self._f = None
            os.replace(self._tmp, self._path)
            if self._stamp is not None:
                with open(self._stamp[0], "w") as f:
                    f.write(self._stamp[1] + "\n")

    def abort(self):
        """Drop what was written (the run did not finish): no cache, no stamp."""
        import os
        if self._f is not None:
            self._f.close()
            self._f = None
            try:
                os.remove(self._tmp)
            except OSError:
                pass

    def __del__(self):
        try:
            self.abort()
        except Exception:
            pass


def packed_cache_complete(path):
    """True if `path` ends with the terminator close() writes (nv = 0, name_bytes = 0): a cheap test that the file was written to its end."""
    import os
    try:
        if os.path.getsize(path) < 8 + 8 + 8 + 16:
            return False
        with open(path, "rb") as f:
            f.seek(-16, 2)
            return f.read(16) == b"\0" * 16
    except (IOError, OSError):
        return False


class _DmaWindows(object):
    """Windows of the packed-cache mapping registered with the HIP runtime (sh_host_register) so that a block's rows go from the page cache
    to the device by DMA, with no CPU copy (4 ns of CPU per 632-byte row against 21 for the copy into pinned staging:
    profiles/r05/host_feed_probe.txt).  Registering is cheap (1.2 ms per 160 MB, also with the device busy) but UNREGISTERING waits for the
    device to drain (28 ms per call under load, tools/gpu_probe_register.py): so a window spans several stored blocks (up to WINDOW bytes,
    cut at block boundaries: windows never share a page), is unregistered once all its blocks have been collected, and by a JANITOR thread of
    its own -- never by the thread that feeds the device, and since the lanes (a fixed-effects device is never idle between blocks: the
    reader thread, which did it until then, stood 0.33 of 1.2 s in hipHostUnregister) not by the reader either."""
    WINDOW = 1 << 30
    # hipHostUnregister waits for the device to drain AND holds new work back while it does: every call is a hole in the device's timeline
    # (16 windows of a 24 M-row fixed-effects job: 155 of 820 ms idle).  So finished windows stay registered (their pages pinned) until the
    # stream ends, or until more than KEEP bytes of them have piled up -- a quarter of the machine's memory, at most 64 GB per stream.
    # KEEP is the PROCESS's budget: a `--gpus G` job runs G streams, each with its own windows, and G x a quarter of the memory would be
    # twice the machine at G = 8: a stream's share is KEEP / (streams alive).
    try:
        KEEP = min(64 << 30, os.sysconf("SC_PHYS_PAGES") * os.sysconf("SC_PAGE_SIZE") // 4)
    except (ValueError, OSError, AttributeError):
        KEEP = 8 << 30
    PAGE = 4096                        # sh_host_register rounds a window out to whole pages
    _live = 0                          # streams with windows, process-wide
    _live_lock = None

    def _keep(self):
        return self.KEEP // max(1, _DmaWindows._live)

    def __init__(self, device):
        import threading
        from . import _abi
        if _DmaWindows._live_lock is None:
            _DmaWindows._live_lock = threading.Lock()
        with _DmaWindows._live_lock:
            _DmaWindows._live += 1
        self._counted = True
        self._lib = _abi.load()
        self._device = int(device)
        self._lock = threading.Lock()
        self._win_of = {}              # stored block -> window
        self._done = []                # windows whose blocks have all been collected, to unregister
        self._reader_done = False
        self.ok = True
        self._wake = threading.Condition(self._lock)
        self._janitor = None
        self._stop = False

    def plan(self, base_addr, extents, lo_blk, hi_blk):
        # Windows are registered rounded out to whole pages and must not overlap (registering a page twice fails): where less than a page
        # of metadata lies between two stored blocks (small blocks), the block that starts on the previous window's last page belongs to no
        # window -- its rows are staged -- and the next window starts at the first block on a page of its own.
        j, prev_end = lo_blk, 0
        while j < hi_blk:
            if ((base_addr + extents[j][0]) & ~(self.PAGE - 1)) < prev_end:
                j += 1
                continue
            k, lo = j, base_addr + extents[j][0]
            hi = lo + extents[j][1]
            while k + 1 < hi_blk and base_addr + extents[k + 1][0] + extents[k + 1][1] - lo <= self.WINDOW:
                k += 1
                hi = base_addr + extents[k][0] + extents[k][1]
            w = {"lo": lo, "n": hi - lo, "blocks": k - j + 1, "taken": 0, "released": 0, "state": 0}     # state 0 new, 1 registered, -1 refused
            for b in range(j, k + 1):
                self._win_of[b] = w
            prev_end = (hi + self.PAGE - 1) & ~(self.PAGE - 1)
            j = k + 1

    def _unregister_done(self):
        with self._lock:
            todo, self._done = self._done, []
        for w in todo:
            self._lib.sh_host_unregister(w["lo"])

    def _janitor_loop(self):
        while True:
            with self._wake:
                while not self._due() and not self._stop:
                    self._wake.wait()
                if not self._done and self._stop:
                    return
                todo = []
                if self._reader_done or self._stop:
                    todo, self._done = self._done, []
                else:
                    while self._done and sum(w["n"] for w in self._done) > self._keep():
                        todo.append(self._done.pop(0))
            for w in todo:
                self._lib.sh_host_unregister(w["lo"])

    def _due(self):
        """(lock held) is there a window to unregister NOW: the stream has ended, or too many finished windows are still pinned"""
        return bool(self._done) and (self._reader_done or sum(w["n"] for w in self._done) > self._keep())

    def _kick(self):
        """(lock held) a window is ready to be unregistered: wake the janitor, starting it at the first need"""
        if self._janitor is None:
            import threading
            self._janitor = threading.Thread(target=self._janitor_loop, daemon=True)
            self._janitor.start()
        self._wake.notify()

    def take(self, j):
        """(dma, release) for stored block j, called on the reader thread just before the block is handed on."""
        w = self._win_of.get(j)
        if w is None or not self.ok:
            return False, None
        if w["state"] == 0:
            if self._lib.sh_host_register(w["lo"], w["n"], self._device) != 0:
                w["state"] = -1
                self.ok = False        # (a file system whose pages cannot be pinned, or the pinning limit: the rows are staged through pinned memory from here on)
                sys.stderr.write("pyseer_amd: the packed-cache mapping cannot be registered for DMA (%s): this stream's rows are staged through pinned "
                                 "memory from here on (slower on the host, same results)\n" % self._lib.sh_last_error().decode())
                return False, None
            w["state"] = 1
        if w["state"] != 1:
            return False, None
        w["taken"] += 1

        def release():
            with self._lock:
                w["released"] += 1
                last = w["released"] == w["blocks"] or (self._reader_done and w["released"] == w["taken"])
                if last:
                    self._done.append(w)
                    self._kick()
        return True, release

    def reader_done(self):
        if self._counted:                                              # (this stream pins nothing new from here on: its share goes back)
            self._counted = False
            with _DmaWindows._live_lock:
                _DmaWindows._live -= 1
        with self._lock:
            self._reader_done = True
            for w in {id(x): x for x in self._win_of.values()}.values():
                if w["state"] == 1 and w["released"] == w["taken"] and w not in self._done and w["released"] < w["blocks"]:
                    self._done.append(w)
            if self._done:
                self._kick()


def check_packed_cache(p, path):
    """The header checks of iter_packed_blocks_cached without reading a block (the library's own block loop, sh_job_run_packed, walks the file
    itself): magic, sample list and order, row width.  Raises what the iterator raises."""
    samples = [str(x) for x in p.index]
    with open(path, "rb") as f:
        head = f.read(24)
        if len(head) < 24 or head[:8] != _PK_MAGIC:
            raise IOError("%s is not a packed k-mer cache" % path)
        ns, rb = np.frombuffer(head, dtype="<u4", count=2, offset=8)
        (ln,) = np.frombuffer(head, dtype="<u8", count=1, offset=16)
        names = f.read(int(ln))
        if len(names) < int(ln):
            raise IOError("truncated packed cache %s (delete it, or run without --load-packed)" % path)
    stored = names.decode().split("\n")
    if stored != samples:
        raise ValueError("packed cache was written for a different sample list / order (%d vs %d samples)" % (len(stored), len(samples)))
    if int(rb) != row_bytes_for(len(samples)):
        raise IOError("packed cache row width mismatch")


def iter_packed_blocks_cached(p, path, min_af, max_af, block_size, want_patterns=False, want_samples=False, part=None, raw=False, device=None, ahead=2):
    """PackedBlock stream from a packed cache written by --save-packed; the samples (and their order) must be the run's own.
    Stored blocks are re-cut to about `block_size` variants (stored blocks are never split, only merged).
    part = (i, n): only range i of n contiguous ranges of the cache's rows (the multi-GPU job: one range per device,
    pyseer_amd/__main__.py).  Ranges are made of whole blocks of the single stream, so the n parts yield exactly its blocks; they are
    as equal as block_size allows (a cache of B blocks over n devices: ceil(B/n) against floor(B/n) blocks).
    raw: RawBlock objects for the job stream; with `device` given, a block that is one stored block (no merge) is handed on as a window of the
    mapping registered for DMA (sh_host_register: the rows go from the page cache to the device with no CPU copy).
    ahead: blocks the reader thread runs ahead of the consumer (the job stream of a fixed-effects model holds 2 + lanes blocks in flight)."""
    samples = [str(x) for x in p.index]
    order = sorted(range(len(samples)), key=lambda i: samples[i])
    n = len(samples)

    def raw_blocks():
        # The cache is memory-mapped and a block's bit rows are a VIEW into the mapping (round 3: f.read() of a 170 MB block was a 40 ms copy
        # per block, more than the GPU needs to test it): the engine's host-to-device copy reads the page cache directly.
        import mmap
        with open(path, "rb") as f:
            size = os.fstat(f.fileno()).st_size
            if size < 24:
                raise IOError("%s is not a packed k-mer cache" % path)
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            mm.madvise(mmap.MADV_SEQUENTIAL)
        except (AttributeError, OSError):
            pass
        if mm[:8] != _PK_MAGIC:
            raise IOError("%s is not a packed k-mer cache" % path)
        ns, rb = np.frombuffer(mm, dtype="<u4", count=2, offset=8)
        (ln,) = np.frombuffer(mm, dtype="<u8", count=1, offset=16)
        pos = 24 + int(ln)
        if pos > size:
            raise IOError("truncated packed cache %s (delete it, or run without --load-packed)" % path)
        stored = mm[24:pos].decode().split("\n")
        if stored != samples:
            raise ValueError("packed cache was written for a different sample list / order (%d vs %d samples)"
                             % (len(stored), n))
        if int(rb) != row_bytes_for(n):
            raise IOError("packed cache row width mismatch")
        rb = int(rb)

        def need(nbytes):
            nonlocal pos
            if pos + nbytes > size:
                raise IOError("truncated packed cache %s (delete it, or run without --load-packed)" % path)
            at = pos
            pos += nbytes
            return at
        lo_blk, hi_blk = 0, None
        extents = None                                     # (offset, bytes) of every stored block's bit rows, when somebody needs them up front
        if part is not None or use_dma:
            # index pass: the 16-byte header of every stored block (a few hundred bytes of the mapping are touched).  The single stream merges
            # stored blocks until it holds block_size rows (merged() below); a part owns whole such groups -- the groups whose first row falls
            # into its share of the rows -- so every part hands the engine exactly the blocks the single stream would, and output that depends
            # on where blocks end (fit_lmm's filtered-first order, its stale-k lineage: pyseer/lmm.py:160-217) does not change
            start = pos
            nvs, extents = [], []
            while True:
                at = need(16)
                nv, nb = (int(x) for x in np.frombuffer(mm, dtype="<u8", count=2, offset=at))
                if nv == 0:
                    break
                need(8 * (nv + 1) + 4 * nv + nb)
                extents.append((need(nv * rb), nv * rb))
                nvs.append(nv)
            pos = start
        if part is not None:
            total = sum(nvs)
            i_, n_ = part
            owner, first_row, rows, row0 = [], 0, 0, 0
            for j, nv in enumerate(nvs):
                if rows == 0:
                    first_row = row0
                owner.append(min(n_ - 1, (first_row * n_) // max(total, 1)))
                rows += nv; row0 += nv
                if rows >= block_size:
                    rows = 0
            mine = [j for j, o in enumerate(owner) if o == i_]
            lo_blk, hi_blk = (mine[0], mine[-1] + 1) if mine else (0, 0)
        if use_dma:
            windows.plan(np.frombuffer(mm, dtype=np.uint8, count=1).ctypes.data, extents, lo_blk, len(extents) if hi_blk is None else hi_blk)
        j = -1
        while True:
            at = need(16)
            nv, nb = (int(x) for x in np.frombuffer(mm, dtype="<u8", count=2, offset=at))
            if nv == 0:
                return
            j += 1
            if hi_blk is not None and not (lo_blk <= j < hi_blk):                          # a stored block of another part: skipped unread
                if j >= hi_blk:
                    return
                need(8 * (nv + 1) + 4 * nv + nb + nv * rb)
                continue
            off = np.frombuffer(mm, dtype="<i8", count=nv + 1, offset=need(8 * (nv + 1)))
            counts = np.frombuffer(mm, dtype="<i4", count=nv, offset=need(4 * nv))
            at = need(nb)
            blob = np.frombuffer(mm, dtype=np.uint8, count=nb, offset=at) if raw else mm[at:at + nb]      # (raw: a view, the names are only read for printed rows)
            bits = np.frombuffer(mm, dtype=np.uint8, count=nv * rb, offset=need(nv * rb)).reshape(nv, rb)
            yield blob, off, counts, bits, j

    def merged():
        acc, rows = [], 0
        for blk in raw_blocks():
            acc.append(blk); rows += blk[2].shape[0]
            if rows >= block_size:
                yield acc
                acc, rows = [], 0
        if acc:
            yield acc

    use_dma = raw and device is not None and _route.route("dma", "1") != "0"
    windows = _DmaWindows(device) if use_dma else None

    def finished():                      # merging and a block's host preparation (6 ms per 262 144 rows) on the reader's thread, ahead of the engine
        try:
            for group in merged():
                if raw and len(group) == 1:
                    blob, off, counts, bits, j = group[0]
                    dma, release = windows.take(j) if windows is not None else (False, None)
                    yield _raw_block(blob, off, counts, bits, dma, release)
                    continue
                if len(group) == 1:
                    blob, off, counts, bits, _ = group[0]
                else:
                    blob = b"".join(bytes(g[0]) for g in group)
                    base = np.cumsum([0] + [len(g[0]) for g in group[:-1]])
                    off = np.concatenate([g[1][:-1] + b for g, b in zip(group, base)] + [np.array([len(blob)], dtype=np.int64)])
                    counts = np.concatenate([g[2] for g in group])
                    bits = np.concatenate([g[3] for g in group], axis=0)
                if raw:
                    yield _raw_block(blob, off, counts, bits)
                    continue
                yield _block_from_raw(n, samples, order, None, blob, off, bits, counts, min_af, max_af, want_patterns, want_samples)
        finally:
            if windows is not None:
                windows.reader_done()

    for blk in prefetched(finished(), depth=max(2, int(ahead))):
        yield blk
