"""`python -m pyseer_amd` -- pyseer's command line for the per-variant association tests, driving the HIP engine.

Mirrors pyseer/__main__.py for the options that reach the hot path: k-mer / Rtab input, fixed effects (SEER) with MDS of
a distance matrix or --no-distances, --lmm with a similarity matrix or --load-lmm cache, covariates, continuous
phenotypes, the AF / missing / p-value filters, --print-samples / --print-filtered / --output-patterns, and the four
end-of-run counters.  Variants are parsed in blocks of --block_size, packed, and tested block-wise on the GPU; rows are
printed in the reference's order (fixed effects: input order; LMM: within a block the filtered variants first, then the
tested ones -- fit_lmm returns them that way, lmm.py:160-224).

--lineage (MDS components or --lineage-clusters) runs fit_lineage_effect on the GPU as well (1 + lineages + covariates <= 16).

Not supported (out of scope): --vcf/--burden (pysam), --wg (enet/rf).
"""
import argparse
import os
import sys
import warnings

from . import _route


def _torch_reachable(argv):
    """torch is used by the command line for one thing: the eigendecomposition of a similarity matrix on the GPU (--lmm without
    --load-lmm / --cpu-eigh; lmm.spectral_decomposition).  Judged on the raw arguments (argparse accepts unique prefixes), because the
    device warm-up starts before the options are parsed; main() sets the same flag from the parsed options."""
    return any(a.startswith("--lm") for a in argv) and not any(a.startswith("--load-l") or a.startswith("--cpu-e") for a in argv)


def _first_device(argv):
    """The device the run will use first, from the raw arguments: `--gpu N`, `--gpu=N`, `--gpus a,b,...` / `--gpus=a,b` (first listed) or
    `--gpus N` (a count: device 0).  Anything unreadable warms device 0 -- a context on a device the run does not use is the cost."""
    dev = 0
    for i, a in enumerate(argv):
        key, _, val = a.partition("=")
        if key not in ("--gpu", "--gpus"):
            continue
        if not val and i + 1 < len(argv):
            val = argv[i + 1]
        try:
            if key == "--gpu":
                dev = int(val)
            else:
                dev = int(val.split(",")[0]) if "," in val else 0
        except ValueError:
            pass
    return dev


def _warm_device():
    """sh_warmup on a side thread: the HIP runtime's start (0.8 s at N = 5000's first context) runs while the interpreter imports
    numpy / pandas and the inputs are read.  Errors are left to the engine, which reports a missing library or device itself."""
    try:
        dev = _first_device(sys.argv)
        import ctypes
        from . import _abi
        # ctypes.CDLL() holds the GIL through dlopen (the HIP runtime and its dependencies: 0.4 s); dlopen called as a foreign function
        # does not, and makes the CDLL() that follows a look-up
        if not _abi.TORCH_FIRST:                           # (with torch in play its libraries must be loaded first: _abi.load does that)
            libc = ctypes.CDLL(None)
            libc.dlopen.restype = ctypes.c_void_p
            libc.dlopen.argtypes = [ctypes.c_char_p, ctypes.c_int]
            libc.dlopen(_abi.LIB_PATH.encode(), os.RTLD_NOW)
        lib = _abi.load()
        # a job's host threads wait for the device asleep (include/seerhip.h sh_set_wait_mode): the runtime's default spins a CPU per waiting stream
        lib.sh_set_wait_mode(0 if _route.route("wait") == "spin" else 1)
        lib.sh_warmup(dev)
    except Exception:
        pass


if __name__ == "__main__" and "--help" not in sys.argv and "-h" not in sys.argv:
    import atexit
    import threading
    from . import _abi as _abi0
    # written once, on this thread, before the warm-up thread exists and before anyone loads the library; main() reads it back and, should
    # the parsed options disagree with this reading of the raw arguments, decomposes on the CPU rather than import torch after libseerhip
    _abi0.TORCH_FIRST = _torch_reachable(sys.argv)
    _warm_thread = threading.Thread(target=_warm_device, daemon=True)
    _warm_thread.start()
    atexit.register(_warm_thread.join, 10.0)               # an early exit (bad options) must not tear the process down under a running dlopen

import numpy as np
import pandas as pd

from . import __version__
from .classes import Seer, LMM, FLAG_FILTER, FLAG_PREFILTER, notes_from_flags
from .input import (check_packed_cache, load_phenotypes, load_structure, load_covariates, load_lineage, open_variant_file,
                    iter_packed_blocks, iter_packed_blocks_native, iter_packed_blocks_native_multi, iter_packed_blocks_cached,
                    PackedCacheWriter, packed_cache_complete)
from .lmm import initialise_lmm, mask_like_fit_lmm
from .model import fit_null, covariate_block
from .utils import format_output


def get_options(argv=None):
    description = 'SEER (doi: 10.1038/ncomms12797), reimplemented in python -- MI355X engine (pyseer_amd)'
    parser = argparse.ArgumentParser(description=description, prog="pyseer_amd")
    ph = parser.add_argument_group('Phenotype')
    ph.add_argument('--phenotypes', required=True, help='Phenotypes file (whitespace separated)')
    ph.add_argument('--phenotype-column', default=None, help='Phenotype file column to use [Default: last column]')
    va = parser.add_argument_group('Variants')
    vg = va.add_mutually_exclusive_group(required=True)
    vg.add_argument('--kmers', default=None, nargs='+',
                    help='Kmers file; several files are tested as one stream in the order given and read concurrently (one gzip stream inflates serially)')
    vg.add_argument('--vcf', default=None, help='VCF file (not supported by this build)')
    vg.add_argument('--pres', default=None, help='Presence/absence .Rtab matrix as produced by roary and piggy')
    va.add_argument('--burden', help='(not supported by this build)')
    di = parser.add_argument_group('Distances')
    dg = di.add_mutually_exclusive_group()
    dg.add_argument('--distances', help='Strains distance square matrix (fixed or lineage effects)')
    dg.add_argument('--load-m', help='Load an existing matrix decomposition')
    sg = di.add_mutually_exclusive_group()
    sg.add_argument('--similarity', help='Strains similarity square matrix (for --lmm)')
    sg.add_argument('--load-lmm', help='Load an existing lmm cache')
    di.add_argument('--save-m', help='Prefix for saving matrix decomposition')
    di.add_argument('--save-lmm', help='Prefix for saving LMM cache')
    di.add_argument('--mds', default="classic", help='Type of multidimensional scaling [Default: classic]')
    di.add_argument('--max-dimensions', type=int, default=10, help='Maximum number of dimensions to consider after MDS')
    di.add_argument('--no-distances', action='store_true', default=False, help='Allow run without a distance matrix')
    asc = parser.add_argument_group('Association options')
    asc.add_argument('--continuous', action='store_true', default=False, help='Force continuous phenotype')
    asc.add_argument('--lmm', action='store_true', default=False, help='Use random instead of fixed effects')
    asc.add_argument('--wg', default=None, help='(not supported by this build)')
    asc.add_argument('--lineage', action='store_true', help='Report lineage effects')
    asc.add_argument('--lineage-clusters', help='Custom clusters to use as lineages [Default: MDS components]')
    asc.add_argument('--lineage-file', default="lineage_effects.txt", help='File to write lineage association to')
    fi = parser.add_argument_group('Filtering options')
    fi.add_argument('--min-af', type=float, default=0.01, help='Minimum AF [Default: 0.01]')
    fi.add_argument('--max-af', type=float, default=0.99, help='Maximum AF [Default: 0.99]')
    fi.add_argument('--max-missing', type=float, default=0.05, help='Maximum missing (vcf/Rtab) [Default: 0.05]')
    fi.add_argument('--filter-pvalue', type=float, default=1, help='Prefiltering t-test pvalue threshold [Default: 1]')
    fi.add_argument('--lrt-pvalue', type=float, default=1, help='Likelihood ratio test pvalue threshold [Default: 1]')
    co = parser.add_argument_group('Covariates')
    co.add_argument('--covariates', default=None, help='User-defined covariates file')
    co.add_argument('--use-covariates', default=None, nargs='*', help="Covariates to use; 'q' suffix = quantitative")
    ot = parser.add_argument_group('Other')
    ot.add_argument('--print-samples', action='store_true', default=False, help='Print sample lists [Default: hide]')
    ot.add_argument('--print-filtered', action='store_true', default=False, help='Print filtered variants [Default: hide]')
    ot.add_argument('--output-patterns', default=False, help='File to print patterns to, useful for finding pvalue threshold')
    ot.add_argument('--uncompressed', action='store_true', default=False, help='Uncompressed kmers file [Default: gzipped]')
    ot.add_argument('--cpu', type=int, default=1, help='Accepted for compatibility; the tests run on the GPU')
    ot.add_argument('--block_size', type=int, default=3000, help='Number of variants parsed and sent to the GPU at a time')
    ot.add_argument('--gpu', type=int, default=0, help='GPU index [Default: 0]')
    ot.add_argument('--gpus', default=None, help='Comma-separated GPU indices, or a count N (= 0..N-1): the variant stream is sharded across them. '
                    'From a packed cache every GPU tests its own contiguous range of the rows through its own pipeline; other input is one stream '
                    'whose blocks go to the GPUs in turn. Output keeps the input order [Default: the single --gpu]')
    ot.add_argument('--save-packed', default=None,
                    help='Also write the parsed k-mer file as packed bit rows (for --load-packed in later runs over the same samples)')
    ot.add_argument('--packed-cache', action='store_true', default=False,
                    help='Keep a packed cache next to the k-mer file (<kmers>.seerpack): written by the first run, read by later runs '
                         'over the same samples while the k-mer file is unchanged (size and modification time); ignored with a warning otherwise')
    ot.add_argument('--load-packed', default=None,
                    help='Read variants from a packed cache written by --save-packed instead of parsing --kmers again')
    ot.add_argument('--cpu-eigh', action='store_true', default=False,
                    help='Decompose the similarity matrix with numpy on the host instead of rocSOLVER on the GPU')
    ot.add_argument('--no-dedup', action='store_true', default=False,
                    help='Test every variant separately [Default: each distinct presence pattern of a block is tested once]')
    ot.add_argument('--python-reader', action='store_true', default=False,
                    help='Parse k-mer files with the Python reader instead of the native C++ one')
    ot.add_argument('--python-sink', action='store_true', default=False,
                    help='Format every output row in Python (one result tuple per variant) instead of the native block sink')
    ot.add_argument('--serial-sink', action='store_true', default=False,
                    help='Format and write each block before the next one is tested [Default: on a worker thread, while the next block is on the GPU]')
    ot.add_argument('--packed-part', default=None, metavar='I/N',
                    help='With --load-packed: run only range I of N contiguous ranges of the cache (one process per device: run N processes, '
                         'each with its own --gpu, and concatenate their outputs in order)')
    ot.add_argument('--lmm-lineage-per-variant', action='store_true', default=False,
                    help='With --lmm --lineage, fit the lineage effect of each variant itself. [Default: reproduce the reference, '
                         'which fits the LAST variant of each block for every variant of that block]')
    ot.add_argument('--version', action='version', version='%(prog)s ' + __version__)
    return parser.parse_args(argv)


def _die(msg):
    sys.stderr.write(msg)
    sys.exit(1)


from .model import host_pre_filtering as _host_prefilter


def main(argv=None):
    options = get_options(argv)
    if __name__ == "__main__":                             # as a program (a caller of main() keeps the library's default: torch first)
        from . import _abi as _abi_mod
        if bool(options.lmm and not options.load_lmm and not options.cpu_eigh) and not _abi_mod.TORCH_FIRST:
            # (the raw-argument reading missed that this run decomposes a kinship matrix: libseerhip may already be loaded without torch's
            # HIP runtime underneath it, and importing torch now would break the device for both)
            sys.stderr.write("pyseer_amd: kinship decomposition on the CPU (numpy) for this run\n")
            options.cpu_eigh = True
    if options.vcf or options.burden:
        _die('VCF / burden input needs pysam and is not supported by pyseer_amd\n')
    if options.wg:
        _die('Whole-genome models (--wg) are out of scope for pyseer_amd\n')
    if options.max_dimensions < 1:
        _die('Minimum number of dimensions after MDS is 1\n')
    if options.lmm and not options.similarity and not options.load_lmm:
        _die('Must provide a similarity matrix or lmm cache for random effects\n')
    if not options.no_distances:
        if (options.lmm and (options.distances or options.load_m) and not options.lineage) or (not options.lmm and (options.similarity or options.load_lmm)):
            sys.stderr.write('Must use distance matrix with fixed effects, or similarity matrix with random effects\n')
            _die('Unless performing a lineage analysis with random effects\n')
        if options.lmm and not (options.distances or options.load_m) and options.lineage:
            _die('Must also provide a distance matrix to report lineage effects\n')
        if not options.lmm and not options.distances and not options.load_m:
            _die('Option --no-distances must be used when no distance matrix is provided\n')
    else:
        if options.distances or options.load_m:
            _die('Cannot use --no-distances with --distances or --load-m\n')
        if not options.lmm and not options.lineage_clusters and options.lineage:
            _die('Must provide a lineage clusters file when --no-distances and --lineage are used together in fixed-effects mode\n')
        if options.lmm:
            _die('Cannot use --no-distances with --lmm\n')
    if options.block_size < 1:
        _die('Block size must be at least 1\n')
    warnings.filterwarnings('ignore')

    p = load_phenotypes(options.phenotypes, options.phenotype_column)
    sys.stderr.write("Read " + str(len(p)) + " phenotypes\n")
    if not options.continuous:
        if p.values[(p.values != 0) & (p.values != 1)].size > 0:
            options.continuous = True
            sys.stderr.write("Detected continuous phenotype\n")
        else:
            sys.stderr.write("Detected binary phenotype\n")

    if options.covariates is not None:
        cov = load_covariates(options.covariates, options.use_covariates, p)
        if cov is None:
            sys.exit(1)
    else:
        cov = pd.DataFrame([])

    m = np.empty(shape=(0, 0))
    null_fit = firth_null = None
    if (options.lineage and not options.lineage_clusters) or not options.lmm:
        if not options.no_distances:
            if options.load_m and os.path.isfile(options.load_m):
                mdf = pd.read_pickle(options.load_m)
                sys.stderr.write("Loaded projection with dimension " + str(mdf.shape) + "\n")
            else:
                seed = os.environ.get('PYSEERSEED', None)
                mdf = load_structure(options.distances, p, options.max_dimensions, options.mds, options.cpu,
                                     int(seed) if seed is not None else None)
                if options.save_m:
                    mdf.to_pickle(options.save_m + ".pkl")
            if options.max_dimensions > mdf.shape[1]:
                sys.stderr.write('Population MDS scaling restricted to %d dimensions instead of requested %d\n' %
                                 (mdf.shape[1], options.max_dimensions))
                options.max_dimensions = mdf.shape[1]
            inter = p.index.intersection(mdf.index)
            sys.stderr.write("Analysing " + str(len(inter)) + " samples found in both phenotype and structure matrix\n")
            p = p.loc[inter]
            m = mdf.loc[p.index].values[:, :options.max_dimensions]
        if cov.shape[1] > 0:
            cov = cov.loc[p.index]
        null_fit = fit_null(p.values, m, cov, options.continuous)
        firth_null = fit_null(p.values, m, cov, options.continuous, True) if not (options.continuous or options.lmm) else True
        if null_fit is None or firth_null is None:
            _die('Could not fit null model, exiting\n')

    # lineage effects (pyseer/__main__.py:388-444)
    lineage_clusters = None
    lineage_dict = None
    if options.lineage_clusters:
        lineage_clusters, lineage_dict = load_lineage(options.lineage_clusters, p)
    if options.lineage:
        from scipy.stats import norm
        lineage_wald = {}
        if options.lineage_clusters:
            # one-hot clusters are not full rank: drop the cluster least associated with the phenotype
            for lineage, design in zip(lineage_dict, lineage_clusters.T):
                lf = fit_null(p.values, design.reshape(-1, 1).astype(float), cov, options.continuous)
                if lf is None:
                    _die('Could not fit lineage null model, exiting\n')
                lineage_wald[lineage] = np.absolute(lf.params[1]) / lf.bse[1]
            min_lineage = min(lineage_wald.items(), key=lambda kv: kv[1])[0]
            drop = lineage_dict.index(min_lineage)
            lineage_clusters = np.delete(lineage_clusters, drop, 1)
            del lineage_dict[drop]
        else:
            lineage_dict = ["MDS" + str(i + 1) for i in range(options.max_dimensions)]
            lineage_clusters = m
            for lineage, slope, se in zip(lineage_dict, null_fit.params[1:], null_fit.bse[1:]):
                lineage_wald[lineage] = np.absolute(slope) / se
        sys.stderr.write('Writing lineage effects to %s\n' % options.lineage_file)
        with open(options.lineage_file, 'w') as lineage_out:
            lineage_out.write("\t".join(["lineage", "wald_test", "p-value"]) + "\n")
            for lineage, wald in sorted(lineage_wald.items(), key=lambda kv: kv[1], reverse=True):
                lineage_out.write("\t".join([lineage, str(wald), str(2 * (1 - norm.cdf(wald)))]) + "\n")

    from .engine import Engine

    def make_engines(n):
        """One context per listed device; the same device may be listed twice (two contexts on one GPU: the test of
        the multi-device job path on a one-GPU box)."""
        wt = globals().get("_warm_thread")
        if wt is not None:
            wt.join()                                      # (the device's wait mode is chosen at its first use: by the warm-up, not by a context racing it)
        from . import _abi as _abi_w
        _abi_w.load().sh_set_wait_mode(0 if _route.route("wait") == "spin" else 1)
        if options.gpus is None:
            return [Engine(n, device=options.gpu)]
        devs = [int(x) for x in options.gpus.split(",")] if "," in options.gpus else list(range(int(options.gpus)))
        return [Engine(n, device=d) for d in devs]
    import time as _time0
    t_inputs = _time0.time()
    since_start = float("nan")
    try:                                                   # seconds since this process started (field 22 of /proc/self/stat against the uptime)
        with open("/proc/self/stat") as fh:
            st_ = fh.read()
        with open("/proc/uptime") as fh:
            since_start = float(fh.read().split()[0]) - int(st_[st_.rindex(")") + 2:].split()[19]) / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        pass
    # One PROCESS per device for a `--gpus` job over a packed cache (round 6; pyseer_amd/_packed_child.py): this process does the host set-up
    # and joins the parts, every device's process runs the library's own block loop over its range of the cache.  SEERHIP_ROUTE procs=0: the
    # devices' streams as threads of this process (round 5).
    devs_ = None
    if options.gpus is not None:
        devs_ = [int(x) for x in options.gpus.split(",")] if "," in options.gpus else list(range(int(options.gpus)))
    proc_mode = bool(devs_ and len(devs_) > 1 and options.load_packed and not options.save_packed and not options.python_sink and not options.serial_sink
                     and not options.packed_part and _route.route("job", "1") not in ("0", "py") and _route.route("procs", "1") != "0")
    if proc_mode:
        try:
            sys.stdout.fileno()
        except Exception:
            proc_mode = False
    if options.lmm:
        sys.stderr.write("Setting up LMM\n")
        # lineage_samples = p.index as the reference passes it (__main__.py:403, 455-456): a similarity matrix over another set of
        # samples is refused here, before p is cut down to it and the lineage design goes out of step with the phenotype
        p, lmm, h2 = initialise_lmm(p, cov, options.similarity, options.load_lmm, options.save_lmm,
                                    lineage_samples=(p.index if options.lineage else None),
                                    use_gpu=not options.cpu_eigh, device=options.gpu)
        sys.stderr.write("h^2 = " + '{0:.2f}'.format(h2) + "\n")
        engs = [] if proc_mode else make_engines(len(p))
        # the first context builds the per-run state (M = U~ diag(1/Sd) U~^T, limbs, tables); the others receive it device to device
        # (sh_lmm_share: 94 MB at N = 5000) instead of each re-deriving it from the host copy of U
        if engs:
            engs[0].lmm_setup(lmm.U, lmm.S, lmm.Y, lmm.X, h2, options.continuous, options.filter_pvalue, options.lrt_pvalue)
        for e_ in engs[1:]:
            e_.lmm_share_from(engs[0])
    else:
        engs = [] if proc_mode else make_engines(len(p))
        for e_ in engs:
            e_.glm_setup(p.values, covariate_block(len(p), m, cov), options.continuous,
                         np.nan if options.continuous else null_fit.llf, None if options.continuous else firth_null,
                         options.filter_pvalue, options.lrt_pvalue)

    t_setup = _time0.time()
    for e_ in engs:
        e_.set_dedup(not options.no_dedup)
        if options.lineage:
            e_.lineage_setup(np.asarray(lineage_clusters, dtype=float), cov.values if cov.shape[1] > 0 else None)
    if not options.lineage:
        lineage_dict = None

    all_strains = set(p.index)
    kmer_files = list(options.kmers) if options.kmers else []
    if options.load_packed and options.save_packed and os.path.realpath(options.load_packed) == os.path.realpath(options.save_packed):
        # (the cache is memory-mapped while it is read: rewriting it under the mapping would end the run with SIGBUS)
        _die('--save-packed and --load-packed name the same file\n')
    if len(kmer_files) > 1 and (options.python_reader or options.load_packed or options.save_packed or options.packed_cache):
        _die('Several --kmers files need the native reader and cannot be combined with a packed cache\n')
    var_type, var_file = ("kmers", kmer_files[0]) if kmer_files else ("Rtab", options.pres)
    native = (var_type == "kmers") and not options.python_reader
    if native or options.load_packed:
        # blocks of the native reader / the packed cache hand every parsed row to the engine (input.py _block_from_raw): the AF window is applied
        # on the device as well, so filtered rows are neither copied out on the host nor contracted on the GPU
        for e_ in engs:
            e_.set_af_filter(options.min_af, options.max_af)
    if native and not options.uncompressed and not options.load_packed:
        for vf in (kmer_files or [var_file]):
            with open(vf, "rb") as fh:                     # the reference's gzip.open raises on plain text (input.py:271-276)
                if fh.read(2) != b"\x1f\x8b":
                    sys.stderr.write("Not a gzipped file (%s): use --uncompressed for plain-text k-mers\n" % vf)
                    sys.exit(1)
    if not native:
        infile, sample_order = open_variant_file(var_type, var_file, None, None, options.uncompressed)
    patterns = open(options.output_patterns, 'wb') if options.output_patterns else None

    header = ['variant', 'af', 'filter-pvalue', 'lrt-pvalue', 'beta', 'beta-std-err']
    if not options.lmm:
        header.append('intercept')
        if not options.no_distances:
            header += ['PC%d' % i for i in range(1, options.max_dimensions + 1)]
        if options.covariates is not None:
            header += [x for x in cov.columns]
    else:
        header.append('variant_h2')
    if options.lineage:
        header.append('lineage')
    if options.print_samples:
        header += ['k-samples', 'nk-samples']
    header.append('notes')
    print('\t'.join(header))

    model = 'lmm' if options.lmm else 'seer'
    pv = p.values.astype(float)
    nan = np.nan

    cache_out, cache_stamp = None, None
    if options.packed_cache and native and not options.load_packed and not options.save_packed:
        # the automatic form of --save-packed / --load-packed: one file next to the input, tied to it by size + mtime (a sidecar stamp),
        # and to the run by the sample list the cache itself stores
        import os as _os
        side = var_file + ".seerpack"
        st = _os.stat(var_file)
        stamp = "%d %d" % (st.st_size, int(st.st_mtime))
        fresh = False
        try:
            fresh = _os.path.exists(side) and open(side + ".stamp").read().strip() == stamp
        except (IOError, OSError):
            fresh = False
        if fresh and not packed_cache_complete(side):
            sys.stderr.write("Packed cache %s is incomplete (an interrupted run); parsing %s again\n" % (side, var_file))
            fresh = False
        if fresh:
            try:
                next(iter(iter_packed_blocks_cached(p, side, options.min_af, options.max_af, 1)), None)    # header + sample list check
                options.load_packed = side
            except ValueError:
                sys.stderr.write("Packed cache %s was written for other samples; parsing %s again\n" % (side, var_file))
                fresh = False
            except IOError as ex:
                sys.stderr.write("Packed cache %s cannot be used (%s); parsing %s again\n" % (side, ex, var_file))
                fresh = False
        if not fresh:
            # the cache is written to a temporary name and renamed when the run has read the whole input; the stamp follows the rename
            # (PackedCacheWriter.close), so an interrupted run leaves neither
            if _os.access(_os.path.dirname(_os.path.abspath(side)) or ".", _os.W_OK):
                options.save_packed = side
                cache_stamp = (side + ".stamp", stamp)
            else:
                sys.stderr.write("Cannot write a packed cache next to %s; continuing without\n" % var_file)
    # ---- the job stream (round 5; include/seerhip.h sh_job_*): unless a cross-check sink is asked for (round 6: --lineage, --output-patterns and --print-samples
    # run inside the stream too: sh_job_set_lineage / _patterns / _samples) a block goes to the library as parsed and comes back as the text of its printed rows: the AF
    # window, the NaN masks, the counters and the choice of rows run on the device, the host formats printed rows only.  Output without
    # --print-filtered does not depend on where blocks end, so short blocks are coalesced (a 3000-row block is 90 us of GPU time).
    job_path = ((native or bool(options.load_packed)) and not options.python_sink and not options.serial_sink and _route.route("job", "1") != "0")
    # (--lineage, round 6: fit_lineage_effect runs inside the stream, for printed rows -- sh_job_set_lineage.  The LMM's lineage is that of each
    # block's LAST variant (pyseer/lmm.py:209-213, the stale `k`), so its blocks must end where the reference's do: no coalescing)
    lmm_block_lineage = bool(options.lineage and options.lmm and not options.lmm_lineage_per_variant)
    # (2^18 rows: a call of 2^16 runs the LMM 7 % slower, the logistic path 27 %, Firth 15 % -- profiles/r06/rows_per_call.txt; round 5 had 2^16)
    job_block = options.block_size if (options.print_filtered or lmm_block_lineage or not job_path) else max(options.block_size, 1 << 18)
    # the reader runs as far ahead as the job stream holds blocks in flight (fixed effects: 2 + lanes, include/seerhip.h sh_job_depth)
    job_ahead = (2 + engs[0].get_lanes()) if (job_path and not options.lmm and engs) else 2
    if options.load_packed:
        blocks = iter_packed_blocks_cached(p, options.load_packed, options.min_af, options.max_af, job_block,
                                           want_patterns=bool(options.output_patterns), want_samples=options.print_samples,
                                           raw=job_path, device=(engs[0].device if (job_path and engs) else None), ahead=job_ahead)
    elif native and len(kmer_files) > 1:
        blocks = iter_packed_blocks_native_multi(p, kmer_files, options.min_af, options.max_af, job_block,
                                                 want_patterns=bool(options.output_patterns), want_samples=options.print_samples, raw=job_path)
    elif native:
        if options.save_packed:
            cache_out = PackedCacheWriter(options.save_packed, [str(x) for x in p.index], stamp=cache_stamp)
        blocks = iter_packed_blocks_native(p, var_file, options.min_af, options.max_af, job_block,
                                           want_patterns=bool(options.output_patterns), want_samples=options.print_samples,
                                           save_to=cache_out, raw=job_path)
    else:
        blocks = iter_packed_blocks(p, var_type, infile, all_strains, sample_order, options.min_af, options.max_af,
                                    options.max_missing, options.uncompressed, options.block_size)
    # ---- block sink (csrc/writer.cpp): results stay arrays from the engine to the TSV text; the per-variant tuple path below
    # is kept for --print-samples, for variants carrying missing calls, and as the cross-check (--python-sink)
    from .sink import RowFormatter, names_blob
    from .packing import pack_variants
    NOTE_AF, NOTE_FIRTH_FAIL = 1, 1 << 6
    q_out = 0

    import collections
    import time as _time
    cli_timing = _route.debug("cli")

    def json_load(path_):
        import json as _json_
        with open(path_) as fh_:
            return _json_.load(fh_)

    def new_tm():
        return {"engine": 0.0, "sink": 0.0, "write": 0.0, "blocks": 0, "t0": _time.perf_counter(), "reader": 0.0, "queue": 0.0, "format": 0.0,
                "t0w": _time.time(), "rows": 0}

    def run_stream_job(engs, blocks, write_text, tm, stop=None, write_patterns=None):
        """One stream of RawBlocks through the library's job stream, block k on context k % len(engs) (each context pipelined Job.depth deep:
        the rows of its next block cross PCIe while the block before runs its kernels and the one before that is written out here).
        Returns (pre-filtered, tested, printed)."""
        from .engine import Job
        jobs = [Job(e_, options.lmm, options.print_filtered, lineage_labels=(lineage_dict if options.lineage else None),
                    lineage_per_variant=options.lmm_lineage_per_variant, patterns=write_patterns is not None,
                    sample_names=([str(x) for x in p.index] if options.print_samples else None)) for e_ in engs]
        prefilter = tested = printed = 0
        order = collections.deque()                           # the job each block in flight went to, in input order

        def take():
            nonlocal prefilter, tested, printed
            jb = order.popleft()
            t_c = _time.perf_counter()
            text, cnt, release = jb.collect()
            tm["engine"] += _time.perf_counter() - t_c
            prefilter += cnt[0]; tested += cnt[1]; printed += cnt[2]
            if write_patterns is not None and cnt[1]:
                write_patterns(jb.patterns())                 # (hash_pattern of the block's tested variants, made on the device)
            if len(text):
                t_w = _time.perf_counter()
                write_text(text)
                tm["write"] += _time.perf_counter() - t_w
            if release is not None:
                release()
        try:
            blocks = iter(blocks)
            k = 0
            while True:
                t_r = _time.perf_counter()
                rb = next(blocks, None)                       # (waiting here = the reader is the slowest stage)
                tm["reader"] += _time.perf_counter() - t_r
                if rb is None or (stop is not None and stop.is_set()):
                    break
                jb = jobs[k % len(jobs)]
                while jb.pending() >= jb.depth:
                    take()
                t_e = _time.perf_counter()
                jb.submit(rb.bits, rb.counts, rb.blob, rb.off, rows_are_dma=rb.dma, keep=rb.release)
                tm["engine"] += _time.perf_counter() - t_e; tm["blocks"] += 1
                order.append(jb)
                # depth - 1 blocks stay in flight per context (one uploading, one computing per lane); the next is collected, formatted and written
                while len(order) > (jobs[0].depth - 1) * len(jobs):
                    take()
                k += 1
            while order:
                take()
        finally:
            for jb in jobs:
                jb.close()
        tm["rows"] = prefilter + tested; tm["loop"] = _time.perf_counter() - tm["t0"]; tm["overlap"] = True; tm["job"] = True
        return prefilter, tested, printed

    def run_stream_packed(eng_, part, out_file, pat_file, tm, stop_c=None):
        """One stream of a packed cache entirely inside the library (round 6: sh_job_run_packed -- csrc/job_run.inc): the block loop, the DMA
        windows, the writes to the two descriptors; this thread holds no interpreter lock while it runs."""
        from .engine import Job
        sys.stdout.flush()                                     # (the header line sits in the text layer's buffer)
        out_file.flush()
        if pat_file is not None:
            pat_file.flush()
        job = Job(eng_, options.lmm, options.print_filtered, lineage_labels=(lineage_dict if options.lineage else None),
                  lineage_per_variant=options.lmm_lineage_per_variant, patterns=pat_file is not None,
                  sample_names=([str(x) for x in p.index] if options.print_samples else None))
        try:
            pf_, te_, pr_, nb_ = job.run_packed(options.load_packed, part, job_block, use_dma=_route.route("dma", "1") != "0", out_fd=out_file.fileno(),
                                                pat_fd=(pat_file.fileno() if pat_file is not None else -1), stop=stop_c)
        finally:
            job.close()
        tm["blocks"] = nb_; tm["rows"] = pf_ + te_; tm["loop"] = _time.perf_counter() - tm["t0"]; tm["overlap"] = True; tm["job"] = True
        tm["engine"] = tm["loop"]
        return pf_, te_, pr_

    def has_fd(f):
        try:
            f.fileno()
            return True
        except Exception:
            return False

    def run_stream(engs, blocks, write_text, write_patterns, tm):
        """One stream of blocks, in order: reader (its own thread inside `blocks`) -> engine calls, block k on engs[k % len(engs)], each engine
        pipelined -> one ordered sink writing through write_text / write_patterns.  Returns (pre-filtered, tested, printed).  The job runs one
        such stream per device when the input can be cut into ranges (a packed cache), else one stream over all devices."""
        prefilter = tested = printed = 0
        eng = engs[0]
        formatter = RowFormatter(lineage_dict if options.lineage else None)     # (its text buffer is valid until its next call: one per stream)

        class out(object):                                   # the tuple path writes text; the streams carry bytes
            @staticmethod
            def write(s):
                write_text(s.encode())

        class _Pat(object):
            @staticmethod
            def write(b):
                write_patterns(b)
        patterns = _Pat if write_patterns is not None else None

        def emit(x):
            nonlocal prefilter, tested, printed
            if x.prefilter:
                prefilter += 1
                if options.print_filtered:
                    printed += 1
                    out.write(format_output(x, lineage_dict, model, options.print_samples) + "\n")
                return
            tested += 1
            if patterns is not None:
                patterns.write(x.pattern)
            if x.filter and not options.print_filtered:
                return
            printed += 1
            out.write(format_output(x, lineage_dict, model, options.print_samples) + "\n")

        def sink_block(blk, r):
            nonlocal prefilter, tested, printed
            t_in = _time.perf_counter()
            nb = len(blk)
            status = np.asarray(blk.status)
            on = status == 0
            row_of = np.asarray(blk.row_of, dtype=np.int64)
            # blocks of the native reader / the packed cache: one engine row per variant, in place (row_of = 0, 1, 2, ...): whole-array selects
            # instead of gathers through an index
            ident = r is not None and isinstance(blk.row_of, np.ndarray) and nb == r["flags"].shape[0]
            j = None if ident else row_of[on]
            flags = np.zeros(nb, dtype=np.uint32)
            flags[status == 1] = NOTE_AF | FLAG_PREFILTER
            keys = ("prep", "pvalue", "beta", "bse", "frac_h2") if options.lmm else ("prep", "pvalue", "kbeta", "bse", "intercept")
            cols = [np.asarray(blk.afs, dtype=np.float64)]
            for kname in keys:
                if ident:
                    c = np.where(on, r[kname], np.nan)
                else:
                    c = np.full(nb, np.nan)
                    if r is not None:
                        c[on] = r[kname][j]
                cols.append(c)
            betas = valid = None
            if r is not None:
                if ident:
                    flags = np.where(on, r["flags"], flags)
                else:
                    flags[on] = r["flags"][j]
                if not options.lmm and r["betas"].shape[1]:
                    if ident:
                        betas = r["betas"]                    # (rows that are not `on` have valid = 0: the formatter never reads their slopes)
                        valid = (on & (np.isfinite(r["kbeta"]) | np.isfinite(r["pvalue"]))).astype(np.uint8)
                    else:
                        betas = np.full((nb, r["betas"].shape[1]), np.nan)
                        betas[on] = r["betas"][j]
                        valid = np.zeros(nb, dtype=np.uint8)
                        valid[on] = np.isfinite(r["kbeta"][j]) | np.isfinite(r["pvalue"][j])
            pf = (flags & FLAG_PREFILTER) != 0
            ft = (flags & FLAG_FILTER) != 0
            lineage = None
            if options.lineage:
                lineage = np.full(nb, -1, dtype=np.int32)
                if options.lmm and not options.lmm_lineage_per_variant:
                    kl = blk.last_k                      # pyseer/lmm.py:209-213: the stale `k` of fit_lmm's first loop
                    ml = -1
                    if kl is not None and not np.isnan(np.asarray(kl, dtype=float)).any():
                        ml = int(eng.lineage_batch(pack_variants(np.asarray(kl).reshape(1, -1)))[0])
                    lineage[~pf & ~ft] = ml
                else:
                    need = on & ~pf & (~ft if options.lmm else ((flags & NOTE_FIRTH_FAIL) == 0))
                    if need.any():
                        lineage[need] = eng.lineage_batch(blk.bits[row_of[need]])
            order = np.arange(nb)
            if options.lmm:                                   # fit_lmm's return order: filtered first, then tested
                order = np.concatenate([order[pf], order[~pf]])
            npf = int(pf.sum())
            prefilter += npf
            tested += nb - npf
            if patterns is not None:
                patterns.write(b''.join(blk.patterns[i] for i in order if not pf[i]))
            show = np.ones(nb, dtype=bool) if options.print_filtered else (~pf & ~ft)
            sel = order[show[order]]
            printed += int(sel.shape[0])
            if sel.shape[0]:
                blob, off = (blk.names_blob, blk.name_off) if getattr(blk, "names_blob", None) is not None else names_blob(blk.names)
                t_f = _time.perf_counter()
                text = formatter.format_view(blob, off, sel, cols, flags, betas, valid, lineage)
                t_w = _time.perf_counter()
                tm["format"] += t_w - t_f
                write_text(text)
                tm["write"] += _time.perf_counter() - t_w
            tm["sink"] += _time.perf_counter() - t_in

        # Three stages run at once (round 3): the block reader (its own thread, `prefetched`), the engine call of block k on this thread (a ctypes
        # call: the GIL is released while the rows go to the GPU and the statistics come back), and the sink of block k-1 on a worker thread
        # (NaN masking, counters, native formatter, the write to stdout -- the reference does all of this per variant between two fits,
        # pyseer/__main__.py:805-827).  One worker and a FIFO of depth 2: the output order is the input order.  The tuple path below (missing
        # calls, --print-samples, --python-sink) and the lineage fits (a second engine call on the same context) stay on this thread, after the
        # worker has drained.
        import queue
        import threading
        sink_q = queue.Queue(maxsize=2)
        sink_err = []
        overlap = not options.lineage and not options.python_sink and not options.print_samples and not options.serial_sink

        def sink_worker():
            while True:
                item = sink_q.get()
                try:
                    if item is None:
                        return
                    if not sink_err:
                        blk_, r_ = item
                        sink_block(blk_, mask_like_fit_lmm(r_) if (options.lmm and r_ is not None) else r_)
                except BaseException as ex:          # re-raised on the main thread
                    sink_err.append(ex)
                finally:
                    sink_q.task_done()
        sink_thread = None
        if overlap:
            sink_thread = threading.Thread(target=sink_worker, daemon=True)
            sink_thread.start()

        def drain_sink():
            if sink_thread is not None:
                sink_q.join()
            if sink_err:
                raise sink_err[0]

        def after_engine(blk, r):
            """Everything that follows a block's engine call: hand it to the sink (array-backed), or build the reference's row objects."""
            if not options.python_sink and not options.print_samples and 2 not in blk.status:
                if overlap:
                    if sink_err:
                        raise sink_err[0]
                    t_q = _time.perf_counter()
                    sink_q.put((blk, r))                      # (waiting here = the sink is the slowest stage)
                    tm["queue"] += _time.perf_counter() - t_q
                else:
                    sink_block(blk, mask_like_fit_lmm(r) if (options.lmm and r is not None) else r)
                return
            drain_sink()
            if options.lmm and r is not None:
                r = mask_like_fit_lmm(r)
            rows = []
            for i, name in enumerate(blk.names):
                st, af, ks, nks = blk.status[i], blk.afs[i], blk.kstrains[i], blk.nkstrains[i]
                if st == 1:                                   # AF / missing filtered (model.py:255-260, lmm.py:160-167)
                    if options.lmm:
                        rows.append(LMM(name, None, af, nan, nan, nan, nan, nan, None, ks, nks, {'af-filter'}, True, False))
                    else:
                        rows.append(Seer(name, blk.patterns[i], af, nan, nan, nan, nan, nan, np.array([]), None, ks, nks,
                                         {'af-filter'}, True, False))
                    continue
                if st == 2:                                   # missing calls inside k: the reference's error path
                    prep, bad = _host_prefilter(pv, blk.ks[i], options.continuous)
                    notes = {'bad-chisq'} if bad else set()
                    thr = options.filter_pvalue
                    failed = (prep >= thr) if options.lmm else (prep > thr)
                    if failed or not np.isfinite(prep):
                        notes.add('pre-filtering-failed')
                        x = (LMM(name, blk.patterns[i], af, prep, nan, nan, nan, nan, None, ks, nks, notes, True, False)
                             if options.lmm else
                             Seer(name, blk.patterns[i], af, prep, nan, nan, nan, nan, np.array([]), None, ks, nks, notes, True, False))
                    elif options.lmm:                         # NaN propagates through fit_lmm_block -> lrt-filtering-failed
                        notes.add('lrt-filtering-failed')
                        x = LMM(name, blk.patterns[i], af, prep, nan, nan, nan, nan, None, ks, nks, notes, False, True)
                    else:                                     # statsmodels MissingDataError, model.py:371-377
                        notes.add('missing-data-error')
                        x = Seer(name, blk.patterns[i], af, prep, nan, nan, nan, nan, np.array([]), None, ks, nks, notes, False, True)
                    rows.append(x)
                    continue
                j = blk.row_of[i]
                fl = int(r["flags"][j])
                notes = notes_from_flags(fl)
                pf, ft = bool(fl & FLAG_PREFILTER), bool(fl & FLAG_FILTER)
                if options.lmm:
                    rows.append(LMM(name, blk.patterns[i], af, r["prep"][j], r["pvalue"][j], r["beta"][j], r["bse"][j],
                                    r["frac_h2"][j], None, ks, nks, notes, pf, ft))
                else:
                    tested_ok = np.isfinite(r["kbeta"][j]) or np.isfinite(r["pvalue"][j])
                    betas = r["betas"][j] if (tested_ok and r["betas"].shape[1]) else np.array([])
                    rows.append(Seer(name, blk.patterns[i], af, r["prep"][j], r["pvalue"][j], r["kbeta"][j], r["bse"][j],
                                     r["intercept"][j], betas, None, ks, nks, notes, pf, ft))
            if options.lineage:
                # fit_lineage_effect: fixed effects -> every variant that reached the fit (model.py:379-382; firth-fail and
                # missing-data return earlier); LMM -> only variants that pass the LRT filter (lmm.py:209-213)
                need = [i for i, x in enumerate(rows) if blk.status[i] == 0 and not x.prefilter and
                        ((not x.filter) if options.lmm else ('firth-fail' not in x.notes))]
                if options.lmm and not options.lmm_lineage_per_variant:
                    # Reference behaviour (pyseer/lmm.py:209-213): inside fit_lmm's second loop `k` still holds the LAST variant
                    # unpacked by the first loop, so every passing variant of a block reports the lineage of that last variant.
                    kl = blk.last_k
                    ml = -1
                    if kl is not None and not np.isnan(np.asarray(kl, dtype=float)).any():
                        from .packing import pack_variants
                        ml = int(eng.lineage_batch(pack_variants(np.asarray(kl).reshape(1, -1)))[0])
                    for i, x in enumerate(rows):
                        if not x.prefilter and not x.filter:
                            rows[i] = x._replace(max_lineage=(None if ml < 0 else ml))
                elif need:
                    ml = eng.lineage_batch(blk.bits[[blk.row_of[i] for i in need]])
                    for i, v in zip(need, ml):
                        rows[i] = rows[i]._replace(max_lineage=(None if v < 0 else int(v)))
            if options.lmm:                                   # fit_lmm's return order: filtered first, then tested
                rows = [x for x in rows if x.prefilter] + [x for x in rows if not x.prefilter]
            for x in rows:
                emit(x)

        # The engine calls themselves are pipelined (sh_*_batch_async, include/seerhip.h): a call returns while its last chunk is still on the
        # device, and that chunk's results arrive while the NEXT call on the same context stages and queues its first chunk -- a block is
        # therefore handed on len(engs) iterations late.  Synchronous calls left the device idle for a third of every call (first chunk's
        # staging + upload at the head, the last chunk's kernels + download at the tail: 12.6 ms per 262 144-row block of which 8 ms were
        # kernels).  With one context the loop also announces the rows of the block it has read ahead (sh_prefetch_rows).
        import collections
        pipe = True
        G = len(engs)
        pend = collections.deque()                        # (block, results, engine) handed to the engine, not yet to the sink
        blocks = iter(blocks)
        t_r = _time.perf_counter()
        ahead = next(blocks, None)
        tm["reader"] += _time.perf_counter() - t_r
        k = 0
        while ahead is not None:
            blk = ahead
            t_r = _time.perf_counter()
            ahead = next(blocks, None)                    # (waiting here = the reader is the slowest stage)
            tm["reader"] += _time.perf_counter() - t_r
            t_e = _time.perf_counter()
            e = engs[k % G]
            r = None
            if pipe and G == 1 and ahead is not None and ahead.bits.shape[0] and ahead.bits.dtype == np.uint8 and ahead.bits.flags.c_contiguous:
                e.prefetch(ahead.bits)                    # block k+1's first chunk goes up while block k is on the device (sh_prefetch_rows)
            if blk.bits.shape[0]:
                if pipe:
                    r = e.lmm_batch(blk.bits, pipelined=True) if options.lmm else e.glm_batch(blk.bits, pipelined=True)
                else:
                    r = e.lmm_batch(blk.bits) if options.lmm else e.glm_batch(blk.bits)
            elif pipe:
                e.wait()                                  # no call for an empty block: complete this context's previous one here
            tm["engine"] += _time.perf_counter() - t_e; tm["blocks"] += 1
            pend.append((blk, r, e))
            # block j is complete once its context has returned from its next call (or wait): the oldest of more than G pending ones is
            while len(pend) > (G if pipe else 0):
                b_, r_, e_ = pend.popleft()
                eng = e_                                  # (the lineage fits of the sink go to the context that holds the block's rows)
                after_engine(b_, r_)
            k += 1
        while pend:
            b_, r_, e_ = pend.popleft()
            t_e = _time.perf_counter()
            if pipe:
                e_.wait()
            tm["engine"] += _time.perf_counter() - t_e
            eng = e_
            after_engine(b_, r_)

        drain_sink()
        if sink_thread is not None:
            sink_q.put(None)
            sink_thread.join()
        tm["rows"] = prefilter + tested; tm["loop"] = _time.perf_counter() - tm["t0"]; tm["overlap"] = overlap
        return prefilter, tested, printed

    if hasattr(sys.stdout, "buffer"):
        def write_stdout(b):
            sys.stdout.flush()
            sys.stdout.buffer.write(b)
    else:
        def write_stdout(b):
            sys.stdout.write(bytes(b).decode())

    # ---- the job: one stream over all contexts, or one stream PER context ------------------------------------------------------------------
    # The reference's --cpu N hands blocks of the variant stream to N workers and takes them back in order (__main__.py:541-568, 777-780).
    # Here every device gets its own pipelined stream when the input can be cut into ranges up front (the packed cache: range i of the
    # rows to context i, each with its own reader thread, engine pipeline, sink thread and output part; the parts are joined in order and the
    # four counters summed at the end -- nothing crosses between devices while the job runs).  Input that is only readable front to back
    # (text, gzip) stays one stream whose blocks go to the contexts in turn.
    tms = [new_tm()]
    import resource as _res
    from . import _abi as _abi_mod
    _lib = _abi_mod.load()
    ru_loop0 = _res.getrusage(_res.RUSAGE_SELF); cpu_stage0 = _abi_mod.host_cpu_seconds(); t_loop0 = _time.perf_counter()

    def task_cpu():
        """(SEERHIP_DEBUG=cli) user / system CPU seconds of every thread of the process by name: /proc/self/task/*/stat"""
        out = {}
        try:
            tck = float(os.sysconf("SC_CLK_TCK"))
            for tid in os.listdir("/proc/self/task"):
                try:
                    st = open("/proc/self/task/%s/stat" % tid).read()
                except (IOError, OSError):
                    continue
                comm = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (comm, int(f[11]) / tck, int(f[12]) / tck)
        except Exception:
            pass
        return out
    task0 = task_cpu() if cli_timing else {}
    thread_cpu = {}                                        # CPU seconds of the Python threads of the block loop, by role (time.thread_time)
    # a packed cache through the job stream: the block loop runs inside the library, one call per device (SEERHIP_ROUTE job=py: the loop in
    # Python, as round 5 had it and as text input still has it)
    packed_c_loop = bool(job_path and options.load_packed and _route.route("job", "1") != "py" and has_fd(sys.stdout)
                         and (patterns is None or has_fd(patterns)))
    if packed_c_loop:
        check_packed_cache(p, options.load_packed)
    if proc_mode:
        import pickle
        import shutil
        import subprocess
        import tempfile
        check_packed_cache(p, options.load_packed)
        G = len(devs_)
        state_dir = tempfile.mkdtemp(prefix="pyseer_amd_job_")
        try:
            arrays = {}
            st_ = {"n": len(p), "lmm": bool(options.lmm), "devices": devs_, "continuous": bool(options.continuous), "filter_pvalue": options.filter_pvalue,
                   "lrt_pvalue": options.lrt_pvalue, "no_dedup": bool(options.no_dedup), "min_af": options.min_af, "max_af": options.max_af,
                   "print_filtered": bool(options.print_filtered), "lineage_labels": (list(lineage_dict) if options.lineage else None),
                   "lineage_per_variant": bool(options.lmm_lineage_per_variant), "patterns": patterns is not None,
                   "sample_names": ([str(x) for x in p.index] if options.print_samples else None), "job_block": int(job_block),
                   "dma": _route.route("dma", "1") != "0", "path": os.path.abspath(options.load_packed)}
            if options.lmm:
                arrays.update(U=lmm.U, S=lmm.S, Y=lmm.Y, X=lmm.X); st_["h2"] = float(h2)
            else:
                arrays.update(y=np.asarray(p.values, dtype=float), W=covariate_block(len(p), m, cov))
                st_["llf"] = float("nan") if options.continuous else float(null_fit.llf)
                st_["firth_null"] = None if options.continuous else float(firth_null)
            if options.lineage:
                arrays.update(lin=np.asarray(lineage_clusters, dtype=float), lin_cov=(np.asarray(cov.values, dtype=float) if cov.shape[1] > 0 else np.zeros((0, 0))))
            np.savez(os.path.join(state_dir, "arrays.npz"), **arrays)
            with open(os.path.join(state_dir, "state.pkl"), "wb") as fh:
                pickle.dump(st_, fh)
            sys.stdout.flush()
            env_c = dict(os.environ)
            env_c["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env_c.get("PYTHONPATH", "")
            t_children0 = _time.perf_counter()
            kids = [subprocess.Popen([sys.executable, "-m", "pyseer_amd._packed_child", state_dir, str(i)], env=env_c,
                                     stdout=(None if i == 0 else subprocess.DEVNULL), stderr=subprocess.PIPE) for i in range(G)]
            # (like the reference's and the single-device run's, the output of a run that fails is partial: what part 0 had printed)
            errs_c = [k_.communicate()[1].decode() for k_ in kids]
            for i, k_ in enumerate(kids):
                for line in errs_c[i].splitlines():
                    if line.startswith("No observations of ") or line.startswith("pyseer_amd:"):
                        sys.stderr.write(line + "\n")
            bad = [i for i, k_ in enumerate(kids) if k_.returncode != 0]
            if bad:
                sys.stderr.write(errs_c[bad[0]][-4000:])
                _die("pyseer_amd: the stream of device %d failed; the output is partial\n" % devs_[bad[0]])
            results = [json_load(os.path.join(state_dir, "result_%d.json" % i)) for i in range(G)]
            for i in range(1, G):                          # the parts, in the order of the input
                with open(os.path.join(state_dir, "out_%d.tsv" % i), "rb") as fh:
                    shutil.copyfileobj(fh, sys.stdout.buffer if hasattr(sys.stdout, "buffer") else sys.stdout, 1 << 22)
            if patterns is not None:
                patterns.flush()
                for i in range(G):
                    with open(os.path.join(state_dir, "pat_%d.txt" % i), "rb") as fh:
                        shutil.copyfileobj(fh, patterns, 1 << 22)
            prefilter, tested, printed = (sum(r_[k_] for r_ in results) for k_ in ("prefilter", "tested", "printed"))
            tms = []
            for i, r_ in enumerate(results):
                tm_ = new_tm(); tm_.update(blocks=r_["blocks"], rows=r_["prefilter"] + r_["tested"], loop=r_["wall_s"], engine=r_["wall_s"], overlap=True, job=True)
                tms.append(tm_)
                thread_cpu["device %d process (block loop, user + sys)" % devs_[i]] = r_["user_s"] + r_["sys_s"]
            child_budget = {"user_s": sum(r_["user_s"] for r_ in results), "sys_s": sum(r_["sys_s"] for r_ in results),
                            "stages": {k_: sum(r_["library_stage_cpu_s"].get(k_, 0.0) for r_ in results) for k_ in results[0]["library_stage_cpu_s"]},
                            "wall_s": max(r_["wall_s"] for r_ in results), "launch_to_join_s": _time.perf_counter() - t_children0}
        finally:
            shutil.rmtree(state_dir, ignore_errors=True)
    elif len(engs) > 1 and options.load_packed:
        import shutil
        import tempfile
        import threading as _th
        G = len(engs)
        _lib.sh_set_host_streams(G)                        # per-stream helpers (stager, readers) take 1/G of the host CPU budget each
        tms = [new_tm() for _ in range(G)]
        # part 0 goes straight to stdout, the others to temporary files ($TMPDIR) joined in order at the end: like the reference's and the
        # single-device run's, the output of a run that fails is partial (what part 0 had printed); the first error stops the other streams
        outs = [None] + [tempfile.TemporaryFile() for _ in range(1, G)]
        pouts = [None] + [tempfile.TemporaryFile() if patterns is not None else None for _ in range(1, G)]
        counts = [None] * G
        errs = [None] * G
        stop = _th.Event()
        import ctypes as _ct
        stop_c = _ct.c_int(0)

        def stream_worker(i):
            try:
                t_th = _time.thread_time()
                if packed_c_loop:
                    counts[i] = run_stream_packed(engs[i], (i, G), (sys.stdout.buffer if (i == 0 and hasattr(sys.stdout, "buffer")) else sys.stdout) if i == 0 else outs[i],
                                                  None if patterns is None else (patterns if i == 0 else pouts[i]), tms[i], stop_c)
                    thread_cpu["stream %d loop thread" % i] = _time.thread_time() - t_th
                    return
                blocks_i = iter_packed_blocks_cached(p, options.load_packed, options.min_af, options.max_af, job_block,
                                                     want_patterns=bool(options.output_patterns), want_samples=options.print_samples, part=(i, G),
                                                     raw=job_path, device=(engs[i].device if job_path else None), ahead=job_ahead)
                wp = None if patterns is None else (patterns.write if i == 0 else pouts[i].write)
                if job_path:
                    counts[i] = run_stream_job([engs[i]], blocks_i, write_stdout if i == 0 else outs[i].write, tms[i], stop, wp)
                else:
                    counts[i] = run_stream([engs[i]], blocks_i, write_stdout if i == 0 else outs[i].write, wp, tms[i])
                thread_cpu["stream %d loop thread" % i] = _time.thread_time() - t_th
            except BaseException as ex:                    # re-raised below, on the main thread
                errs[i] = ex
                stop.set()
                stop_c.value = 1
        workers = [_th.Thread(target=stream_worker, args=(i,)) for i in range(G)]
        for w_ in workers:
            w_.start()
        for w_ in workers:
            w_.join()
        for ex in errs:
            if ex is not None:
                raise ex
        sys.stdout.flush()
        for i in range(1, G):                              # the parts, in the order of the input
            outs[i].seek(0)
            if hasattr(sys.stdout, "buffer"):
                shutil.copyfileobj(outs[i], sys.stdout.buffer, 1 << 22)
            else:
                sys.stdout.write(outs[i].read().decode())
            outs[i].close()
            if patterns is not None:
                pouts[i].seek(0)
                shutil.copyfileobj(pouts[i], patterns, 1 << 22)
                pouts[i].close()
        prefilter, tested, printed = (sum(c[j] for c in counts) for j in range(3))
    elif packed_c_loop:
        t_th = _time.thread_time()
        part01 = (0, 1)
        if options.packed_part:
            part01 = tuple(int(x) for x in options.packed_part.split("/"))
        prefilter, tested, printed = run_stream_packed(engs[0], part01, sys.stdout.buffer if hasattr(sys.stdout, "buffer") else sys.stdout, patterns, tms[0])
        thread_cpu["stream 0 loop thread"] = _time.thread_time() - t_th
    elif job_path:
        t_th = _time.thread_time()
        prefilter, tested, printed = run_stream_job(engs, blocks, write_stdout, tms[0], None, None if patterns is None else patterns.write)
        thread_cpu["stream 0 loop thread"] = _time.thread_time() - t_th
    else:
        prefilter, tested, printed = run_stream(engs, blocks, write_stdout, None if patterns is None else patterns.write, tms[0])

    if cli_timing:
        # the host budget of the block loop (tools/gpu_e2e_job.py -> profiles/r05/host_budget.json): process CPU seconds (user + sys, every
        # thread: getrusage) over the loop, the library's own per-stage thread CPU (csrc/host_pool.h), the loop threads' CPU (time.thread_time)
        ru1 = _res.getrusage(_res.RUSAGE_SELF); st1 = _abi_mod.host_cpu_seconds()
        rows_all = sum(tm_["rows"] for tm_ in tms)
        if proc_mode:                                      # the block loops ran in the devices' processes: their own rusage over their loops
            ru1 = type("R", (), {"ru_utime": ru1.ru_utime + child_budget["user_s"], "ru_stime": ru1.ru_stime + child_budget["sys_s"]})()
            st1 = {k_: st1.get(k_, 0.0) + child_budget["stages"].get(k_, 0.0) for k_ in set(st1) | set(child_budget["stages"])}
        budget = {"rows": int(rows_all), "wall_s": (child_budget["wall_s"] if proc_mode else _time.perf_counter() - t_loop0), "streams": len(tms), "job_path": bool(job_path),
                  "host_cpus": int(_lib.sh_host_cpus()), "pool_workers": int(_lib.sh_host_pool_workers()),
                  "process_cpu_s": (ru1.ru_utime - ru_loop0.ru_utime) + (ru1.ru_stime - ru_loop0.ru_stime),
                  "process_user_s": ru1.ru_utime - ru_loop0.ru_utime, "process_sys_s": ru1.ru_stime - ru_loop0.ru_stime,
                  "library_stage_cpu_s": {k_: st1[k_] - cpu_stage0.get(k_, 0.0) for k_ in st1},
                  "loop_thread_cpu_s": thread_cpu}
        by_name = {}
        for tid, (comm, ut, stt) in task_cpu().items():
            u0, s0 = task0.get(tid, (comm, 0.0, 0.0))[1:]
            a = by_name.setdefault(comm, [0, 0.0, 0.0]); a[0] += 1; a[1] += ut - u0; a[2] += stt - s0
        budget["threads_by_name"] = {k_: {"threads": v_[0], "user_s": round(v_[1], 3), "sys_s": round(v_[2], 3)} for k_, v_ in by_name.items() if v_[1] + v_[2] >= 0.01}
        import json as _json
        sys.stderr.write("[cli budget] " + _json.dumps(budget) + "\n")
    if cli_timing:
        for i, tm in enumerate(tms):
            if tm.get("job"):
                sys.stderr.write("[cli timing] stream %d of %d (job stream): %d blocks, %d rows in %.2f s of the block loop = %.3g rows/s; in library calls (submit + "
                                 "collect: launches, the wait for the oldest block, formatting of printed rows) %.2f s, write %.2f s; this thread waited %.2f s for the reader\n"
                                 % (i, len(tms), tm["blocks"], tm["rows"], tm["loop"], tm["rows"] / max(tm["loop"], 1e-9), tm["engine"], tm["write"], tm["reader"]))
                continue
            sys.stderr.write("[cli timing] stream %d of %d: %d blocks, %d rows in %.2f s of the block loop = %.3g rows/s; engine calls (H2D + GPU + D2H) %.2f s, sink (masking, "
                             "counters, formatting, write) %.2f s of which formatting %.2f s and write %.2f s; sink %s; this thread waited %.2f s for the reader and %.2f s for the sink's queue\n"
                             % (i, len(tms), tm["blocks"], tm["rows"], tm["loop"], tm["rows"] / max(tm["loop"], 1e-9), tm["engine"], tm["sink"], tm["format"], tm["write"],
                                "on a worker thread" if tm["overlap"] else "serial", tm["reader"], tm["queue"]))
        sys.stderr.write("[cli timing] before the block loop: %.2f s from process start to the model set-up (interpreter, imports, phenotypes, "
                         "structure), %.2f s model set-up (kinship cache / null fit, device context, engine set-up), %.2f s to the first block; "
                         "%.2f s from process start to the end of the loop\n"
                         % (since_start, t_setup - t_inputs, tms[0]["t0w"] - t_setup, since_start + _time.time() - t_inputs))
    if patterns is not None:
        patterns.close()
    if cache_out is not None:
        cache_out.close()
    for e_ in engs:
        e_.close()
    sys.stderr.write('%d loaded variants\n' % (prefilter + tested))
    sys.stderr.write('%d pre-filtered variants\n' % prefilter)
    sys.stderr.write('%d tested variants\n' % tested)
    sys.stderr.write('%d printed variants\n' % printed)


if __name__ == "__main__":
    main()
