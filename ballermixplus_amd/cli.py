"""Command line of the MI355X build: every flag of the reference's main()
(BalLeRMix+_v1.py:715-802) with the same spelling, defaults and pipeline order
InputData -> NeutralSFS -> get_neut_probs -> Grids -> NormalizedBetaBinom -> Scan (v1:777-799).

Additions (do not change any reference command line):
  --device K        GPU index for a single-process run (default 0)
  -i a.txt,b.txt,... | --inputs LIST.txt     several input files (a whole genome, one file per chromosome) in ONE process
                    on one scan context: the selection table is built once (and rebuilt only when a file's sample sizes
                    or minCount differ), the next file is read while the current one is scanned, and -o names a directory
                    (created; outputs are <dir>/<input basename>.out.txt) or a pattern containing {} (replaced by the
                    input's basename without extension).  The reference handles one file per process
                    (BalLeRMix+_v1.py:777-799); looping it pays process start, HIP start-up and the table once per file.
Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N -m ballermixplus_amd.cli ...`;
test sites are sharded over the ranks (rank r computes on GPU LOCAL_RANK), rank 0 gathers the 16-byte records
(one RCCL gather) and writes the output file.  BMX_DIST_BACKEND=gloo BMX_SINGLE_DEVICE=1 lets several ranks
share GPU 0 with a CPU gather: a rehearsal of the multi-rank control flow on a 1-GPU box.
"""
import argparse
import os
import sys
from datetime import datetime


def build_parser():
    """Flags, destinations, types and defaults are the reference's (v1:718-753): its command lines run unchanged.
    The help texts are this build's own wording."""
    parser = argparse.ArgumentParser(description='BalLeRMix+ B-statistic scan on AMD Instinct MI355X (libbmxscan).')
    parser.add_argument('-i', '--input', dest='infile', required=False, default=None,
                        help='input file: header line, then tab-separated physPos, genPos, derived (or minor) allele count x, sample size n')
    parser.add_argument('-o', '--output', dest='outfile', help='output file (7 tab-separated columns, one row per test site)')
    parser.add_argument('--spect', dest='spectfile', required=True,
                        help='neutral helper file to read: frequency spectrum (k n fraction) or, with --noFreq, the '
                             'substitution/polymorphism configuration. With --getSpect / --getConfig: the file to WRITE')
    parser.add_argument('--minCount', dest='minCount', default=1,
                        help='smallest allele count present in the input when rare variants were filtered out (default 1; '
                             'used as given only by the B_1 statistic, otherwise taken from the data)')
    parser.add_argument('--getSpect', dest='getSpec', action='store_true', default=False,
                        help='helper step: tabulate the (k, n) frequency spectrum of the concatenated input -i into --spect and exit (combine with --MAF / --noSub as for the scan)')
    parser.add_argument('--getConfig', dest='getConfig', action='store_true', default=False,
                        help='helper step: tabulate the substitution : polymorphism proportions per sample size of -i into --spect and exit')
    parser.add_argument('--findBal', dest='bal', action='store_true', default=False,
                        help='restrict the (x, alpha) grid to shapes typical of balancing selection')
    parser.add_argument('--findPos', dest='pos', action='store_true', default=False,
                        help='restrict the (x, alpha) grid to shapes typical of positive selection (ignored when --findBal is given, as in the reference)')
    parser.add_argument('--noFreq', dest='nofreq', action='store_true', default=False,
                        help='B_1: ignore allele frequencies; a site is a substitution (count 0) or a polymorphism (any other count)')
    parser.add_argument('--noSub', dest='nosub', action='store_true', default=False,
                        help='B_0 / B_0,MAF: the input holds polymorphic sites only, no substitutions')
    parser.add_argument('--MAF', dest='MAF', action='store_true', default=False,
                        help='B_2,MAF / B_0,MAF: fold counts to minor-allele counts (default: polarised counts, B_2)')
    parser.add_argument('--usePhysPos', action='store_true', dest='phys', default=False,
                        help='measure distances on physical positions times --rec instead of on the genPos column')
    parser.add_argument('--rec', dest='Rrate', default=1e-6, type=float,
                        help='uniform recombination rate in cM per nucleotide for --usePhysPos (default 1e-6)')
    parser.add_argument('--fixWinSize', action='store_true', dest='size', default=False,
                        help='windows of a fixed physical length; give the length in nucleotides with -w')
    parser.add_argument('-w', '--window', dest='w', type=int, default=0,
                        help='number of informative sites on either side of the test site, or with --fixWinSize the window length in nucleotides (default 0: every site with alpha >= 1e-8)')
    parser.add_argument('--noCenter', action='store_true', dest='noCenter', default=False,
                        help='with --fixWinSize: test positions every -s nucleotides instead of at informative sites')
    parser.add_argument('-s', '--step', dest='step', type=float, default=1,
                        help='test every s-th informative site, or every s nucleotides with --noCenter (default 1)')
    parser.add_argument('--fixX', dest='x', help='fix the equilibrium frequency x instead of searching the x grid')
    parser.add_argument('--fixAlpha', dest='abeta', type=float, default=None,
                        help='fix the beta-binomial alpha parameter instead of searching the alpha grid')
    parser.add_argument('--rangeA', dest='seqA', help='linkage parameter grid as <Amin>,<Amax>,<Astep> (no spaces)')
    parser.add_argument('--listA', dest='listA', help='linkage parameter grid as a comma-separated list (no spaces)')
    # additions
    parser.add_argument('--inputs', dest='inputs', default=None,
                        help='MI355X build only: a text file naming several input files, one per line (whole genome in one process); '
                             'equivalent to -i a.txt,b.txt,...; -o is then a directory or a pattern containing {}')
    parser.add_argument('--device', dest='device', type=int, default=None,
                        help='GPU index for a single-process run (MI355X build only; not allowed under torch.distributed.run, where every rank uses GPU LOCAL_RANK)')
    return parser


def main(argv=None):
    import time
    t_start = time.time()
    stamp = (lambda what: print('[bmx cli] %-28s %.3f s' % (what, time.time() - t_start), file=sys.stderr)) \
        if os.environ.get('BMX_TRACE') else (lambda what: None)
    argv = sys.argv[1:] if argv is None else argv
    parser = build_parser()
    if len(argv) == 0:
        parser.print_help()
        sys.exit()
    opt = parser.parse_args(argv)
    if opt.infile is None and opt.inputs is None:
        parser.error('the following arguments are required: -i/--input')
    files = None
    if opt.inputs is not None:
        if opt.getSpec or opt.getConfig:
            print('--inputs lists files to scan; --getSpect / --getConfig take ONE concatenated input with -i (as in the reference).')
            sys.exit(1)
        files = input_list(opt.inputs)
    elif ',' in opt.infile and not os.path.exists(opt.infile):
        files = [p for p in opt.infile.split(',') if p]
    if files is not None and not (opt.getSpec or opt.getConfig):
        return main_many(opt, files, stamp)

    from . import helpers
    if opt.getSpec:
        print('You\'ve chosen to generate site frequency spectrum...')
        print(('Concatenated input: %s \nSpectrum file: %s' % (opt.infile, opt.spectfile)))
        helpers.getSpect(opt.infile, opt.spectfile, opt.MAF, opt.nosub)
        sys.exit()
    elif opt.getConfig:
        print('You\'ve chosen to generate the substitution-polymorphism configuration...')
        print(('Concatenated input: %s \nConfiguration file: %s' % (opt.infile, opt.spectfile)))
        helpers.getConfig(opt.infile, opt.spectfile)
        sys.exit()

    from . import distributed, engine
    from .hostmodel import Grids, InputData, NeutralSFS
    from .scan import Scan

    world = distributed.World.from_env(backend=os.environ.get('BMX_DIST_BACKEND'))
    if world.distributed and opt.device is not None:
        # torch tensors of the gather and the scan context must live on the same GPU: one rank, one GPU
        print('--device cannot be combined with a multi-process launch: each rank uses GPU LOCAL_RANK.')
        sys.exit(1)
    device = opt.device if opt.device is not None else world.device_index
    verbose = world.rank == 0

    def say(*a):
        if verbose:
            print(*a)

    stamp('imports, process group')
    # The HIP runtime and the context take ~0.2 s to come up: start them now, on a helper thread, while this thread reads
    # the input and the helper file (the native calls release the GIL); engine.NormalizedBetaBinom picks the context up.
    import threading
    warm = {}

    def _warm():
        try:
            warm['ctx'] = engine.Context(device)
        except Exception as e:          # reported by the main thread where the reference-style flow creates the context
            warm['err'] = e

    warm_thread = threading.Thread(target=_warm)
    warm_thread.start()
    say(f"\n{datetime.now()}. Reading input from {opt.infile}")
    data = InputData(opt.infile, opt.nofreq, opt.MAF, opt.nosub, opt.minCount, phys=opt.phys, Rrate=opt.Rrate)
    Neutral = NeutralSFS(opt.spectfile, opt.nofreq, opt.MAF, opt.nosub)
    say(f'\n{datetime.now()}. Initializing...')
    say('Retrieving per-site neutral probabilities...')
    Neutral.get_neut_probs(data)
    grid = Grids(opt.x, opt.abeta, opt.bal, opt.pos, opt.seqA, opt.listA)
    say('\nOptimizing over x= ' + ', '.join(['%g' % (x) for x in grid.x]))
    say('\n \t alpha= ' + ', '.join([str(a) for a in grid.abeta]))
    say('\n \t A= ' + ', '.join([str(A) for A in grid.A]))
    stamp('input, neutral model, grids')
    warm_thread.join()
    if 'err' in warm:
        raise warm['err']
    stamp('HIP context ready')
    Sel_Probs = engine.NormalizedBetaBinom(data, grid, opt.nofreq, opt.MAF, opt.nosub, device=device, ctx=warm['ctx'])
    say(("\n%s. Start computing likelihood ratios..." % (datetime.now())))
    # BMX_SHARD_BLOCK: test sites per shard block (default distributed.BLOCK = 4096; a multiple of 16 keeps every window's
    # arithmetic independent of the number of ranks) -- lets small inputs exercise real sharding in the tests
    runner = world.sharded_runner(block=shard_block(), balance=os.environ.get('BMX_SHARD_BALANCE') == '1') if world.distributed else None
    Scan(data, Neutral, Sel_Probs, grid, opt.outfile if world.rank == 0 else None, fixSize=opt.size, r=opt.w,
         s=opt.step, phys=opt.phys, noCenter=opt.noCenter, runner=runner, verbose=verbose, keep_results=False)
    stamp('table, scan, output')
    world.finish()
    say(f'\n{datetime.now()}. Pipeline finished.')


def shard_block():
    """BMX_SHARD_BLOCK: test sites per shard block of a multi-rank run (default distributed.BLOCK = 4096).  A multiple of 16
    keeps group boundaries -- hence every window's arithmetic -- independent of the number of ranks; small values let small
    inputs exercise real sharding in the tests."""
    v = os.environ.get('BMX_SHARD_BLOCK')
    if not v:
        return None
    try:
        block = int(v)
    except ValueError:
        block = 0
    if block < 16 or block % 16:
        print('BMX_SHARD_BLOCK must be a positive multiple of 16.')
        sys.exit(1)
    return block


def input_list(path):
    """The files named in an --inputs list: one per line, blank lines and lines starting with # skipped, relative names taken
    relative to the list's own directory."""
    here = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        names = [l.strip() for l in f]
    return [n if os.path.isabs(n) else os.path.join(here, n) for n in names if n and not n.startswith('#')]


def output_name(outspec, infile):
    """-o of the multi-file form: a pattern containing {} (the input's basename without extension goes there) or a directory."""
    base = os.path.basename(infile)
    if '{}' in outspec:
        return outspec.replace('{}', os.path.splitext(base)[0])
    return os.path.join(outspec, base + '.out.txt')


def main_many(opt, files, stamp=lambda what: None):
    """Several input files through the reference's stages (InputData -> NeutralSFS.get_neut_probs -> NormalizedBetaBinom ->
    Scan, BalLeRMix+_v1.py:777-799) in one process on one scan context: file i + 1 is read and given its neutral
    probabilities on a helper thread while file i is scanned and written (the native calls release the GIL)."""
    import threading
    from . import distributed, engine
    from .hostmodel import Grids, InputData, NeutralSFS
    from .scan import Scan
    if not files:
        print('No input files given.')
        sys.exit(1)
    if not opt.outfile:
        print('Several input files need -o <directory> or -o <pattern with {}>.')
        sys.exit(1)
    world = distributed.World.from_env(backend=os.environ.get('BMX_DIST_BACKEND'))
    if world.distributed and opt.device is not None:
        print('--device cannot be combined with a multi-process launch: each rank uses GPU LOCAL_RANK.')
        sys.exit(1)
    device = opt.device if opt.device is not None else world.device_index
    verbose = world.rank == 0

    def say(*a):
        if verbose:
            print(*a)

    outs = [output_name(opt.outfile, f) for f in files]
    if len(set(outs)) != len(outs):
        print('Two input files map to the same output name; give -o a pattern with {} or distinct basenames.')
        sys.exit(1)
    if world.rank == 0:
        for d in sorted(set(os.path.dirname(os.path.abspath(o)) for o in outs)):
            os.makedirs(d, exist_ok=True)
    grid = Grids(opt.x, opt.abeta, opt.bal, opt.pos, opt.seqA, opt.listA)
    say('\nOptimizing over x= ' + ', '.join(['%g' % (x) for x in grid.x]))
    say('\n \t alpha= ' + ', '.join([str(a) for a in grid.abeta]))
    say('\n \t A= ' + ', '.join([str(A) for A in grid.A]))
    runner = world.sharded_runner(block=shard_block(), balance=os.environ.get('BMX_SHARD_BALANCE') == '1') if world.distributed else None
    nxt = {}

    def host_stage(i):
        try:
            data = InputData(files[i], opt.nofreq, opt.MAF, opt.nosub, opt.minCount, phys=opt.phys, Rrate=opt.Rrate)
            neut = NeutralSFS(opt.spectfile, opt.nofreq, opt.MAF, opt.nosub)
            neut.get_neut_probs(data)
            # the host half of NormalizedBetaBinom (model arrays, every site's table row) here too: no device call
            sel = engine.NormalizedBetaBinom(data, grid, opt.nofreq, opt.MAF, opt.nosub, device=device).prepare(neut)
            nxt[i] = (data, neut, sel)
        except BaseException as e:          # incl. the reference-style sys.exit() of the readers: re-raised on the main thread
            nxt[i] = e

    th = threading.Thread(target=host_stage, args=(0,))
    th.start()
    ctx = engine.Context(device)        # HIP start-up (0.2-0.3 s) while the first file is being read
    tables = 0
    kernel_ms = 0.0
    for i, (infile, outfile) in enumerate(zip(files, outs)):
        th.join()
        got = nxt.pop(i)
        if isinstance(got, BaseException):
            raise got
        data, neut, sel = got
        if i + 1 < len(files):
            th = threading.Thread(target=host_stage, args=(i + 1,))
            th.start()
        say(f"\n{datetime.now()}. {infile} -> {outfile}")
        Scan(data, neut, sel, grid, outfile if world.rank == 0 else None, fixSize=opt.size, r=opt.w, s=opt.step, phys=opt.phys,
             noCenter=opt.noCenter, runner=runner, verbose=verbose, keep_results=False, reuse_ctx=ctx)
        ctx = sel.ctx
        tables += 0 if sel.table_reused else 1
        try:
            kernel_ms += ctx.last_scan_ms()
        except Exception:           # a file without test sites
            pass
        stamp('file %d of %d' % (i + 1, len(files)))
    world.finish()
    say(f'\n{datetime.now()}. Pipeline finished: {len(files)} files, selection table built {tables} time(s), '
        f'scan kernels {kernel_ms / 1e3:.2f} s.')


if __name__ == '__main__':
    main()
    sys.stdout.flush()
    sys.stderr.flush()
    if os.environ.get('BMX_FAST_EXIT', '1') != '0':       # see BalLeRMixPlus_amd.py
        os._exit(0)
