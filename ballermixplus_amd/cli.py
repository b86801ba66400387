"""Command line of the MI355X build: every flag of the reference's main()
(BalLeRMix+_v1.py:715-802) with the same spelling, defaults and pipeline order
InputData -> NeutralSFS -> get_neut_probs -> Grids -> NormalizedBetaBinom -> Scan (v1:777-799).

Additions (do not change any reference command line):
  --device K        GPU index for a single-process run (default 0, or LOCAL_RANK under torchrun)
Multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N -m ballermixplus_amd.cli ...`;
test sites are sharded over the ranks, rank 0 gathers (RCCL) and writes the output file.
"""
import argparse
import os
import sys
from datetime import datetime


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('-i', '--input', dest='infile', help='Path and name of your input file.\n', required=True)
    parser.add_argument('-o', '--output', dest='outfile', help='Path and name of your output file.\n')
    parser.add_argument('--spect', dest='spectfile', help='Path and name of the allele frequency spectrum file or configuration file.\n', required=True)
    parser.add_argument('--minCount', dest='minCount', default=1, help='If rare variants are removed from the input, please provide the smallest allele count included in the input. Default value is 1.')
    parser.add_argument('--getSpect', dest='getSpec', action='store_true', default=False, help='Option to generate frequency spectrum file from the concatenated input file. Use "-i" and "--spect" commands to provide names and paths to input and output files, respectively. Indicate the input type with "--MAF".\n')
    parser.add_argument('--getConfig', dest='getConfig', action='store_true', default=False, help='Option to generate configuration file from the concatenated input file. Use "-i" and "--spect" commands to provide names and paths to input and output files, respectively.\n\n')
    parser.add_argument('--findBal', dest='bal', action='store_true', default=False, help="Option to only look for footprints of balancing selection.\n")
    parser.add_argument('--findPos', dest='pos', action='store_true', default=False, help="Option to only look for footprints of positive selection.\n")
    parser.add_argument('--noFreq', dest='nofreq', action='store_true', default=False, help='Option to compute B_1 statistic and ignore allele frequency information (if given). All polymorphic sites (non-zero counts) will be considered as equivalent. Substitutions should be represented as having zero count in the input.')
    parser.add_argument('--noSub', dest='nosub', action='store_true', default=False, help='Option to not include substitution in input data. B_0 or B_0maf will be computed.')
    parser.add_argument('--MAF', dest='MAF', action='store_true', default=False, help='Option to compute B_2maf statistic and use minor allele frequency instead of polarized allele frequency (if given). The latter is default (B_2 statisitc).')
    parser.add_argument('--usePhysPos', action='store_true', dest='phys', default=False, help='Option to use physical positions instead of genetic positions (in cM). Default is using genetic positions.\n')
    parser.add_argument('--rec', dest='Rrate', default=1e-6, type=float, help='The uniform recombination rate in cM/nt. Default value is 1e-6 cM/nt. Only useful when choose to use physical positions as coordinates.\n\n')
    parser.add_argument('--fixWinSize', action='store_true', dest='size', default=False, help='Option to fix the size (in nt) of sliding windows during scan. When true, please also provide the length of window in neucleotide (nt) with "-w" or "--window" command.\n')
    parser.add_argument('-w', '--window', dest='w', type=int, default=0, help='Number of sites flanking the test locus on either side. When choose to fix window size ("--fixSize"), input the length of window in bp.\n')
    parser.add_argument('--noCenter', action='store_true', dest='noCenter', default=False, help='Option to have the scanning windows not centered on informative sites. Require that the window size ("-w") in physical positions ("--usePhysPos") is provided. Default is True.\n')
    parser.add_argument('-s', '--step', dest='step', type=float, default=1, help='Step size in bp (when using "--noCenter") or the number of informative sites. Default value is one site or one nucleotide.\n\n')
    parser.add_argument('--fixX', dest='x', help='Option to fix the presumed equilibrium frequency.\n')
    parser.add_argument('--fixAlpha', dest='abeta', type=float, default=None, help='Option to fix the alpha parameter in the beta-binomial distribution.\n')
    parser.add_argument('--rangeA', dest='seqA', help='Range of the values of the linkage parameter A to optimize over. Format should follow <Amin>,<Amax>,<Astep> with no space around commas.\n')
    parser.add_argument('--listA', dest='listA', help='Manually provide a list of A values to optimize over. Please separate the values with comma, no space.\n')
    # additions
    parser.add_argument('--device', dest='device', type=int, default=None, help='GPU index (MI355X build only).')
    return parser


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    parser = build_parser()
    if len(argv) == 0:
        parser.print_help()
        sys.exit()
    opt = parser.parse_args(argv)

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

    world = distributed.World.from_env()
    device = opt.device if opt.device is not None else world.local_rank
    verbose = world.rank == 0

    def say(*a):
        if verbose:
            print(*a)

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
    Sel_Probs = engine.NormalizedBetaBinom(data, grid, opt.nofreq, opt.MAF, opt.nosub, device=device)
    say(("\n%s. Start computing likelihood raito..." % (datetime.now())))
    runner = world.sharded_runner() if world.distributed else None
    Scan(data, Neutral, Sel_Probs, grid, opt.outfile if world.rank == 0 else None, fixSize=opt.size, r=opt.w,
         s=opt.step, phys=opt.phys, noCenter=opt.noCenter, runner=runner)
    world.finish()
    say(f'\n{datetime.now()}. Pipeline finished.')


if __name__ == '__main__':
    main()
