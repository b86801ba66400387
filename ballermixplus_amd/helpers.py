"""Helper-file generation: --getSpect / --getConfig (reference BalLeRMix+_v1.py:645-710).

Byte-compatible restatement: same tabulation, same '%s' formatting of Python floats, same
messages.  Not on the GPU path (one O(N) text pass)."""
import sys


def getConfig(infile, configfile):
    """v1:645-664"""
    Config = {}
    numSites = 0
    with open(infile, 'r') as sites:
        next(sites)
        for l in sites:
            x, n = [int(v) for v in l.strip().split('\t')[2:]]
            if x == 0:
                print('Please make sure the input has derived allele frequency. Sites with 0 observed allele count (k=0) will be ignored.\n')
                continue
            if n not in Config:
                Config[n] = [0, 0]
            Config[n][0] += int(x == n)
            Config[n][1] += 1 - int(x == n)
            numSites += 1
    with open(configfile, 'w') as config:
        for N in sorted(Config.keys()):
            config.write('%s\t%s\t%s\n' % (N, Config[N][0] / float(numSites), Config[N][1] / float(numSites)))
    print('Done')


def getSpect(infile, spectfile, MAF, nosub):
    """v1:667-710"""
    Spect = {}
    numSites = 0
    translate = False
    skip_report = False
    with open(infile, 'r') as sites:
        next(sites)
        for l in sites:
            (x, n) = [int(i) for i in l.strip().split('\t')[2:]]
            if MAF:
                if not x <= n / 2:
                    if not translate:
                        print('Input data includes non-MAF site/s (frequency >= 0.5) despite choosing to use B_maf (with --MAF). These frequencies will be folded for following analyses.')
                        translate = True
                    x = n - x
                x = min(x, n - x)
            elif x == 0:
                print('Please make sure the input has derived allele frequency. Sites with 0 observed allele count (k=0) should not be included.\n')
                sys.exit()
            if nosub:
                if not x != (n * (1 - MAF)):
                    if not skip_report:
                        print('Input includes substitutions despite choosing to use B_0 or B_0maf (with --noSub). These sites will not be accounted for.')
                        skip_report = True
                    continue
            if (x, n) in Spect:
                Spect[(x, n)] += 1
            else:
                Spect[(x, n)] = 1
            numSites += 1
    with open(spectfile, 'w') as spec:
        for x, n in sorted(Spect.keys()):
            spec.write('%s\t%s\t%s\n' % (x, n, float(Spect[(x, n)]) / float(numSites)))
    print('Done.')
