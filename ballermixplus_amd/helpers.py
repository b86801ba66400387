"""Helper-file generation: --getSpect / --getConfig (reference BalLeRMix+_v1.py:645-710).

Byte-compatible restatement: same tabulation, same '%s' formatting of Python floats, same
messages.  Not on the GPU path (one O(N) text pass)."""
import sys


def _columns(infile):
    """(k, n) columns through the native reader, or None (library not built / odd file)."""
    try:
        from . import _lib
        _, _, k, n = _lib.read_input(infile, 1)
        return k, n
    except Exception:
        return None


def getConfig(infile, configfile):
    """v1:645-664"""
    cols = _columns(infile)
    if cols is not None:        # same tabulation, vectorised (a 40M-line genome is minutes in the text loop)
        import numpy as np
        k, n = cols
        zero = k == 0
        for _ in range(int(zero.sum())):
            print('Please make sure the input has derived allele frequency. Sites with 0 observed allele count (k=0) will be ignored.\n')
        k, n = k[~zero], n[~zero]
        numSites = len(k)
        with open(configfile, 'w') as config:
            for N in np.unique(n).tolist():
                sel = n == N
                sub = int(np.sum(k[sel] == N))
                config.write('%s\t%s\t%s\n' % (N, sub / float(numSites), (int(sel.sum()) - sub) / float(numSites)))
        print('Done')
        return
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
    cols = _columns(infile)
    if cols is not None:
        import numpy as np
        x, n = cols
        if MAF:
            if np.any(x > n / 2):
                print('Input data includes non-MAF site/s (frequency >= 0.5) despite choosing to use B_maf (with --MAF). These frequencies will be folded for following analyses.')
            x = np.where(x > n / 2, n - x, x)
            x = np.minimum(x, n - x)
        elif np.any(x == 0):
            print('Please make sure the input has derived allele frequency. Sites with 0 observed allele count (k=0) should not be included.\n')
            sys.exit()
        if nosub:
            sub = x == (n * (1 - MAF))
            if np.any(sub):
                print('Input includes substitutions despite choosing to use B_0 or B_0maf (with --noSub). These sites will not be accounted for.')
            x, n = x[~sub], n[~sub]
        numSites = len(x)
        base = int(n.max()) + 1 if numSites else 1
        key, cnt = np.unique(x * base + n, return_counts=True)     # sorted by (x, n), as sorted(Spect.keys())
        with open(spectfile, 'w') as spec:
            for kk, c in zip(key.tolist(), cnt.tolist()):
                spec.write('%s\t%s\t%s\n' % (kk // base, kk % base, float(c) / float(numSites)))
        print('Done.')
        return
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
