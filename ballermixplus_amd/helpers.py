"""Helper-file generation: --getSpect / --getConfig (reference BalLeRMix+_v1.py:645-710).

Byte-compatible: same tabulation, same '%s' formatting of Python floats, same messages -- computed on the (k, n) columns
as arrays (a 40M-line genome is seconds, not minutes).  Not on the GPU path (one O(N) pass over the input)."""
import sys

import numpy as np

from .hostmodel import read_columns


def _columns(infile):
    """(k, n) of every data line (native reader; Python's converters for files it declines)."""
    _, _, k, n = read_columns(infile, 1)
    return k, n


def getConfig(infile, configfile):
    """v1:645-664: per sample size, the fractions of substitutions (k == n) and polymorphisms among all sites."""
    k, n = _columns(infile)
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


def getSpect(infile, spectfile, MAF, nosub):
    """v1:667-710: the (k, n) frequency table of the input, folded with --MAF, without substitutions with --noSub."""
    x, n = _columns(infile)
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
    key, cnt = np.unique(x * base + n, return_counts=True)     # ascending (x, n): the order the reference writes
    with open(spectfile, 'w') as spec:
        for kk, c in zip(key.tolist(), cnt.tolist()):
            spec.write('%s\t%s\t%s\n' % (kk // base, kk % base, float(c) / float(numSites)))
    print('Done.')
