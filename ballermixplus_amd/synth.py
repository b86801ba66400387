"""Deterministic synthetic inputs for the BASELINE.json configurations.

The reference ships no large inputs, so configs 3-5 are
generated here from a seed.  The recipe is fixed so that this container and the
GPU box produce byte-identical files:

  per chromosome c, seed 1000+c, numpy Generator(PCG64(seed)), drawn in this order:
    gaps    = geometric(p = 1/72.5, N)          (>= 1  => strictly increasing positions)
    physPos = cumsum(gaps)
    kpoly   = choice(1..n-1, p ~ 1/k)
    issub   = random(N) < 0.70
    k       = n if issub else kpoly
    genPos  = physPos * 1e-6, written '%.6f'    (== physPos / 1e6, correctly rounded)

The 4-column text layout is the one InputData.readCounts parses
(reference BalLeRMix+_v1.py:113-131): header line, then physPos, genPos, x, n.
"""
import numpy as np

GRCH37_AUTOSOME_LEN = [
    249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663,
    146364022, 141213431, 135534747, 135006516, 133851895, 115169878, 107349540,
    102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566,
]


def synth_chromosome(N, n=100, chrom=1):
    """Return (physPos int64[N], genPos f64[N], k int64[N], n int64[N])."""
    rng = np.random.Generator(np.random.PCG64(1000 + int(chrom)))
    gaps = rng.geometric(1.0 / 72.5, int(N))
    phys = np.cumsum(gaps).astype(np.int64)
    ks = np.arange(1, n)
    w = 1.0 / ks
    kpoly = rng.choice(ks, size=int(N), p=w / w.sum())
    issub = rng.random(int(N)) < 0.70
    k = np.where(issub, n, kpoly).astype(np.int64)
    gen = phys / 1e6
    return phys, gen, k, np.full(int(N), n, dtype=np.int64)


def write_input(path, phys, gen, k, n):
    """Write the reference's 4-column input format."""
    with open(path, 'w') as f:
        f.write('physPos\tgenPos\tx\tn\n')
        for p, g, kk, nn in zip(phys.tolist(), gen.tolist(), k.tolist(), n.tolist()):
            f.write('%d\t%.6f\t%d\t%d\n' % (p, g, kk, nn))


def config4_sizes(total=40_000_000):
    """Per-chromosome SNP counts of BASELINE config 4 (proportional to GRCh37 lengths)."""
    L = np.array(GRCH37_AUTOSOME_LEN, dtype=np.float64)
    return [int(round(total * l / L.sum())) for l in L]


def spect_from_counts(k, n):
    """(k, n, fraction) rows exactly as getSpect would tabulate them for a DAF input
    (reference BalLeRMix+_v1.py:699-708): sorted by (k, n), fraction = count/numSites."""
    k = np.asarray(k, dtype=np.int64)
    n = np.asarray(n, dtype=np.int64)
    key = k * (int(n.max()) + 1) + n
    uniq, cnt = np.unique(key, return_counts=True)
    kk = uniq // (int(n.max()) + 1)
    nn = uniq % (int(n.max()) + 1)
    tot = float(len(k))
    return [(int(a), int(b), float(c) / tot) for a, b, c in zip(kk, nn, cnt)]
