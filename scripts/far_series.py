"""Economised coefficients of log1p(x) = sum_k (-1)^(k+1) w_k x^k on [-eps, eps] (the far-field series of the scan kernels):
the Taylor polynomial of degree 40 re-expanded in Chebyshev polynomials of x/eps and cut after T_K, constant term dropped.
    python scripts/far_series.py [K=12] [eps=0.15]
Prints w_1..w_K as C doubles, the max error of the cut polynomial on the interval (against mpmath's log1p, 60 digits) and, for
comparison, the plain Taylor cut's."""
import sys
import mpmath as mp

mp.mp.dps = 80
K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
eps = mp.mpf(sys.argv[2]) if len(sys.argv) > 2 else mp.mpf('0.15')
D = 40
# Taylor coefficients in t = x / eps: c_k t^k, c_k = (-1)^(k+1) eps^k / k
c = [mp.mpf(0)] + [(-1) ** (k + 1) * eps ** k / k for k in range(1, D + 1)]


def mono_to_cheb(c):
    """monomial coefficients -> Chebyshev coefficients (exact in mp arithmetic), via t^k = 2^(1-k) sum' binom(k, (k-j)/2) T_j"""
    n = len(c)
    a = [mp.mpf(0)] * n
    for k, ck in enumerate(c):
        if ck == 0:
            continue
        for j in range(k, -1, -2):          # j = k, k-2, ...
            coef = mp.binomial(k, (k - j) // 2) / mp.mpf(2) ** (k - 1) if k > 0 else mp.mpf(1)
            if j == 0 and k > 0:
                coef /= 2
            a[j] += ck * coef
    return a


def cheb_to_mono(a):
    """Chebyshev coefficients -> monomial coefficients, T_{n+1} = 2 t T_n - T_{n-1}"""
    n = len(a)
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for k in range(2, n):
        nxt = [mp.mpf(0)] + [2 * v for v in T[-1]]
        for i, v in enumerate(T[-2]):
            nxt[i] -= v
        T.append(nxt)
    out = [mp.mpf(0)] * n
    for k, ak in enumerate(a):
        for i, v in enumerate(T[k]):
            out[i] += ak * v
    return out


a = mono_to_cheb(c)
cut = cheb_to_mono(a[:K + 1])          # degree-K polynomial in t
w = [abs(cut[k] / eps ** k) for k in range(1, K + 1)]        # back to x; signs alternate as in the Taylor series
for k in range(1, K + 1):
    assert (cut[k] > 0) == (k % 2 == 1)


def err(coefs_x):
    worst = mp.mpf(0)
    for i in range(-2000, 2001):
        x = eps * i / 2000
        p = sum((-1) ** (k + 1) * coefs_x[k - 1] * x ** k for k in range(1, K + 1))
        worst = max(worst, abs(p - mp.log1p(x)))
    return worst


print('K = %d, eps = %s: constant term dropped %s' % (K, mp.nstr(eps, 6), mp.nstr(cut[0], 3)))
print('economised: max |error| on the interval %s    plain Taylor cut: %s' % (mp.nstr(err(w), 3), mp.nstr(err([mp.mpf(1) / k for k in range(1, K + 1)]), 3)))
print('{' + ', '.join(repr(float(v)) for v in w) + '}')
