#!/usr/bin/env python3
"""Coefficients of exp_pos() in unconfined_amd/csrc/ucf_fastpath.h: e^r = 1 + r + r^2 s(r) on |r| <= ln2/2, s of degree
9 interpolating (e^r - 1 - r)/r^2 at the Chebyshev nodes (near-minimax), computed in 70-digit decimal arithmetic and
rounded to double; prints the maximum relative error of the rounded polynomial in exact arithmetic."""
from decimal import Decimal as D, getcontext
getcontext().prec = 70
a = D("0.34665")
PI = D("3.14159265358979323846264338327950288419716939937510")


def g(r):
    return D("0.5") if abs(r) < D("1e-30") else (r.exp() - 1 - r) / (r * r)


def dcos(x):
    s, t, n = D(0), D(1), 0
    while abs(t) > D("1e-60"):
        s += t
        n += 2
        t = -t * x * x / (n * (n - 1))
    return s


deg = 9
n = deg + 1
nodes = [a * dcos((2 * j + 1) * PI / (2 * n)) for j in range(n)]
A = [[x ** k for k in range(n)] + [g(x)] for x in nodes]
for i in range(n):
    p = max(range(i, n), key=lambda r: abs(A[r][i]))
    A[i], A[p] = A[p], A[i]
    for r in range(i + 1, n):
        f = A[r][i] / A[i][i]
        for c in range(i, n + 1):
            A[r][c] -= f * A[i][c]
coef = [D(0)] * n
for i in reversed(range(n)):
    coef[i] = (A[i][n] - sum(A[i][c] * coef[c] for c in range(i + 1, n))) / A[i][i]
cd = [float(c) for c in coef]
err = D(0)
for k in range(-2000, 2001):
    r = a * D(k) / 2000
    s = D(0)
    for c in reversed(cd):
        s = s * r + D(c)
    err = max(err, abs(1 + r + r * r * s - r.exp()) / r.exp())
print("max rel err", float(err))
for i, c in enumerate(cd):
    print(f"s{i} = {c!r}")
