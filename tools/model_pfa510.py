#!/usr/bin/env python
"""In-place prime-factor (Good-Thomas) 510-point transform, 510 = 2 . 3 . 5 . 17, as the wave kernel of
nmx_k_timeosc_w510.h runs it: no twiddles, every small DFT reads and writes the SAME positions.
  input  x[n] sits at position n (natural order)
  a position p has coordinates c_d = (p * inv_d) mod N_d, inv_d = (N / N_d)^-1 mod N_d   (p = sum_d c_d N / N_d mod N)
  the DFT along dimension d runs over the positions (g + c N / N_d) mod N, c = 0 .. N_d - 1, g = a position with c_d = 0
  afterwards position p holds X[k] with k = the CRT index of p's coordinates: k = sum_d c_d e_d mod N,
  e_d = (N / N_d) * ((N / N_d)^-1 mod N_d)
Phases as in the kernel: (2 x 5) in registers (51 groups of 10), 3 (170 groups), 17 (30 groups)."""
import numpy as np

N = 510
DIMS = (2, 3, 5, 17)
STR = {d: N // d for d in DIMS}
INV = {d: pow(N // d, -1, d) for d in DIMS}


def coords(p):
    return {d: (p * INV[d]) % d for d in DIMS}


def k_of_pos(p):
    c = coords(p)
    return sum(c[d] * STR[d] * INV[d] for d in DIMS) % N


def dft_along(buf, d):
    out = buf.copy()
    done = set()
    for g in range(N):
        if coords(g)[d] != 0 or g in done:
            continue
        pos = [(g + c * STR[d]) % N for c in range(d)]
        done.update(pos)
        v = buf[pos]
        k = np.arange(d)
        out[pos] = np.exp(-2j * np.pi * np.outer(k, k) / d) @ v
    return out


def main():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    buf = x.copy()
    for d in (2, 5, 3, 17):
        buf = dft_along(buf, d)
    X = np.fft.fft(x)
    kk = np.array([k_of_pos(p) for p in range(N)])
    assert sorted(kk.tolist()) == list(range(N))
    err = np.abs(buf - X[kk]).max()
    print("max err", err)
    assert err < 1e-9
    # group bases of the three phases (positions with the phase's coordinates zero), as the host tables hold them
    g10 = [p for p in range(N) if coords(p)[2] == 0 and coords(p)[5] == 0]
    g3 = [p for p in range(N) if coords(p)[3] == 0]
    g17 = [p for p in range(N) if coords(p)[17] == 0]
    print(len(g10), len(g3), len(g17))
    assert (len(g10), len(g3), len(g17)) == (51, 170, 30)


if __name__ == "__main__":
    main()
