#!/usr/bin/env python
"""Lane / register model of the M = 2048 channel-pair FIR transform (nmx_k_bank_w64e.h): the 32-point-per-lane sibling
of tools/model_w64c.py.

64 lanes x 32 registers, three register passes (radix 32 = 4 x 8, 8, 8) and two exchanges; forward = decimation in
frequency (A, B, C), inverse = the mirror (C', B', A'): the spectrum is consumed in the order the forward transform leaves
it and the series comes out in natural order (sample l + 64 j in register j of lane l).  Checks the index maps, the
exchange-tile addresses (bank-conflict freedom per 32-lane half) and the twiddle tables against numpy.fft, and the two
prunings the kernel uses: zero inputs j >= 16 (a window of <= 1024 samples) and outputs j < 16 only."""
import numpy as np

N, L, R, G = 2048, 64, 32, 4
S1 = 72                      # complex points per k_a row of exchange 1 (64 + 8 pad)
G2, S2, Q2 = 576, 72, 9      # exchange 2: g stride, u stride, q_a stride


def ka_of_reg(rho):          # pass-A output register rho = 8 r + p holds k_a = 4 p + r
    return G * (rho & 7) + (rho >> 3)


def dft(a, sign, axis):
    n = a.shape[axis]
    k = np.arange(n)
    Wm = np.exp(sign * 2j * np.pi * np.outer(k, k) / n)
    return np.moveaxis(np.tensordot(Wm, np.moveaxis(a, axis, 0), axes=(1, 0)), 0, axis)


def tables():
    l = np.arange(L)
    twa = np.exp(-2j * np.pi * np.outer([ka_of_reg(r) for r in range(R)], l) / N)      # [reg][lane]
    twb = np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 64)                 # [q][lane & 7]
    return twa, twb


def k_of(lane, reg):         # spectrum layout after forward pass C
    qa, u, g, qb = lane & 7, lane >> 3, reg >> 3, reg & 7
    return (u + 8 * g) + R * (qa + 8 * qb)


def check_banks(addr):
    for h in range(2):
        a = addr[32 * h:32 * h + 32] % 32
        assert len(set(a.tolist())) == 32, a


def forward(v, twa, twb):
    lanes = np.arange(L)
    # pass A: DFT-4 over jb (regs ja + 8 jb), twiddle W32^(ja r), DFT-8 over ja (regs 8 r + ja) -> reg 8 r + p: k_a = 4 p + r
    a = v.reshape(L, G, 8)                                   # [lane][jb][ja]
    t = dft(a, -1, 1)                                        # [lane][r][ja]
    t = t * np.exp(-2j * np.pi * np.outer(np.arange(G), np.arange(8)) / R)[None]
    a = dft(t, -1, 2).reshape(L, R)
    a = a * twa.T
    X = np.zeros(R * S1, complex)
    for rho in range(R):
        addr = ka_of_reg(rho) * S1 + lanes
        check_banks(addr)
        X[addr] = a[:, rho]
    b = np.zeros((L, R), complex)
    llo, u = lanes & 7, lanes >> 3
    for g in range(G):
        for lhi in range(8):
            addr = (u + 8 * g) * S1 + llo + 8 * lhi
            check_banks(addr)
            b[:, 8 * g + lhi] = X[addr]
    b = dft(b.reshape(L, G, 8), -1, 2)
    b = b * twb[:, llo].T[:, None, :]
    b = b.reshape(L, R)
    X2 = np.zeros(G * G2, complex)
    for g in range(G):
        for qa in range(8):
            addr = g * G2 + u * S2 + Q2 * qa + llo
            check_banks(addr)
            X2[addr] = b[:, 8 * g + qa]
    c = np.zeros((L, R), complex)
    qa_l = lanes & 7
    for g in range(G):
        for lo in range(8):
            addr = g * G2 + u * S2 + Q2 * qa_l + lo
            check_banks(addr)
            c[:, 8 * g + lo] = X2[addr]
    return dft(c.reshape(L, G, 8), -1, 2).reshape(L, R)


def inverse(z, twa, twb):
    lanes = np.arange(L)
    u = lanes >> 3
    c = dft(z.reshape(L, G, 8), +1, 2)
    qa_l = lanes & 7
    c = c * np.conj(twb[:, qa_l].T[:, None, :])
    c = c.reshape(L, R)
    X2 = np.zeros(G * G2, complex)
    for g in range(G):
        for lo in range(8):
            X2[g * G2 + u * S2 + Q2 * qa_l + lo] = c[:, 8 * g + lo]
    b = np.zeros((L, R), complex)
    llo = lanes & 7
    for g in range(G):
        for qa in range(8):
            b[:, 8 * g + qa] = X2[g * G2 + u * S2 + Q2 * qa + llo]
    b = dft(b.reshape(L, G, 8), +1, 2).reshape(L, R)
    X = np.zeros(R * S1, complex)
    for g in range(G):
        for lhi in range(8):
            X[(u + 8 * g) * S1 + llo + 8 * lhi] = b[:, 8 * g + lhi]
    a = np.zeros((L, R), complex)
    for rho in range(R):
        a[:, rho] = X[ka_of_reg(rho) * S1 + lanes]
    a = a * np.conj(twa.T)
    a = a.reshape(L, G, 8)                                   # [lane][r][p]
    t = dft(a, +1, 2)                                        # over p -> ja
    t = t * np.exp(+2j * np.pi * np.outer(np.arange(G), np.arange(8)) / R)[None]
    return dft(t, +1, 1).reshape(L, R)                       # over r -> jb: reg ja + 8 jb = j


def main():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    twa, twb = tables()
    v = x.reshape(R, L).T.copy()
    z = forward(v, twa, twb)
    Xref = np.fft.fft(x)
    kk = np.array([[k_of(l, r) for r in range(R)] for l in range(L)])
    assert sorted(kk.ravel().tolist()) == list(range(N))
    err = np.abs(z - Xref[kk]).max()
    y = inverse(z, twa, twb) / N
    err2 = np.abs(y - v).max()
    print("forward max err", err, "round trip max err", err2)
    assert err < 1e-9 and err2 < 1e-12


if __name__ == "__main__":
    main()
