"""Index-level model of the multi-pass LDS FFT used by the fast kernels
(sdr_channelizer_amd/csrc/pfb_fast.hpp).  Executable documentation: it checks
the position / twiddle formulas against numpy.fft for every plan the kernels
instantiate, and reports LDS bank conflicts of each pass under the MI355X
banking rules (ds_read_b64: 2x32-lane groups, 64 banks; ds_write_b64: 4x16-lane
groups, 32 banks).

Plan R = [R0..R_{n-1}], M = prod(R).  S_i = prod(R[i+1:]), K_i = prod(R[:i]).
Input of pass i lives at pos_i(n_i, item) = n_i*RS_i + item, item = kk*S_i + rest.
"""
import itertools
import sys
import numpy as np


def strides(R):
    n = len(R)
    S = [int(np.prod(R[i + 1:])) for i in range(n)]
    K = [int(np.prod(R[:i])) for i in range(n)]
    return S, K


def run_plan(x, R, RS):
    """x: (M,) complex in natural order. Returns y[k] = sum_n x[n] e^{+2 pi j n k / M}."""
    M = x.size
    S, K = strides(R)
    FS = max(R[i] * RS[i] for i in range(len(R)))
    buf = np.zeros(FS, dtype=complex)
    # FIR writes element n at pos_0
    for n in range(M):
        n0, rest = divmod(n, S[0])
        buf[n0 * RS[0] + rest] = x[n]
    out = np.zeros(M, dtype=complex)
    for i, Ri in enumerate(R):
        nxt = np.zeros_like(buf)
        for item in range(M // Ri):
            kk, rest = divmod(item, S[i])
            a = np.array([buf[n * RS[i] + item] for n in range(Ri)])
            b = np.array([sum(a[n] * np.exp(2j * np.pi * n * k / Ri) for n in range(Ri)) for k in range(Ri)])
            if i + 1 < len(R):
                b = b * np.exp(2j * np.pi * rest * np.arange(Ri) / (Ri * S[i]))
                n1, rest2 = divmod(rest, S[i + 1])
                for k in range(Ri):
                    item2 = (kk + k * K[i]) * S[i + 1] + rest2
                    nxt[n1 * RS[i + 1] + item2] = b[k]
            else:
                for k in range(Ri):
                    out[kk + k * K[i]] = b[k]
        buf = nxt
    return out


def conflicts(addrs_by_lane, kind):
    """addrs_by_lane: list of 64 element addresses (8-byte units) or None for idle lanes.
    Returns worst-case serialisation factor over the lane groups."""
    if kind == "read":   # ds_read_b64
        groups, nb = [range(0, 32), range(32, 64)], 64
    else:                # ds_write_b64
        groups, nb = [range(g * 16, g * 16 + 16) for g in range(4)], 32
    worst = 1
    for g in groups:
        per_bank = {}
        for l in g:
            a = addrs_by_lane[l]
            if a is None:
                continue
            for d in (0, 1):
                bank = (2 * a + d) % nb
                per_bank.setdefault(bank, set()).add(2 * a + d)
        if per_bank:
            worst = max(worst, max(len(s) for s in per_bank.values()))
    return worst


def check_conflicts(M, R, RS, FS, C, NT):
    S, K = strides(R)
    rep = []
    for i, Ri in enumerate(R):
        ipf = M // Ri
        items = C * ipf
        for it in range((items + NT - 1) // NT):
            for wave in range(NT // 64):
                lanes = [wave * 64 + l + it * NT for l in range(64)]
                for n in range(Ri):
                    ad = [None if w >= items else (w // ipf) * FS + n * RS[i] + (w % ipf) for w in lanes]
                    c = conflicts(ad, "read")
                    if c > 1:
                        rep.append((f"pass{i} read n={n}", c))
                if i + 1 < len(R):
                    for k in range(Ri):
                        ad = []
                        for w in lanes:
                            if w >= items:
                                ad.append(None); continue
                            fc, item = divmod(w, ipf)
                            kk, rest = divmod(item, S[i])
                            n1, rest2 = divmod(rest, S[i + 1])
                            item2 = (kk + k * K[i]) * S[i + 1] + rest2
                            ad.append(fc * FS + n1 * RS[i + 1] + item2)
                        c = conflicts(ad, "write")
                        if c > 1:
                            rep.append((f"pass{i} write k={k}", c))
    return rep


PLANS = {
    # name: (M, R, RS, FS, C, NT)
    "m64": (64, [8, 8], [8, 9], 72, 8, 64),
    "m128": (128, [16, 8], [8, 17], 136, 8, 64),     # final-pass reads 2-way
    "m256": (256, [16, 16], [17, 17], 272, 4, 64),
    "m1024": (1024, [16, 16, 4], [64, 68, 260], 1088, 8, 512),   # 8-wave lockstep plan and (C = 4) the team plan
    "m1024_16w": (1024, [8, 8, 16], [128, 128, 65], 1040, 8, 1024),
    "m56": (56, [8, 7], [7, 9], 71, 8, 64),           # the reference's fs*1e-6; 2-way on ~half the accesses
    "m560_9w": (560, [10, 8, 7], [56, 71, 82], 600, 7, 576),     # round(fs/0.1e6); 2-way on ~a third
    "m560_teams": (560, [14, 10, 4], [40, 60, 140], 600, 4, 320),
    "m32": (32, [8, 4], [4, 9], 36, 8, 64),
    "m16": (16, [4, 4], [4, 5], 20, 8, 64),
    "m8": (8, [4, 2], [2, 5], 10, 64, 64),
    "m10": (10, [5, 2], [2, 5], 10, 48, 64),
    "m20": (20, [10, 2], [2, 15], 42, 24, 64),
    "m40": (40, [8, 5], [5, 9], 45, 8, 64),
    # the other plausible radio rates (pfb_kernels_mixed.hip); row strides from tools/fft_plan_search.py
    "m12": (12, [6, 2], [2, 7], 38, 8, 64),
    "m24": (24, [12, 2], [2, 13], 26, 8, 64),
    "m25": (25, [5, 5], [5, 5], 25, 8, 64),
    "m30": (30, [10, 3], [3, 13], 39, 8, 64),
    "m48": (48, [16, 3], [3, 19], 57, 8, 64),
    "m50": (50, [10, 5], [5, 11], 55, 8, 64),
    "m80": (80, [16, 5], [5, 17], 101, 8, 64),
    "m96": (96, [16, 6], [6, 17], 102, 8, 64),
    "m100": (100, [10, 10], [10, 10], 106, 4, 64),
    "m112": (112, [16, 7], [7, 17], 135, 8, 64),
    "m120": (120, [12, 10], [10, 13], 138, 4, 64),
    "m160": (160, [16, 10], [10, 17], 170, 8, 128),
    "m200": (200, [10, 10, 2], [20, 20, 104], 210, 8, 256),
    "m250": (250, [10, 5, 5], [25, 51, 50], 274, 4, 256),
    "m280": (280, [7, 10, 4], [40, 28, 70], 296, 8, 320),
    "m320": (320, [8, 10, 4], [40, 34, 84], 340, 8, 320),
    "m400": (400, [10, 10, 4], [40, 40, 100], 424, 8, 448),
    "m500": (500, [10, 10, 5], [50, 51, 102], 516, 4, 256),
    "m512": (512, [16, 16, 2], [32, 33, 264], 528, 8, 256),
}

if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for name, (M, R, RS, FS, C, NT) in PLANS.items():
        x = rng.standard_normal(M) + 1j * rng.standard_normal(M)
        y = run_plan(x, R, RS)
        ref = np.fft.ifft(x) * M
        err = abs(y - ref).max() / abs(ref).max()
        rep = check_conflicts(M, R, RS, FS, C, NT)
        print(f"{name}: M={M} R={R} RS={RS} FS={FS} err={err:.2e} conflicts={rep if rep else 'none'}")
        assert err < 1e-12
