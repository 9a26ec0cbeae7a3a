"""CPU check of the ALGORITHM behind k_exposure_sum (csrc/ycge_post.hip): the reference's serial binary32 sum
(ToneMapper.cs:63-77) evaluated chunk by chunk as integer arithmetic inside one binade, with the serial loop as the fallback.
This is a line-for-line Python twin of the kernel's three phases; the kernel itself is held to the oracle's exposure bits by
the -m gpu post-stage tests."""
import math
import struct

import numpy as np
import pytest


def f2u(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def u2f(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def chunked_sum(terms, CH=512):
    n = len(terms)
    nch = (n + CH - 1) // CH
    csum = [float(np.sum(terms[c * CH:(c + 1) * CH].astype(np.float64))) for c in range(nch)]       # phase A
    chunks, pre = [], 0.0
    for c in range(nch):                                                                            # phase B
        s0 = pre
        pre += csum[c]
        neg = 1 if s0 < 0 else 0
        e = math.frexp(abs(s0))[1] - 1 if abs(s0) >= 1e-30 else -1000
        C = dict(e=e, neg=neg, d=[0, 0], lo=[0, 0], hi=[0, 0])
        if not np.any(terms[c * CH:(c + 1) * CH]):
            C["e"] = -2000                      # nothing but + 0.0f: leaves any sum as it is
        elif -12 <= e < 100:
            inv_u = (2.0 ** (23 - e)) * (-1.0 if neg else 1.0)
            d, lo, hi, p = [0, 0], [0, 0], [0, 0], [0, 1]
            for t in terms[c * CH:(c + 1) * CH]:
                x = float(t) * inv_u
                fx = math.floor(x)
                fr = x - fx
                ifx = int(fx)
                for k in (0, 1):
                    inc = ((p[k] + ifx) & 1) if fr == 0.5 else (1 if fr > 0.5 else 0)
                    d[k] += ifx + inc
                    p[k] = (p[k] + ifx + inc) & 1
                    lo[k] = min(lo[k], d[k])
                    hi[k] = max(hi[k], d[k])
            C.update(d=d, lo=lo, hi=hi)
        chunks.append(C)
    # groups of 16 chunks composed into one map (the kernel's two-level walk)
    groups = []
    for g0 in range(0, nch - 15, 16):
        ok = all(chunks[g0 + j]["e"] == -2000 or (chunks[g0 + j]["e"] == chunks[g0]["e"] and chunks[g0 + j]["neg"] == chunks[g0]["neg"] and chunks[g0]["e"] > -1000)
                 for j in range(16))
        G = dict(ok=ok, d=[0, 0], lo=[0, 0], hi=[0, 0])
        for pin in (0, 1):
            D, L, H, P = 0, 0, 0, pin
            for j in range(16):
                Ck = chunks[g0 + j]
                if Ck["e"] == -2000:
                    continue
                L = min(L, D + Ck["lo"][P]); H = max(H, D + Ck["hi"][P])
                dk = Ck["d"][P]
                D += dk
                P = (P + dk) & 1
            G["d"][pin], G["lo"][pin], G["hi"][pin] = D, L, H
        groups.append(G)
    s, n_serial = np.float32(0), 0
    c = 0
    while c < nch:                                                                                  # phase C
        C = chunks[c]
        bits = f2u(float(s))
        if C["e"] == -2000:
            c += 1
            continue
        if c % 16 == 0 and c // 16 < len(groups) and groups[c // 16]["ok"]:
            G = groups[c // 16]
            ex0 = (bits >> 23) & 0xff
            m0 = (bits & 0x7fffff) | 0x800000
            p0 = m0 & 1
            if ex0 != 0 and ex0 - 127 == C["e"] and (bits >> 31) == C["neg"] and m0 + G["lo"][p0] > (1 << 23) and m0 + G["hi"][p0] < (1 << 24):
                s = np.float32(u2f((bits & 0xff800000) | ((m0 + G["d"][p0]) & 0x7fffff)))
                c += 16
                continue
        ex = (bits >> 23) & 0xff
        m = (bits & 0x7fffff) | 0x800000
        p = m & 1
        fast = ex != 0 and ex - 127 == C["e"] and (bits >> 31) == C["neg"] and m + C["lo"][p] > (1 << 23) and m + C["hi"][p] < (1 << 24)
        if fast:
            s = np.float32(u2f((bits & 0xff800000) | ((m + C["d"][p]) & 0x7fffff)))
        else:
            for t in terms[c * CH:(c + 1) * CH]:
                s = np.float32(s + t)
            n_serial += 1
        c += 1
    return s, n_serial, nch


@pytest.mark.parametrize("trial", range(6))
def test_chunked_evaluation_equals_the_serial_binary32_sum(trial):
    rng = np.random.default_rng(5 + trial)
    n = 30000 + trial * 3777
    lum = rng.random(n).astype(np.float32) ** np.float32(2 + trial)
    terms = np.log(np.float32(1e-6) + lum).astype(np.float32)
    if trial % 2:
        terms[rng.random(n) < 0.3] = np.float32(0)                      # skipped samples contribute +0
        terms[:7000] = np.float32(0); terms[20000:23000] = np.float32(0)  # whole chunks of sky, at the start and inside
    if trial == 4:
        terms = np.abs(terms) * np.float32(0.01)                        # a positive, slowly growing sum
    if trial == 5:
        terms = (terms + np.float32(5.0)).astype(np.float32)            # mixed signs
    idx = rng.integers(0, n, 1500)                                      # exact ties: dyadic terms
    terms[idx] = (rng.integers(-64, 64, 1500) / np.float32(32.0)).astype(np.float32)
    ref = np.float32(0)
    for t in terms:
        ref = np.float32(ref + t)
    got, n_serial, nch = chunked_sum(terms)
    assert f2u(float(ref)) == f2u(float(got))
    assert n_serial < nch // 3              # the fast path carries the bulk
