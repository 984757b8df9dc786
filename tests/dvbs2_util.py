"""Test helper: valid DVB-S2 LDPC code words, built from the parity-check structure itself (satdump_amd/csrc/dvbs2_tables.inc = the
standard's address tables): check row r (natural order) = its data bits + parity r + parity r-1, so parity[r] = parity[r-1] ^ XOR(data
bits of row r) -- the IRA accumulator of ETSI EN 302 307 5.3.2. (The reference's own BBFrameLDPC::encode is not used for this: for the
tables in which a bit group addresses the same check residue twice its output does not satisfy the reference DECODER's checks --
measured: 1/3, 1/2, 3/5, 8/9 normal and most short codes come back 'not converged' noise-free -- so it cannot make test vectors.)"""
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def _tables():
    if "t" not in _cache:
        src = open(os.path.join(_HERE, "..", "satdump_amd", "csrc", "dvbs2_tables.inc")).read()
        grp = [int(v) for v in re.findall(r"\d+", re.search(r"S2_GRP\[\] = \{(.*?)\};", src, re.S).group(1).split("\n", 1)[1])]
        pos_body = re.sub(r"//.*", "", re.search(r"S2_POS\[\] = \{(.*?)\};", src, re.S).group(1))
        pos = [int(v) for v in re.findall(r"\d+", pos_body)]
        tabs = []
        for m in re.finditer(r'\{"(\w+)", (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}', src):
            name, M, N, K, cn, lt, ng, go, po, npos = m.group(1), *[int(v) for v in m.groups()[1:]]
            tabs.append(dict(name=name, M=M, N=N, K=K, groups=[(grp[2 * (go + i)], grp[2 * (go + i) + 1]) for i in range(ng)], pos=pos[po:po + npos]))
        _cache["t"] = tabs
    return _cache["t"]


def table(framesize, rate_code):
    normal = [0, 1, 2, 3, 4, 5, 6, 7, 8, -1, 9, 10]
    short = [11, 12, 13, 14, 15, 16, 17, 18, 19, -1, 20, -1]
    return _tables()[(normal if framesize == 0 else short)[rate_code]]


def check_rows(framesize, rate_code):
    """(row index, data bit) pairs of every edge between a data bit and a check row (natural row order)."""
    key = ("rows", framesize, rate_code)
    if key not in _cache:
        t = table(framesize, rate_code)
        M, R = t["M"], t["N"] - t["K"]
        q = R // M
        rows, bits = [], []
        bit, p = 0, 0
        for deg, nrows in t["groups"]:
            for _ in range(nrows):
                a = np.array(t["pos"][p:p + deg], dtype=np.int64)
                p += deg
                for m in range(M):
                    rows.append((a + m * q) % R)
                    bits.append(np.full(deg, bit, dtype=np.int64))
                    bit += 1
        _cache[key] = (np.concatenate(rows), np.concatenate(bits), R)
    return _cache[key]


def encode(framesize, rate_code, data_bits):
    """data_bits uint8 [nframes, K] -> code words uint8 [nframes, N] (systematic, parity by accumulation)."""
    rows, bits, R = check_rows(framesize, rate_code)
    nf = data_bits.shape[0]
    acc = np.zeros((nf, R), dtype=np.uint8)
    for f in range(nf):
        np.bitwise_xor.at(acc[f], rows, data_bits[f, bits])
    par = np.bitwise_xor.accumulate(acc, axis=1)
    return np.concatenate([data_bits, par], axis=1)
