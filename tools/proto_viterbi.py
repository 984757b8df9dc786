"""numpy model of the wave-parallel ACS used by the HIP kernel (rotating lane<->state layout,
lane-order decision ballots, segment-parallel traceback). Validated against the oracle."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import pyref

def rotl6(x, n):
    n %= 6
    return ((x << n) | (x >> (6 - n))) & 63
def rotr6(x, n):
    n %= 6
    return ((x >> n) | (x << (6 - n))) & 63
def par(x): return bin(x).count("1") & 1

lanes = np.arange(64)

def decode_block(syms, F, start_state, first):
    nsteps = F + 6
    X = np.full(64, 31 if first else 63, dtype=np.int64)
    if not first:
        X[start_state] = 0            # lane l holds state l at t=0
    ballots = np.zeros(nsteps, dtype=np.uint64)
    for t in range(nsteps):
        p = t % 6
        st = rotl6(lanes, p)          # old state held by each lane
        i = st & 31
        hi = st >> 5
        b0 = np.array([par((2 * int(ii)) & 79) for ii in i]); b1 = np.array([par((2 * int(ii)) & 109) for ii in i])
        s0, s1 = int(syms[2 * t]), int(syms[2 * t + 1])
        metric = (1 + np.where(b0, 255 - s0, s0) + np.where(b1, 255 - s1, s1)) >> 3
        partner = lanes ^ (32 >> p)
        mine = (X + metric) & 255
        other = (X[partner] + 63 - metric) & 255
        Y = np.minimum(mine, other)
        dec = np.where(hi == 1, other >= mine, mine >= other)
        Y = Y - Y.min()
        bal = 0
        for l in range(64):
            if dec[l]: bal |= (1 << l)
        ballots[t] = bal
        X = Y
    # end state: lane l holds state rotl6(l, nsteps)
    stf = rotl6(lanes, nsteps % 6)
    mn = X.min()
    endstate = int(stf[X == mn].min())
    return ballots, endstate

def dec_bit(ballots, t, state):
    lane = rotr6(state, (t + 1) % 6)   # new state at step t lives in lane sigma_{t+1}^{-1}(state)
    return (int(ballots[t]) >> lane) & 1

def traceback_serial(ballots, F, endstate):
    out = np.zeros(F, dtype=np.uint8); st = endstate; ret = 0
    for n in range(F - 1, -1, -1):
        k = dec_bit(ballots, n + 6, st)
        st = (st >> 1) | (k << 5)
        out[n] = k
        if n == F - 6: ret = st
    return out, ret

def traceback_parallel(ballots, F, endstate, D=96):
    L = -(-F // 64); L = -(-L // 32) * 32
    out = np.zeros(F, dtype=np.uint8)
    entry = np.full(64, -1); exitst = np.full(64, -1)
    top_step = F + 5
    for l in range(64):
        lo = 6 + l * L; hi_ = min(6 + (l + 1) * L, F + 6) - 1   # own steps [lo, hi_]
        if lo > F + 5: continue
        tstart = min(hi_ + D, top_step)
        st = endstate if tstart == top_step else 0
        for t in range(tstart, lo - 1, -1):
            if t == hi_: entry[l] = st
            k = dec_bit(ballots, t, st)
            if t <= hi_: out[t - 6] = k
            st = (st >> 1) | (k << 5)
        exitst[l] = st
    ok = True
    for l in range(1, 64):
        if entry[l - 1] >= 0 and exitst[l] >= 0 and entry[l - 1] != exitst[l]: ok = False
    return out, ok

if __name__ == "__main__":
    rng = np.random.default_rng(1)
    R = pyref.ref()
    for F in (4096, 1024, 12288):
        nb = 3
        stride = 2 * (F + 6)
        bits = rng.integers(0, 2, nb * F + 64).astype(np.uint8)
        from satdump_amd import synth
        coded = synth.conv_encode(bits)
        soft = np.clip(np.rint((coded.astype(float) * 2 - 1) * 60 + rng.standard_normal(len(coded)) * 40), -127, 127).astype(np.int64)
        u = soft + 127; u[u == 128] = 127
        syms = np.zeros(nb * stride, dtype=np.uint8)
        for b in range(nb):
            syms[b * stride: b * stride + 2 * F] = u[b * 2 * F:(b + 1) * 2 * F]
            syms[b * stride + 2 * F:(b + 1) * stride] = 128
        ref = R.ccdecoder(F, syms)
        st, first = 0, True
        allok = True
        for b in range(nb):
            ballots, e = decode_block(syms[b * stride:(b + 1) * stride], F, st, first)
            out, ret = traceback_serial(ballots, F, e)
            outp, ok = traceback_parallel(ballots, F, e)
            allok &= np.array_equal(out, ref[b * F:(b + 1) * F]) and np.array_equal(outp, out) and ok
            st, first = ret, False
        print("F", F, "parity", allok)
