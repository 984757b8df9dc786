"""satdump_amd/synth_dvbs2.py (the DVB-S2 workload generator of tests and bench.py) pinned against the compiled reference: its BCH encoder and BB
scrambler equal the reference's (BBFrameBCH::encode, BBFrameDescrambler::work), and what it makes decodes on the reference's own receive chain
(S2PLSyncBlock -> S2PLLBlock -> S2BBToSoft -> BBFrameLDPC -> BBFrameBCH -> BBFrameDescrambler) to the BBFRAMEs that went in."""
import numpy as np
import pytest

from oracle import pyref
from satdump_amd import synth_dvbs2 as sd


def _need_ref():
    if not (pyref.Dvbs2Ref.available(False) and pyref.S2FrontRef.available()):
        pytest.skip("oracle/_ref/libsdref_dvbs2.so not built (needs /root/reference at build time)")


@pytest.mark.parametrize("fs,rate", [(0, 5), (1, 3), (0, 10), (0, 8)])
def test_bch_encoder_and_bb_scrambler_equal_the_reference(fs, rate):
    _need_ref()
    fec = pyref.Dvbs2Ref(False)
    kb, nb, _t = sd.bch_dims(fs, rate)
    _n, k = fec.dims(fs, rate)
    assert k == nb and kb == fec.bch_kbch(fs, rate)
    bb = np.zeros((3, k // 8), dtype=np.uint8)
    bb[:, :kb // 8] = np.random.default_rng(fs * 16 + rate).integers(0, 256, (3, kb // 8), dtype=np.uint8)
    mine = np.packbits(sd.bch_encode(fs, rate, np.unpackbits(bb[:, :kb // 8], axis=1)), axis=1)
    assert np.array_equal(mine, fec.bch_encode(fs, rate, bb.copy()))
    assert np.array_equal(fec.bb_descramble(fs, rate, bb.copy())[:, :kb // 8], sd.bb_scramble(bb[:, :kb // 8]))


@pytest.mark.parametrize("modcod,short,esn0_db", [(13, 0, 9.5), (12, 1, 9.0), (4, 1, 6.0)])
def test_plframes_decode_on_the_reference_chain(modcod, short, esn0_db):
    _need_ref()
    from tests.test_dvbs2_gpu import _s2_reference_chain
    c = sd.modcod_cfg(modcod, short)
    assert c == {**c, **{k: v for k, v in pyref.S2FrontRef().cfg(modcod, short, 0).items() if k in c}}
    bb = sd.bbframes_random(short, c["rate"], 4, seed=3)
    x = sd.symbol_stream(sd.plframes(modcod, short, bb), seed=5, lead=300, cfo=0.0003, esn0_db=esn0_db, amplitude=0.7 if c["bits"] != 2 else 2.0 / 3.0)
    out, _tr, _corr, fr, _rp, _st = _s2_reference_chain(modcod, short, x)
    assert len(fr) >= 4 and [bytes(r) for r in out[:4]] == [bytes(r) for r in bb]
