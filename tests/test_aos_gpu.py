"""The first step behind the CADUs on the device (satdump_amd/csrc/aos_demux.hip): VCDU header parse, virtual-channel selection and the
M_PDU packet demultiplexer against the reference's own ccsds_aos helpers compiled in place (oracle/ref_wrap_aos.cpp): byte work, bit-exact -- the same
packets (header, payload bytes, the frame whose work() call hands each out) for well-formed streams, streams with missing frames, idle frames, headers split
across frames, packets ending exactly on a zone's end, pointers outside the zone; any cut of the stream into calls."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    from satdump_amd import capi as c
    c.lib()
    return c


def make_stream(nframes, size=884, seed=0, cadu_bytes=1024, insert=0, drop=(), damage=(), lens=None, sec_ext=False):
    """CADUs of several virtual channels: VCID 5 carries Space Packets of random lengths packed into M_PDUs the standard's way (first header pointer =
    offset of the first packet header that STARTS in the zone, 2047 when none does), VCID 9 and idle frames (63) in between; `drop` = indices (of VCID 5's
    frames) left out, `damage` = indices whose pointer is overwritten with a value outside the zone."""
    rng = np.random.default_rng(seed)
    # the packet byte stream of VCID 5
    stream = bytearray()
    starts = []
    apids = [64, 65, 1200]
    count = 0
    while len(stream) < nframes * size + 4000:
        ln = int(lens[count % len(lens)]) if lens is not None else int(rng.choice([1, 2, 7, 20, 100, 400, 878, 879, 880, 1500, 3000], p=[.05, .05, .1, .15, .2, .15, .05, .05, .05, .1, .05]))
        apid = apids[count % 3]
        shf = int(rng.integers(0, 2))
        field = ln - 1 - (8 if (sec_ext and shf) else 0)
        if field < 0:
            field, ln = 0, 1 + (8 if (sec_ext and shf) else 0)
        hdr = bytes([(0 << 5) | (0 << 4) | (shf << 3) | (apid >> 8), apid & 0xFF, (3 << 6) | ((count >> 8) & 0x3F), count & 0xFF, field >> 8, field & 0xFF])
        starts.append(len(stream))
        stream += hdr + bytes(rng.integers(0, 256, ln, dtype=np.uint8))
        count += 1
    starts = np.array(starts)
    frames = []
    zoff = 10 + insert
    k5 = 0
    for f in range(nframes * 2):
        c = np.zeros(cadu_bytes, dtype=np.uint8)
        c[:4] = [0x1A, 0xCF, 0xFC, 0x1D]
        which = f % 3
        vcid = 5 if which != 1 else (9 if (f // 3) % 2 else 63)
        scid = 0x2A
        c[4] = (1 << 6) | (scid >> 2)
        c[5] = ((scid & 3) << 6) | vcid
        cnt = f * 7 + 3
        c[6], c[7], c[8] = (cnt >> 16) & 0xFF, (cnt >> 8) & 0xFF, cnt & 0xFF
        c[9] = (f % 5 == 0) << 7
        if vcid == 5:
            a = k5 * size
            zone = np.frombuffer(bytes(stream[a:a + size]), dtype=np.uint8)
            s = starts[(starts >= a) & (starts < a + size)]
            fhp = int(s[0] - a) if len(s) else 2047
            if k5 in damage:
                fhp = size + 5
            c[zoff] = (fhp >> 8) & 7
            c[zoff + 1] = fhp & 0xFF
            c[zoff + 2:zoff + 2 + size] = zone
            if k5 not in drop:
                frames.append(c)
            k5 += 1
            if k5 >= nframes:
                break
        else:
            c[zoff:] = rng.integers(0, 256, cadu_bytes - zoff, dtype=np.uint8)
            frames.append(c)
    return np.array(frames)


def _torch_helpers():
    import torch

    def to_dev(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
        return (t, t.data_ptr())

    def zeros_dev(n, dt):
        t = torch.zeros(n, dtype={np.uint8: torch.uint8, np.int32: torch.int32}[dt], device="cuda")
        return (t, t.data_ptr())

    return to_dev, (lambda d: d[0].cpu().numpy()), zeros_dev


def check_vcdu_and_select(capi, to_dev, from_dev, zeros_dev):
    ref = pyref.AosRef()
    cadus = make_stream(40, seed=1)
    L = capi.lib()
    d_c = to_dev(cadus)
    want = ref.vcdu(cadus)
    out = zeros_dev(len(cadus) * C.sizeof(capi.Vcdu), np.uint8)
    assert L.sdhip_aos_parse_vcdu_dev(0, C.c_void_p(d_c[1]), cadus.shape[1], len(cadus), C.c_void_p(out[1])) == 0
    got = np.frombuffer(from_dev(out).tobytes(), dtype=np.dtype([("version", "u1"), ("pad0", "u1"), ("scid", "<u2"), ("vcid", "u1"), ("pad1", "u1", 3), ("counter", "<u4"), ("replay", "u1"),
                                                                  ("pad2", "u1", 3)]))
    assert C.sizeof(capi.Vcdu) == got.dtype.itemsize
    assert np.array_equal(got["version"], want[:, 0]) and np.array_equal(got["scid"], want[:, 1]) and np.array_equal(got["vcid"], want[:, 2])
    assert np.array_equal(got["counter"], want[:, 3]) and np.array_equal(got["replay"], want[:, 4])
    for vcid in (5, 9, 63, 17):
        sel = zeros_dev(cadus.size, np.uint8)
        idx = zeros_dev(len(cadus), np.int32)
        n = L.sdhip_aos_select_vcid_dev(0, C.c_void_p(d_c[1]), cadus.shape[1], len(cadus), vcid, C.c_void_p(sel[1]), len(cadus), C.c_void_p(idx[1]))
        w = np.flatnonzero(want[:, 2] == vcid)
        assert n == len(w)
        assert np.array_equal(from_dev(idx)[:n], w) and np.array_equal(from_dev(sel)[: n * cadus.shape[1]].reshape(n, cadus.shape[1]), cadus[w])


def run_demux(capi, to_dev, from_dev, zeros_dev, frames, cuts, size=884, insert=None, sec_ext=False):
    L = capi.lib()
    h = L.sdhip_aos_demux_create(0, size, int(insert is not None), insert or 2, int(sec_ext))
    assert h, capi.last_error()
    hdrs, metas, pools = [], [], []
    base = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = frames[a:b]
        cap_p = len(part) * 140 + 16
        tab = (capi.AosPacket * cap_p)()
        pool = zeros_dev(len(part) * frames.shape[1] * 2 + 8192, np.uint8)
        used = C.c_uint64(0)
        d_c = to_dev(part) if len(part) else (None, 0)
        n = L.sdhip_aos_demux_work_dev(h, C.c_void_p(d_c[1]), frames.shape[1], len(part), tab, cap_p, C.c_void_p(pool[1]), len(part) * frames.shape[1] * 2 + 8192, C.byref(used))
        assert n >= 0, capi.last_error()
        pl = from_dev(pool)[: used.value]
        off = 0
        for k in range(n):
            p = tab[k]
            hdrs.append(bytes(p.header))
            metas.append((a + p.frame, p.payload_size, p.apid, p.packet_sequence_count, p.packet_length, p.sequence_flag))
            assert p.payload_offset == off
            off += p.payload_size
        pools.append(pl)
        base += len(part)
    L.sdhip_aos_demux_destroy(h)
    return hdrs, metas, (np.concatenate(pools) if pools else np.zeros(0, np.uint8))


CASES = [
    dict(nframes=60, seed=2),
    dict(nframes=60, seed=3, drop=(7, 8, 30)),
    dict(nframes=60, seed=4, damage=(5, 22), drop=(40,)),
    dict(nframes=50, seed=5, insert=2),
    dict(nframes=50, seed=6, sec_ext=True),
    dict(nframes=40, seed=7, lens=[878, 872, 1, 2, 3, 4, 5, 6, 7, 871, 2000, 877]),   # packets ending on / next to a zone's end, headers split across frames
    dict(nframes=30, seed=8, lens=[1] * 50 + [300]),                                   # more packet headers in a frame than the scan record holds
    dict(nframes=30, seed=9, size=400, cadu_bytes=512),
]


def check_demux(capi, to_dev, from_dev, zeros_dev, case):
    ref = pyref.AosRef()
    kw = dict(case)
    size, insert, sec_ext = kw.get("size", 884), kw.get("insert", 0), kw.get("sec_ext", False)
    cadus = make_stream(**kw)
    vc = cadus[(cadus[:, 5] & 0x3F) == 5]
    wh, wm, wp = ref.demux(vc, size, insert > 0, insert or 2, sec_ext)
    assert len(wh) > 20
    n = len(vc)
    for cuts in ([0, n], [0, 1, 2, 7, n // 2, n // 2 + 1, n], list(range(0, n + 1))):
        gh, gm, gp = run_demux(capi, to_dev, from_dev, zeros_dev, vc, cuts, size, insert if insert else None, sec_ext)
        assert len(gh) == len(wh), (len(gh), len(wh), cuts[:4])
        assert gh == [bytes(r) for r in wh]
        assert np.array_equal(np.array(gm, dtype=np.uint32).reshape(-1, 6), wm)
        assert np.array_equal(gp, wp)


def test_vcdu_and_select(capi):
    if not pyref.AosRef.available():
        pytest.skip("oracle/_ref/libsdref_aos.so not built")
    check_vcdu_and_select(capi, *_torch_helpers())


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_demux(capi, case):
    if not pyref.AosRef.available():
        pytest.skip("oracle/_ref/libsdref_aos.so not built")
    check_demux(capi, *_torch_helpers(), case)
