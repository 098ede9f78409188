"""CPU-side checks of the product library: it loads, exports every symbol include/*.h
declares, and its host range coder is bit-exact against the golden known answers and the
oracle.  No GPU compute is called here."""
import os
import re
import subprocess

import numpy as np
import pytest

from util import ROOT, kat_plane, sha


@pytest.fixture(scope="module")
def api():
    from waverange_amd import api as a
    from waverange_amd import build
    build.build(verbose=False)
    a.lib()
    return a


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "waverange_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)  # preprocessor lines
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_header_symbols_exported(api):
    syms = declared_symbols()
    assert {"setup_wr", "encoding_wrap", "decoding_wrap", "setup_wr_f", "encoding_wrap_f",
            "decoding_wrap_f", "wr_encode_device", "wr_decode_device"} <= set(syms)
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH]).decode()
    exported = set(line.split()[-1] for line in out.splitlines() if " T " in line)
    missing = [s for s in syms if s not in exported]
    assert not missing, missing


def test_drop_in_alias_exists(api):
    assert os.path.exists(os.path.join(os.path.dirname(api.LIB_PATH), "libwaverange.so"))


def test_setup_wr(api):
    assert api.setup_wr(64, 64, 64) == (8, 8 * 64 ** 3)
    assert api.setup_wr(3, 3, 3) == (8, 8 * 1024)


def test_range_coder_golden(api, golden):
    for key, g in golden["G3"].items():
        kind, n = key.rsplit("_", 1)
        p = kat_plane(kind, int(n))
        s = api.range_encode(p)
        assert s.size == g["length"] and sha(s) == g["stream_sha256"], key
        back, got = api.range_decode(s, p.size)
        assert got == p.size and np.array_equal(back, p), key


def test_range_coder_vs_oracle_random(api, oracle):
    rs = np.random.RandomState(11)
    for n in (1, 2, 3, 255, 59999, 60000, 60001, 120000, 180001, 300007):
        for kind in range(4):
            if kind == 0:
                p = rs.randint(0, 256, n)
            elif kind == 1:
                p = np.minimum(rs.geometric(0.3, n), 255)
            elif kind == 2:
                p = np.full(n, 255)
                p[::7] = 254
            else:
                p = np.zeros(n)
            p = p.astype(np.uint8)
            s = api.range_encode(p)
            assert np.array_equal(s, oracle.range_encode(p)), (n, kind)
            back, got = api.range_decode(s, n)
            assert got == n and np.array_equal(back, p), (n, kind)
            ob, og = oracle.range_decode(s, n)
            assert og == n and np.array_equal(ob, p)


@pytest.mark.parametrize("count", [1, 2, 3, 4, 7])
def test_interleaved_planes_same_bytes_as_oracle(api, oracle, count):
    """wr_range_encode_multi / wr_range_decode_multi (several planes in one symbol loop) against the
    oracle's coder plane by plane; the planes are shaped like real bit planes (one dominant symbol, two
    symbols, noise) so that the decoder's division-free shortcut, its bucket table and its padded-tail
    path all run.  n is not a multiple of the block size and > 3 blocks of margin."""
    n = 60000 * 7 + 4321
    rs = np.random.RandomState(count)
    u = rs.random_sample(n)
    shapes = [np.where(u < 0.9997, 121, rs.randint(0, 256, n)),
              np.where(u < 0.737, 189, np.where(u < 0.994, 190, rs.randint(0, 256, n))),
              rs.randint(0, 256, n), np.where(u < 0.5, 255, 254),  # 255 = largest symbol is also the most probable
              np.clip(np.rint(rs.normal(128, 12, n)), 0, 255), rs.randint(0, 4, n), np.full(n, 7)]
    planes = [shapes[i % len(shapes)].astype(np.uint8) for i in range(count)]
    streams = api.range_encode_multi(planes)
    for p, s in zip(planes, streams):
        assert np.array_equal(s, oracle.range_encode(p))
    back, got = api.range_decode_multi(streams, n)
    assert got == [n] * count
    for p, b in zip(planes, back):
        assert np.array_equal(p, b)


def test_range_decoder_rejects_garbage(api):
    rs = np.random.RandomState(5)
    junk = rs.randint(0, 256, 4096).astype(np.uint8)
    # must neither crash nor write beyond the n symbols it was given room for, whatever the stream claims to hold
    n, guard = 1000, 4096
    out = np.full(n + guard, 0xEE, dtype=np.uint8)
    got = api.lib().wr_range_decode(junk.ctypes.data, junk.size, out.ctypes.data, n)
    assert np.all(out[n:] == 0xEE), "decoder wrote past the room it was given"
    assert got == 2 ** 64 - 1 or got >= 0   # a failure mark or a symbol count; the codec layer compares it with nx*ny*nz
    p = kat_plane("skewed", 1000)
    s = api.range_encode(p)
    _, got = api.range_decode(s[: s.size // 2], 1000)  # truncated stream: no out-of-bounds read
    assert isinstance(got, int)


def test_no_gpu_fails_loudly(api):
    """Without a usable GPU the compute path must raise, not fall back to a CPU path."""
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(api.WaveRangeError):
        api.Context(0)


def test_reference_export_set_is_complete(api):
    """The 24 unmangled symbols the reference's libwaverange.so exports (nm -D)."""
    want = {"setup_wr", "encoding_wrap", "decoding_wrap", "setup_wr_f", "encoding_wrap_f", "decoding_wrap_f",
            "waveletcdf97_3d", "ind_p2w_3d", "start_encoding", "encode_freq", "encode_shift", "done_encoding",
            "start_decoding", "decode_culfreq", "decode_culshift", "decode_update", "decode_byte", "decode_short",
            "done_decoding", "init_databuf", "free_databuf", "countblock", "readcounts", "coderversion"}
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH]).decode()
    have = set(line.split()[-1] for line in out.splitlines())
    assert want <= have, want - have
    refso = os.path.join(ROOT, "oracle", "_ref", "libwaverange_ref.so")
    if os.path.exists(refso):
        r = subprocess.check_output(["nm", "-D", "--defined-only", refso]).decode()
        ref_syms = set(line.split()[-1] for line in r.splitlines() if line.split()[-2] in "TBD")
        assert {s for s in ref_syms if not s.startswith("_")} <= have
    # ... and nothing beyond them but the wr_* API the header declares: no C++ internals, no kernel stubs, no unprefixed
    # helper a host program's own symbols could collide with (-fvisibility=hidden + csrc/exports.map)
    extra = {s for s in have if s not in want and not s.startswith("wr_")}
    assert not extra, sorted(extra)
    undeclared = {s for s in have if s.startswith("wr_")} - set(declared_symbols())
    assert not undeclared, sorted(undeclared)


def test_rngcod13_primitives_and_index_map(api, oracle, golden):
    """Part 1b symbols: drive the exported primitives with the reference's block model
    (wrappers.cpp:68-149) and compare with the oracle's stream; decode it back; ind_p2w_3d."""
    import ctypes as C
    L = api.lib()

    class RC(C.Structure):
        _fields_ = [("low", C.c_uint), ("range", C.c_uint), ("help", C.c_uint), ("buffer", C.c_ubyte),
                    ("bytecount", C.c_uint), ("databuf", C.POINTER(C.c_ubyte)), ("datalen", C.c_ulong),
                    ("datapos", C.c_ulong)]
    for f in ("start_encoding", "encode_freq", "encode_shift", "init_databuf", "free_databuf", "decode_update"):
        getattr(L, f).restype = None
    L.start_encoding.argtypes = [C.POINTER(RC), C.c_char, C.c_ulong]
    L.encode_freq.argtypes = L.encode_shift.argtypes = L.decode_update.argtypes = [C.POINTER(RC)] + [C.c_uint] * 3
    L.done_encoding.argtypes = L.start_decoding.argtypes = L.done_decoding.argtypes = [C.POINTER(RC)]
    L.init_databuf.argtypes = [C.POINTER(RC), C.c_ulong]
    L.free_databuf.argtypes = [C.POINTER(RC)]
    L.decode_culfreq.argtypes = [C.POINTER(RC), C.c_uint]
    L.decode_culfreq.restype = C.c_uint
    L.decode_short.argtypes = [C.POINTER(RC)]
    L.decode_short.restype = C.c_ushort
    p = kat_plane("skewed", 1000)
    rc = RC()
    L.init_databuf(C.byref(rc), 4096)
    L.start_encoding(C.byref(rc), b"\x00", 0)
    L.encode_freq(C.byref(rc), 1, 1, 2)
    counts = np.bincount(p, minlength=256)
    for c in counts:
        L.encode_shift(C.byref(rc), 1, int(c), 16)
    cum = np.concatenate([[0], np.cumsum(counts)])
    for ch in p:
        L.encode_freq(C.byref(rc), int(counts[ch]), int(cum[ch]), p.size)
    L.encode_freq(C.byref(rc), 1, 0, 2)
    L.done_encoding(C.byref(rc))
    stream = bytes(rc.databuf[: rc.datapos])
    assert stream.hex() == golden["G3"]["skewed_1000"]["stream_hex"]
    # decode with the primitives
    rc.datapos = 0
    L.start_decoding(C.byref(rc))
    assert L.decode_culfreq(C.byref(rc), 2) == 1
    L.decode_update(C.byref(rc), 1, 1, 2)
    got_counts = [L.decode_short(C.byref(rc)) for _ in range(256)]
    assert got_counts == [int(c) for c in counts]
    back = []
    for _ in range(p.size):
        cf = L.decode_culfreq(C.byref(rc), p.size)
        s = int(np.searchsorted(cum, cf, side="right") - 1)
        L.decode_update(C.byref(rc), int(counts[s]), int(cum[s]), p.size)
        back.append(s)
    assert back == [int(v) for v in p]
    L.free_databuf(C.byref(rc))
    L.ind_p2w_3d.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)] * 4
    for n1, n2, n3, i1, i2, i3, l, w1, w2, w3 in golden["ind_p2w_3d"]:
        o = [C.c_int() for _ in range(4)]
        L.ind_p2w_3d(4, n1, n2, n3, i1, i2, i3, *[C.byref(v) for v in o])
        assert tuple(v.value for v in o) == (l, w1, w2, w3)


def test_coder_pool_same_bytes_as_oracle(oracle):
    """The process-wide coder pool (wr_set_coder_pool): planes of many lengths and kinds, submitted from several
    threads at once, coded by 1, 2 and 3 workers with 2, 3 and 4 decoder streams per loop -- streams join and
    leave the interleaved loops at block boundaries.  Every stream must equal the oracle's bytes, every plane
    must come back."""
    import threading
    from waverange_amd import api
    rs = np.random.RandomState(5)
    planes = []
    for n in (1, 59999, 60000, 60001, 120000, 150000, 333333, 600000, 777777, 1200000):
        planes.append(rs.randint(0, 256, n).astype(np.uint8))                                     # noise
        planes.append(rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]))            # two symbols
        planes.append(np.where(rs.random_sample(n) < 0.999, 7, rs.randint(0, 16, n)).astype(np.uint8))  # one dominant
        planes.append(rs.choice(np.array([3, 4, 5, 250], np.uint8), size=n))                      # four symbols
    want = [oracle.range_encode(p) for p in planes]
    queued0 = api.stat(api.STAT_POOL_QUEUE_MS)
    try:
        for workers, streams in ((1, 2), (2, 3), (3, 4)):
            api.set_coder_pool(workers, streams)
            errors = []

            def client(sel):
                try:
                    mine = [planes[i] for i in sel]
                    enc = api.range_encode_pool(mine)
                    for i, e in zip(sel, enc):
                        assert np.array_equal(e, want[i]), ("encode", i, planes[i].size)
                    dec, got = api.range_decode_pool(enc, [p.size for p in mine])
                    for i, d, g in zip(sel, dec, got):
                        assert g == planes[i].size and np.array_equal(d, planes[i]), ("decode", i)
                except Exception as exc:  # noqa: BLE001
                    errors.append(exc)

            idx = list(range(len(planes)))
            rs.shuffle(idx)
            ths = [threading.Thread(target=client, args=(idx[k::4],)) for k in range(4)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            assert not errors, errors[:2]
        # damaged and truncated streams next to good ones: reported per plane, the others unharmed
        api.set_coder_pool(2, 4)
        good = [planes[30], planes[25], planes[20]]
        enc = api.range_encode_pool(good)
        bad = enc[1].copy()
        bad[len(bad) // 2: len(bad) // 2 + 40] ^= 0xA5
        dec, got = api.range_decode_pool([enc[0], bad, enc[2][: len(enc[2]) // 2]], [p.size for p in good])
        assert got[0] == good[0].size and np.array_equal(dec[0], good[0])
        assert got[2] != good[2].size
        # 40 planes on 1-3 workers: most of them waited in a queue before a worker had room (wr_stat: milliseconds, summed)
        assert api.stat(api.STAT_POOL_QUEUE_MS) > queued0
    finally:
        api.set_coder_pool(0)


def test_avx512_decoder_loop_same_symbols(oracle):
    """The 16-lane AVX-512 decoder loop for dominant-symbol planes (wr_rangecoder_avx512.cpp): two-symbol planes with
    the pair anywhere in the alphabet (incl. the largest symbol present, whose interval is open-ended), one
    dominant symbol with rare others (scalar look-up on that lane), a noise plane (every block falls back to
    the scalar loop), lengths around the block size, more planes than lanes.  Streams come from the oracle."""
    from waverange_amd import api
    rs = np.random.RandomState(3)
    planes = []
    for n in (1, 59999, 60000, 60001, 300000, 300123, 420000, 419999):
        planes.append(rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]))
        planes.append(rs.choice(np.array([254, 255], np.uint8), size=n))
        planes.append(rs.choice(np.array([0, 255], np.uint8), size=n, p=[0.05, 0.95]))
        planes.append(np.where(rs.random_sample(n) < 0.9997, 128, rs.randint(120, 136, n)).astype(np.uint8))
        planes.append(np.full(n, 9, np.uint8))
    planes.append(rs.randint(0, 256, 300000).astype(np.uint8))
    planes.append(rs.choice(np.array([3, 4, 5, 250], np.uint8), size=300000))
    enc = [oracle.range_encode(p) for p in planes]
    try:
        dec, got = api.range_decode_vec(enc, [p.size for p in planes])
    except api.WaveRangeError:
        pytest.skip("no AVX-512 on this CPU")
    for i, (d, g, p) in enumerate(zip(dec, got, planes)):
        assert g == p.size and np.array_equal(d, p), (i, p.size)
    # the same planes and noise planes of several shapes (every block of those falls back to the scalar loop of its lane):
    # more planes than lanes, streams joining and leaving at block boundaries
    more = [rs.randint(0, 256, 300000 + 777 * k).astype(np.uint8) for k in range(6)]
    more += [np.clip(np.rint(rs.normal(128, 30, 360000)), 0, 255).astype(np.uint8), rs.randint(0, 2, 240000).astype(np.uint8) * 255,
             np.where(rs.random_sample(300000) < 0.5, 255, rs.randint(0, 256, 300000)).astype(np.uint8)]  # the largest symbol is frequent
    aplanes, aenc = planes + more, enc + [oracle.range_encode(p) for p in more]
    dec, got = api.range_decode_vec(aenc, [p.size for p in aplanes])
    for i, (d, g, p) in enumerate(zip(dec, got, aplanes)):
        assert g == p.size and np.array_equal(d, p), ("more planes than lanes", i, p.size)
    # the encoder's vector loop (candidate compares; rare other symbols through the lane's table; noise planes and
    # partial blocks through the scalar code of their stream)
    planes.append(np.where(rs.random_sample(300000) < 0.995, 255, rs.randint(0, 256, 300000)).astype(np.uint8))
    enc.append(oracle.range_encode(planes[-1]))
    venc = api.range_encode_vec(planes)
    for i, (a, b) in enumerate(zip(venc, enc)):
        assert np.array_equal(a, b), ("vector encode", i, planes[i].size)
    # through the pool as well (dominant-symbol planes take the decoder's vector route there, every plane the encoder's)
    api.set_coder_pool(2, 4)
    try:
        penc = api.range_encode_pool(planes)
        for i, (a, b) in enumerate(zip(penc, enc)):
            assert np.array_equal(a, b), ("pool encode", i, planes[i].size)
        dec, got = api.range_decode_pool(enc, [p.size for p in planes])
        for i, (d, g, p) in enumerate(zip(dec, got, planes)):
            assert g == p.size and np.array_equal(d, p), ("pool", i, p.size)
    finally:
        api.set_coder_pool(0)


@pytest.mark.parametrize("n", [1, 59999, 60000, 120000, 180001, 420000, 433333])
def test_windowed_symbol_access_same_bytes_as_oracle(oracle, n):
    """Planes that live in device memory reach the coder through windows of a pinned ring (wr_rangecoder.h,
    PlaneWindow).  Here the windows are 1 or 2 blocks of plain host buffers (the one handed out before is poisoned):
    streams and symbols must equal the oracle's on every loop -- scalar groups, the pool, the 16-lane loops --
    for noise, dominant-symbol and constant planes, lengths at and around block and window boundaries (incl. the
    empty trailing block of a plane that ends on a block boundary), and a corrupted stream next to healthy ones."""
    from waverange_amd import api
    rs = np.random.RandomState(n % 1000)
    planes = [rs.randint(0, 256, n).astype(np.uint8),
              rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.9, 0.1]),
              np.where(rs.random_sample(n) < 0.999, 255, rs.randint(0, 256, n)).astype(np.uint8),
              np.full(n, 7, np.uint8),
              rs.randint(100, 140, n).astype(np.uint8)]
    want = [oracle.range_encode(p) for p in planes]
    api.set_coder_pool(3, 4)
    try:
        for chunk in (60000, 120000):
            for mode in (0, 1, 2):
                try:
                    enc = api.range_encode_windowed(planes, chunk, mode)
                    dec, got = api.range_decode_windowed(want, n, chunk, mode)
                except api.WaveRangeError:
                    assert mode >= 2  # no AVX-512 on this CPU
                    continue
                for i, (a, b) in enumerate(zip(enc, want)):
                    assert np.array_equal(a, b), ("encode", chunk, mode, i)
                for i, (d, g, p) in enumerate(zip(dec, got, planes)):
                    assert g == n and np.array_equal(d, p), ("decode", chunk, mode, i)
        if n > 100000:  # one damaged stream among healthy ones: the others are untouched, nothing crashes
            bad = [w.copy() for w in want]
            bad[0][len(bad[0]) // 2:] ^= 0x5A
            for mode in (0, 1):
                dec, got = api.range_decode_windowed(bad, n, 60000, mode)
                for i in range(1, len(planes)):
                    assert got[i] == n and np.array_equal(dec[i], planes[i]), ("healthy next to damaged", mode, i)
    finally:
        api.set_coder_pool(0)


def test_streams_change_workers_same_bytes(oracle):
    """The coder pool's idle workers take over half of the streams of the fullest session at a block boundary (wr_rangecoder.cpp,
    Pool::offer / next).  Many long planes submitted at once to a pool whose first worker grabs them all into one 16-lane
    session: the other workers must end up with streams (the counter moves) and every stream and every decoded plane must
    equal the oracle's -- encoder sessions, decoder sessions of both kinds, whole planes and windowed ones."""
    from waverange_amd import api
    rs = np.random.RandomState(11)
    n = 60000 * 12 + 4321
    planes = []
    for k in range(14):
        if k % 3 == 0:
            planes.append(rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]))
        elif k % 3 == 1:
            planes.append(np.where(rs.random_sample(n) < 0.999, 255, rs.randint(0, 256, n)).astype(np.uint8))
        else:
            planes.append(rs.randint(0, 256, n).astype(np.uint8))
    want = [oracle.range_encode(p) for p in planes]
    api.set_coder_pool(4, 4)
    try:
        before = api.stat(api.STAT_POOL_STREAMS_MOVED)
        for rep in range(3):
            enc = api.range_encode_pool(planes)
            for i, (a, b) in enumerate(zip(enc, want)):
                assert np.array_equal(a, b), ("encode", rep, i)
            dec, got = api.range_decode_pool(want, [n] * len(planes))
            for i, (d, g, p) in enumerate(zip(dec, got, planes)):
                assert g == n and np.array_equal(d, p), ("decode", rep, i)
            enc = api.range_encode_windowed(planes, 120000, 1)
            for i, (a, b) in enumerate(zip(enc, want)):
                assert np.array_equal(a, b), ("windowed encode", rep, i)
            dec, got = api.range_decode_windowed(want, n, 120000, 1)
            for i, (d, g, p) in enumerate(zip(dec, got, planes)):
                assert g == n and np.array_equal(d, p), ("windowed decode", rep, i)
        moved = api.stat(api.STAT_POOL_STREAMS_MOVED) - before
        if api.lib().wr_range_decode_vec(0, None, None, None, None, None) == 0:   # AVX-512 sessions exist on this CPU
            assert moved > 0, "no stream ever changed workers"
    finally:
        api.set_coder_pool(0)


SMALL_SESSIONS = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from waverange_amd import api
from oracle.loader import Oracle
o = Oracle()
rs = np.random.RandomState(17)
lens = (60000 * 5 + 17, 60000 * 3, 60000 * 7 + 59999, 60000 * 4 + 1, 60000 * 6)
planes = [rs.choice(np.array([127, 128], np.uint8), size=lens[0], p=[0.8, 0.2]),
          np.where(rs.random_sample(lens[1]) < 0.999, 255, rs.randint(0, 256, lens[1])).astype(np.uint8),
          rs.randint(0, 256, lens[2]).astype(np.uint8),
          rs.choice(np.array([3, 4, 5, 250], np.uint8), size=lens[3]),
          np.full(lens[4], 9, np.uint8)]
want = [o.range_encode(p) for p in planes]
if api.lib().wr_range_decode_vec(0, None, None, None, None, None) != 0:
    print("no AVX-512"); sys.exit(0)
for k in (1, 2, 3, 4, 5):   # sessions of 1 .. 5 streams that shrink to nothing as the planes end one after the other
    for first in range(len(planes)):
        sel = [(first + j) %% len(planes) for j in range(k)]
        enc = api.range_encode_vec([planes[i] for i in sel])
        for i, e in zip(sel, enc):
            assert np.array_equal(e, want[i]), ("encode", k, i)
        dec, got = api.range_decode_vec([want[i] for i in sel], [planes[i].size for i in sel])
        for i, d, g in zip(sel, dec, got):
            assert g == planes[i].size and np.array_equal(d, planes[i]), ("decode", k, i)
n = 60000 * 2 + 54321
ps = [p[:n] for p in planes[:3]]
ws = [o.range_encode(p) for p in ps]
# the same through windows of two blocks (planes in device memory reach the coder that way)
enc = api.range_encode_windowed(ps, 120000, 2)
dec, got = api.range_decode_windowed(ws, n, 120000, 2)
for a, b, d, g, p in zip(enc, ws, dec, got, ps):
    assert np.array_equal(a, b) and g == n and np.array_equal(d, p), "windowed"
print("ok")
"""


def test_small_vector_sessions_same_bytes():
    """A 16-lane session that is down to three streams or fewer runs them through the scalar interleaved loops
    (wr_rangecoder.cpp, VecEncGroup::step / VecDecGroup::prepare): sessions of 1-5 planes of all kinds and lengths -- the
    streams change loops in mid-plane as their neighbours end -- must give the oracle's bytes and symbols, whole planes and
    windowed ones."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", SMALL_SESSIONS % root], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] in ("ok", "no AVX-512")


DUAL_GROUPS = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from waverange_amd import api
from oracle.loader import Oracle
o = Oracle()
if api.lib().wr_range_decode_vec(0, None, None, None, None, None) != 0:
    print("no AVX-512"); sys.exit(0)
rs = np.random.RandomState(23)
planes = []
for k in range(37):   # more than two groups hold; lengths around block boundaries, streams leave one after the other
    n = 60000 * (2 + k %% 5) + (0, 1, 59999, 777)[k %% 4]
    kind = k %% 4
    if kind == 0:
        planes.append(rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]))
    elif kind == 1:
        planes.append(np.where(rs.random_sample(n) < 0.9997, 128, rs.randint(120, 136, n)).astype(np.uint8))
    elif kind == 2:
        planes.append(rs.choice(np.array([0, 254, 255], np.uint8), size=n, p=[0.05, 0.05, 0.9]))
    else:
        planes.append(rs.randint(0, 256, n).astype(np.uint8))   # noise: every block falls back to the scalar loop of its lane
want = [o.range_encode(p) for p in planes]
dec, got = api.range_decode_vec(want, [p.size for p in planes])
for i, (d, g, p) in enumerate(zip(dec, got, planes)):
    assert g == p.size and np.array_equal(d, p), ("hook", i)
api.set_coder_pool(2, 4)
try:
    for rep in range(2):
        dec, got = api.range_decode_pool(want, [p.size for p in planes])
        for i, (d, g, p) in enumerate(zip(dec, got, planes)):
            assert g == p.size and np.array_equal(d, p), ("pool", rep, i)
        n = 60000 * 3 + 4321
        ps = [p[:n] for p in planes if p.size >= n][:20]
        ws = [o.range_encode(p) for p in ps]
        dec, got = api.range_decode_windowed(ws, n, 120000, 2)
        for i, (d, g, p) in enumerate(zip(dec, got, ps)):
            assert g == n and np.array_equal(d, p), ("windowed", rep, i)
finally:
    api.set_coder_pool(0)
print("ok")
"""


def test_more_decoder_streams_than_lanes_same_symbols():
    """37 planes of all kinds through the 16-lane decoder's measurement hook, the pool and windows: more streams than a
    session has lanes, so streams join as others end: the oracle's symbols, whatever lane a stream lands in and whenever
    its neighbours end."""
    import sys
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", DUAL_GROUPS % ROOT], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] in ("ok", "no AVX-512")


def test_stream_length_bound_from_histograms(api):
    """wr_range_encode_bound_hist: what lets the planes of a field be coded side by side straight into data_enc.  It must
    never be below the stream's length -- whatever the statistics: noise, few symbols, a single value, skewed blocks that put
    the coder's rounding loss at its worst, a plane ending on a block boundary (extra empty block) -- and it must be tight
    (the gaps between the planes are what is moved afterwards)."""
    rs = np.random.RandomState(7)
    planes = []
    for n in (1, 59999, 60000, 60001, 180000, 60000 * 17 + 4321):
        planes.append(rs.randint(0, 256, size=n).astype(np.uint8))                                  # noise
        planes.append(rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]))               # two symbols
        planes.append(np.full(n, 200, np.uint8))                                                     # one value: 0 bits per symbol
        planes.append((rs.standard_normal(n) * 3 + 128).clip(0, 255).astype(np.uint8))               # a quantizer's middle plane
        p = np.full(n, 7, np.uint8); p[rs.randint(0, n, size=max(1, n // 30000))] = 255               # counts of 1-2 per block: r * sy << range * sy / tot
        planes.append(p)
        planes.append((np.arange(n) % 251).astype(np.uint8))                                         # 251 equally likely symbols
    for p in planes:
        real = api.range_encode(p).size
        est = api.range_encode_bound_hist(p)
        assert est >= real, (p.size, int(p[0]), est, real)
        assert est <= api.lib().wr_range_encode_bound(p.size)
        # tight in absolute terms: the coder's worst-case rounding loss (0.0104 bit per symbol, of which a real stream uses a
        # twentieth) plus the flush and store-ahead slack -- 1.4 MB for a plane of 2^30 symbols, whatever its entropy
        assert est - real <= 80 + 0.0016 * p.size, (p.size, est, real)
