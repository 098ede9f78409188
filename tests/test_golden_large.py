"""Parity at the headline sizes against what the REFERENCE ITSELF produced (tests/golden/large.json, written by
tools/make_golden_large.py from oracle/_ref in the build container): 512^3 (BASELINE configs[1]) and 1024^3
(configs[2]) -- plane streams far past 2^24 bytes, where rngcod13's 24-bit length trailer wraps
(src/rangecod/rangecod.c:254-276), 17 896 coding blocks per plane.

CPU: the oracle restatement against the pins (512^3 always; 1024^3 with WR_GOLDEN_1024=1: ~10 minutes, 40 GiB).
GPU: the product, through the C ABI, against the pins -- on per-call coder threads AND on the configuration bench.py
times (coder pool of 16, AVX-512 encoder and decoder sessions, windowed planes, two-call decode)."""
import os
import threading

import numpy as np
import pytest

from util import check_large_record, large_golden, sha_big
from waverange_amd import synth

CASES_512 = [(512, 1e-5), (512, 1e-7)]
CASES_1024 = [(1024, 1e-3), (1024, 1e-7)]
SEED = 12345


def key(n, tol, seed=SEED):
    """the bench's own field (seed 12345) has the short key; rank r of an N-rank run codes seed 12345 + r"""
    return "%d^3_tol%g" % (n, tol) + ("" if seed == SEED else "_seed%d" % seed)


def host_field(n, seed=12345, out=None):
    f = np.empty((n, n, n)) if out is None else out
    step = max(1, (1 << 24) // (n * n))
    for z in range(0, n, step):
        f[z:z + step] = synth.field(n, n, n, seed, z, min(n, z + step))
    return f


def oracle_vs_pins(oracle, n, tols, seed=SEED):
    g = large_golden()
    f = host_field(n, seed)
    assert sha_big(f) == g[key(n, tols[0], seed)]["input_sha256"], "the synthetic field is not the one the pins were made from"
    errs = []

    def one(tol):
        try:
            rec = g[key(n, tol, seed)]
            e = oracle.encode(f, tol)
            check_large_record(e, rec, key(n, tol, seed))
            assert sha_big(e["residual"]) == rec["residual_sha256"]
            e.pop("residual")
            assert sha_big(oracle.decode(e, f.shape)) == rec["decoded_sha256"], "reconstruction differs from the reference's"
        except BaseException as exc:  # noqa: BLE001
            errs.append((tol, exc))

    ths = [threading.Thread(target=one, args=(t,)) for t in tols]  # ctypes drops the GIL: one core per tolerance
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0][1]


def test_oracle_equals_the_reference_at_512(oracle):
    oracle_vs_pins(oracle, 512, [t for _, t in CASES_512])


def test_oracle_equals_the_reference_at_512_tol_1e16_and_on_another_rank_s_field(oracle):
    """BASELINE configs[4]'s near-lossless tolerance: eight planes, six of them noise -- the plane loop runs to its last index
    (NLAYMAX, wrappers.cpp:333), where on this field the natural step also falls below the tolerance for the first time (deps :=
    tolabs, :326-330): seven planes of recomputation depth in k_quant_blk; and configs[3]'s field of rank 1 (seed 12346)."""
    g = large_golden()
    for n in (512, 1024):
        rec = g[key(n, 1e-16)]
        deps, tolabs = [float.fromhex(v) for v in rec["deps_vec"]], float.fromhex(rec["tolabs"])
        assert rec["nlay"] == 8 and deps[6] > tolabs and deps[7] == tolabs
    errs = []

    def guarded(fn, *a):
        try:
            fn(*a)
        except BaseException as exc:  # noqa: BLE001
            errs.append(exc)
    ths = [threading.Thread(target=guarded, args=(oracle_vs_pins, oracle, 512, [1e-16])),
           threading.Thread(target=guarded, args=(oracle_vs_pins, oracle, 512, [1e-5], 12346))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]


def test_every_rank_of_config4_has_a_pin():
    g = large_golden()
    shas = set()
    for r in range(8):
        rec = g[key(512, 1e-5, SEED + r)]
        assert rec["seed"] == SEED + r and rec["n"] == 512 and rec["nlay"] == 4
        shas.add(rec["input_sha256"])
    assert len(shas) == 8
    # ... and of a weak-scaling bench.py run at 1024^3 (every rank round-trips its own field at tol 1e-3 and 1e-7)
    shas = set()
    for r in range(8):
        for tol, nlay in ((1e-3, 3), (1e-7, 4)):
            rec = g[key(1024, tol, SEED + r)]
            assert rec["seed"] == SEED + r and rec["n"] == 1024 and rec["nlay"] == nlay, (r, tol)
        shas.add(g[key(1024, 1e-3, SEED + r)]["input_sha256"])
    assert len(shas) == 8


@pytest.mark.skipif(not os.environ.get("WR_GOLDEN_1024"), reason="ten minutes of CPU and 40 GiB: WR_GOLDEN_1024=1 (log of a run: profiles/r03/oracle_vs_reference_pins_1024.log)")
def test_oracle_equals_the_reference_at_1024(oracle):
    oracle_vs_pins(oracle, 1024, [t for _, t in CASES_1024])


def test_pins_are_past_the_24_bit_trailer():
    g = large_golden()
    assert g[key(512, 1e-7)]["planes_past_2p24_bytes"] == [2, 3]
    assert g[key(1024, 1e-7)]["planes_past_2p24_bytes"] == [1, 2, 3] and g[key(1024, 1e-3)]["planes_past_2p24_bytes"] == [1, 2]
    for k, rec in g.items():
        if not k.startswith("_"):
            # (tol 1e-16 is below what eight planes resolve: the error there is what the eighth plane leaves, ~6e-15)
            assert sum(rec["len_enc_vec"]) == rec["ntot_enc"] and float(rec["linf_rel"]) < max(1.15 * float(rec["tol"]), 1e-14)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def api():
    from waverange_amd import api as a
    a.set_verbosity(0)
    return a


def device_field_on_host(api, ctx, n, seed=12345):
    dbuf = ctx.alloc(n ** 3 * 8)
    ctx.synth_field(dbuf, n, n, n, seed)
    ctx.sync()
    f = api.pinned_array((n, n, n))
    api._check(api.lib().wr_dev_download(ctx.h, f.ctypes.data, dbuf.ptr, f.nbytes))
    dbuf.free()
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["per_call_threads", "bench_pool_vector_two_call"])
@pytest.mark.parametrize("n", [512, 1024])
def test_product_equals_the_reference_pins(api, mode, n):
    """Every coded byte, every header bit and the reconstruction of the GPU path == the reference's, at full size."""
    g = large_golden()
    cases = CASES_512 if n == 512 else CASES_1024
    pooled = mode != "per_call_threads"
    if pooled:
        api.set_coder_pool(16, 4)   # what bench.py runs on a 16-CPU rank: vector encoder + decoder sessions, scalar loops of 4
    else:
        api.set_coder_pool(0)
        api.set_threads(8)
    try:
        with api.Context(0) as ctx, api.Context(0) as ctx2:
            f = device_field_on_host(api, ctx, n)
            assert sha_big(f) == g[key(n, cases[0][1])]["input_sha256"]
            out = api.pinned_array(f.shape)
            encs, errs = {}, []

            def encode(c, tol):
                try:
                    e, _ = c.encode_host(f, tol)
                    e["data"] = e["data"].copy()
                    encs[tol] = e
                except BaseException as exc:  # noqa: BLE001
                    errs.append(exc)

            # both tolerances in flight at once: with the pool their planes share the vector sessions
            ths = [threading.Thread(target=encode, args=(c, tol)) for c, (_, tol) in zip((ctx, ctx2), cases)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            if errs:
                raise errs[0]
            for _, tol in cases:
                rec = g[key(n, tol)]
                check_large_record(encs[tol], rec, "%s %s" % (mode, key(n, tol)))
                if pooled:   # the two-call decode of the bench
                    ctx.decode_begin(f.shape, encs[tol])
                    ctx.decode_finish_host(out)
                else:
                    ctx.decode_host(out, encs[tol])
                assert sha_big(out) == rec["decoded_sha256"], "reconstruction differs from the reference's (%s)" % key(n, tol)
            if pooled:
                loops = api.pool_loop_stats()
                assert loops["vector_encoder"][1] > 0 and loops["vector_decoder"][1] > 0, "the AVX-512 sessions did not run"
    finally:
        api.set_coder_pool(0)
        api.set_threads(8)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024])
def test_product_equals_the_reference_pin_at_tol_1e16(api, n):
    """BASELINE configs[4]'s tolerance at size: eight planes (k_quant_blk's recomputation seven deep), six noise planes through
    the hand-allotted four-stream decoder loop, 6 GB of coded bytes at 1024^3 -- every byte and the reconstruction == the
    reference's, on the bench's coder configuration."""
    g = large_golden()
    rec = g[key(n, 1e-16)]
    api.set_coder_pool(16, 4)
    try:
        with api.Context(0) as ctx:
            f = device_field_on_host(api, ctx, n)
            assert sha_big(f) == rec["input_sha256"]
            e, _ = ctx.encode_host(f, 1e-16)
            check_large_record(e, rec, "pool " + key(n, 1e-16))
            out = api.pinned_array(f.shape)
            ctx.decode_begin(f.shape, e)
            ctx.decode_finish_host(out)
            assert sha_big(out) == rec["decoded_sha256"], "reconstruction differs from the reference's"
    finally:
        api.set_coder_pool(0)
        api.set_threads(8)
