"""MSSG front-end (wrenc_mssg / wrdec_mssg): SURVEY.md section 8 "next" row (after N4).

Golden outputs come from the REFERENCE's wrmssgenc / wrmssgdec compiled from /root/reference/src/mssg
(tools/make_golden_mssg.py -> tests/golden/mssg.json) on synthetic MSSG data sets (tests/mssg_cases.py):
regular GrADS output with and without undefined points (the mask is the reference's only wtflag = 0
caller), restart sets coded as global fields and subdomain by subdomain; parameters through arguments,
both "inmeta" formats and stdin.  Every output file must be byte-identical:
  * CPU: OUR front-end sources linked with the REFERENCE codec (oracle/_ref/*_mssg_ours_refcodec);
  * GPU: our front-end on the GPU library, and the REFERENCE's front-end linked against our library."""
import hashlib
import json
import os
import sys
import tempfile

import pytest

from util import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden_mssg  # noqa: E402  (run_case: drives an encoder / decoder pair over a case)
import mssg_cases  # noqa: E402

REFDIR = os.path.join(ROOT, "oracle", "_ref")
BINDIR = os.path.join(ROOT, "waverange_amd", "bin")


@pytest.fixture(scope="module")
def golden_mssg():
    with open(os.path.join(GOLDEN, "mssg.json")) as fh:
        return json.load(fh)


def check_case(case, enc, dec, g):
    if not (os.path.exists(enc) and os.path.exists(dec)):
        pytest.skip("%s / %s not built" % (enc, dec))
    os.environ.setdefault("WR_QUIET", "1")
    with tempfile.TemporaryDirectory() as d:
        files = make_golden_mssg.run_case(case, enc, dec, d)
    assert sorted(files) == sorted(g)
    assert hashlib.sha256(files["__inputs__"]).hexdigest() == g["__inputs__"]["sha256"], "synthetic inputs differ"
    for name, data in files.items():
        if "text" in g[name]:
            assert data.decode() == g[name]["text"], name
        assert len(data) == g[name]["size"], name
        assert hashlib.sha256(data).hexdigest() == g[name]["sha256"], name


@pytest.mark.parametrize("case", sorted(mssg_cases.CASES))
def test_our_mssg_frontend_on_reference_codec(case, golden_mssg):
    check_case(case, os.path.join(REFDIR, "wrenc_mssg_ours_refcodec"), os.path.join(REFDIR, "wrdec_mssg_ours_refcodec"),
               golden_mssg[case])


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(mssg_cases.CASES))
def test_our_mssg_frontend_on_gpu(case, golden_mssg):
    check_case(case, os.path.join(BINDIR, "wrenc_mssg"), os.path.join(BINDIR, "wrdec_mssg"), golden_mssg[case])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["regout_f32_be_masked", "restart_united"])
def test_reference_mssg_frontend_on_our_library(case, golden_mssg):
    check_case(case, os.path.join(REFDIR, "wrmssgenc_ref_dyn"), os.path.join(REFDIR, "wrmssgdec_ref_dyn"), golden_mssg[case])
