"""N>1 path on CPU: world_size-2 gloo processes shard the fields of a generic file, code their own
fields, gather to rank 0 and write one .wrh/.wrb pair that must equal the reference CLI's golden
output.  The codec injected here is the oracle (test infrastructure); on a GPU node the same driver
takes waverange_amd.api.Context.encode."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import pytest

import cli_cases
from util import GOLDEN, ROOT
from waverange_amd import sharded

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch.distributed as dist
from oracle.loader import Oracle
from waverange_amd import sharded
import cli_cases
case, workdir = sys.argv[1], sys.argv[2]
dist.init_process_group("gloo")
o = Oracle()
c = cli_cases.CASES[case]
specs = []
for fd in c["fields"]:
    nbytes, nx, ny, nz, nh, idinv = fd["spec"]
    specs.append(dict(nbytes=nbytes, nx=nx, ny=ny, nz=nz, nh=nh, idinv=idinv, icomp=fd["icomp"], tol_base=float(fd["tol"])))
codec = lambda fld, tol: o.encode(fld, tol)
sharded.wrenc_sharded(os.path.join(workdir, "data.bin"), os.path.join(workdir, "data.wrb"),
                      os.path.join(workdir, "data.wrh"), specs, c["file_type"], bool(c["flip"]), codec, dist)
# and back: every rank decodes its fields and writes them at their offsets of the output file
sharded.wrdec_sharded(os.path.join(workdir, "data.wrb"), os.path.join(workdir, "data.wrh"), os.path.join(workdir, "datarec.bin"),
                      c["file_type"], bool(c["flip"]), lambda enc, shape: o.decode(enc, shape), dist)
print("rank", dist.get_rank(), "fields", sharded.plan(len(specs), dist.get_world_size())[dist.get_rank()])
dist.destroy_process_group()
'''


def test_plan_round_robin():
    assert sharded.plan(8, 8) == [[i] for i in range(8)]
    assert sharded.plan(5, 2) == [[0, 2, 4], [1, 3]]
    assert sharded.plan(1, 4) == [[0], [], [], []]
    assert sorted(sum(sharded.plan(13, 4), [])) == list(range(13))


@pytest.mark.parametrize("case", ["inmeta_new_type0", "argv_two_fp32", "inmeta_old_type1_bigendian"])
def test_sharded_wrenc_world2_gloo(case):
    with open(os.path.join(GOLDEN, "cli.json")) as fh:
        g = json.load(fh)[case]
    with tempfile.TemporaryDirectory() as d:
        cli_cases.write_inputs(case, d)
        script = os.path.join(d, "worker.py")
        with open(script, "w") as fh:
            fh.write(WORKER % dict(root=ROOT))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29653", script, case, d],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert open(os.path.join(d, "data.wrh")).read() == g["wrh"]
        assert hashlib.sha256(open(os.path.join(d, "data.wrb"), "rb").read()).hexdigest() == g["wrb_sha256"]
        assert os.path.getsize(os.path.join(d, "datarec.bin")) == g["rec_size"]
        assert hashlib.sha256(open(os.path.join(d, "datarec.bin"), "rb").read()).hexdigest() == g["rec_sha256"], "decoded file differs"


def test_single_process_matches_too():
    """world == 1 (no process group) takes the same code path minus the gather."""
    from oracle.loader import Oracle
    o = Oracle()
    case = "inmeta_new_type0"
    with open(os.path.join(GOLDEN, "cli.json")) as fh:
        g = json.load(fh)[case]
    c = cli_cases.CASES[case]
    specs = []
    for fd in c["fields"]:
        nbytes, nx, ny, nz, nh, idinv = fd["spec"]
        specs.append(dict(nbytes=nbytes, nx=nx, ny=ny, nz=nz, nh=nh, idinv=idinv, icomp=fd["icomp"], tol_base=float(fd["tol"])))
    with tempfile.TemporaryDirectory() as d:
        cli_cases.write_inputs(case, d)
        sharded.wrenc_sharded(os.path.join(d, "data.bin"), os.path.join(d, "data.wrb"), os.path.join(d, "data.wrh"),
                              specs, c["file_type"], bool(c["flip"]), lambda f, t: o.encode(f, t))
        assert open(os.path.join(d, "data.wrh")).read() == g["wrh"]
        assert hashlib.sha256(open(os.path.join(d, "data.wrb"), "rb").read()).hexdigest() == g["wrb_sha256"]
        sharded.wrdec_sharded(os.path.join(d, "data.wrb"), os.path.join(d, "data.wrh"), os.path.join(d, "datarec.bin"),
                              c["file_type"], bool(c["flip"]), lambda enc, shape: o.decode(enc, shape))
        assert hashlib.sha256(open(os.path.join(d, "datarec.bin"), "rb").read()).hexdigest() == g["rec_sha256"]
