"""Synthetic MSSG data sets and command lines for the wrenc_mssg / wrdec_mssg parity cases.
Shared by tools/make_golden_mssg.py (which runs the compiled REFERENCE wrmssgenc / wrmssgdec on them
and stores their outputs under tests/golden/mssg.json) and by tests/test_mssg.py.

File formats as the reference reads them (src/mssg/ctrl_aux.cpp): a GrADS control file + one flat
binary file of nt fields (regular output), or a namelist + one flat binary file of records per
subdomain (restart set; record 1 is the time record)."""
import os

import numpy as np

from waverange_amd import synth

UNDEF = -9.99e33

CASES = {
    # regular output, single precision, big-endian, undefined points in two of three fields; arguments
    "regout_f32_be_masked": dict(kind="regout", prefix="n_tm", nx=40, ny=24, nz=8, nt=3, nbytes=4, flip=1, tol="1e-5",
                                 masked=[0, 2], enc_mode="argv", dec_mode="argv"),
    # regular output, double precision, native byte order, odd box, no undefined points;
    # encoder parameters from an "inmeta" file (new format), decoder answers on stdin
    "regout_f64_odd_inmeta": dict(kind="regout", prefix="uvel", nx=17, ny=9, nz=5, nt=2, nbytes=8, flip=0, tol="1e-9",
                                  masked=[], enc_mode="inmeta_new", dec_mode="stdin"),
    # restart set of 2 x 2 subdomains coded as global fields (one constant record among them)
    "restart_united": dict(kind="restart", prefix="rst", nx=32, ny=24, nz=6, npx=2, npy=2, nbytes=8, flip=1, tol="1e-7",
                           records=["time", "u", "v", "tconst", "w"], filetype=1, procs=[0], enc_mode="argv", dec_mode="argv"),
    # the same set coded subdomain by subdomain; encoder parameters from an old-format "inmeta"
    "restart_divided": dict(kind="restart", prefix="rst", nx=32, ny=24, nz=6, npx=2, npy=2, nbytes=8, flip=1, tol="1e-7",
                            records=["time", "u", "v", "tconst", "w"], filetype=2, procs=[0, 1, 2, 3], enc_mode="inmeta_old",
                            dec_mode="argv"),
}


def _dtype(nbytes, flip):
    # "flip" = the file is in the byte order opposite to this (little-endian) host
    return np.dtype("f4" if nbytes == 4 else "f8").newbyteorder(">" if flip else "<")


def _regout_fields(c):
    out = []
    for it in range(c["nt"]):
        f = synth.field(c["nx"], c["ny"], c["nz"], seed=500 + it) * 3.0 + 280.0
        if it in c["masked"]:
            z, y, x = np.meshgrid(np.arange(c["nz"]), np.arange(c["ny"]), np.arange(c["nx"]), indexing="ij")
            land = ((x - 11) ** 2 + (y - 9) ** 2 < 30 + 4 * it) | ((x > 30) & (y < 5 + z))
            f = np.where(land, UNDEF, f)
        out.append(f)
    return out


def _restart_global(c):
    recs = []
    for i, name in enumerate(c["records"]):
        if name == "time":
            recs.append(None)
        elif name == "tconst":
            recs.append(np.full((c["nz"], c["ny"], c["nx"]), 300.0))
        else:
            recs.append(synth.field(c["nx"], c["ny"], c["nz"], seed=700 + i) * (1.0 + i))
    return recs


def write_inputs(case, workdir):
    """Create the data set in workdir.  Returns the list of input file names (relative)."""
    c = CASES[case]
    names = []
    if c["kind"] == "regout":
        ctl = ("DSET ^%s.grd\nTITLE synthetic MSSG regular output\nOPTIONS big_endian\nUNDEF %s\n"
               "XDEF %d LINEAR 0.0 1.0\nYDEF %d LINEAR 0.0 1.0\nZDEF %d LEVELS 1 2 3\nTDEF %d LINEAR 00Z01JAN2000 1hr\n"
               "VARS 1\nt %d 99 temperature\nENDVARS\n" % (c["prefix"], "-9.99E33", c["nx"], c["ny"], c["nz"], c["nt"], c["nz"]))
        with open(os.path.join(workdir, c["prefix"] + ".ctl"), "w") as fh:
            fh.write(ctl)
        with open(os.path.join(workdir, c["prefix"] + ".grd"), "wb") as fh:
            for f in _regout_fields(c):
                fh.write(np.ascontiguousarray(f).astype(_dtype(c["nbytes"], c["flip"])).tobytes())
        names = [c["prefix"] + ".ctl", c["prefix"] + ".grd"]
    else:
        nml = ("&grid_size\n nx = %d, ny = %d, nr = %d\n/\n&parallel\n nproc = %d, dim_size = %d, %d\n/\n"
               % (c["nx"], c["ny"], c["nz"], c["npx"] * c["npy"], c["npx"], c["npy"]))
        for i, name in enumerate(c["records"]):
            nml += "&record\n var = '%s', rec = %d\n/\n" % (name, i + 1)
        with open(os.path.join(workdir, c["prefix"] + ".nmlst"), "w") as fh:
            fh.write(nml)
        names.append(c["prefix"] + ".nmlst")
        nxl, nyl = c["nx"] // c["npx"], c["ny"] // c["npy"]
        recs = _restart_global(c)
        for py in range(c["npy"]):
            for px in range(c["npx"]):
                proc = px + c["npx"] * py
                name = "%s.p_%04d" % (c["prefix"], proc)
                with open(os.path.join(workdir, name), "wb") as fh:
                    for r in recs:
                        if r is None:   # time record: 15 meaningful values, then zeros
                            t = np.zeros(c["nz"] * nyl * nxl)
                            t[:15] = 86400.0 * 3 + np.arange(15) * 0.125 + 1.0 / 3.0
                            blk = t
                        else:
                            blk = r[:, py * nyl:(py + 1) * nyl, px * nxl:(px + 1) * nxl]
                        fh.write(np.ascontiguousarray(blk).astype(_dtype(c["nbytes"], c["flip"])).tobytes())
                names.append(name)
    return names


def enc_invocation(case, proc):
    """(argv, stdin text, inmeta text) for the encoder run of subdomain `proc`."""
    c = CASES[case]
    ftype = 0 if c["kind"] == "regout" else c["filetype"]
    vals = [c["prefix"], ".enc", str(ftype), "1" if c["nbytes"] == 4 else "2", str(c["flip"]), c["tol"], str(proc)]
    if c["enc_mode"] == "argv":
        return vals, None, None
    if c["enc_mode"] == "inmeta_new":
        keys = ["prefix_name", "ext_name", "file_type", "input_data_type", "endian_conversion", "tolerance", "id_of_proc"]
        text = "# parameters of the encoder\n" + "".join("&%s = %s\n" % (k, v) for k, v in zip(keys, vals))
        return [], None, text
    if c["enc_mode"] == "inmeta_old":
        return [], None, "".join(v + "\n" for v in vals)
    return [], "".join(v + "\n" for v in vals), None


def dec_invocation(case, proc):
    c = CASES[case]
    ftype = 0 if c["kind"] == "regout" else c["filetype"]
    vals = [c["prefix"], ".enc", "dec_" + c["prefix"], str(ftype), "1" if c["nbytes"] == 4 else "2", str(c["flip"]), str(proc)]
    if c["dec_mode"] == "argv":
        return vals, None
    return [], "".join(v + "\n" for v in vals)


def output_files(case):
    """(encoder outputs, decoder outputs) relative names."""
    c = CASES[case]
    p = c["prefix"]
    if c["kind"] == "regout":
        return [p + "_h.enc", p + "_f.enc"], ["dec_%s.grd" % p, "dec_%s.ctl" % p]
    if c["filetype"] == 1:
        enc = [p + "_h.enc", p + "_f.enc"]
        dec = ["dec_%s.p_%04d" % (p, k) for k in range(c["npx"] * c["npy"])] + ["dec_%s.nmlst" % p]
    else:
        enc = [f for k in c["procs"] for f in ("%s_h%04d.enc" % (p, k), "%s_f%04d.enc" % (p, k))]
        dec = ["dec_%s.p_%04d" % (p, k) for k in c["procs"]] + ["dec_%s.nmlst" % p]
    return enc, dec
