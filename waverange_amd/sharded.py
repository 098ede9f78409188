"""Field sharding across GPUs: independent fields, one process per GPU, no
data-path collective -- only a gather of the coded fields to rank 0, which writes the
`.wrh` / `.wrb` pair in field order (the container of reference src/generic/gen_aux.cpp).

The codec is passed in (``codec(field_zyx, tolrel) -> dict`` with the outputs of
``encoding_wrap``): on a GPU node it is ``Context.encode`` of waverange_amd.api; the CPU tests
inject another one to exercise the plan / gather / container logic under gloo.
"""
import struct

import numpy as np

CODER_VERSION = 31503  # reference src/core/defs.h:34


def plan(nf, world):
    """Round-robin field -> rank assignment: rank r codes fields r, r+world, ..."""
    return [list(range(r, nf, world)) for r in range(world)]


def field_offsets(specs, file_type):
    """Byte offset of every field record in a generic input file.
    specs: list of dicts with nbytes, nx, ny, nz, nh (reference gen_aux.cpp:230-397)."""
    ml = 4 if file_type == 0 else (8 if file_type == 1 else 0)
    offs, pos = [], 0
    for s in specs:
        offs.append(pos)
        pos += 2 * ml + s["nbytes"] * s["nx"] * s["ny"] * s["nz"] * s["nh"]
    return offs


def read_field(path, offset, spec, file_type, flip):
    """One field as float64 (nz*nh, ny, nx) + its 8 record-marker bytes (gen_aux.cpp:230-397)."""
    ml = 4 if file_type == 0 else (8 if file_type == 1 else 0)
    n = spec["nx"] * spec["ny"] * spec["nz"] * spec["nh"]
    recl = bytearray(8)
    with open(path, "rb") as fh:
        fh.seek(offset)
        if ml:
            m = fh.read(ml)
            recl[:ml] = m[::-1] if flip else m
        dt = np.dtype("f4" if spec["nbytes"] == 4 else "f8").newbyteorder(">" if flip else "<")
        a = np.frombuffer(fh.read(n * spec["nbytes"]), dtype=dt).astype(np.float64)
    if spec.get("idinv", 0):  # file loops ix outermost ... ih innermost
        a = a.reshape(spec["nx"], spec["ny"], spec["nz"], spec["nh"]).transpose(3, 2, 1, 0)
    return np.ascontiguousarray(a).reshape(spec["nz"] * spec["nh"], spec["ny"], spec["nx"]), bytes(recl)


def encode_my_fields(path, specs, file_type, flip, my_ids, codec, effective_tol):
    """Code the fields of this rank.  Returns {field id: record} with record = dict(recl, enc)
    (enc is None for fields stored uncompressed, then `raw` holds their bytes)."""
    offs = field_offsets(specs, file_type)
    out = {}
    for i in my_ids:
        fld, recl = read_field(path, offs[i], specs[i], file_type, flip)
        if specs[i].get("icomp", 1):
            enc = codec(fld, effective_tol)
            out[i] = dict(recl=recl, enc={k: enc[k] for k in ("tolabs", "midval", "halfspanval", "wlev", "nlay",
                                                              "ntot_enc", "deps_vec", "minval_vec", "len_enc_vec")},
                          payload=bytes(np.ascontiguousarray(enc["data"][:enc["ntot_enc"]])))
        else:
            dt = "<f4" if specs[i]["nbytes"] == 4 else "<f8"
            out[i] = dict(recl=recl, enc=None, payload=fld.astype(dt).tobytes())
    return out


def gather_fields(mine, nf, dist=None):
    """All coded fields on rank 0 in field order (None elsewhere).  `dist` is torch.distributed
    (already initialised) or None for a single process."""
    if dist is None or dist.get_world_size() == 1:
        return [mine[i] for i in range(nf)]
    rank, world = dist.get_rank(), dist.get_world_size()
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(mine, bucket, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return [merged[i] for i in range(nf)]


def _g19(v):
    return "%.19g" % v  # 19 significant digits, reference gen_aux.cpp:532 (quirk Q4)


def write_container(wrh_path, wrb_path, wrb_name, specs, file_type, flip, records):
    """`.wrh` text + `.wrb` payloads exactly as the reference writes them
    (gen_enc.cpp:509-520, gen_aux.cpp:401-408, 419-468, 505-556), including quirk Q2."""
    lines = [" ===== Header file for compressed data =====", " Coder version: %d" % CODER_VERSION,
             " Encoded data file name: %s" % wrb_name,
             " File type (0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++): %d" % file_type,
             " Converted big endian to little endian or vice versa" if flip else " No endian conversion",
             " Number of fields in the file, nf: %d" % len(specs)]
    prev_ntot = 0
    with open(wrb_path, "wb") as wrb:
        for i, (s, r) in enumerate(zip(specs, records)):
            e = r["enc"]
            icomp = 1 if e is not None else 0
            reminder = e["ntot_enc"] if icomp else prev_ntot
            head = " nbytes; recl; nx; ny; nz; nh; idinv; icomp;"
            if icomp:
                head += " tol_base; tolabs; midval; halfspanval; wlev; nlay; ntot_enc;"
            if reminder > 0:
                head += " deps_vec(1:nlay); minval_vec(1:nlay); len_enc_vec(1:nlay)"
            lines += [" -----", str(i), head, str(s["nbytes"]),
                      "".join("%x " % b for b in r["recl"]), str(s["nx"]), str(s["ny"]), str(s["nz"]), str(s["nh"]),
                      str(s.get("idinv", 0)), str(icomp)]
            if icomp:
                lines += [_g19(s["tol_base"]), _g19(e["tolabs"]), _g19(e["midval"]), _g19(e["halfspanval"]),
                          str(e["wlev"]), str(e["nlay"]), str(e["ntot_enc"])]
                if e["ntot_enc"] > 0:
                    lines += ["".join(_g19(v) + " " for v in e["deps_vec"]),
                              "".join(_g19(v) + " " for v in e["minval_vec"]),
                              "".join("%d " % v for v in e["len_enc_vec"])]
                prev_ntot = e["ntot_enc"]
            wrb.write(r["payload"])
    with open(wrh_path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def wrenc_sharded(in_path, wrb_path, wrh_path, specs, file_type, flip, codec, dist=None):
    """Sharded generic encoder: every rank codes its fields, rank 0 writes the container.
    Quirk Q1 of the reference CLI is kept: the tolerance of the LAST field is applied to all."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    eff_tol = specs[-1]["tol_base"]
    mine = encode_my_fields(in_path, specs, file_type, flip, plan(len(specs), world)[rank], codec, eff_tol)
    records = gather_fields(mine, len(specs), dist)
    if rank == 0:
        import os
        write_container(wrh_path, wrb_path, os.path.basename(wrb_path), specs, file_type, flip, records)
    if dist is not None and world > 1:
        dist.barrier()


def read_container_header(wrh_path):
    """Parse a `.wrh` file (reference gen_dec.cpp:160-168, gen_aux.cpp:559-644) into (nf, [field dicts])."""
    with open(wrh_path) as fh:
        lines = fh.read().split("\n")
    nf = int(lines[5].split(":")[-1])
    pos = 6
    fields = []
    for i in range(nf):
        assert lines[pos].strip() == "-----" and int(lines[pos + 1]) == i, "encoding header file read error"
        f = dict(nbytes=int(lines[pos + 3]), recl=bytes(int(t, 16) for t in lines[pos + 4].split()))
        f["nx"], f["ny"], f["nz"], f["nh"], f["idinv"], f["icomp"] = (int(lines[pos + 5 + k]) for k in range(6))
        pos += 11
        if f["icomp"]:
            f["tol_base"], f["tolabs"], f["midval"], f["halfspanval"] = (float(lines[pos + k]) for k in range(4))
            f["wlev"], f["nlay"], f["ntot_enc"] = (int(lines[pos + 4 + k]) for k in range(3))
            pos += 7
            if f["ntot_enc"] > 0:
                f["deps_vec"] = np.array([float(t) for t in lines[pos].split()])
                f["minval_vec"] = np.array([float(t) for t in lines[pos + 1].split()])
                f["len_enc_vec"] = [int(t) for t in lines[pos + 2].split()]
                pos += 3
        fields.append(f)
    return nf, fields


def wrdec_sharded(wrb_path, wrh_path, out_path, file_type, flip, decoder, dist=None):
    """Sharded generic decoder: every rank decodes its fields (round robin) and writes them at their byte
    offsets of the output file, which has the layout of the original input (reference gen_aux.cpp:49-226:
    record markers from the header, fp32 / fp64, endian flip, inverted dimension order).
    decoder(enc dict incl. data, shape (nz*nh, ny, nx)) -> float64 array."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    nf, fields = read_container_header(wrh_path)
    ml = 4 if file_type == 0 else (8 if file_type == 1 else 0)
    src, dst, a, b = [], [], 0, 0
    for f in fields:
        n = f["nx"] * f["ny"] * f["nz"] * f["nh"]
        src.append(a)
        dst.append(b)
        a += f["ntot_enc"] if f["icomp"] else f["nbytes"] * n
        b += 2 * ml + f["nbytes"] * n
    if rank == 0:
        with open(out_path, "wb") as fh:
            fh.truncate(b)
    if dist is not None and world > 1:
        dist.barrier()
    with open(wrb_path, "rb") as fin, open(out_path, "r+b") as fout:
        for i in plan(nf, world)[rank]:
            f = fields[i]
            shape = (f["nz"] * f["nh"], f["ny"], f["nx"])
            n = shape[0] * shape[1] * shape[2]
            fin.seek(src[i])
            if not f["icomp"]:
                fld = np.frombuffer(fin.read(f["nbytes"] * n), dtype="<f4" if f["nbytes"] == 4 else "<f8").astype(np.float64)
            elif f["ntot_enc"] == 0:
                fld = np.full(n, f["midval"])  # gen_dec.cpp:201
            else:
                enc = {k: f[k] for k in ("tolabs", "midval", "halfspanval", "wlev", "nlay", "ntot_enc", "deps_vec", "minval_vec", "len_enc_vec")}
                enc["data"] = np.frombuffer(fin.read(f["ntot_enc"]), dtype=np.uint8)
                fld = np.asarray(decoder(enc, shape), dtype=np.float64)
            arr = fld.reshape(f["nh"], f["nz"], f["ny"], f["nx"])
            if f["idinv"]:
                arr = arr.transpose(3, 2, 1, 0)
            dt = np.dtype("f4" if f["nbytes"] == 4 else "f8").newbyteorder(">" if flip else "<")
            marker = f["recl"][:ml][::-1] if flip else f["recl"][:ml]
            fout.seek(dst[i])
            fout.write(marker + np.ascontiguousarray(arr).astype(dt).tobytes() + marker)
    if dist is not None and world > 1:
        dist.barrier()


def gpu_codec(device):
    """The product codec for wrenc_sharded: wr_encode_host on this rank's GPU (field and coded bytes in host
    memory, exactly what the drop-in encoding_wrap runs)."""
    from . import api
    ctx = api.Context(device)

    def codec(fld, tol):
        enc, _ = ctx.encode_host(np.ascontiguousarray(fld, dtype=np.float64), tol)
        enc["data"] = enc["data"].copy()
        return enc

    codec.close = ctx.close
    return codec


def gpu_decoder(device):
    """The product decoder for wrdec_sharded: wr_decode_host on this rank's GPU."""
    from . import api
    ctx = api.Context(device)

    def decoder(enc, shape):
        out = np.empty(shape, dtype=np.float64)
        ctx.decode_host(out, enc)
        return out

    decoder.close = ctx.close
    return decoder


def main(argv=None):
    """Sharded generic encoder / decoder, one process per GPU:

        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
            -m waverange_amd.sharded IN OUT HDR TYPE ENDIANFLIP NF PRECISION NX NY NZ TOL

            -m waverange_amd.sharded ENC HDR OUT TYPE ENDIANFLIP

    (the generic wrenc / wrdec command lines, reference src/generic/gen_enc.cpp:365-412, gen_dec.cpp:105-117;
    every field has the same shape and tolerance in the encoder's argument mode).  Rank r codes fields r, r + N,
    ... on GPU LOCAL_RANK (modulo the visible GPUs).  Encoder: rank 0 gathers the coded fields over gloo -- host
    bytes, no GPU tensors travel -- and writes the .wrh / .wrb pair in field order.  Decoder: every rank writes
    its fields at their offsets of the output file.  Without a launcher it runs as a single process."""
    import os
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) not in (5, 11):
        sys.stderr.write(main.__doc__ + "\n")
        return 2
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if len(argv) == 5:
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        from . import api
        api.set_verbosity(0)
        ndev = api.device_count()
        if ndev < 1:
            raise SystemExit("waverange_amd.sharded: no GPU visible (libwaverange_amd has no CPU fallback)")
        decoder = gpu_decoder(local_rank % ndev)
        try:
            wrdec_sharded(argv[0], argv[1], argv[2], int(argv[3]), bool(int(argv[4])), decoder, dist)
        finally:
            decoder.close()
            if dist is not None:
                dist.destroy_process_group()
        return 0
    in_path, wrb_path, wrh_path = argv[0], argv[1], argv[2]
    file_type, flip, nf, prec = int(argv[3]), int(argv[4]), int(argv[5]), int(argv[6])
    nx, ny, nz, tol = int(argv[7]), int(argv[8]), int(argv[9]), float(argv[10])
    specs = [dict(nbytes=4 if prec == 1 else 8, nx=nx, ny=ny, nz=nz, nh=1, idinv=0, icomp=1, tol_base=tol) for _ in range(nf)]
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
    from . import api
    api.set_verbosity(0)
    ndev = api.device_count()
    if ndev < 1:
        raise SystemExit("waverange_amd.sharded: no GPU visible (libwaverange_amd has no CPU fallback)")
    codec = gpu_codec(local_rank % ndev)
    try:
        wrenc_sharded(in_path, wrb_path, wrh_path, specs, file_type, bool(flip), codec, dist)
    finally:
        codec.close()
        if dist is not None:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
