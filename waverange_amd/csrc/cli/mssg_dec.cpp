// wrdec_mssg -- restore MSSG output compressed by wrenc_mssg / wrmssgenc; command line, prompts and files
// of the reference's wrmssgdec (src/mssg/mssg_dec.cpp):
//
//   wrdec_mssg ENCODED_NAME_PREFIX ENCODED_NAME_EXT EXTRACTED_NAME_PREFIX TYPE PRECISION ENDIANFLIP PROCID
//
// TYPE 0: PREFIX.ctl + PREFIX_h.enc + PREFIX_f.enc -> OUT.grd (+ a copy of the control file as OUT.ctl)
// TYPE 1: PREFIX.nmlst + PREFIX_h.enc/_f.enc      -> OUT.p_NNNN for every subdomain
// TYPE 2: PREFIX.nmlst + PREFIX_hNNNN.enc/_fNNNN.enc -> OUT.p_NNNN for subdomain PROCID
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/waverange_amd.h"
#include "mssg_io.h"

using namespace wrmssg;

namespace {

struct Params {
    std::string in_prefix, ext, out_prefix;
    int filetype = 0, outtype = 1, flip = 0, proc = 0;
};

void get_params(int argc, char** argv, Params* p)
{
    std::cout << "usage: ./wrmssgdec ENCODED_NAME_PREFIX ENCODED_NAME_EXT EXTRACTED_NAME_PREFIX TYPE PRECISION ENDIANFLIP PROCID\n";
    std::cout << "where TYPE=(0: regular output; 1: backup united; 2: backup divided), PRECISION=(1:single; 2:double), "
                 "ENDIANFLIP=(0:no; 1:yes) and PROCID=(this proc id)\n";
    std::cout << "interactive mode if not enough arguments are passed.\n";
    std::string v[4];
    if (argc == 8) {  // mssg_dec.cpp:103-114
        std::cout << "automatic mode.";
        p->in_prefix = argv[1];
        p->ext = argv[2];
        p->out_prefix = argv[3];
        for (int k = 0; k < 4; k++) v[k] = argv[4 + k];
    } else {  // mssg_dec.cpp:116-131
        const char* ask[7] = {"Enter encoded data file name prefix []: ", "Enter encoded data file extension name [.enc]: ",
                              "Enter extracted data file name prefix []: ",
                              "Enter file type (0: regular output; 1: backup merged; 2: backup separated) [0]: ",
                              "Enter extracted data type (1: float; 2: double) [2]: ",
                              "Enter endian conversion (0: do not perform; 1: inversion) [1]: ", "Enter id of this proc [0]: "};
        std::string ans[7];
        for (int k = 0; k < 7; k++) { std::cout << ask[k]; std::getline(std::cin, ans[k]); }
        p->in_prefix = ans[0];
        p->ext = ans[1];
        p->out_prefix = ans[2];
        for (int k = 0; k < 4; k++) v[k] = ans[3 + k];
    }
    std::stringstream(v[0]) >> p->filetype;
    std::stringstream(v[1]) >> p->outtype;
    std::stringstream(v[2]) >> p->flip;
    std::stringstream(v[3]) >> p->proc;
}

void open_or_die(std::ifstream& f, const std::string& path, std::ios::openmode mode)
{
    f.open(path.c_str(), mode);
    if (!f) { std::cout << "Cannot read from " << path << std::endl; std::exit(1); }
}

// read ntot_enc payload bytes and decode them into fld (nothing to read for a constant field)
void decode_dataset(std::ifstream& payload, const Coding& c, int nx, int ny, int nz, double* fld, std::vector<unsigned char>& buf)
{
    if (buf.size() < c.ntot_enc) buf.resize(c.ntot_enc);
    payload.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)c.ntot_enc);
    Coding m = c;  // decoding_wrap takes non-const pointers
    decoding_wrap(nx, ny, nz, fld, &m.tolabs, &m.midval, &m.halfspanval, &m.wlev, &m.nlay, &m.ntot_enc, m.deps_vec, m.minval_vec,
                  m.len_enc_vec, buf.data());
}

// ---- TYPE 0 (mssg_dec.cpp:152-331)
int regular_output(const Params& p, int nbytes)
{
    const std::string control = p.in_prefix + ".ctl";
    const GradsControl g = read_grads_control(control);
    std::cout << std::endl << "=== Parameters read from control file ===" << std::endl;
    std::cout << " dset=" << g.dset << " nx=" << g.nx << " ny=" << g.ny << " nz=" << g.nz << " nt=" << g.nt << " undef=" << g.undef
              << std::endl;
    const size_t ntot = (size_t)g.nx * (size_t)g.ny * (size_t)g.nz;
    if (p.in_prefix != p.out_prefix) copy_text_file(control, p.out_prefix + ".ctl");
    const std::string out_name = p.out_prefix + ".grd";
    std::ifstream header, payload;
    open_or_die(header, p.in_prefix + "_h" + p.ext, std::ios::in);
    open_or_die(payload, p.in_prefix + "_f" + p.ext, std::ios::in | std::ios::binary);
    std::string line;
    for (int j = 0; j < 8; j++) std::getline(header, line);  // the preamble written by the encoder
    std::vector<double> fld(ntot), mask;
    std::vector<unsigned char> buf;
    for (int it = 0; it < g.nt; it++) {
        Coding c;
        std::string name = read_header_record(header, it, &c);
        for (size_t j = 0; j < ntot; j++) fld[j] = c.midval;
        bool masked = false;
        double mask_mid = 0;
        if (name == "mask") {  // a mask record precedes the field record of the same number (mssg_dec.cpp:233-273)
            masked = true;
            mask.assign(ntot, c.midval);
            if (c.ntot_enc > 0) {
                std::cout << "  decoding mask_1d_rec, it=" << it << std::endl;
                decode_dataset(payload, c, g.nx, g.ny, g.nz, mask.data(), buf);
                mask_mid = c.midval;
                // the two mask values sit on either side of the middle value: back to {undef, 0}
                for (size_t j = 0; j < ntot; j++) mask[j] = (mask[j] < c.midval) ? g.undef : 0;
                std::cout << "        min=" << g.undef << " max=" << 0 << std::endl;
                name = read_header_record(header, it, &c);
            }
        }
        if (c.ntot_enc > 0) {
            std::cout << "  decoding fld_1d_rec, it=" << it << std::endl;
            decode_dataset(payload, c, g.nx, g.ny, g.nz, fld.data(), buf);
            double lo = fld[0], hi = fld[0];
            for (size_t j = 0; j < ntot; j++) { lo = std::fmin(lo, fld[j]); hi = std::fmax(hi, fld[j]); }
            std::cout << "        min=" << lo << " max=" << hi << std::endl;
        }
        if (masked)
            for (size_t j = 0; j < ntot; j++)
                if (mask[j] < mask_mid) fld[j] = mask[j];
        write_field(out_name, p.flip != 0, nbytes, it, g.nx, g.ny, g.nz, g.nx, g.ny, 0, 0, fld.data());
        std::cout << "  wrote: fld_1d_rec[0]=" << fld[0] << " fld_1d_rec[last]=" << fld[ntot - 1] << std::endl;
    }
    return 0;
}

// ---- TYPE 1 and 2 (mssg_dec.cpp:334-548)
int restart_set(const Params& p, int nbytes)
{
    const std::string control = p.in_prefix + ".nmlst";
    const RestartControl r = read_restart_control(control);
    const int nxloc = r.nx / r.nprocx, nyloc = r.ny / r.nprocy;
    const int ndset = (int)r.dsets.size();
    std::cout << std::endl << "=== Parameters read from control file ===" << std::endl;
    std::cout << "nx = " << r.nx << "; ny = " << r.ny << "; nr(=nz) = " << r.nz << "; dim_size(=nprocx,nprocy) = " << r.nprocx << ", "
              << r.nprocy << "; ndset = " << ndset << std::endl;
    for (int j = 0; j < ndset; j++) std::cout << "record number = " << j + 1 << "; field = " << r.dsets[j] << std::endl;
    const bool united = p.filetype == 1;
    const int fx = united ? r.nx : nxloc, fy = united ? r.ny : nyloc;
    const size_t ntot = (size_t)fx * (size_t)fy * (size_t)r.nz;
    if (p.in_prefix != p.out_prefix) copy_text_file(control, p.out_prefix + ".nmlst");
    const std::string lbl = subdomain_label(p.proc);
    std::ifstream header, payload;
    open_or_die(header, p.in_prefix + "_h" + (united ? "" : lbl) + p.ext, std::ios::in);
    open_or_die(payload, p.in_prefix + "_f" + (united ? "" : lbl) + p.ext, std::ios::in | std::ios::binary);
    std::vector<double> fld(ntot);
    std::vector<unsigned char> buf;
    for (int idset = 0; idset < ndset; idset++) {
        std::fill(fld.begin(), fld.end(), 0.0);
        if (united) std::cout << " dset=" << r.dsets[idset] << " nx=" << r.nx << " ny=" << r.ny << " nz=" << r.nz << std::endl;
        else std::cout << " dset=" << r.dsets[idset] << " nxloc=" << nxloc << " nyloc=" << nyloc << " nz=" << r.nz << std::endl;
        if (idset == 0) {
            // the time record comes back from the header text; every subdomain file starts with it
            // (mssg_dec.cpp:415-446)
            std::string line;
            for (int j = 0; j < 12; j++) std::getline(header, line);
            for (int j = 0; j < kTimeRecLen; j++) header >> fld[j];
            std::getline(header, line);
            std::cout << "  'time' record = ";
            for (int j = 0; j < kTimeRecLen; j++)
                std::cout << " " << std::setprecision(std::numeric_limits<long double>::digits10 + 1) << fld[j];
            std::cout << std::endl;
            if (united)
                for (int py = 0; py < r.nprocy; py++)
                    for (int px = 0; px < r.nprocx; px++)
                        if (px + py > 0)
                            for (int ix = 0; ix < kTimeRecLen; ix++)
                                fld[(size_t)(ix + px * nxloc) + (size_t)r.nx * (size_t)(py * nyloc)] = fld[ix];
        } else {
            Coding c;
            (void)read_header_record(header, idset, &c);
            if (c.ntot_enc > 0) {
                decode_dataset(payload, c, fx, fy, r.nz, fld.data(), buf);
                std::cout << "  decode: fld_1d_rec[0]=" << fld[0] << " fld_1d_rec[last]=" << fld[ntot - 1] << std::endl;
            } else {
                std::fill(fld.begin(), fld.end(), c.midval);  // all elements are equal (mssg_dec.cpp:492-496)
            }
        }
        if (united) {
            for (int py = 0; py < r.nprocy; py++)
                for (int px = 0; px < r.nprocx; px++)
                    write_field(p.out_prefix + ".p_" + subdomain_label(px + r.nprocx * py), p.flip != 0, nbytes, idset, r.nx, r.ny, r.nz,
                                nxloc, nyloc, px * nxloc, py * nyloc, fld.data());
        } else {
            write_field(p.out_prefix + ".p_" + lbl, p.flip != 0, nbytes, idset, nxloc, nyloc, r.nz, nxloc, nyloc, 0, 0, fld.data());
        }
        std::cout << "  wrote: fld_1d_rec[0]=" << fld[0] << " fld_1d_rec[last]=" << fld[ntot - 1] << std::endl;
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    Params p;
    get_params(argc, argv, &p);
    const int nbytes = p.outtype == 1 ? 4 : 8;
    std::cout << std::endl << "=== Decoding parameters ===" << std::endl;
    std::cout << "Encoded file name prefix: " << p.in_prefix << std::endl;
    std::cout << "Encoded file extension name: " << p.ext << std::endl;
    std::cout << "Extracted file name prefix: " << p.out_prefix << std::endl;
    std::cout << "File type (0: regular output; 1: backup merged; 2: backup separated): " << p.filetype << std::endl;
    std::cout << "Output files contain " << nbytes << "-byte floating point data" << std::endl;
    if (p.flip) std::cout << "Convert big endian to little endian or vice versa" << std::endl;
    std::cout << "This proc id: " << p.proc << std::endl;
    int rc = 0;
    if (p.filetype == 0) rc = regular_output(p, nbytes);
    else if (p.filetype == 1 || p.filetype == 2) rc = restart_set(p, nbytes);
    else std::cout << "Error: unknown file type" << std::endl;
    std::cout << "=== End of decompression ===\n";
    return rc;
}
