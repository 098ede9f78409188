// wrenc_mssg -- compress MSSG output with the GPU codec; command line, prompts, parameter file and
// output files of the reference's wrmssgenc (src/mssg/mssg_enc.cpp):
//
//   wrenc_mssg FILE_NAME_PREFIX ENCODED_NAME_EXT TYPE PRECISION ENDIANFLIP TOLERANCE PROCID
//
// TYPE 0: regular output  PREFIX.ctl (GrADS) + its data file -> PREFIX_h.enc (text) + PREFIX_f.enc; a field
//         with undefined points is stored as a mask (coded without the transform) plus the padded field.
// TYPE 1: restart set, all subdomain files PREFIX.p_NNNN assembled into global fields  -> PREFIX_h.enc/_f.enc
// TYPE 2: restart set, the one subdomain PROCID on its own                              -> PREFIX_hNNNN.enc/_fNNNN.enc
// A file "inmeta" in the working directory replaces the arguments ("&name=value" lines or the seven
// values one per line); with neither, the values are asked for on stdin.
#include <algorithm>
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
    std::string prefix, ext = ".enc";
    int filetype = 0, intype = 2, flip = 1, proc = 0;
    double tol = 1e-16;
};

std::string trimmed(const std::string& s)
{
    const char* ws = " \t\v\r\n";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return std::string();
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
}

// The pieces of an "&name=value" line as the reference's split() yields them (src/core/trim_split.h): pieces
// between '=' signs, empty ones included, nothing after a trailing '='; a line without any '=' counts as
// two pieces (the line, twice), so it passes as an unknown name rather than as an error.
std::vector<std::string> split_at_equals(const std::string& l)
{
    std::vector<std::string> parts;
    if (l.find('=') == std::string::npos) return {l, l};
    size_t a = 0;
    while (a < l.size()) {
        const size_t b = std::min(l.find('=', a), l.size());
        parts.push_back(l.substr(a, b - a));
        a = b + 1;
    }
    return parts;
}

// mssg_enc.cpp:104-272: "inmeta", else 7 arguments, else questions.  Returns false on a malformed file.
bool get_params(int argc, char** argv, Params* p)
{
    std::string v[5];  // file type, precision, endian flip, tolerance, proc id -- as text, parsed below
    std::ifstream meta("inmeta");
    if (meta) {
        std::cout << "==== inmeta exists. ====" << std::endl;
        std::vector<std::string> lines;
        for (std::string l; std::getline(meta, l);) lines.push_back(l);
        bool keyed = false;
        for (const std::string& raw : lines) {
            const std::string l = trimmed(raw);
            if (l.empty() || l[0] != '&') continue;  // anything else is a comment
            const std::vector<std::string> parts = split_at_equals(l);
            if (parts.size() != 2) {
                if (parts.size() > 1) std::cout << "==== Error : '=' exists twice in a sentence :" << l << "====" << std::endl;
                else std::cout << "==== Error 'value' is missing in a sentence :" << l << "====" << std::endl;
                return false;
            }
            keyed = true;
            std::string name = trimmed(parts[0]);
            const std::string val = trimmed(parts[1]);
            std::transform(name.begin(), name.end(), name.begin(), ::tolower);
            if (name == "&prefix_name") p->prefix = val;
            if (name == "&ext_name") p->ext = val;
            if (name == "&file_type") v[0] = val;
            if (name == "&input_data_type") v[1] = val;
            if (name == "&endian_conversion") v[2] = val;
            if (name == "&tolerance") v[3] = val;
            if (name == "&id_of_proc") v[4] = val;
        }
        if (!keyed) {  // old format: the seven values, one per line
            std::cout << "==== read parameters from inmeta as old format. ====" << std::endl;
            auto line = [&](size_t i) { return i < lines.size() ? lines[i] : std::string(); };
            p->prefix = line(0);
            p->ext = line(1);
            for (int k = 0; k < 5; k++) v[k] = line(2 + k);
        }
    } else if (argc == 8) {
        std::cout << "automatic mode.";
        p->prefix = argv[1];
        p->ext = argv[2];
        for (int k = 0; k < 5; k++) v[k] = argv[3 + k];
    } else {
        std::cout << "usage: ./wrmssgenc FILE_NAME_PREFIX ENCODED_NAME_EXT TYPE PRECISION ENDIANFLIP TOLERANCE PROCID\n";
        std::cout << "where TYPE=(0: regular output; 1: backup united; 2: backup divided), PRECISION=(1:single; 2:double), "
                     "ENDIANFLIP=(0:no; 1:yes), TOLERANCE=(e.g. 1.0e-16) and PROCID=(this proc id)\n";
        std::cout << "interactive mode if not enough arguments are passed.\n";
        const char* ask[7] = {"Enter data file name prefix []: ", "Enter encoded file extension name [.enc]: ",
                              "Enter file type (0: regular output; 1: backup merged; 2: backup separated) [0]: ",
                              "Enter input data type (1: float; 2: double) [2]: ",
                              "Enter endian conversion (0: do not perform; 1: inversion) [1]: ",
                              "Enter base cutoff relative tolerance [1e-16]: ", "Enter id of this proc [0]: "};
        std::string ans[7];
        for (int k = 0; k < 7; k++) { std::cout << ask[k]; std::getline(std::cin, ans[k]); }
        p->prefix = ans[0];
        p->ext = ans[1];
        for (int k = 0; k < 5; k++) v[k] = ans[2 + k];
    }
    // an empty or unreadable answer keeps the default (mssg_enc.cpp:265-271)
    std::stringstream(v[0]) >> p->filetype;
    std::stringstream(v[1]) >> p->intype;
    std::stringstream(v[2]) >> p->flip;
    std::stringstream(v[3]) >> p->tol;
    std::stringstream(v[4]) >> p->proc;
    return true;
}

void open_header(const std::string& path, const char* title, const Params& p, int nbytes, const char* no_flip_text)
{
    std::ofstream h(path.c_str(), std::ios::out | std::ios::trunc);
    if (!h) { std::cout << "Cannot write to " << path << std::endl; std::exit(1); }
    h << title << std::endl;
    h << " Coder version: " << kCoderVersion << std::endl;
    h << " File name prefix: " << p.prefix << std::endl;
    h << " Encoded file extension name: " << p.ext << std::endl;
    h << " File type (0: regular output; 1: backup merged; 2: backup separated): " << p.filetype << std::endl;
    h << " Input files contained " << nbytes << "-byte floating point data" << std::endl;
    if (p.flip) h << " Converted big endian to little endian or vice versa" << std::endl;
    else h << no_flip_text << std::endl;
    h << " Base cutoff relative tolerance: " << p.tol << std::endl;
}

void truncate_file(const std::string& path)
{
    std::ofstream f(path.c_str(), std::ios::binary | std::ios::out | std::ios::trunc);
    if (!f) { std::cout << "Cannot write to " << path << std::endl; std::exit(1); }
}

// one data set through the codec and into the two output files
void code_dataset(int nx, int ny, int nz, double* fld, int wtflag, double tolrel, std::vector<unsigned char>& buf,
                  const std::string& header, const std::string& payload, int idset, const std::string& name)
{
    Coding c;
    double cutoff = tolrel;
    unsigned char nlaymax;
    unsigned long cap;
    setup_wr(nx, ny, nz, &nlaymax, &cap);
    if (buf.size() < cap) buf.resize(cap);
    encoding_wrap(nx, ny, nz, fld, wtflag, 1, 1, 1, &cutoff, &c.tolabs, &c.midval, &c.halfspanval, &c.wlev, &c.nlay,
                  &c.ntot_enc, c.deps_vec, c.minval_vec, c.len_enc_vec, buf.data());
    append_header_record(header, idset, name, c);
    if (c.ntot_enc > 0) append_bytes(payload, buf.data(), c.ntot_enc);
    if (wtflag) std::cout << "        tolabs=" << c.tolabs << std::endl;
}

void min_max(const double* f, size_t n, double* lo, double* hi)
{
    double a = f[0], b = f[0];
    for (size_t j = 0; j < n; j++) { a = std::fmin(a, f[j]); b = std::fmax(b, f[j]); }  // mssg_enc.cpp:311-318
    *lo = a; *hi = b;
}

// ---- TYPE 0 (mssg_enc.cpp:277-414)
int regular_output(const Params& p, int nbytes)
{
    const GradsControl g = read_grads_control(p.prefix + ".ctl");
    const size_t ntot = (size_t)g.nx * (size_t)g.ny * (size_t)g.nz;
    std::cout << " dset=" << g.dset << " nx=" << g.nx << " ny=" << g.ny << " nz=" << g.nz << " nt=" << g.nt << " undef=" << g.undef
              << std::endl;
    const std::string header = p.prefix + "_h" + p.ext, payload = p.prefix + "_f" + p.ext;
    open_header(header, " ===== Header file for compressed MSSG regular output data =====", p, nbytes, " No endian conversion");
    truncate_file(payload);
    std::vector<double> fld(ntot), mask;
    std::vector<unsigned char> buf;
    for (int it = 0; it < g.nt; it++) {
        std::cout << "Field number it=" << it << std::endl;
        read_field(g.dset, p.flip != 0, nbytes, it, g.nx, g.ny, g.nz, g.nx, g.ny, 0, 0, fld.data());
        std::cout << "  read: fld_1d[0]=" << fld[0] << " fld_1d[last]=" << fld[ntot - 1] << std::endl;
        double lo, hi;
        min_max(fld.data(), ntot, &lo, &hi);
        std::cout << "        min=" << lo << " max=" << hi << std::endl;
        // Undefined points carry a value at (or, after lossy storage, near) `undef`, far below the data: they
        // go into a two-valued mask field {min, 0}, coded without the transform at a tolerance that keeps
        // the two values apart, and are padded with the mean of the defined points (mssg_enc.cpp:323-381).
        const double thresh = g.undef + std::fabs(g.undef) * kMaskThresholdAcc;
        if (lo < thresh) {
            double pad = 0;
            int defined = 0;
            for (size_t j = 0; j < ntot; j++)
                if (fld[j] >= thresh) { pad += fld[j]; defined++; }
            pad /= defined;
            mask.resize(ntot);
            for (size_t j = 0; j < ntot; j++) {
                if (fld[j] < thresh) { fld[j] = pad; mask[j] = lo; }
                else mask[j] = 0;
            }
            std::cout << " Masking detected, padding with fld_pad=" << pad << ", mask min=" << lo << std::endl;
            code_dataset(g.nx, g.ny, g.nz, mask.data(), 0, kMaskTolRel, buf, header, payload, it, "mask");
            std::cout << " Mask done, encoding the main field..." << std::endl;
        }
        code_dataset(g.nx, g.ny, g.nz, fld.data(), 1, p.tol, buf, header, payload, it, g.dset);
    }
    return 0;
}

// ---- TYPE 1 and 2 (mssg_enc.cpp:417-598)
int restart_set(const Params& p, int nbytes)
{
    const RestartControl r = read_restart_control(p.prefix + ".nmlst");
    const int nxloc = r.nx / r.nprocx, nyloc = r.ny / r.nprocy;
    const int ndset = (int)r.dsets.size();
    std::cout << std::endl << "=== Parameters read from control file ===" << std::endl;
    std::cout << "nx(=nlg+i_over*2) = " << r.nx << "; ny(=npg+j_over*2) = " << r.ny << "; nr(=nz) = " << r.nz
              << "; dim_size(=nprocx,nprocy) = " << r.nprocx << ", " << r.nprocy << "; ndset = " << ndset << std::endl;
    for (int j = 0; j < ndset; j++) std::cout << "record number = " << j + 1 << "; field = " << r.dsets[j] << std::endl;
    const bool united = p.filetype == 1;
    const int fx = united ? r.nx : nxloc, fy = united ? r.ny : nyloc;  // size of the fields that get coded
    const size_t ntot = (size_t)fx * (size_t)fy * (size_t)r.nz;
    const std::string lbl = subdomain_label(p.proc);
    const std::string header = p.prefix + "_h" + (united ? "" : lbl) + p.ext;
    const std::string payload = p.prefix + "_f" + (united ? "" : lbl) + p.ext;
    open_header(header, " ===== Header file for compressed MSSG restart data =====", p, nbytes, " Did not perform endian conversion");
    // record 1 is the time record: its first values go into the header as text (mssg_enc.cpp:488-508)
    const std::string own_file = p.prefix + ".p_" + lbl;
    {
        std::vector<double> t((size_t)nxloc * nyloc * r.nz);
        read_field(own_file, p.flip != 0, nbytes, 0, nxloc, nyloc, r.nz, nxloc, nyloc, 0, 0, t.data());
        std::ofstream h(header.c_str(), std::ios::out | std::ios::app);
        h << " -----" << std::endl;
        h << "1" << std::endl;
        h << " Data set name = " << (ndset ? r.dsets[0] : std::string()) << std::endl;
        h << " first " << kTimeRecLen << " elements of time record" << std::endl;
        for (int j = 0; j < kTimeRecLen; j++) h << std::setprecision(std::numeric_limits<long double>::digits10 + 1) << t[j] << " ";
        h << std::endl;
    }
    truncate_file(payload);
    std::vector<double> fld(ntot);
    std::vector<unsigned char> buf;
    for (int idset = 1; idset < ndset; idset++) {
        if (united) {
            for (int py = 0; py < r.nprocy; py++)
                for (int px = 0; px < r.nprocx; px++)
                    read_field(p.prefix + ".p_" + subdomain_label(px + r.nprocx * py), p.flip != 0, nbytes, idset, r.nx, r.ny, r.nz,
                               nxloc, nyloc, px * nxloc, py * nyloc, fld.data());
            std::cout << " dset=" << r.dsets[idset] << " nx=" << r.nx << " ny=" << r.ny << " nz=" << r.nz << std::endl;
        } else {
            read_field(own_file, p.flip != 0, nbytes, idset, nxloc, nyloc, r.nz, nxloc, nyloc, 0, 0, fld.data());
            std::cout << " dset=" << r.dsets[idset] << " nxloc=" << nxloc << " nyloc=" << nyloc << " nz=" << r.nz << std::endl;
        }
        std::cout << "  read: fld_1d[0]=" << fld[0] << " fld_1d[last]=" << fld[ntot - 1] << std::endl;
        double lo, hi;
        min_max(fld.data(), ntot, &lo, &hi);
        std::cout << "        min=" << lo << " max=" << hi << std::endl;
        code_dataset(fx, fy, r.nz, fld.data(), 1, p.tol, buf, header, payload, idset, r.dsets[idset]);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    setenv("WR_WRITEBACK_RESIDUAL", "0", 0);  // the residual encoding_wrap leaves in the field array is not used here
    Params p;
    if (!get_params(argc, argv, &p)) return -1;
    const int nbytes = p.intype == 1 ? 4 : 8;
    std::cout << std::endl << "=== Compression parameters ===" << std::endl;
    std::cout << "Data file name prefix: " << p.prefix << std::endl;
    std::cout << "Encoded file extension name: " << p.ext << std::endl;
    std::cout << "File type (0: regular output; 1: backup merged; 2: backup separated): " << p.filetype << std::endl;
    std::cout << "Input files contain " << nbytes << "-byte floating point data" << std::endl;
    if (p.flip) std::cout << "Convert big endian to little endian or vice versa" << std::endl;
    std::cout << "Base cutoff relative tolerance: " << p.tol << std::endl;
    std::cout << "This proc id: " << p.proc << std::endl;
    int rc = 0;
    if (p.filetype == 0) rc = regular_output(p, nbytes);
    else if (p.filetype == 1 || p.filetype == 2) rc = restart_set(p, nbytes);
    else std::cout << "Error: unknown file type" << std::endl;
    std::cout << "=== End of compression ===\n";
    return rc;
}
