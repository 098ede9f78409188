// gen_io.cpp -- see gen_io.h.
#include "gen_io.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace wrio {

namespace {

[[noreturn]] void die(const std::string& msg)
{
    std::cout << msg << std::endl;
    throw std::runtime_error(msg);
}

void check_nbytes(int nbytes)
{
    if (nbytes != 4 && nbytes != 8) die("Generic input nbytes must be equal to 4 or 8");  // gen_aux.cpp:57-62
}

int marker_len(int file_type) { return file_type == 0 ? 4 : (file_type == 1 ? 8 : 0); }

inline void reverse_bytes(unsigned char* p, int n)
{
    for (int a = 0, b = n - 1; a < b; a++, b--) { unsigned char t = p[a]; p[a] = p[b]; p[b] = t; }
}

// position in the x-fastest array of the k-th element of the file
// idinv == 0: file order == memory order (ih, iz, iy outer; ix inner)      gen_aux.cpp:284-326
// idinv != 0: file loops ix outermost ... ih innermost                      gen_aux.cpp:329-371
struct Inverted {
    size_t nx, ny, nz, nh;
    size_t mem_index(size_t k) const
    {
        size_t ih = k % nh; k /= nh;
        size_t iz = k % nz; k /= nz;
        size_t iy = k % ny; k /= ny;
        size_t ix = k;
        return ix + nx * (iy + ny * (iz + nz * ih));
    }
};

}  // namespace

void read_field(const std::string& path, int file_type, bool flip, const FieldSpec& spec, unsigned char recl[8],
                long* pos, std::vector<double>& fld)
{
    check_nbytes(spec.nbytes);
    std::ifstream in(path, std::ios::in | std::ios::binary);
    if (!in.is_open()) die("Cannot open " + path);
    in.seekg(*pos);
    const int ml = marker_len(file_type);
    if (ml) {  // leading record marker, kept (endian-corrected) for the decoder  gen_aux.cpp:265-282
        unsigned char m[8] = {0};
        in.read(reinterpret_cast<char*>(m), ml);
        if (flip) reverse_bytes(m, ml);
        memcpy(recl, m, ml);
        *pos += ml;
    }
    const size_t n = spec.count();
    const size_t nb = (size_t)spec.nbytes;
    // the common layout (doubles in memory order, native endianness) goes straight into the field array
    const bool direct = spec.nbytes == 8 && !flip && !spec.idinv;
    std::unique_ptr<unsigned char[]> raw(direct ? nullptr : new unsigned char[n * nb]);
    fld.resize(n);
    in.read(direct ? reinterpret_cast<char*>(fld.data()) : reinterpret_cast<char*>(raw.get()), (std::streamsize)(n * nb));
    *pos += (long)(n * nb);
    if (ml) {  // trailing marker: read and discard  gen_aux.cpp:374-383
        unsigned char m[8];
        in.read(reinterpret_cast<char*>(m), ml);
        *pos += ml;
    }
    if (in.fail()) die("Cannot read from " + path);  // gen_aux.cpp:389-395
    if (direct) return;
    const Inverted inv{(size_t)spec.nx, (size_t)spec.ny, (size_t)spec.nz, (size_t)spec.nh};
    for (size_t k = 0; k < n; k++) {
        unsigned char* p = raw.get() + k * nb;
        if (flip) reverse_bytes(p, spec.nbytes);
        double v;
        if (spec.nbytes == 4) { float f; memcpy(&f, p, 4); v = f; }
        else memcpy(&v, p, 8);
        fld[spec.idinv ? inv.mem_index(k) : k] = v;
    }
}

void write_field(const std::string& path, bool first, int file_type, bool flip, const FieldSpec& spec,
                 const unsigned char recl[8], const double* fld)
{
    check_nbytes(spec.nbytes);
    std::ofstream out(path, std::ios::out | std::ios::binary | (first ? std::ios::trunc : std::ios::app));
    if (!out.is_open()) die("Cannot open " + path);
    const int ml = marker_len(file_type);
    unsigned char m[8] = {0};
    if (ml) {  // gen_aux.cpp:92-107
        memcpy(m, recl, ml);
        if (flip) reverse_bytes(m, ml);
        out.write(reinterpret_cast<char*>(m), ml);
    }
    const size_t n = spec.count();
    const size_t nb = (size_t)spec.nbytes;
    if (spec.nbytes == 8 && !flip && !spec.idinv) {  // the common layout: the field array is the record
        out.write(reinterpret_cast<const char*>(fld), (std::streamsize)(n * nb));
    } else {
        std::unique_ptr<unsigned char[]> raw(new unsigned char[n * nb]);
        const Inverted inv{(size_t)spec.nx, (size_t)spec.ny, (size_t)spec.nz, (size_t)spec.nh};
        for (size_t k = 0; k < n; k++) {
            const double v = fld[spec.idinv ? inv.mem_index(k) : k];
            unsigned char* p = raw.get() + k * nb;
            if (spec.nbytes == 4) { float f = (float)v; memcpy(p, &f, 4); }  // gen_aux.cpp:134-139
            else memcpy(p, &v, 8);
            if (flip) reverse_bytes(p, spec.nbytes);
        }
        out.write(reinterpret_cast<char*>(raw.get()), (std::streamsize)(n * nb));
    }
    if (ml) out.write(reinterpret_cast<char*>(m), ml);  // gen_aux.cpp:207-222
}

void append_bytes(const std::string& path, const unsigned char* data, size_t n)
{
    std::ofstream out(path, std::ios::binary | std::ios::out | std::ios::app);
    if (!out.is_open()) die("Cannot open " + path);
    out.write(reinterpret_cast<const char*>(data), (std::streamsize)n);
}

void append_raw_field(const std::string& path, int nbytes, const double* fld, size_t n)
{
    check_nbytes(nbytes);
    std::ofstream out(path, std::ios::binary | std::ios::out | std::ios::app);
    if (!out.is_open()) die("Cannot open " + path);
    if (nbytes == 8) { out.write(reinterpret_cast<const char*>(fld), (std::streamsize)(n * 8)); return; }
    std::vector<float> f(n);
    for (size_t j = 0; j < n; j++) f[j] = (float)fld[j];
    out.write(reinterpret_cast<const char*>(f.data()), (std::streamsize)(n * 4));
}

void read_raw_field(std::istream& in, int nbytes, double* fld, size_t n)
{
    check_nbytes(nbytes);
    if (nbytes == 8) { in.read(reinterpret_cast<char*>(fld), (std::streamsize)(n * 8)); return; }
    std::vector<float> f(n);
    in.read(reinterpret_cast<char*>(f.data()), (std::streamsize)(n * 4));
    for (size_t j = 0; j < n; j++) fld[j] = f[j];
}

void write_header_preamble(const std::string& path, const std::string& wrb_name, int file_type, bool flip, int nf)
{
    std::ofstream h(path, std::ios::out | std::ios::trunc);  // gen_enc.cpp:509-520
    if (!h.is_open()) die("Cannot open " + path);
    h << " ===== Header file for compressed data =====" << std::endl;
    h << " Coder version: " << kCoderVersion << std::endl;
    h << " Encoded data file name: " << wrb_name << std::endl;
    h << " File type (0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++): "
      << file_type << std::endl;
    h << (flip ? " Converted big endian to little endian or vice versa" : " No endian conversion") << std::endl;
    h << " Number of fields in the file, nf: " << nf << std::endl;
}

namespace {
// 19 significant digits: numeric_limits<long double>::digits10 + 1 on x86-64 (gen_aux.cpp:532, Q4)
std::string d19(double v)
{
    std::ostringstream s;
    s << std::setprecision(19) << v;
    return s.str();
}
}  // namespace

void append_field_header(const std::string& path, int id, const FieldHeader& h, unsigned long reminder_ntot_enc)
{
    std::ofstream fs(path, std::ios::out | std::ios::app);  // gen_aux.cpp:505-556
    if (!fs.is_open()) die("Cannot open " + path);
    const FieldSpec& s = h.spec;
    fs << " -----" << std::endl;
    fs << id << std::endl;
    fs << " nbytes; recl; nx; ny; nz; nh; idinv; icomp;";
    if (s.icomp) fs << " tol_base; tolabs; midval; halfspanval; wlev; nlay; ntot_enc;";
    if (reminder_ntot_enc > 0) fs << " deps_vec(1:nlay); minval_vec(1:nlay); len_enc_vec(1:nlay)" << std::endl;
    else fs << std::endl;
    fs << s.nbytes << std::endl;
    for (int j = 0; j < 8; j++) fs << std::hex << static_cast<unsigned>(h.recl[j]) << " ";
    fs << std::dec << std::endl;
    fs << s.nx << std::endl << s.ny << std::endl << s.nz << std::endl << s.nh << std::endl;
    fs << s.idinv << std::endl << s.icomp << std::endl;
    if (s.icomp > 0) {
        fs << d19(s.tol_base) << std::endl << d19(h.tolabs) << std::endl;
        fs << d19(h.midval) << std::endl << d19(h.halfspanval) << std::endl;
        fs << h.wlev << std::endl << h.nlay << std::endl << h.ntot_enc << std::endl;
        if (h.ntot_enc > 0) {
            for (unsigned j = 0; j < h.nlay; j++) fs << d19(h.deps_vec[j]) << " ";
            fs << std::endl;
            for (unsigned j = 0; j < h.nlay; j++) fs << d19(h.minval_vec[j]) << " ";
            fs << std::endl;
            for (unsigned j = 0; j < h.nlay; j++) fs << h.len_enc_vec[j] << " ";
            fs << std::endl;
        }
    }
}

int read_header_preamble(std::istream& in)
{
    std::string line;
    for (int j = 0; j < 5; j++) std::getline(in, line);  // gen_dec.cpp:160-163
    std::getline(in, line);
    line.erase(0, 34);  // " Number of fields in the file, nf:"
    int nf = 0;
    std::stringstream(line) >> nf;
    return nf;
}

void read_field_header(std::istream& fs, int id, FieldHeader& h)
{
    std::string line;  // gen_aux.cpp:559-624
    FieldSpec& s = h.spec;
    std::getline(fs, line);
    int found = -1;
    fs >> found;
    if (found != id) {
        std::cout << "Encoding header file read error" << std::endl;
        std::cout << "Reading field " << id << ", found field " << found << std::endl;
        throw std::runtime_error("header field id mismatch");
    }
    std::getline(fs, line);
    std::getline(fs, line);
    fs >> s.nbytes;
    for (int j = 0; j < 8; j++) { int b = 0; fs >> std::hex >> b; h.recl[j] = (unsigned char)b; }
    fs >> std::dec;
    std::getline(fs, line);
    fs >> s.nx >> s.ny >> s.nz >> s.nh >> s.idinv >> s.icomp;
    if (s.icomp > 0) {
        fs >> s.tol_base >> h.tolabs >> h.midval >> h.halfspanval;
        fs >> h.wlev >> h.nlay >> h.ntot_enc;
        std::getline(fs, line);
        if (h.nlay > (unsigned)kNlayMax) throw std::runtime_error("header: nlay > 8");
        if (h.ntot_enc > 0) {
            for (unsigned j = 0; j < h.nlay; j++) fs >> h.deps_vec[j];
            std::getline(fs, line);
            for (unsigned j = 0; j < h.nlay; j++) fs >> h.minval_vec[j];
            std::getline(fs, line);
            for (unsigned j = 0; j < h.nlay; j++) fs >> h.len_enc_vec[j];
            std::getline(fs, line);
        }
    } else {
        std::getline(fs, line);
    }
    if (fs.fail()) throw std::runtime_error("encoding header file is truncated or malformed");
}

}  // namespace wrio
