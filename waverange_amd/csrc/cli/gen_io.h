// gen_io.h -- generic (raw / Fortran-sequential) field files and the .wrh/.wrb container of
// the wrenc / wrdec command-line tools.
//
// Format contract = reference src/generic/gen_aux.cpp (file:line cited at each function in
// gen_io.cpp).  The I/O is done in bulk (whole records, then converted in memory) instead of
// the reference's one-ifstream-read-per-element loops; the bytes read and written are the same.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <string>
#include <vector>

namespace wrio {

constexpr int kCoderVersion = 31503;  // reference src/core/defs.h:34
constexpr int kNlayMax = 8;           // reference src/core/defs.h:38

// parameters of one field, as the reference keeps them in its *_vec arrays (gen_enc.cpp:86-88)
struct FieldSpec {
    int nbytes = 8;  // 4: single, 8: double
    int nx = 16, ny = 16, nz = 16, nh = 1;
    int idinv = 0;   // 1: the file stores the dimensions in inverted (C) order
    int icomp = 1;   // 0: store uncompressed
    double tol_base = 1e-16;
    size_t count() const { return (size_t)nx * (size_t)ny * (size_t)nz * (size_t)nh; }
};

// per-field record of the .wrh header (gen_aux.cpp:505-556)
struct FieldHeader {
    FieldSpec spec;
    unsigned char recl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double tolabs = 0, midval = 0, halfspanval = 0;
    unsigned wlev = 0, nlay = 0;
    unsigned long ntot_enc = 0;
    double deps_vec[kNlayMax] = {0};
    double minval_vec[kNlayMax] = {0};
    unsigned long len_enc_vec[kNlayMax] = {0};
};

// file_type: 0 = Fortran sequential, 4-byte record markers; 1 = 8-byte markers; 2 = raw C.
// Reads field `spec` starting at byte *pos of `path` into fld (x fastest), returns the record
// marker bytes in recl and advances *pos (gen_aux.cpp:230-397).
void read_field(const std::string& path, int file_type, bool flip_endian, const FieldSpec& spec,
                unsigned char recl[8], long* pos, std::vector<double>& fld);

// Appends (truncates if first) the field to `path` in the original layout (gen_aux.cpp:49-226).
void write_field(const std::string& path, bool first, int file_type, bool flip_endian,
                 const FieldSpec& spec, const unsigned char recl[8], const double* fld);

// .wrb payloads (gen_aux.cpp:401-408, 419-468)
void append_bytes(const std::string& path, const unsigned char* data, size_t n);
void append_raw_field(const std::string& path, int nbytes, const double* fld, size_t n);
void read_raw_field(std::istream& in, int nbytes, double* fld, size_t n);

// .wrh text (gen_enc.cpp:509-520, gen_aux.cpp:505-556, 559-644, gen_dec.cpp:160-168)
void write_header_preamble(const std::string& path, const std::string& wrb_name, int file_type,
                           bool flip_endian, int nf);
// `reminder_ntot_enc` reproduces quirk Q2: the reference tests the ntot_enc
// variable of the PREVIOUS field when the current one is not compressed.
void append_field_header(const std::string& path, int id, const FieldHeader& h, unsigned long reminder_ntot_enc);
int read_header_preamble(std::istream& in);  // returns nf
void read_field_header(std::istream& in, int id, FieldHeader& h);

}  // namespace wrio
