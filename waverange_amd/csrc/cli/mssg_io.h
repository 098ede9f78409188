// mssg_io.h -- files of the MSSG front-end (wrenc_mssg / wrdec_mssg): GrADS control files of regular
// output, namelists of restart ("backup") sets, the flat binary field files, and the text header /
// byte payload pair the coder writes next to them.
//
// Format contract = reference src/mssg/ctrl_aux.cpp and the file handling of src/mssg/mssg_enc.cpp,
// mssg_dec.cpp (file:line cited at each function in mssg_io.cpp).  Field files are read and written in
// bulk (one subdomain row at a time, converted in memory) instead of one stream call per element; the
// bytes are the same.
#pragma once
#include <iosfwd>
#include <string>
#include <vector>

namespace wrmssg {

constexpr int kCoderVersion = 31503;      // reference src/core/defs.h:34
constexpr int kNlayMax = 8;               // defs.h:38
constexpr int kFileDigits = 4;            // defs.h:54  MSSG_FILE_DIG: width of the subdomain number in ".p_0012"
constexpr int kTimeRecLen = 15;           // defs.h:56  MSSG_TIME_REC_LEN
constexpr double kMaskTolRel = 0.126;     // defs.h:58  MSSG_MASK_TOLREL
constexpr double kMaskThresholdAcc = 1e-4;  // defs.h:60  MSSG_MASK_THRESHOLD_ACC

// GrADS control file of a regular-output set (ctrl_aux.cpp:199-298)
struct GradsControl {
    int nx = 0, ny = 0, nz = 0, nt = 0;
    double undef = 0;
    std::string dset;  // data file name
};
GradsControl read_grads_control(const std::string& path);

// namelist of a restart set (ctrl_aux.cpp:49-195): global size, subdomain grid, record names
struct RestartControl {
    int nx = 0, ny = 0, nz = 0, nprocx = 0, nprocy = 0;
    std::vector<std::string> dsets;  // by record number
};
RestartControl read_restart_control(const std::string& path);

// One field of a flat binary file: record `idset` of nz * nyloc * nxloc values of `nbytes` bytes, placed
// at (ixst, iyst) of a global nx * ny * nz array, x fastest (ctrl_aux.cpp:386-457 / 301-383).
void read_field(const std::string& path, bool flip_endian, int nbytes, int idset, int nx, int ny, int nz,
                int nxloc, int nyloc, int ixst, int iyst, double* fld);
void write_field(const std::string& path, bool flip_endian, int nbytes, int idset, int nx, int ny, int nz,
                 int nxloc, int nyloc, int ixst, int iyst, const double* fld);

// coding attributes of one data set, as encoding_wrap returns them
struct Coding {
    double tolabs = 0, midval = 0, halfspanval = 0;
    unsigned char wlev = 0, nlay = 0;
    unsigned long ntot_enc = 0;
    double deps_vec[kNlayMax] = {0};
    double minval_vec[kNlayMax] = {0};
    unsigned long len_enc_vec[kNlayMax] = {0};
};
// header records (ctrl_aux.cpp:478-515, 518-582) and payload (ctrl_aux.cpp:460-475)
void append_header_record(const std::string& path, int idset, const std::string& dsetname, const Coding& c);
std::string read_header_record(std::istream& fs, int idset, Coding* c);  // returns the data set name
void append_bytes(const std::string& path, const unsigned char* data, unsigned long n);

// the parameters both tools take from argv, from stdin, and (encoder only) from a file "inmeta"
std::string subdomain_label(int proc);  // "0007"
void copy_text_file(const std::string& from, const std::string& to);

}  // namespace wrmssg
