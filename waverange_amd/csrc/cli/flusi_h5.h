// flusi_h5.h -- HDF5 container of the FluSI front-end (reference src/flusi/hdf5_interfaces.cpp).
//
// Layout contract (file:line of the reference at each function in flusi_h5.cpp):
//   original / reconstructed file : dataset <name>, 3-D (nz, ny, nx), float or double, plus
//       regular output:  attributes time, viscosity, epsi (double[1]), domain_size (double[3]), nxyz (int[3])
//       backup:          attribute bckp (double[8]) = (time, dt1, dt0, n1, it, nx, ny, nz)
//   compressed file : dataset <name>, 1-D uchar[ntot_enc], the same attributes, plus coder_version (int),
//       tolabs, midval, halfspanval (double[1]), wlev, nlay (uchar[1]), ntot_enc (ulong[1]),
//       deps_vec, minval_vec (double[nlay]), len_enc_vec (ulong[nlay])
#pragma once
#include <string>
#include <vector>

#include "../../../include/waverange_amd.h"

namespace flusi {

constexpr int kCoderVersion = 31503;  // reference src/core/defs.h:34

// the 50 dataset names a FluSI backup may hold, in the reference's order (main_enc.cpp:319-330)
extern const char* const kBackupNames[50];

void create_file(const std::string& path);                         // truncate / create
std::vector<std::string> dataset_names(const std::string& path);   // root-level datasets, name order
bool has_dataset(const std::string& path, const std::string& name);

bool read_attr_double(const std::string& path, const std::string& dset, const char* attr, double* out, int n);
bool read_attr_int(const std::string& path, const std::string& dset, const char* attr, int* out, int n);
void write_attr_double(const std::string& path, const std::string& dset, const char* attr, const double* v, int n);
void write_attr_int(const std::string& path, const std::string& dset, const char* attr, const int* v, int n);

// field data, converted to / from double by the library as the reference does (hdf5_interfaces.cpp:716-738, 671-701)
void read_field(const std::string& path, const std::string& dset, std::vector<double>& fld, size_t expect);
void write_field(const std::string& path, const std::string& dset, const double* fld, int nx, int ny, int nz, bool single);

// coded bytes + coding attributes (hdf5_interfaces.cpp:283-441, 741-815)
void write_coded(const std::string& path, const std::string& dset, const unsigned char* data, const wr_enc_info& info);
void read_coded(const std::string& path, const std::string& dset, std::vector<unsigned char>& data, wr_enc_info& info);

}  // namespace flusi
