// mssg_io.cpp -- see mssg_io.h.  All file:line citations are relative to the reference tree.
#include "mssg_io.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>

namespace wrmssg {

namespace {

[[noreturn]] void die(const std::string& msg)
{
    std::cout << msg << std::endl;
    std::exit(1);
}

std::string slurp(const std::string& path, const char* what)
{
    std::ifstream is(path.c_str(), std::ios::in | std::ios::binary);
    if (!is) die(std::string("Unable to open ") + what + " " + path);  // ctrl_aux.cpp:57-65, 207-215
    std::ostringstream ss;
    ss << is.rdbuf();
    return ss.str();
}

bool one_of(const std::string& s, std::initializer_list<const char*> names)
{
    for (const char* n : names)
        if (s == n) return true;
    return false;
}

// last value recorded under `name`, or null
const std::string* last_value(const std::vector<std::pair<std::string, std::string>>& tab, const char* name)
{
    const std::string* v = nullptr;
    for (const auto& kv : tab)
        if (kv.first == name) v = &kv.second;
    return v;
}

}  // namespace

// ctrl_aux.cpp:199-298.  Words end at a blank, a '^' or a line end (a word the file ends in without
// one of those is never seen).  On every line the first word that is one of the six keywords takes
// the next word OF THAT LINE as its value; the rest of the line is ignored.  A later line overrides
// an earlier one.
GradsControl read_grads_control(const std::string& path)
{
    const std::string text = slurp(path, "control file");
    std::vector<std::pair<std::string, std::string>> tab;
    std::string word, pending;
    enum { kIgnore, kWantName, kWantValue } mode = kWantName;
    for (char ch : text) {
        const bool sep = ch == '\n' || ch == '^' || ch == ' ';
        if (!sep) { word.push_back(ch); continue; }
        if (!word.empty()) {
            if (mode == kWantName) {
                if (one_of(word, {"DSET", "UNDEF", "XDEF", "YDEF", "ZDEF", "TDEF"})) { pending = word; mode = kWantValue; }
            } else if (mode == kWantValue) {
                tab.emplace_back(pending, word);
                mode = kIgnore;
            }
            word.clear();
        }
        if (ch == '\n') mode = kWantName;  // a keyword without a value on its line is dropped
    }
    GradsControl c;
    if (const std::string* v = last_value(tab, "DSET")) c.dset = *v;
    if (const std::string* v = last_value(tab, "UNDEF")) c.undef = atof(v->c_str());
    if (const std::string* v = last_value(tab, "XDEF")) c.nx = atoi(v->c_str());
    if (const std::string* v = last_value(tab, "YDEF")) c.ny = atoi(v->c_str());
    if (const std::string* v = last_value(tab, "ZDEF")) c.nz = atoi(v->c_str());
    if (const std::string* v = last_value(tab, "TDEF")) c.nt = atoi(v->c_str());
    return c;
}

// ctrl_aux.cpp:49-195.  Words end at a line end, '&', blank, quote or comma.  A word that is one of the
// ten names and is FOLLOWED BY A SEPARATOR (so "nx = 4", not "nx=4": '=' discards the word before it)
// opens an entry; the first word after the next '=' closes it.  Then: nx/ny (regional grid) or
// npg/i_over/j_over (Yin-Yang grid, sizes as hard-wired in MSSG), nr, nproc, dim_size; every
// (var, rec) pair from the first "var" on names record `rec`.
RestartControl read_restart_control(const std::string& path)
{
    const std::string text = slurp(path, "namelist file");
    std::vector<std::pair<std::string, std::string>> tab;
    std::string word, pending;
    bool open = false;        // an entry waits for its value
    bool after_equal = false; // the word being read follows an '='
    for (char ch : text) {
        const bool sep = ch == '\n' || ch == '&' || ch == ' ' || ch == '\'' || ch == ',';
        if (ch == '=') { after_equal = true; word.clear(); continue; }
        if (!sep) { word.push_back(ch); continue; }
        if (word.empty()) continue;  // runs of separators change nothing, not even "after '='"
        if (after_equal) {
            if (open) { tab.emplace_back(pending, word); open = false; }
        } else if (one_of(word, {"nx", "ny", "nr", "npg", "i_over", "j_over", "nproc", "dim_size", "var", "rec"})) {
            pending = word;
            open = true;
        }
        after_equal = false;
        word.clear();
    }
    RestartControl c;
    int nproc = 0;
    for (const auto& kv : tab) {
        const int v = atoi(kv.second.c_str());
        if (kv.first == "nx") {
            c.nx = v;
            if (const std::string* y = last_value(tab, "ny")) c.ny = atoi(y->c_str());
        } else if (kv.first == "npg") {
            const int nlg = 3 * v - 4;  // as hard-wired in MSSG (ctrl_aux.cpp:155)
            int i_over = 0, j_over = 0;
            if (const std::string* s = last_value(tab, "i_over")) i_over = atoi(s->c_str());
            if (const std::string* s = last_value(tab, "j_over")) j_over = atoi(s->c_str());
            c.nx = nlg + i_over * 2;
            c.ny = (v + j_over * 2) * 2;  // two grids stacked in y
        } else if (kv.first == "nr") {
            c.nz = v;
        } else if (kv.first == "nproc") {
            nproc = v;
        } else if (kv.first == "dim_size") {
            c.nprocx = v;
        }
    }
    if (c.nprocx <= 0) die("namelist " + path + ": dim_size missing");
    c.nprocy = nproc / c.nprocx;
    size_t i = 0;
    while (i < tab.size() && tab[i].first != "var") i++;
    if (i == tab.size()) die("namelist " + path + ": no record list");
    size_t count = 0;
    for (; i + 1 < tab.size(); i += 2, count++) {  // (var, rec) pairs to the end of the table
        const int rec = atoi(tab[i + 1].second.c_str());
        if (rec < 1) die("namelist " + path + ": bad record number");
        if (c.dsets.size() < (size_t)rec) c.dsets.resize(rec);
        c.dsets[rec - 1] = tab[i].second;
    }
    c.dsets.resize(count);  // the reference counts pairs (ndset) and indexes records 0 .. ndset-1
    return c;
}

namespace {

inline void put_value(unsigned char* dst, double v, int nbytes, bool flip)
{
    unsigned char raw[8];
    if (nbytes == 4) { const float f = float(v); memcpy(raw, &f, 4); }
    else memcpy(raw, &v, 8);
    if (flip) for (int k = 0; k < nbytes; k++) dst[k] = raw[nbytes - 1 - k];
    else memcpy(dst, raw, nbytes);
}

inline double get_value(const unsigned char* src, int nbytes, bool flip)
{
    unsigned char raw[8];
    if (flip) for (int k = 0; k < nbytes; k++) raw[k] = src[nbytes - 1 - k];
    else memcpy(raw, src, nbytes);
    if (nbytes == 4) { float f; memcpy(&f, raw, 4); return f; }
    double d; memcpy(&d, raw, 8); return d;
}

void check_nbytes(int nbytes)
{
    if (nbytes != 4 && nbytes != 8) die("MSSG input nbytes must be equal to 4 or 8");  // ctrl_aux.cpp:310-315, 394-399
}

}  // namespace

// ctrl_aux.cpp:386-457
void read_field(const std::string& path, bool flip_endian, int nbytes, int idset, int nx, int ny, int nz,
                int nxloc, int nyloc, int ixst, int iyst, double* fld)
{
    check_nbytes(nbytes);
    std::ifstream in(path.c_str(), std::ios::in | std::ios::binary);
    if (!in) die("Cannot read from " + path);
    in.seekg((long)idset * (long)nz * (long)nyloc * (long)nxloc * (long)nbytes);
    std::vector<unsigned char> row((size_t)nxloc * nbytes);
    for (int iz = 0; iz < nz; iz++)
        for (int iy = iyst; iy < iyst + nyloc; iy++) {
            in.read(reinterpret_cast<char*>(row.data()), (std::streamsize)row.size());
            if (!in) die("Cannot read from " + path);
            double* dst = fld + (size_t)ixst + (size_t)nx * (size_t)iy + (size_t)nx * (size_t)ny * (size_t)iz;
            for (int ix = 0; ix < nxloc; ix++) dst[ix] = get_value(row.data() + (size_t)ix * nbytes, nbytes, flip_endian);
        }
}

// ctrl_aux.cpp:301-383: record 0 truncates the file, later records append
void write_field(const std::string& path, bool flip_endian, int nbytes, int idset, int nx, int ny, int nz,
                 int nxloc, int nyloc, int ixst, int iyst, const double* fld)
{
    check_nbytes(nbytes);
    std::ofstream out(path.c_str(), std::ios::out | std::ios::binary | (idset == 0 ? std::ios::trunc : std::ios::app));
    if (!out) die("Cannot write to " + path);
    std::vector<unsigned char> row((size_t)nxloc * nbytes);
    for (int iz = 0; iz < nz; iz++)
        for (int iy = iyst; iy < iyst + nyloc; iy++) {
            const double* src = fld + (size_t)ixst + (size_t)nx * (size_t)iy + (size_t)nx * (size_t)ny * (size_t)iz;
            for (int ix = 0; ix < nxloc; ix++) put_value(row.data() + (size_t)ix * nbytes, src[ix], nbytes, flip_endian);
            out.write(reinterpret_cast<const char*>(row.data()), (std::streamsize)row.size());
        }
}

// ctrl_aux.cpp:478-515: 19 significant digits, default float format
void append_header_record(const std::string& path, int idset, const std::string& dsetname, const Coding& c)
{
    std::ofstream fs(path.c_str(), std::ios::out | std::ios::app);
    if (!fs) die("Cannot write to " + path);
    const int prec = std::numeric_limits<long double>::digits10 + 1;
    fs << " -----" << std::endl;
    fs << idset + 1 << std::endl;
    fs << " Data set name = " << dsetname << std::endl;
    fs << " tolabs; midval; halfspanval; wlev; nlay; ntot_enc;";
    if (c.ntot_enc > 0) fs << " deps_vec(1:nlay); minval_vec(1:nlay); len_enc_vec(1:nlay)";
    fs << std::endl;
    fs << std::setprecision(prec) << c.tolabs << std::endl;
    fs << std::setprecision(prec) << c.midval << std::endl;
    fs << std::setprecision(prec) << c.halfspanval << std::endl;
    fs << static_cast<unsigned>(c.wlev) << std::endl;
    fs << static_cast<unsigned>(c.nlay) << std::endl;
    fs << c.ntot_enc << std::endl;
    if (c.ntot_enc > 0) {
        for (int j = 0; j < c.nlay; j++) fs << std::setprecision(prec) << c.deps_vec[j] << " ";
        fs << std::endl;
        for (int j = 0; j < c.nlay; j++) fs << std::setprecision(prec) << c.minval_vec[j] << " ";
        fs << std::endl;
        for (int j = 0; j < c.nlay; j++) fs << c.len_enc_vec[j] << " ";
        fs << std::endl;
    }
}

// ctrl_aux.cpp:518-582
std::string read_header_record(std::istream& fs, int idset, Coding* c)
{
    std::string line;
    std::getline(fs, line);  // " -----"
    int id1 = 0;
    fs >> id1;
    if (id1 != idset + 1) {
        std::cout << "Encoding header file does not match with the control file" << std::endl;
        std::cout << "idset+1 = " << idset + 1 << " idset1 = " << id1 << std::endl;
        std::exit(1);
    }
    std::getline(fs, line);  // rest of the id line
    std::getline(fs, line);  // " Data set name = NAME"
    std::string name = line.size() > 17 ? line.substr(17) : std::string();
    std::getline(fs, line);  // reminder line
    int wlev = 0, nlay = 0;
    fs >> c->tolabs >> c->midval >> c->halfspanval >> wlev >> nlay >> c->ntot_enc;
    c->wlev = (unsigned char)wlev;
    c->nlay = (unsigned char)nlay;
    std::getline(fs, line);
    if (c->ntot_enc > 0) {
        if (nlay < 0 || nlay > kNlayMax) die("Encoding header file: nlay out of range");
        for (int j = 0; j < nlay; j++) fs >> c->deps_vec[j];
        std::getline(fs, line);
        for (int j = 0; j < nlay; j++) fs >> c->minval_vec[j];
        std::getline(fs, line);
        for (int j = 0; j < nlay; j++) fs >> c->len_enc_vec[j];
        std::getline(fs, line);
    }
    return name;
}

// ctrl_aux.cpp:460-468
void append_bytes(const std::string& path, const unsigned char* data, unsigned long n)
{
    std::ofstream out(path.c_str(), std::ios::binary | std::ios::out | std::ios::app);
    if (!out) die("Cannot write to " + path);
    out.write(reinterpret_cast<const char*>(data), (std::streamsize)n);
}

std::string subdomain_label(int proc)
{
    std::ostringstream s;
    s << std::setw(kFileDigits) << std::setfill('0') << proc;  // mssg_enc.cpp:455-456
    return s.str();
}

// mssg_dec.cpp:172-186, 358-372: character by character
void copy_text_file(const std::string& from, const std::string& to)
{
    std::ifstream in(from.c_str(), std::ios::in);
    std::ofstream out(to.c_str(), std::ios::out | std::ios::trunc);
    if (!in || !out) die("Cannot copy " + from + " to " + to);
    char ch;
    while (in.get(ch)) out << ch;
}

}  // namespace wrmssg
