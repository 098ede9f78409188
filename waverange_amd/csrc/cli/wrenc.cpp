// wrenc -- generic encoder command-line tool on top of libwaverange_amd.
//
// Same command line, `inmeta` control file (new key=value and old line formats), interactive
// prompts and .wrh/.wrb output as the reference's generic encoder (src/generic/gen_enc.cpp):
//   wrenc INPUT_FILE ENCODED_FILE HEADER_FILE TYPE ENDIANFLIP NF PRECISION NX NY NZ TOLERANCE
// Parameter sources in the reference's priority order (gen_enc.cpp:112-486): a file named
// `inmeta` in the working directory, then 11 arguments, then interactive prompts.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <future>
#include <thread>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/waverange_amd.h"
#include "batch.h"
#include "gen_io.h"

using std::cout;
using std::endl;
using std::string;

namespace {

struct Job {
    string in_name = "data.bin", out_name = "data.wrb", header_name = "data.wrh";
    int file_type = 0, flip = 0, nf = 1;
    std::vector<wrio::FieldSpec> fields;
    // the value the reference leaves in its scalar `tol_base` when parsing ends; it alone
    // feeds cutoffvec[0] for EVERY field (quirk Q1, gen_enc.cpp:499-503)
    double effective_tol = 1e-16;
};

string trim(const string& s)
{
    const char* ws = " \t\v\r\n";
    size_t a = s.find_first_not_of(ws);
    if (a == string::npos) return "";
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
}

string lower(string s) { std::transform(s.begin(), s.end(), s.begin(), ::tolower); return s; }

template <class T> void parse_into(const string& s, T& v) { if (!s.empty()) std::stringstream(s) >> v; }

// "key = value" with exactly one '='; returns false otherwise
bool key_value(const string& line, string& key, string& val, bool& malformed)
{
    size_t n = std::count(line.begin(), line.end(), '=');
    malformed = n != 1;
    if (malformed) return false;
    size_t p = line.find('=');
    key = lower(trim(line.substr(0, p)));
    val = trim(line.substr(p + 1));
    if (val.empty()) { malformed = true; return false; }
    return true;
}

// running per-field defaults: a field inherits whatever the previous one set (gen_enc.cpp:61-101)
struct Running {
    int intype = 2;
    wrio::FieldSpec s;
    void set(const string k[8])
    {
        parse_into(k[0], intype); parse_into(k[1], s.nx); parse_into(k[2], s.ny); parse_into(k[3], s.nz);
        parse_into(k[4], s.nh); parse_into(k[5], s.idinv); parse_into(k[6], s.icomp); parse_into(k[7], s.tol_base);
        s.nbytes = intype == 1 ? 4 : 8;
    }
};

void echo_field(const string k[8])
{
    static const char* names[8] = {"input_data_type", "nx", "ny", "nz", "nh", "order", "compress", "tolerance"};
    for (int i = 0; i < 8; i++) cout << names[i] << " = " << k[i] << endl;
    cout << "" << endl;
}

void echo_common(const Job& j, const string& ft, const string& ec, const string& nf)
{
    cout << "in_name = " << j.in_name << endl << "out_name = " << j.out_name << endl;
    cout << "header_name = " << j.header_name << endl << "file_type = " << ft << endl;
    cout << "endian_conversion = " << ec << endl << "number_of_field = " << nf << endl << "" << endl;
}

// `inmeta`, new format: "&key = value" lines, "%field = k" blocks closed by "/"  (gen_enc.cpp:119-279)
// returns 1 parsed, 0 not this format, -1 error
int parse_inmeta_new(const std::vector<string>& lines, Job& job)
{
    string ft, ec, nf;
    bool any = false;
    for (const string& raw : lines) {
        string l = trim(raw);
        if (l.empty() || l[0] != '&') continue;
        string k, v; bool bad;
        if (!key_value(l, k, v, bad)) {
            if (std::count(l.begin(), l.end(), '=') > 1) cout << "==== Error : '=' exists twice in a sentence :" << l << " ====" << endl;
            else cout << "==== Error : 'value' is missing in a sentence :" << l << " ====" << endl;
            return -1;
        }
        any = true;
        if (k == "&in_name") job.in_name = v;
        if (k == "&out_name") job.out_name = v;
        if (k == "&header_name") job.header_name = v;
        if (k == "&file_type") ft = v;
        if (k == "&endian_conversion") ec = v;
        if (k == "&number_of_field") nf = v;
    }
    if (!any) return 0;
    echo_common(job, ft, ec, nf);
    parse_into(ft, job.file_type); parse_into(ec, job.flip); parse_into(nf, job.nf);
    job.fields.assign(job.nf, wrio::FieldSpec());
    Running run;
    string k8[8];
    int blocks = 0, field_id = -1;
    for (const string& raw : lines) {
        string l = trim(raw);
        if (l.empty()) continue;
        string k, v; bool bad;
        if (l[0] == '%' && key_value(l, k, v, bad) && k == "%field") {
            std::stringstream(v) >> field_id;
            cout << "==== read parameters for field " << field_id << " ====" << endl;
            blocks++;
        }
        if (l[0] == '&' && key_value(l, k, v, bad)) {
            static const char* keys[8] = {"&input_data_type", "&nx", "&ny", "&nz", "&nh", "&order", "&compress", "&tolerance"};
            for (int i = 0; i < 8; i++) if (k == keys[i]) k8[i] = v;
        }
        if (l[0] == '/') {
            echo_field(k8);
            run.set(k8);
            if (field_id >= 0 && field_id < job.nf) job.fields[field_id] = run.s;
        }
    }
    if (blocks != job.nf) {
        cout << "==== Number of fields is " << job.nf << " ====" << endl;
        cout << "==== Number of blocks for field parameters are not sufficient. ====" << endl;
        cout << "==== Numbef of block = " << blocks << " ====" << endl;
        return -1;
    }
    job.effective_tol = run.s.tol_base;
    return 1;
}

// `inmeta`, old format: one value per line in prompt order (gen_enc.cpp:282-349)
void parse_inmeta_old(const std::vector<string>& lines, Job& job)
{
    cout << "==== read parameters from inmeta as old format. ====" << endl;
    size_t p = 0;
    auto next = [&]() { return p < lines.size() ? lines[p++] : string(); };
    job.in_name = next(); job.out_name = next(); job.header_name = next();
    string ft = next(), ec = next(), nf = next();
    echo_common(job, ft, ec, nf);
    if (job.in_name.empty()) job.in_name = "data.bin";
    if (job.out_name.empty()) job.out_name = "data.wrb";
    if (job.header_name.empty()) job.header_name = "data.wrh";
    parse_into(ft, job.file_type); parse_into(ec, job.flip); parse_into(nf, job.nf);
    Running run;
    for (int it = 0; it < job.nf; it++) {
        string k8[8];
        for (auto& s : k8) s = next();
        echo_field(k8);
        run.set(k8);
        job.fields.push_back(run.s);
    }
    job.effective_tol = run.s.tol_base;
}

void usage()
{
    cout << "usage: ./wrenc INPUT_FILE ENCODED_FILE HEADER_FILE TYPE ENDIANFLIP NF PRECISION NX NY NZ TOLERANCE\n";
    cout << "where TYPE=(0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++),\n";
    cout << "      ENDIANFLIP=(0:no; 1:yes), NF=(how many fields, e.g. 1), PRECISION=(1:single; 2:double),\n";
    cout << "      NX=(e.g. 16), NY=(e.g. 16), NZ=(e.g. 16) and TOLERANCE=(e.g. 1.0e-16)\n";
    cout << "interactive mode if not enough arguments are passed.\n";
}

void parse_argv(char** argv, Job& job)  // gen_enc.cpp:365-412
{
    cout << "automatic mode.";
    job.in_name = argv[1]; job.out_name = argv[2]; job.header_name = argv[3];
    Running run;
    parse_into(string(argv[4]), job.file_type); parse_into(string(argv[5]), job.flip);
    parse_into(string(argv[6]), job.nf); parse_into(string(argv[7]), run.intype);
    parse_into(string(argv[8]), run.s.nx); parse_into(string(argv[9]), run.s.ny);
    parse_into(string(argv[10]), run.s.nz); parse_into(string(argv[11]), run.s.tol_base);
    run.s.nbytes = run.intype == 1 ? 4 : 8;
    job.fields.assign(job.nf > 0 ? job.nf : 0, run.s);
    job.effective_tol = run.s.tol_base;
}

void parse_interactive(Job& job)  // gen_enc.cpp:413-486
{
    auto ask = [](const char* prompt) { cout << prompt; string s; std::getline(std::cin, s); return s; };
    string s;
    s = ask("Enter input data file name [data.bin]: "); if (!s.empty()) job.in_name = s;
    s = ask("Enter encoded data file name [data.wrb]: "); if (!s.empty()) job.out_name = s;
    s = ask("Enter encoding header file name [data.wrh]: "); if (!s.empty()) job.header_name = s;
    parse_into(ask("Enter file type (0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++) [0]: "), job.file_type);
    parse_into(ask("Enter endian conversion (0: do not perform; 1: inversion) [0]: "), job.flip);
    parse_into(ask("Enter the number of fields in the file, nf [1]: "), job.nf);
    Running run;
    for (int it = 0; it < job.nf; it++) {
        cout << "Field number " << it << endl;
        parse_into(ask("Enter input data type (1: float; 2: double) [2]: "), run.intype);
        run.s.nbytes = run.intype == 1 ? 4 : 8;
        parse_into(ask("Enter the number of data points in the first dimension, nx [16]: "), run.s.nx);
        parse_into(ask("Enter the number of data points in the second dimension, ny [16]: "), run.s.ny);
        parse_into(ask("Enter the number of data points in the third dimension, nz [16]: "), run.s.nz);
        parse_into(ask("Enter the number of data points in the higher (slowest) dimensions, nh [1]: "), run.s.nh);
        parse_into(ask("Invert the order of the dimensions? (0: no; 1: yes) [0]: "), run.s.idinv);
        parse_into(ask("Enter compression flag (0: do not compress; 1: compress) [1]: "), run.s.icomp);
        wrio::FieldSpec f = run.s;
        if (run.s.icomp) { parse_into(ask("Enter base cutoff relative tolerance [1e-16]: "), run.s.tol_base); f.tol_base = run.s.tol_base; }
        else f.tol_base = 0;
        job.fields.push_back(f);
    }
    job.effective_tol = run.s.tol_base;
}

}  // namespace

int main(int argc, char** argv)
{
    Job job;
    std::ifstream meta("inmeta");
    if (!meta.fail()) {
        cout << "==== inmeta exists. ====" << endl;
        std::vector<string> lines;
        for (string l; std::getline(meta, l);) lines.push_back(l);
        int rc = parse_inmeta_new(lines, job);
        if (rc < 0) return -1;
        if (rc == 0) parse_inmeta_old(lines, job);
    } else {
        cout << "==== inmeta doesn't exists. ====" << endl;
        usage();
        if (argc == 12) parse_argv(argv, job); else parse_interactive(job);
    }

    cout << endl << "=== Compression parameters ===" << endl;
    cout << "Input data file name: " << job.in_name << endl;
    cout << "Encoded data file name: " << job.out_name << endl;
    cout << "Encoding header file name: " << job.header_name << endl;
    cout << "File type (0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++): " << job.file_type << endl;
    if (job.flip) cout << "Convert big endian to little endian or vice versa" << endl;
    cout << "Number of fields in the file, nf: " << job.nf << endl;

    wrio::write_header_preamble(job.header_name, job.out_name, job.file_type, job.flip != 0, job.nf);
    { std::ofstream trunc(job.out_name, std::ios::binary | std::ios::out | std::ios::trunc); }
    if (job.file_type < 0 || job.file_type > 2) { cout << "Error: unknown file type" << endl; return 0; }

    // Field pipeline.  The reference codes one field after the other (gen_enc.cpp:538-605); here up to `depth` fields
    // are in flight: the main thread reads field k from the input file and hands it to a worker thread (min/max for the
    // log, encoding_wrap: upload, GPU kernels, host range coder), a writer thread appends the header records and coded
    // bytes of finished fields in field order.  With more than one field the library's coder pool codes the plane
    // streams of all fields in flight on one worker per CPU, and says how many fields fit (wr_autotune_batch;
    // WR_CLI_PIPELINE overrides; 0 = strictly one after the other with the reference's order of log lines).
    size_t max_elems = 0;
    for (const wrio::FieldSpec& s : job.fields) max_elems = std::max(max_elems, s.count());
    int depth = wrcli::fields_in_flight(max_elems, job.nf);
    if (depth > 0) setenv("WR_QUIET", "1", 0);  // the library's progress lines of concurrent fields would interleave
    setenv("WR_WRITEBACK_RESIDUAL", "0", 0);    // the residual encoding_wrap leaves in the field array is not used here

    const double cutoff = job.effective_tol;  // quirk Q1: one cutoff for all fields
    long pos = 0;
    unsigned long prev_ntot_enc = 0;  // quirk Q2: stale value reused for uncompressed fields
    struct Item {
        wrio::FieldHeader h;
        std::vector<double> fld;
        wrcli::RawBuffer data_enc;   // setup_wr's worst case, untouched beyond the coded bytes
        std::future<void> done;
        std::ostringstream log;      // this field's lines, printed when it is written (pipelined mode)
    };
    std::vector<Item> items(job.nf);
    auto scan = [](const Item& im, std::ostream& os) {
        const size_t ntot = im.h.spec.count();
        os << "  read: fld_1d[0]=" << im.fld[0] << " fld_1d[last]=" << im.fld[ntot - 1] << endl;
        double lo, hi;
        wrcli::minmax(im.fld.data(), ntot, &lo, &hi);
        os << "        min=" << lo << " max=" << hi << endl;
    };
    auto finish = [&](int it) {  // in field order: wait for the codec, then append to .wrh / .wrb
        Item& im = items[it];
        const wrio::FieldSpec& s = im.h.spec;
        if (im.done.valid()) im.done.get();
        cout << im.log.str();
        if (s.icomp) {
            cout << "        tolabs=" << im.h.tolabs << endl;
            wrio::append_field_header(job.header_name, it, im.h, im.h.ntot_enc);
            if (im.h.ntot_enc > 0) wrio::append_bytes(job.out_name, im.data_enc.data(), im.h.ntot_enc);
            prev_ntot_enc = im.h.ntot_enc;
        } else {
            wrio::append_field_header(job.header_name, it, im.h, prev_ntot_enc);
            wrio::append_raw_field(job.out_name, s.nbytes, im.fld.data(), s.count());
        }
        std::vector<double>().swap(im.fld);
        im.data_enc.release();
    };
    wrcli::InFlight gate(depth);      // fields between "read" and "written"
    std::thread writer;
    std::exception_ptr writer_error;
    if (depth > 0)
        writer = std::thread([&]() {
            try {
                for (int it = 0; it < job.nf; it++) { gate.wait_launched(it); finish(it); gate.leave(); }
            } catch (...) { writer_error = std::current_exception(); gate.abort(); }
        });
    unsigned char recl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    try {
    for (int it = 0; it < job.nf; it++) {
        if (depth > 0 && !gate.enter()) break;  // waits while `depth` fields are in flight
        Item& im = items[it];
        const wrio::FieldSpec& s = job.fields[it];
        std::ostream& os = depth > 0 ? static_cast<std::ostream&>(im.log) : cout;
        os << "Field number " << it << endl;
        os << "  contains " << s.nbytes << "-byte floating point data" << endl;
        os << "  nx=" << s.nx << "  ny=" << s.ny << "  nz=" << s.nz << "  nh=" << s.nh;
        if (s.idinv) os << " and reordering" << endl; else os << endl;
        wrio::read_field(job.in_name, job.file_type, job.flip != 0, s, recl, &pos, im.fld);
        if (depth == 0) scan(im, cout);

        im.h.spec = s;
        for (int j = 0; j < 8; j++) im.h.recl[j] = recl[j];
        if (s.icomp) {
            unsigned char nlaymax; unsigned long cap;
            setup_wr(s.nx, s.ny, s.nz * s.nh, &nlaymax, &cap);
            im.data_enc.allocate(cap);
        }
        Item* ip = &im;
        const bool pipelined = depth > 0;
        auto work = [ip, cutoff, pipelined, &scan]() {
            const wrio::FieldSpec& sp = ip->h.spec;
            if (pipelined) scan(*ip, ip->log);
            if (!sp.icomp) { (pipelined ? static_cast<std::ostream&>(ip->log) : cout) << "  Compression disabled" << endl; return; }
            (pipelined ? static_cast<std::ostream&>(ip->log) : cout) << "  Compression enabled with base relative tolerance " << sp.tol_base << endl;
            unsigned char wlev = 0, nlay = 0;
            double cut = cutoff;
            // nh > 1 folds into z (gen_enc.cpp:559,596)
            encoding_wrap(sp.nx, sp.ny, sp.nz * sp.nh, ip->fld.data(), 1, 1, 1, 1, &cut, &ip->h.tolabs, &ip->h.midval,
                          &ip->h.halfspanval, &wlev, &nlay, &ip->h.ntot_enc, ip->h.deps_vec, ip->h.minval_vec,
                          ip->h.len_enc_vec, ip->data_enc.data());
            ip->h.wlev = wlev; ip->h.nlay = nlay;
        };
        if (depth > 0) { im.done = std::async(std::launch::async, work); gate.launched(it); }
        else { work(); finish(it); }
    }
    } catch (...) {
        gate.abort();
        if (writer.joinable()) writer.join();
        throw;
    }
    if (writer.joinable()) writer.join();
    if (writer_error) std::rethrow_exception(writer_error);
    cout << "=== End of compression ===\n";
    return 0;
}
