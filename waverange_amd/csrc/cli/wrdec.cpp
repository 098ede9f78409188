// wrdec -- generic decoder command-line tool on top of libwaverange_amd.
//
// Same command line, prompts and output files as the reference's generic decoder
// (src/generic/gen_dec.cpp):   wrdec ENCODED_FILE HEADER_FILE EXTRACTED_FILE TYPE ENDIANFLIP
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <future>
#include <memory>
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

int main(int argc, char** argv)
{
    string in_name = "data.wrb", header_name = "data.wrh", out_name = "datarec.bin";
    int file_type = 0, flip = 0;
    cout << "usage: ./wrdec ENCODED_FILE HEADER_FILE EXTRACTED_FILE TYPE ENDIANFLIP\n";
    cout << "where TYPE=(0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++) and ENDIANFLIP=(0:no; 1:yes)\n";
    cout << "interactive mode if not enough arguments are passed.\n";
    if (argc == 6) {  // gen_dec.cpp:105-117
        cout << "automatic mode.";
        in_name = argv[1]; header_name = argv[2]; out_name = argv[3];
        std::stringstream(string(argv[4])) >> file_type;
        std::stringstream(string(argv[5])) >> flip;
    } else {  // gen_dec.cpp:118-135
        auto ask = [](const char* prompt) { cout << prompt; string s; std::getline(std::cin, s); return s; };
        string s;
        s = ask("Enter encoded data file name [data.wrb]: "); if (!s.empty()) in_name = s;
        s = ask("Enter encoding header file name [data.wrh]: "); if (!s.empty()) header_name = s;
        s = ask("Enter extracted (output) data file name [datarec.bin]: "); if (!s.empty()) out_name = s;
        s = ask("Enter file type (0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++) [0]: ");
        if (!s.empty()) std::stringstream(s) >> file_type;
        s = ask("Enter endian conversion (0: do not perform; 1: inversion) [0]: ");
        if (!s.empty()) std::stringstream(s) >> flip;
    }
    cout << endl << "=== Decoding parameters ===" << endl;
    cout << "Encoded data file name " << in_name << endl;
    cout << "Encoding header file name " << header_name << endl;
    cout << "Extracted (output) data file name: " << out_name << endl;
    cout << "File type (0: Fortran sequential w 4-byte recl; 1: Fortran sequential w 8-byte recl; 2: C/C++): " << file_type << endl;
    if (flip) cout << "Convert big endian to little endian or vice versa" << endl;
    if (file_type < 0 || file_type > 2) {
        cout << "Error: unknown file type" << endl;
        cout << "=== End of decompression ===\n";
        return 0;
    }

    std::ifstream fheader(header_name);
    if (!fheader.is_open()) { cout << "Cannot open " << header_name << endl; return 1; }
    const int nf = wrio::read_header_preamble(fheader);
    std::ifstream finput(in_name, std::ios::binary | std::ios::in);
    if (!finput.is_open()) { cout << "Cannot open " << in_name << endl; return 1; }

    // Field pipeline, as in wrenc (the reference decodes one field after the other, gen_dec.cpp:180-260): the main
    // thread reads field k's header record and coded bytes and hands them to a worker thread (decoding_wrap: host range
    // decoder, GPU kernels, download; min/max for the log), a writer thread writes finished fields to the output file
    // in field order; up to `depth` fields in flight (wr_autotune_batch, which also starts the library's coder pool;
    // WR_CLI_PIPELINE overrides, 0 = strictly one after the other).
    struct Item {
        wrio::FieldHeader h;
        std::unique_ptr<double[]> fld;   // not zero-filled: decoding_wrap writes every element
        wrcli::RawBuffer data_enc;
        std::future<void> done;
        bool decoded = false;
        std::ostringstream log;
    };
    std::vector<Item> items(nf);
    // the field sizes are only known record by record; the first record sizes the pipeline
    int depth = -1;
    wrcli::InFlight* gate = nullptr;
    std::unique_ptr<wrcli::InFlight> gate_owner;
    std::thread writer;
    std::exception_ptr writer_error;
    auto tail = [&](Item& im, std::ostream& os) {  // after the decode: what the reference prints about the field
        const size_t ntot = im.h.spec.count();
        if (im.decoded) os << "  decode: fld_1d_rec[0]=" << im.fld[0] << " fld_1d_rec[last]=" << im.fld[ntot - 1] << endl;
        double lo, hi;
        wrcli::minmax(im.fld.get(), ntot, &lo, &hi);
        os << "        min=" << lo << " max=" << hi << endl;
    };
    auto finish = [&](int it) {
        Item& im = items[it];
        const wrio::FieldSpec& s = im.h.spec;
        const size_t ntot = s.count();
        if (im.done.valid()) im.done.get();
        cout << im.log.str();
        wrio::write_field(out_name, it == 0, file_type, flip != 0, s, im.h.recl, im.fld.get());
        cout << "  wrote: fld_1d_rec[0]=" << im.fld[0] << " fld_1d_rec[last]=" << im.fld[ntot - 1] << endl;
        im.fld.reset();
        im.data_enc.release();
    };
    try {
    for (int it = 0; it < nf; it++) {
        Item& im = items[it];
        wrio::FieldHeader& h = im.h;
        wrio::read_field_header(fheader, it, h);
        const wrio::FieldSpec& s = h.spec;
        if (depth < 0) {
            depth = wrcli::fields_in_flight(s.count(), nf);
            if (depth > 0) {
                setenv("WR_QUIET", "1", 0);
                gate_owner.reset(new wrcli::InFlight(depth));
                gate = gate_owner.get();
                writer = std::thread([&]() {
                    try {
                        for (int k = 0; k < nf; k++) { gate->wait_launched(k); finish(k); gate->leave(); }
                    } catch (...) { writer_error = std::current_exception(); gate->abort(); }
                });
            }
        }
        if (gate && !gate->enter()) break;
        std::ostream& os = depth > 0 ? static_cast<std::ostream&>(im.log) : cout;
        // echo of the header values, gen_aux.cpp:626-643
        os << "  tolabs; midval; halfspanval; wlev; nlay; ntot_enc;";
        if (h.ntot_enc > 0) os << " deps_vec(1:nlay); minval_vec(1:nlay); len_enc_vec(1:nlay)" << endl; else os << endl;
        os << "  " << h.tolabs << " " << h.midval << " " << h.halfspanval << " " << h.wlev << " " << h.nlay << " " << h.ntot_enc << endl;
        if (h.ntot_enc > 0) {
            os << "  "; for (unsigned j = 0; j < h.nlay; j++) os << h.deps_vec[j] << " "; os << endl;
            os << "  "; for (unsigned j = 0; j < h.nlay; j++) os << h.minval_vec[j] << " "; os << endl;
            os << "  "; for (unsigned j = 0; j < h.nlay; j++) os << h.len_enc_vec[j] << " "; os << endl;
        }
        os << "  contains " << s.nbytes << "-byte floating point data" << endl;
        os << "  nx=" << s.nx << "  ny=" << s.ny << "  nz=" << s.nz << "  nh=" << s.nh;
        if (s.idinv) os << " and reordering" << endl; else os << endl;
        const size_t ntot = s.count();
        im.fld.reset(new double[ntot]);
        if (s.icomp) {
            if (h.ntot_enc > 0) {
                im.data_enc.allocate(h.ntot_enc);
                finput.read(reinterpret_cast<char*>(im.data_enc.data()), (std::streamsize)h.ntot_enc);
                if (finput.fail()) { cout << "Cannot read from " << in_name << endl; throw std::runtime_error("short .wrb"); }
            } else
                for (size_t j = 0; j < ntot; j++) im.fld[j] = h.midval;  // gen_dec.cpp:201: a trivial field is its mid value
        } else {
            wrio::read_raw_field(finput, s.nbytes, im.fld.get(), ntot);
        }
        im.decoded = s.icomp && h.ntot_enc > 0;
        if (im.decoded) os << "  decoding fld_1d_rec, field number " << it << endl;
        Item* ip = &im;
        const bool pipelined = depth > 0;
        auto work = [ip, pipelined, &tail]() {
            if (ip->decoded) {
                wrio::FieldHeader& hh = ip->h;
                const wrio::FieldSpec& sp = hh.spec;
                unsigned char wlev = (unsigned char)hh.wlev, nlay = (unsigned char)hh.nlay;
                decoding_wrap(sp.nx, sp.ny, sp.nz * sp.nh, ip->fld.get(), &hh.tolabs, &hh.midval, &hh.halfspanval, &wlev, &nlay,
                              &hh.ntot_enc, hh.deps_vec, hh.minval_vec, hh.len_enc_vec, ip->data_enc.data());
            }
            tail(*ip, pipelined ? static_cast<std::ostream&>(ip->log) : cout);
        };
        if (depth > 0) { im.done = std::async(std::launch::async, work); gate->launched(it); }
        else { work(); finish(it); }
    }
    } catch (...) {
        if (gate) gate->abort();
        if (writer.joinable()) writer.join();
        cout << "=== decompression failed ===\n";
        return 1;
    }
    if (writer.joinable()) writer.join();
    if (writer_error) { cout << "=== decompression failed ===\n"; return 1; }
    cout << "=== End of decompression ===\n";
    return 0;
}
