// wrdec -- generic decoder command-line tool on top of libwaverange_amd.
//
// Same command line, prompts and output files as the reference's generic decoder
// (src/generic/gen_dec.cpp):   wrdec ENCODED_FILE HEADER_FILE EXTRACTED_FILE TYPE ENDIANFLIP
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <future>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/waverange_amd.h"
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

    // Field pipeline, as in wrenc: while field k is inside decoding_wrap on a worker thread, the main thread
    // reads field k+1's header record and coded bytes and writes field k-1 to the output file, in field
    // order (WR_CLI_PIPELINE fields in flight, default 2; 0 = strictly one after the other).
    int depth = 2;
    if (const char* e = getenv("WR_CLI_PIPELINE")) depth = atoi(e);
    if (depth > nf - 1) depth = nf - 1;
    if (depth < 0) depth = 0;
    if (depth > 0) setenv("WR_QUIET", "1", 0);
    struct Item {
        wrio::FieldHeader h;
        std::vector<double> fld;
        std::vector<unsigned char> data_enc;
        std::future<void> done;
        bool decoded = false;
    };
    std::vector<Item> items(nf);
    auto finish = [&](int it) {
        Item& im = items[it];
        const wrio::FieldSpec& s = im.h.spec;
        const size_t ntot = s.count();
        if (im.decoded) {
            im.done.get();
            cout << "  decode: fld_1d_rec[0]=" << im.fld[0] << " fld_1d_rec[last]=" << im.fld[ntot - 1] << endl;
        }
        double lo = im.fld[0], hi = im.fld[0];
        for (size_t j = 0; j < ntot; j++) { lo = fmin(lo, im.fld[j]); hi = fmax(hi, im.fld[j]); }
        cout << "        min=" << lo << " max=" << hi << endl;
        wrio::write_field(out_name, it == 0, file_type, flip != 0, s, im.h.recl, im.fld.data());
        cout << "  wrote: fld_1d_rec[0]=" << im.fld[0] << " fld_1d_rec[last]=" << im.fld[ntot - 1] << endl;
        std::vector<double>().swap(im.fld);
        std::vector<unsigned char>().swap(im.data_enc);
    };
    for (int it = 0; it < nf; it++) {
        Item& im = items[it];
        wrio::FieldHeader& h = im.h;
        wrio::read_field_header(fheader, it, h);
        const wrio::FieldSpec& s = h.spec;
        // echo of the header values, gen_aux.cpp:626-643
        cout << "  tolabs; midval; halfspanval; wlev; nlay; ntot_enc;";
        if (h.ntot_enc > 0) cout << " deps_vec(1:nlay); minval_vec(1:nlay); len_enc_vec(1:nlay)" << endl; else cout << endl;
        cout << "  " << h.tolabs << " " << h.midval << " " << h.halfspanval << " " << h.wlev << " " << h.nlay << " " << h.ntot_enc << endl;
        if (h.ntot_enc > 0) {
            cout << "  "; for (unsigned j = 0; j < h.nlay; j++) cout << h.deps_vec[j] << " "; cout << endl;
            cout << "  "; for (unsigned j = 0; j < h.nlay; j++) cout << h.minval_vec[j] << " "; cout << endl;
            cout << "  "; for (unsigned j = 0; j < h.nlay; j++) cout << h.len_enc_vec[j] << " "; cout << endl;
        }
        cout << "  contains " << s.nbytes << "-byte floating point data" << endl;
        cout << "  nx=" << s.nx << "  ny=" << s.ny << "  nz=" << s.nz << "  nh=" << s.nh;
        if (s.idinv) cout << " and reordering" << endl; else cout << endl;
        const size_t ntot = s.count();
        im.fld.assign(ntot, 0.0);
        if (s.icomp) {
            for (size_t j = 0; j < ntot; j++) im.fld[j] = h.midval;  // gen_dec.cpp:201
            if (h.ntot_enc > 0) {
                im.data_enc.resize(h.ntot_enc);
                finput.read(reinterpret_cast<char*>(im.data_enc.data()), (std::streamsize)h.ntot_enc);
                if (finput.fail()) { cout << "Cannot read from " << in_name << endl; return 1; }
            }
        } else {
            wrio::read_raw_field(finput, s.nbytes, im.fld.data(), ntot);
        }
        if (depth > 0 && it - depth >= 0) finish(it - depth);
        if (s.icomp && h.ntot_enc > 0) {
            cout << "  decoding fld_1d_rec, field number " << it << endl;
            im.decoded = true;
            Item* ip = &im;
            auto work = [ip]() {
                wrio::FieldHeader& hh = ip->h;
                const wrio::FieldSpec& sp = hh.spec;
                unsigned char wlev = (unsigned char)hh.wlev, nlay = (unsigned char)hh.nlay;
                decoding_wrap(sp.nx, sp.ny, sp.nz * sp.nh, ip->fld.data(), &hh.tolabs, &hh.midval, &hh.halfspanval, &wlev, &nlay,
                              &hh.ntot_enc, hh.deps_vec, hh.minval_vec, hh.len_enc_vec, ip->data_enc.data());
            };
            if (depth > 0) im.done = std::async(std::launch::async, work);
            else { work(); std::promise<void> p; p.set_value(); im.done = p.get_future(); }
        }
        if (depth == 0) finish(it);
    }
    if (depth > 0)
        for (int it = std::max(0, nf - depth); it < nf; it++) finish(it);
    cout << "=== End of decompression ===\n";
    return 0;
}
