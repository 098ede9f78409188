// wrdec (FluSI) -- HDF5 front-end of the decoder, reference src/flusi/main_dec.cpp.
//   wrdec compressed_000.h5 decompressed_000.h5 TYPE PRECISION   PRECISION 1: single, 2: double
#include <cmath>
#include <memory>

#include "flusi_common.h"
#include "flusi_h5.h"

using std::cout;
using std::endl;
using std::string;

namespace {
struct Item {
    string name;
    int nx = 0, ny = 0, nz = 0;
    bool backup = false;
    double bckp[8] = {0};
    double time = 0, nu = 0, epsi = 0, domain[3] = {0, 0, 0};
    int nxyz[3] = {0, 0, 0};
    std::vector<unsigned char> data;
    std::unique_ptr<double[]> rec;   // not zero-filled: the decoder writes every element
    size_t n = 0;
    double lo = 0, hi = 0;
    flusi::Phases ph;
    wr_enc_info info;
    std::future<int> done;
};
}  // namespace

int main(int argc, char** argv)
{
    static const char* const keys[4] = {"&in_name", "&out_name", "&file_type", "&precision"};
    static const char* const prompts[4] = {"Enter compressed data file name []: ", "Enter reconstructed file name []: ",
                                           "Enter file type (0: regular output; 1: backup) [0]: ",
                                           "Enter output data type (1: float; 2: double) [2]: "};
    string p[4];
    if (!flusi::get_params(argc, argv, false, keys, prompts,
                           "usage: ./wrdec compressed_000.h5 decompressed_000.h5 TYPE PRECISION\n"
                           "where TYPE=(0: regular output; 1: backup) and PRECISION=(1:single; 2:double)\n"
                           "interactive mode if not enough arguments are passed.\n"
                           "note: the HDF5 container layout follows the sources of the reference's FluSI tools (src/flusi); it could not be\n"
                           "      compared with files written by them (they do not compile with current g++ and ship no sample files): the coded\n"
                           "      bytes and coding attributes inside are bit-identical to the reference codec's, the container itself is unpinned.\n", p))
        return -1;
    int file_type = 0, outtype = 1;  // main_dec.cpp:66 defaults
    std::stringstream(p[2]) >> file_type;
    std::stringstream(p[3]) >> outtype;
    const string in_name = p[0], out_name = p[1];
    cout << endl << "=== Decoding parameters ===" << endl;
    cout << "Input file name: " << in_name << endl << "Output file name: " << out_name << endl;
    cout << "File type (0: regular output; 1: backup): " << file_type << endl;
    cout << "Output data type (1: float; 2: double): " << outtype << endl;
    if (file_type != 0 && file_type != 1) { cout << "Error: unknown file type" << endl; return 0; }

    flusi::create_file(out_name);
    std::vector<string> names;
    if (file_type == 0) {
        std::vector<string> all = flusi::dataset_names(in_name);
        if (all.empty()) { cout << "no dataset in " << in_name << endl; return 1; }
        names.push_back(all.back());
    } else {
        for (const char* n : flusi::kBackupNames)
            if (flusi::has_dataset(in_name, n)) names.push_back(n);
    }
    flusi::Pipeline pipe;
    flusi::Clock clk;
    std::vector<Item> items(names.size());
    auto finish = [&](Item& it) {
        if (it.done.get() != WR_OK) { std::cerr << "wrdec: " << it.name << ": decode failed" << endl; exit(1); }
        it.ph.write0 = clk.now();
        const size_t n = it.n;
        cout << "  decode: fld_1d_rec[0]=" << it.rec[0] << " fld_1d_rec[last]=" << it.rec[n - 1] << endl;
        cout << "        min=" << it.lo << " max=" << it.hi << endl;
        flusi::write_field(out_name, it.name, it.rec.get(), it.nx, it.ny, it.nz, outtype == 1);
        if (it.backup) flusi::write_attr_double(out_name, it.name, "bckp", it.bckp, 8);
        else {
            flusi::write_attr_double(out_name, it.name, "time", &it.time, 1);
            flusi::write_attr_double(out_name, it.name, "viscosity", &it.nu, 1);
            flusi::write_attr_double(out_name, it.name, "epsi", &it.epsi, 1);
            flusi::write_attr_double(out_name, it.name, "domain_size", it.domain, 3);
            flusi::write_attr_int(out_name, it.name, "nxyz", it.nxyz, 3);
        }
        it.rec.reset();
        it.ph.write1 = clk.now();
        if (clk.on) it.ph.print(it.name);
    };
    size_t next_to_finish = 0;
    for (size_t k = 0; k < items.size(); k++) {
        Item& it = items[k];
        it.name = names[k];
        it.backup = file_type == 1;
        if (it.backup) {
            if (!flusi::read_attr_double(in_name, it.name, "bckp", it.bckp, 8)) { cout << it.name << ": no bckp attribute" << endl; return 1; }
            it.nx = int(it.bckp[5]); it.ny = int(it.bckp[6]); it.nz = int(it.bckp[7]);
        } else {
            bool ok = flusi::read_attr_double(in_name, it.name, "time", &it.time, 1);
            ok &= flusi::read_attr_double(in_name, it.name, "viscosity", &it.nu, 1);
            ok &= flusi::read_attr_double(in_name, it.name, "epsi", &it.epsi, 1);
            ok &= flusi::read_attr_double(in_name, it.name, "domain_size", it.domain, 3);
            ok &= flusi::read_attr_int(in_name, it.name, "nxyz", it.nxyz, 3);
            if (!ok) { cout << it.name << ": regular-output attributes missing" << endl; return 1; }
            it.nx = it.nxyz[0]; it.ny = it.nxyz[1]; it.nz = it.nxyz[2];
        }
        cout << " dset=" << it.name << " nx=" << it.nx << " ny=" << it.ny << " nz=" << it.nz << endl;
        const size_t n = (size_t)it.nx * it.ny * it.nz;
        if (k == 0 && !pipe.open(n, (int)items.size())) return 1;
        if (k - next_to_finish >= (size_t)pipe.depth()) finish(items[next_to_finish++]);
        it.ph.read0 = clk.now();
        flusi::read_coded(in_name, it.name, it.data, it.info);
        it.ph.read1 = clk.now();
        it.n = n;
        it.rec.reset(new double[n]);
        wr_ctx* c = pipe.ctx[k % pipe.depth()];
        Item* ip = &it;
        const flusi::Clock* ck = &clk;
        it.done = std::async(std::launch::async, [c, ip, ck]() {
            ip->ph.call0 = ck->now();
            const int rc = wr_decode_host(c, ip->rec.get(), ip->nx, ip->ny, ip->nz, &ip->info, ip->data.data(), ip->data.size(), &ip->ph.tm);
            ip->ph.call1 = ck->now();
            std::vector<unsigned char>().swap(ip->data);
            if (rc == WR_OK) flusi::minmax(ip->rec.get(), ip->n, &ip->lo, &ip->hi);
            return rc;
        });
    }
    while (next_to_finish < items.size()) finish(items[next_to_finish++]);
    cout << "=== End of decompression ===\n";
    return 0;
}
