// wrenc (FluSI) -- HDF5 front-end of the encoder, reference src/flusi/main_enc.cpp.
//   wrenc original_000.h5 compressed_000.h5 TYPE TOLERANCE     TYPE 0: regular output, 1: backup
// Regular files hold one dataset with attributes time / viscosity / epsi / domain_size / nxyz;
// backup files hold up to 50 named datasets with a `bckp` attribute (main_enc.cpp:319-330).
// The datasets of a file are in flight together (BASELINE config 5: ux, uy, uz overlap -- dataset k+1's read, upload and
// transform run under dataset k's range coding; flusi_common.h::Pipeline).
#include <cmath>

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
    std::vector<double> fld;
    flusi::RawBytes data;
    wr_enc_info info;
    double lo = 0, hi = 0;
    flusi::Phases ph;
    std::future<int> done;
};

}  // namespace

int main(int argc, char** argv)
{
    static const char* const keys[4] = {"&in_name", "&out_name", "&file_type", "&tolerance"};
    static const char* const prompts[4] = {"Enter input file name []: ", "Enter output file name []: ",
                                           "Enter file type (0: regular output; 1: backup) [0]: ",
                                           "Enter base cutoff relative tolerance [1e-16]: "};
    string p[4];
    if (!flusi::get_params(argc, argv, true, keys, prompts,
                           "usage: ./wrenc original_000.h5 compressed_000.h5 TYPE TOLERANCE\n"
                           "where TYPE=(0: regular output; 1: backup) and TOLERANCE=(e.g. 1.0e-5)\n"
                           "interactive mode if not enough arguments are passed.\n"
                           "note: the HDF5 container layout follows the sources of the reference's FluSI tools (src/flusi); it could not be\n"
                           "      compared with files written by them (they do not compile with current g++ and ship no sample files): the coded\n"
                           "      bytes and coding attributes inside are bit-identical to the reference codec's, the container itself is unpinned.\n", p))
        return -1;
    int file_type = 0;
    double tol = 1e-16;
    std::stringstream(p[2]) >> file_type;
    std::stringstream(p[3]) >> tol;
    const string in_name = p[0], out_name = p[1];
    cout << endl << "=== Compression parameters ===" << endl;
    cout << "Input file name: " << in_name << endl << "Output file name: " << out_name << endl;
    cout << "File type (0: regular output; 1: backup): " << file_type << endl;
    cout << "Base cutoff relative tolerance: " << tol << endl;
    if (file_type != 0 && file_type != 1) { cout << "Error: unknown file type" << endl; return 0; }

    flusi::create_file(out_name);
    std::vector<string> names;
    if (file_type == 0) {
        std::vector<string> all = flusi::dataset_names(in_name);
        if (all.empty()) { cout << "no dataset in " << in_name << endl; return 1; }
        names.push_back(all.back());  // the reference keeps the last dataset its H5Ovisit callback sees
    } else {
        for (const char* n : flusi::kBackupNames)
            if (flusi::has_dataset(in_name, n)) names.push_back(n);
    }

    flusi::Pipeline pipe;
    flusi::Clock clk;
    std::vector<Item> items(names.size());
    auto finish = [&](Item& it) {  // in dataset order: wait for the codec, then write (main thread owns HDF5)
        if (it.done.get() != WR_OK) { std::cerr << "wrenc: " << it.name << ": encode failed" << endl; exit(1); }
        it.ph.write0 = clk.now();
        cout << " dset=" << it.name << "  min=" << it.lo << " max=" << it.hi << endl;
        cout << "        tolabs=" << it.info.tolabs << endl;
        flusi::write_coded(out_name, it.name, it.data.p, it.info);
        if (it.backup) flusi::write_attr_double(out_name, it.name, "bckp", it.bckp, 8);
        else {
            flusi::write_attr_double(out_name, it.name, "time", &it.time, 1);
            flusi::write_attr_double(out_name, it.name, "viscosity", &it.nu, 1);
            flusi::write_attr_double(out_name, it.name, "epsi", &it.epsi, 1);
            flusi::write_attr_double(out_name, it.name, "domain_size", it.domain, 3);
            flusi::write_attr_int(out_name, it.name, "nxyz", it.nxyz, 3);
        }
        it.data.release();
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
            it.nx = int(it.bckp[5]); it.ny = int(it.bckp[6]); it.nz = int(it.bckp[7]);  // main_enc.cpp:469-474
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
        if (k == 0 && !pipe.open(n, (int)items.size())) return 1;  // the first dataset sizes the pipeline
        if (k - next_to_finish >= (size_t)pipe.depth()) finish(items[next_to_finish++]);  // frees the oldest dataset's context
        it.ph.read0 = clk.now();
        flusi::read_field(in_name, it.name, it.fld, n);
        it.ph.read1 = clk.now();
        cout << "  read: fld_1d[0]=" << it.fld[0] << " fld_1d[last]=" << it.fld[n - 1] << endl;
        unsigned char nl; unsigned long cap;
        setup_wr(it.nx, it.ny, it.nz, &nl, &cap);
        if (!it.data.allocate(cap)) { std::cerr << "wrenc: out of memory" << endl; return 1; }
        wr_ctx* c = pipe.ctx[k % pipe.depth()];
        Item* ip = &it;
        const flusi::Clock* ck = &clk;
        it.done = std::async(std::launch::async, [c, ip, tol, n, ck]() {
            flusi::minmax(ip->fld.data(), n, &ip->lo, &ip->hi);
            const double cutoff = tol;
            ip->ph.call0 = ck->now();
            const int rc = wr_encode_host(c, ip->fld.data(), ip->nx, ip->ny, ip->nz, 1, 1, 1, 1, &cutoff, &ip->info, ip->data.p,
                                          ip->data.bytes, &ip->ph.tm);
            ip->ph.call1 = ck->now();
            std::vector<double>().swap(ip->fld);
            return rc;
        });
    }
    while (next_to_finish < items.size()) finish(items[next_to_finish++]);
    cout << "=== End of compression ===\n";
    return 0;
}
