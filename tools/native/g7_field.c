/* The field of the reference's Fortran example (examples/fortran/example_fort.f90:82-91), evaluated in C with libm:
 *   fld(ix,iy,iz) = 10 * sin((ix-1)/nx) * sin((iy-1)/ny)**2 * cos((iz-1)/nz),  ix fastest, 1-based indices
 * usage: g7_field nx ny nz out.raw   (doubles, x fastest).  gcc -O2 -ffp-contract=off g7_field.c -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char** argv)
{
    if (argc != 5) return 2;
    const int nx = atoi(argv[1]), ny = atoi(argv[2]), nz = atoi(argv[3]);
    FILE* f = fopen(argv[4], "wb");
    if (!f) return 1;
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                const double sy = sin((double)iy / (double)ny);
                /* Fortran evaluates a*b*c*d left to right; x**2 is x*x */
                const double v = ((10.0 * sin((double)ix / (double)nx)) * (sy * sy)) * cos((double)iz / (double)nz);
                fwrite(&v, sizeof v, 1, f);
            }
    fclose(f);
    return 0;
}
