// ref_driver.cpp — argv driver for the compiled reference (oracle/_ref builds only).
// The reference's own main.cpp:28-36 hard-codes its paths; this calls the same entry
// point, ConvertJpgFile (openjpg.h:23), with paths from the command line.
#include "openjpg.h"
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    return ConvertJpgFile(argv[1], argv[2]) == 1 ? 0 : 1;
}
