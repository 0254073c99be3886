// hjd_convert — jpg in, bmp out: the reference's main.cpp:28-36 with the paths taken from argv
// (its own are hard-coded relative to the Vivado csim directory).
#include <stdio.h>

#include "../../include/hjd.h"
#include "openjpg_compat.h"

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.jpg out.bmp\n", argv[0]);
        return 2;
    }
    if (ConvertJpgFile(argv[1], argv[2]) != 1) {
        fprintf(stderr, "%s: %s\n", argv[1], hjd_status_string(hjd_last_status()));
        return 1;
    }
    return 0;
}
