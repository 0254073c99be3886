// reference_decodehw.cpp — the ONE translation unit a maintainer of the reference adds for the drop-in at the HW/SW cut
// (INTEGRATION.md §2): JpegDecodeHW (loadjpg.h:180, the HLS top function, called at openjpg.cpp:655) defined through
// include/hjd_reference_shim.h on top of libhjd.so.  Not part of the library: it needs the reference's headers.  The test recipe links it with the reference's main.cpp and openjpg.cpp,
// compiled where they lie and unchanged, in place of loadjpg.cpp (refhost_hjd; refhost_hjd_big with the capacity
// constants of loadjpg.h:55-57 enlarged and an argv driver); tests/test_gpu_parity.py runs both on the GPU and compares the
// BMP that the reference's own WriteBMP24 (openjpg.cpp:504-570) wrote with the hashes the compiled reference produced.
// Width, height and the sampling factors are the call's own arguments; Ri is not available at this cut (openjpg.cpp:441-446).
#include "hjd_reference_shim.h"

int JpegDecodeHW(stJpegData *jdata, unsigned int jpeg_img_height, unsigned int jpeg_img_width, unsigned char hFactor, unsigned char vFactor)
{
    return hjd_shim_jpeg_decode_hw(jdata, jpeg_img_height, jpeg_img_width, hFactor, vFactor);
}
