// openjpg_compat.h — the reference's C++ entry point, unchanged in name and signature
// (reference src/openjpg.h:23), so that its main.cpp:33 links against this library as is.
#pragma once

// Pass in the filename and it creates a bitmap.  Returns 1 on success, 0 on failure
// (openjpg.cpp:607,643,665,683).  Forwards to the C ABI hjd_convert_jpg_file().
int ConvertJpgFile(char *szJpgFileInName, char *szBmpFileOutName);
