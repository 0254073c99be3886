#include "openjpg_compat.h"

#include "../../include/hjd.h"

int ConvertJpgFile(char *szJpgFileInName, char *szBmpFileOutName)
{
    return hjd_convert_jpg_file(szJpgFileInName, szBmpFileOutName);
}
