// ABI version / build info of libmfc.so.
#include "mfc_common.h"

extern "C" int mfc_abi_version(void) { return MFC_ABI_VERSION; }

extern "C" const char* mfc_build_info(void) {
    return "libmfc gfx950 (CDNA4) wave64; MFMA f32 16x16x4 / bf16 16x16x16; built " __DATE__;
}
