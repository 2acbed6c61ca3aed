// ABI bookkeeping of libemlight_hip.so.
#include "eml_common.h"

extern "C" int eml_abi_version(void) { return EML_ABI_VERSION; }
extern "C" const char* eml_last_error(void) { return eml::err_buf(); }
