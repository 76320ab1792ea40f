// Version entry point of libautomoe_hip.so.
#include "am_common.h"
extern "C" int am_version(void) { return 1; }
