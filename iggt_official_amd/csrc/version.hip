#include "../../include/iggt_hip.h"
extern "C" int iggt_hip_abi_version(void) { return 26; }
