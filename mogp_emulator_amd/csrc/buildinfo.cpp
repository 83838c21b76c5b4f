// Which commit this libmogp_hip.so was built from (Makefile: BUILD_COMMIT).  tools/collect_profiles.py refuses to file profiles of a
// library whose stamp is not the checked-out HEAD.
#include "../../include/mogp_hip.h"
extern "C" const char* mogp_build_commit(void) { return MOGP_BUILD_COMMIT; }
