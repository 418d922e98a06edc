// forwards to the ABI header of the repository (the rewritten source copies of _build/csrc include "../../include/f2n_abi.h")
#include "../../../include/f2n_abi.h"
