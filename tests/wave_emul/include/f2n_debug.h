#include "../../../include/f2n_debug.h"
