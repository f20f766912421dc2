#include "orbhip_common.h"
