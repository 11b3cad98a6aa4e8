/* Forwarding header: same include path as the reference's libepic/include/epic/constants.h.
 * The definitions live in epic/epic_abi.h. */
#ifndef EPIC_FWD_CONSTANTS_H
#define EPIC_FWD_CONSTANTS_H
#include "epic_abi.h"
#endif
