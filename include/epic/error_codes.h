/* Forwarding header: same include path as the reference's libepic/include/epic/error_codes.h.
 * The definitions live in epic/epic_abi.h. */
#ifndef EPIC_FWD_ERROR_CODES_H
#define EPIC_FWD_ERROR_CODES_H
#include "epic_abi.h"
#endif
