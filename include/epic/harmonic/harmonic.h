/* Forwarding header: same include path as the reference's libepic/include/epic/harmonic/harmonic.h.
 * The declarations live in epic/epic_abi.h. */
#ifndef EPIC_FWD_HARMONIC_H
#define EPIC_FWD_HARMONIC_H
#include "../epic_abi.h"
#endif
