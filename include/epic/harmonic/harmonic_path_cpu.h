/* Forwarding header: same include path as the reference's libepic/include/epic/harmonic/harmonic_path_cpu.h.
 * The declarations live in epic/epic_abi.h. */
#ifndef EPIC_FWD_HARMONIC_PATH_CPU_H
#define EPIC_FWD_HARMONIC_PATH_CPU_H
#include "../epic_abi.h"
#endif
