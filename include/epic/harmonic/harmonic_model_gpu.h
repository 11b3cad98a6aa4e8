/* Forwarding header: same include path as the reference's libepic/include/epic/harmonic/harmonic_model_gpu.h.
 * The declarations live in epic/epic_abi.h. */
#ifndef EPIC_FWD_HARMONIC_MODEL_GPU_H
#define EPIC_FWD_HARMONIC_MODEL_GPU_H
#include "../epic_abi.h"
#endif
