/* world/cheaptrick.h -- drop-in for the reference header of the same name: a caller that says
 * #include "world/cheaptrick.h" compiles against this repository's include/ directory unchanged.
 * Declares CheapTrickOption, CheapTrick, InitializeCheapTrickOption, GetFFTSizeForCheapTrick, GetF0FloorForCheapTrick (reference src/world/cheaptrick.h:16-80);
 * all declarations live in ../world_hip.h (Part 1), which cites the reference line of each. */
#ifndef WORLD_HIP_FORWARD_CHEAPTRICK_H_
#define WORLD_HIP_FORWARD_CHEAPTRICK_H_
#include "../world_hip.h"
#endif
