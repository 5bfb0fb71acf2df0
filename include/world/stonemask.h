/* world/stonemask.h -- drop-in for the reference header of the same name: a caller that says
 * #include "world/stonemask.h" compiles against this repository's include/ directory unchanged.
 * Declares StoneMask (reference src/world/stonemask.h:27);
 * all declarations live in ../world_hip.h (Part 1), which cites the reference line of each. */
#ifndef WORLD_HIP_FORWARD_STONEMASK_H_
#define WORLD_HIP_FORWARD_STONEMASK_H_
#include "../world_hip.h"
#endif
