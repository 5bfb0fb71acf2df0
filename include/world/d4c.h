/* world/d4c.h -- drop-in for the reference header of the same name: a caller that says
 * #include "world/d4c.h" compiles against this repository's include/ directory unchanged.
 * Declares D4COption, D4C, InitializeD4COption (reference src/world/d4c.h:16-46);
 * all declarations live in ../world_hip.h (Part 1), which cites the reference line of each. */
#ifndef WORLD_HIP_FORWARD_D4C_H_
#define WORLD_HIP_FORWARD_D4C_H_
#include "../world_hip.h"
#endif
