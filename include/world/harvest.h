/* world/harvest.h -- drop-in for the reference header of the same name: a caller that says
 * #include "world/harvest.h" compiles against this repository's include/ directory unchanged.
 * Declares HarvestOption, Harvest, InitializeHarvestOption, GetSamplesForHarvest (reference src/world/harvest.h:16-59);
 * all declarations live in ../world_hip.h (Part 1), which cites the reference line of each. */
#ifndef WORLD_HIP_FORWARD_HARVEST_H_
#define WORLD_HIP_FORWARD_HARVEST_H_
#include "../world_hip.h"
#endif
