/* world/codec.h -- drop-in for the reference header of the same name: a caller that says
 * #include "world/codec.h" compiles against this repository's include/ directory unchanged.
 * Declares GetNumberOfAperiodicities, CodeAperiodicity, DecodeAperiodicity, CodeSpectralEnvelope, DecodeSpectralEnvelope (reference src/world/codec.h:23-86);
 * all declarations live in ../world_hip.h (Part 1), which cites the reference line of each. */
#ifndef WORLD_HIP_FORWARD_CODEC_H_
#define WORLD_HIP_FORWARD_CODEC_H_
#include "../world_hip.h"
#endif
