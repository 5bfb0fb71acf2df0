/* world/dio.h -- drop-in for the reference header of the same name: a caller that says
 * #include "world/dio.h" compiles against this repository's include/ directory unchanged.
 * Declares DioOption, Dio, InitializeDioOption, GetSamplesForDIO (reference src/world/dio.h:16-61);
 * all declarations live in ../world_hip.h (Part 1), which cites the reference line of each. */
#ifndef WORLD_HIP_FORWARD_DIO_H_
#define WORLD_HIP_FORWARD_DIO_H_
#include "../world_hip.h"
#endif
