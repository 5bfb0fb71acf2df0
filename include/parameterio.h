/* parameterio.h -- drop-in for the reference's tools/parameterio.h (its examples say #include "parameterio.h" with -I tools):
 * WriteF0, ReadF0, GetHeaderInformation, Write/ReadSpectralEnvelope, Write/ReadAperiodicity (reference tools/parameterio.h:24-114), declared in world_hip.h (Part 1). */
#ifndef WORLD_HIP_FORWARD_PARAMETERIO_H_
#define WORLD_HIP_FORWARD_PARAMETERIO_H_
#include "world_hip.h"
#endif
