/* audioio.h -- drop-in for the reference's tools/audioio.h (its examples say #include "audioio.h" with -I tools):
 * wavwrite, GetAudioLength, wavread (reference tools/audioio.h:25-47), declared in world_hip.h (Part 1). */
#ifndef WORLD_HIP_FORWARD_AUDIOIO_H_
#define WORLD_HIP_FORWARD_AUDIOIO_H_
#include "world_hip.h"
#endif
