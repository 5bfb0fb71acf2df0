/* world/macrodefinitions.h -- the C-linkage macros of the reference header of the same name
 * (reference src/world/macrodefinitions.h:66-74), for callers that use them in their own headers. */
#ifndef WORLD_HIP_FORWARD_MACRODEFINITIONS_H_
#define WORLD_HIP_FORWARD_MACRODEFINITIONS_H_
#ifdef __cplusplus
#define WORLD_BEGIN_C_DECLS extern "C" {
#define WORLD_END_C_DECLS }
#else
#define WORLD_BEGIN_C_DECLS
#define WORLD_END_C_DECLS
#endif
#endif
