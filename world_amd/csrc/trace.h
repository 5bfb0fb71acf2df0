// trace.h -- development aid, compiled out of the product.  With -DWH_TRACE a kernel can
// stamp the shader clock at phase boundaries of ONE chosen workgroup (`trace_me`); the stamps
// are read back through world_hip_trace_read_<unit>() (tools/trace.py prints the deltas).
// In-situ phase latencies are what rocprofv3's per-kernel totals cannot show.
#pragma once
// which workgroup of the frame kernels stamps: frame WH_TRACE_FRAME of utterance WH_TRACE_UTT (tools/trace.py: a lone 10 s
// utterance; tools/trace_batch.py: the middle of a 64 x 5 s batch, every CU loaded with other frames' workgroups)
#ifndef WH_TRACE_FRAME
#define WH_TRACE_FRAME 1000
#endif
#ifndef WH_TRACE_UTT
#define WH_TRACE_UTT 0
#endif
#if defined(WH_TRACE) && !defined(WORLD_EMU)
namespace world_hip { static __device__ long long wh_trace[128]; }   // one per translation unit
#define WH_TRACE_DEFINE(unit)                                                                        \
  extern "C" __attribute__((visibility("default"))) int world_hip_trace_read_##unit(long long *out, int n) { \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(world_hip::wh_trace), sizeof(long long) * n);      \
  }
#define WH_STAMP(base, k) do { if (trace_me && threadIdx.x == 0) wh_trace[(base) + (k)] = clock64(); } while (0)
// accumulating form for phases that repeat inside loops: WH_ACC_DECL once, then BEGIN/END pairs
#define WH_ACC_DECL long long wh_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wh_t0 = 0; (void)wh_t0
#define WH_ACC_BEGIN do { if (trace_me) wh_t0 = clock64(); } while (0)
#define WH_ACC_END(k) do { if (trace_me) wh_acc[k] += clock64() - wh_t0; } while (0)
#define WH_ACC_COUNT(k) do { if (trace_me) wh_acc[k] += 1; } while (0)
#define WH_ACC_SET(k, v) do { if (trace_me) wh_acc[k] = (v); } while (0)
#define WH_ACC_FLUSH(base, lane0) do { if (trace_me && (lane0)) for (int k_ = 0; k_ < 8; ++k_) wh_trace[(base) + k_] = wh_acc[k_]; } while (0)
#else
#define WH_ACC_DECL
#define WH_ACC_BEGIN do { } while (0)
#define WH_ACC_END(k) do { } while (0)
#define WH_ACC_COUNT(k) do { } while (0)
#define WH_ACC_SET(k, v) do { } while (0)
#define WH_ACC_FLUSH(base, lane0) do { } while (0)
#define WH_TRACE_DEFINE(unit)
#define WH_STAMP(base, k) do { } while (0)
#endif
