// fi_common.h -- shared host-side helpers for libfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fi_capi.h"

// Every fp32 multiply/add in this library is rounded separately: sampling
// coordinates feed floorf/ceilf and IoUs feed a threshold compare, both of which
// decide integer outputs that must match the CPU specification bit for bit.
#pragma clang fp contract(off)

namespace fi {

template <int V>
struct Int {
    static constexpr int value = V;      // a compile-time integer as a function argument (generic lambdas)
};

void set_error(const char *fmt, ...);

// conv1x1_ring.hip (called from fi_conv2d_forward with weight_layout 3)
struct RingArgs {
    const float *x, *wF, *bias, *scale, *residual, *gate;
    float *y;
    const float *zero;
    int N, Cin, HW, Cout, relu;
};
int launch_conv1x1_ring(const RingArgs &a, hipStream_t st);

#define FI_HIP_CHECK(expr)                                                         \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            fi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                          __FILE__, __LINE__);                                     \
            return FI_ERR_HIP;                                                     \
        }                                                                          \
    } while (0)

#define FI_REQUIRE(cond, msg)                                                      \
    do {                                                                           \
        if (!(cond)) {                                                             \
            fi::set_error("invalid argument: %s (%s)", msg, #cond);                \
            return FI_ERR_INVALID_ARG;                                             \
        }                                                                          \
    } while (0)

// RAII kernel timer: when profiling is enabled records a start event on `stream`
// at construction and a stop event at destruction (i.e. around the launch).
class ProfScope {
  public:
    ProfScope(int kernel_id, hipStream_t stream);
    ~ProfScope();

  private:
    int id_;
    hipStream_t stream_;
    hipEvent_t start_;
    bool active_;
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace fi
