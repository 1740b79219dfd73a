// Library-level entry points of libcp2hip.so.
#include <hip/hip_runtime.h>
#include "common.hpp"

extern "C" __attribute__((visibility("default"))) int cp2_version(void) { return 200; }   // 0.2.0: split workspaces of the dense kernels, cp2_sgd_flat, cp2_wgrad1x1

extern "C" __attribute__((visibility("default"))) const char* cp2_error_string(int code) {
    switch (code) {
        case CP2_OK: return "ok";
        case CP2_ERR_NULL: return "a required pointer is NULL";
        case CP2_ERR_SHAPE: return "a size is zero, negative or inconsistent";
        case CP2_ERR_UNSUPPORTED: return "size outside the supported range of this kernel";
        case CP2_ERR_ALIGN: return "pointer is not 16-byte aligned";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown cp2 error";
    }
}

// Arm start / stop events for the next profiled launch of the calling thread (common.hpp); NULL, NULL disarms.
extern "C" __attribute__((visibility("default"))) int cp2_profile_next_launch(void* start_event, void* stop_event) {
    if ((start_event == nullptr) != (stop_event == nullptr)) return CP2_ERR_NULL;
    cp2_next_events.start = reinterpret_cast<hipEvent_t>(start_event);
    cp2_next_events.stop = reinterpret_cast<hipEvent_t>(stop_event);
    return CP2_OK;
}
