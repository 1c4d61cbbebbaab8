// pvd_common.hip -- ABI version, status strings and the per-thread last-HIP-error slot.
#include "pvd_device.h"

namespace pvd {
static thread_local hipError_t g_last_error = hipSuccess;
void set_last_error(hipError_t e) { g_last_error = e; }
}  // namespace pvd

extern "C" {

int pvd_abi_version(void) { return 6; }  // 2: pvd_adamw_extras gained snapshot / replay; 3: + warm_zero_grad_from, zero_grad_after, arrivals; 4: pvd_head_dw_rider.found_inf; 5: pvd_adamw_extras.compact_grad / compact_param_out / tail_clear, pvd_composite_objective_forward(arrive), pvd_segments_gather_zero_check; 6: pvd_vm_forward_pack_rider

const char *pvd_status_string(int status) {
    switch (status) {
        case PVD_OK: return "ok";
        case PVD_ERR_INVALID: return "invalid argument (null pointer, misaligned buffer or bad size)";
        case PVD_ERR_UNSUPPORTED: return "unsupported D / C / L / degree / dtype";
        case PVD_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown status";
    }
}

const char *pvd_last_hip_error(void) {
    return pvd::g_last_error == hipSuccess ? "" : hipGetErrorName(pvd::g_last_error);
}

}  // extern "C"
