// Default build: the experiment kernels of round 2 (gg_p2.hip, gg_bd.hip, gg_wg2.hip -- bit-exact, measured slower than
// the defaults, see their headers and DESIGN.md section 9) are NOT part of libpai_hip.so.  Building with
// PAI_EXPERIMENTAL=1 (thesis-pai-reconstruction_amd/build.py) compiles them instead of this file.
#include "common.h"

int fwd_p2_rows(const GG&) { return 0; }
int launch_fwd_p2(const GG&, const FwdArgs&, hipStream_t) { pai_set_error("gg_p2.hip is not in this build (PAI_EXPERIMENTAL=1)"); return 1; }
const char* fwd_p2_kernel_name(const GG&) { return ""; }
int fwd_bd_rows(const GG&) { return 0; }
int launch_fwd_bd(const GG&, const FwdArgs&, hipStream_t) { pai_set_error("gg_bd.hip is not in this build (PAI_EXPERIMENTAL=1)"); return 1; }
const char* fwd_bd_kernel_name(const GG&) { return ""; }
bool wgrad2_ok(const GG&) { return false; }
int launch_wgrad2(const GG&, const WgradArgs&, hipStream_t) { pai_set_error("gg_wg2.hip is not in this build (PAI_EXPERIMENTAL=1)"); return 1; }

extern "C" int pai_pack_frag(const void*, int, int, void*, void*) {
    pai_set_error("pai_pack_frag: the register-direct weight kernel (gg_bd.hip) is not in this build; rebuild with PAI_EXPERIMENTAL=1");
    return 1;
}
