// Version / probing entry points of libsfron.so.
#include "common.h"
#include "../../include/sfron.h"
extern "C" {
int sfron_abi_version(void) { return 16; }
const char* sfron_build_arch(void) { return "gfx950"; }
}

// the pending completion event of the next producing launch (common.h SFRON_LAUNCH_EV); one slot per host thread
static thread_local hipEvent_t g_stop_event = nullptr;
void sfron_arm_stop_event(hipEvent_t ev) { g_stop_event = ev; }
hipEvent_t sfron_take_stop_event() { hipEvent_t e = g_stop_event; g_stop_event = nullptr; return e; }
